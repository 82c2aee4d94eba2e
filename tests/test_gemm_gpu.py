"""Parity of the MFMA GEMM family (ark_amd/csrc/gemm.hip) against plain torch fp32 matmul."""
import pytest
import torch

from ark_amd import _lib as L

pytestmark = pytest.mark.gpu


def run_gemm(prec, a_lay, b_lay, M, N, K, epi=L.EPI_NONE, seed=0, accumulate=False):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    aux = torch.randn(M, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    dev = torch.device("cuda:0")
    Ad = (A if a_lay == L.LAY_KMAJ else A.t().contiguous()).to(dev)
    Bd = (B if b_lay == L.LAY_KMAJ else B.t().contiguous()).to(dev)
    lda = K if a_lay == L.LAY_KMAJ else M
    ldb = K if b_lay == L.LAY_KMAJ else N
    C = C0.to(dev).clone()
    C2 = torch.zeros(M, N, device=dev)
    bias_d, aux_d = bias.to(dev), aux.to(dev)  # keep alive across the async launch
    rc = L.lib().ark_gemm(L.i32(prec), L.i32(a_lay), L.i32(b_lay), L.i32(epi), L.ptr(Ad), L.i64(lda), L.ptr(Bd),
                          L.i64(ldb), L.ptr(C), L.i64(N), L.ptr(C2), L.ptr(bias_d), L.ptr(aux_d),
                          L.i32(M), L.i32(N), L.i32(K), L.i32(1 if accumulate else 0), L.cur_stream())
    L.check(rc, "ark_gemm")
    torch.cuda.synchronize()
    if prec == L.PREC_BF16:
        ref = A.bfloat16().float().double() @ B.bfloat16().float().double().t()
    elif prec == L.PREC_F16:
        ref = A.half().float().double() @ B.half().float().double().t()
    else:
        ref = A.double() @ B.double().t()
    if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU):
        ref = ref + bias.double()
    out2 = None
    if epi == L.EPI_BIAS_GELU:
        out2 = torch.nn.functional.gelu(ref)
    if epi == L.EPI_MUL_DGELU:
        x = aux.double().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    if epi == L.EPI_MUL_AUX:
        ref = ref * aux.double()
    if accumulate:
        ref = ref + C0.double()
    return C.cpu().double(), ref, (C2.cpu().double(), out2)


SHAPES = [(64, 64, 32), (128, 128, 64), (100, 55, 72), (1024, 1536, 1536), (333, 20, 1536), (1536, 512, 1000),
          (70, 130, 55), (2048, 2048, 96)]


@pytest.mark.parametrize("prec", [L.PREC_F32, L.PREC_BF16, L.PREC_F16])
@pytest.mark.parametrize("a_lay", [L.LAY_KMAJ, L.LAY_MMAJ])
@pytest.mark.parametrize("b_lay", [L.LAY_KMAJ, L.LAY_MMAJ])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_layouts(prec, a_lay, b_lay, shape):
    M, N, K = shape
    out, ref, _ = run_gemm(prec, a_lay, b_lay, M, N, K)
    scale = K ** 0.5
    # tolerance: fp32 accumulation-order noise only (bf16 reference uses the same rounded operands)
    assert (out - ref).abs().max().item() <= 2e-5 * scale + 1e-5, (out - ref).abs().max().item()


@pytest.mark.parametrize("prec", [L.PREC_F32, L.PREC_BF16])
@pytest.mark.parametrize("epi", [L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_MUL_DGELU, L.EPI_MUL_AUX])
def test_gemm_epilogues(prec, epi):
    out, ref, (o2, r2) = run_gemm(prec, L.LAY_KMAJ, L.LAY_KMAJ, 200, 136, 264, epi=epi)
    assert (out - ref).abs().max().item() <= 2e-4
    if r2 is not None:
        assert (o2 - r2).abs().max().item() <= 2e-4


def test_gemm_accumulate():
    out, ref, _ = run_gemm(L.PREC_F32, L.LAY_MMAJ, L.LAY_MMAJ, 96, 160, 300, accumulate=True)
    assert (out - ref).abs().max().item() <= 2e-4


def test_gemm_bad_args():
    rc = L.lib().ark_gemm(L.i32(7), L.i32(0), L.i32(0), L.i32(0), L.ptr(None), L.i64(1), L.ptr(None), L.i64(1),
                          L.ptr(None), L.i64(1), L.ptr(None), L.ptr(None), L.ptr(None), L.i32(1), L.i32(1), L.i32(1),
                          L.i32(0), L.cur_stream())
    assert rc < 0


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_F16])
@pytest.mark.parametrize("tile,nbuf,waves", [(64, 4, 8), (64, 2, 8), (128, 3, 8), (128, 2, 8), (128, 2, 4), (128, 3, 4)])
@pytest.mark.parametrize("shape", [(64, 64, 64), (192, 128, 640), (1536, 512, 2048), (128, 256, 10240)])
def test_wgrad16_tr_read_kernel(prec, tile, nbuf, waves, shape):
    """C += A^T B with both operands reduction-major 16-bit (LDS-DMA + ds_read_b64_tr_b16)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16 if prec == L.PREC_BF16 else torch.float16
    A = torch.randn(K, M, generator=g).to(dt)
    B = torch.randn(K, N, generator=g).to(dt)
    C0 = torch.randn(M, N, generator=g)
    dev = torch.device("cuda:0")
    Ad, Bd, C = A.to(dev), B.to(dev), C0.to(dev).clone()
    tn = L.wgrad_tuning(tile=tile, nbuf=nbuf, target_wgs=512, waves=waves)
    L.check(L.lib().ark_wgrad16(L.i32(prec), L.ptr(Ad), L.i64(M), L.ptr(Bd), L.i64(N), L.ptr(C), L.i64(N), L.i32(M), L.i32(N),
                                L.i32(K), tn, L.cur_stream()), "ark_wgrad16")
    torch.cuda.synchronize()
    ref = C0.double() + A.double().t() @ B.double()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 3e-5 * (K ** 0.5) + 1e-4, err


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_F16])
@pytest.mark.parametrize("B,Lq,V,ncols,ld", [(16, 3, 55, 192, 256), (48, 10, 130, 384, 512), (1024, 10, 55, 1536, 2048)])
def test_token_sums16(prec, B, Lq, V, ncols, ld):
    """ark_token_sums16: S[v] += sum of the panel rows whose INPUT token is v (rows time-major; += semantics; columns
    beyond n_cols of the panel untouched) against index_add_ in fp64"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3 + B)
    seq = torch.randint(0, V, (B, Lq + 1), generator=g)
    Vp = (V + 63) // 64 * 64
    R = B * Lq
    dt = torch.bfloat16 if prec == L.PREC_BF16 else torch.float16
    X = (torch.randn(R, ld, generator=g) * 0.01).to(dt)
    S = torch.full((Vp, ncols), 0.5, device=dev)
    seq_d, X_d = seq.to(dev), X.to(dev)     # (named: a temporary would be freed before the launch reads it)
    seq_d[:, 0] = 1                         # every row of step 0 carries the same token (BOS), as in the real decoder input
    seq[:, 0] = 1
    scratch = torch.empty(4 * R + 32 + 4 * (R // 64 + 1), dtype=torch.uint8, device=dev)
    L.check(L.lib().ark_token_sums16(L.i32(prec), L.ptr(seq_d), L.i64(Lq + 1), L.ptr(X_d), L.i64(ld), L.ptr(S),
                                     L.i64(ncols), L.ptr(scratch), L.i64(scratch.numel()), L.i32(B), L.i32(Lq), L.i32(Vp),
                                     L.i32(ncols), L.cur_stream()), "ark_token_sums16")
    torch.cuda.synchronize()
    tok = seq[:, :Lq].t().reshape(-1)            # row (t, b) -> seq[b, t]
    want = torch.full((Vp, ncols), 0.5, dtype=torch.float64)
    want.index_add_(0, tok, X[:, :ncols].double())
    assert (S.double().cpu() - want).abs().max().item() <= 2e-6 * (Lq * B / V + 1) ** 0.5 + 1e-6


@pytest.mark.parametrize("B,Z,D,H,with_ext", [(8, 10, 64, 192, False), (37, 10, 512, 1536, True), (64, 32, 128, 384, False),
                                              (16, 128, 512, 1536, False), (21, 100, 256, 768, True), (40, 64, 128, 384, False),
                                              (256, 24, 1024, 3072, False)])   # (syn-types: the dA phase as its own 2-D launch)
def test_latent_chain_bwd_matches_autograd(B, Z, D, H, with_ext):
    """ark_latent_chain_bwd (dh0 -> dz -> dhead -> dA in one launch, + bias gradient of the last MLP layer) and the
    batch reductions behind it (ark_zproj_bwd_dw; ark_latent_reduce_bwd, which fuses them) against torch autograd of
    h0 = tanh(z Wz^T + bz), z = mu + eps*exp(0.5*clamp(logv)), loss = <dh0, h0> + beta*kl_norm-scaled KL (+ <ext, head>)"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)
    head = (rn(B, 2 * Z) * 2).double()
    head[0, Z] = 11.0   # outside the clamp: no gradient through logv there
    head[1, Z + 1] = -12.0
    eps, Wz, bz = rn(B, Z).double(), (rn(D, Z) * 0.3).double(), rn(D).double()
    Whead, pre, act_in = (rn(2 * Z, H) * 0.1).double(), rn(B, H).double(), rn(B, H).double()
    dh0_up, ext = rn(B, D).double(), rn(B, 2 * Z).double()
    beta, kl_norm = 0.3, 1.0 / (B * Z)
    # torch reference: act_in -> gelu(pre stand-in) is not needed; dA is the gradient wrt `pre` of head = gelu(pre) Whead^T
    pre_t = pre.clone().requires_grad_(True)
    Wz_t, bz_t, Wh_t = Wz.clone().requires_grad_(True), bz.clone().requires_grad_(True), Whead.clone().requires_grad_(True)
    a = torch.nn.functional.gelu(pre_t)
    head_t = a @ Wh_t.t() + (head - (torch.nn.functional.gelu(pre) @ Whead.t())).detach()   # value == head, gradient flows
    mu, lv = head_t[:, :Z], head_t[:, Z:].clamp(-10, 10)
    z = mu + eps * torch.exp(0.5 * lv)
    h0 = torch.tanh(z @ Wz_t.t() + bz_t)
    kl = -0.5 * (1 + lv - mu * mu - lv.exp()).sum() * kl_norm
    loss = (dh0_up * h0).sum() + beta * kl + ((ext * head_t).sum() if with_ext else 0.0)
    loss.backward()
    f = lambda t: t.float().to(dev).contiguous()
    hyper = torch.zeros(16, device=dev)
    hyper[1], hyper[2] = beta, kl_norm    # ARK_HP_BETA, ARK_HP_KL_NORM
    dh0_d, h0_d, z_d = f(dh0_up), f(h0.detach()), f(z.detach())
    bufs = dict(Wz=f(Wz), head=f(head), eps=f(eps), ext=f(ext), Wh=f(Whead), pre=f(pre))
    dhead, dA = torch.zeros(B, 2 * Z, device=dev), torch.zeros(B, H, device=dev)
    dA16 = torch.zeros(B, H, dtype=torch.int16, device=dev)
    dbA = torch.full((H,), 0.5, device=dev)          # += semantics
    L.check(L.lib().ark_latent_chain_bwd(L.ptr(dh0_d), L.ptr(h0_d), L.ptr(bufs["Wz"]), L.ptr(bufs["head"]), L.ptr(bufs["eps"]),
                                         L.ptr(hyper), L.ptr(bufs["ext"] if with_ext else None), L.ptr(bufs["Wh"]),
                                         L.ptr(bufs["pre"]), L.ptr(dhead), L.ptr(dA), L.ptr(dA16), L.i32(L.PREC_BF16), L.ptr(dbA),
                                         L.i32(B), L.i32(Z), L.i32(D), L.i32(H), L.cur_stream()), "ark_latent_chain_bwd")
    dWz, dbz = torch.zeros(D, Z, device=dev), torch.zeros(D, device=dev)
    L.check(L.lib().ark_zproj_bwd_dw(L.ptr(dh0_d), L.ptr(z_d), L.ptr(dWz), L.ptr(dbz), L.i32(B), L.i32(Z), L.i32(D), L.i32(0),
                                     L.cur_stream()), "ark_zproj_bwd_dw")
    # the fused reduction: same dWz / dbz plus the head weight / bias gradients from dhead and a 16-bit activation copy
    act16 = torch.nn.functional.gelu(pre).to(torch.bfloat16)
    dWz2, dbz2 = torch.full((D, Z), 0.25, device=dev), torch.full((D,), 0.25, device=dev)
    dWh, dbh = torch.full((2 * Z, H), 0.25, device=dev), torch.full((2 * Z,), 0.25, device=dev)
    act16_d = act16.to(dev)
    L.check(L.lib().ark_latent_reduce_bwd(L.ptr(dh0_d), L.ptr(z_d), L.ptr(dWz2), L.ptr(dbz2), L.ptr(dhead), L.ptr(act16_d),
                                          L.i32(L.PREC_BF16), L.ptr(dWh), L.ptr(dbh), L.i32(B), L.i32(Z), L.i32(D), L.i32(H),
                                          L.cur_stream()), "ark_latent_reduce_bwd")
    torch.cuda.synchronize()
    close = lambda got, want, tol: (got.double().cpu() - want).abs().max().item() <= tol * (want.abs().max().item() + 1e-12)
    assert close(dA, pre_t.grad, 2e-5)
    assert close(dA16.view(torch.bfloat16).float(), pre_t.grad, 1e-2)
    assert close(dbA - 0.5, pre_t.grad.sum(0), 2e-5)
    assert close(dWz, Wz_t.grad, 2e-5) and close(dbz, bz_t.grad, 2e-5)
    assert close(dWz2 - 0.25, Wz_t.grad, 2e-5) and close(dbz2 - 0.25, bz_t.grad, 2e-5)
    dh = dhead.double().cpu()
    assert close(dWh - 0.25, dh.t() @ act16.double(), 2e-5) and close(dbh - 0.25, dh.sum(0), 2e-5)
    # dhead^T gelu(pre) is the head weight gradient: checks dhead itself
    assert close(dhead.double().cpu().t() @ torch.nn.functional.gelu(pre), Wh_t.grad, 2e-5)


@pytest.mark.parametrize("V", [55, 4099, 24101, 40000])
@pytest.mark.parametrize("only16", [False, True])
def test_cross_entropy_kernels_match_torch(V, only16):
    """ark_ce_fwd_bwd against F.cross_entropy(ignore_index=PAD) and its autograd: the wave-per-row kernel (small and
    very large V) and the LDS-cached persistent kernel (4096 <= V, row <= 144 KB), with the fp32 gradient or the
    16-bit copy alone as the output"""
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(V)
    B, Lq = 6, 5
    R = B * Lq
    ld = (V + 3) // 4 * 4
    Vp = (V + 63) // 64 * 64
    logits = torch.zeros(R, ld)
    logits[:, :V] = torch.randn(R, V, generator=g) * 3
    seq = torch.randint(0, V, (B, Lq + 1), generator=g)
    seq[1, 3:] = 0                     # PAD targets are ignored
    tgt = seq[:, 1:].t().reshape(-1)   # rows are time-major: row (t, b) -> seq[b, t+1]
    x = logits[:, :V].double().clone().requires_grad_(True)
    ref = F.cross_entropy(x, tgt, ignore_index=0, reduction="sum")
    count = int((tgt != 0).sum())
    (ref / count).backward()
    hyper = torch.zeros(16, device=dev)
    hyper[3], hyper[4] = 1.0 / count, count          # ARK_HP_CE_INV_COUNT, ARK_HP_CE_COUNT
    lg, sq = logits.to(dev), seq.to(dev)
    row_loss = torch.zeros(R, device=dev)
    d16 = torch.full((R, Vp), 7, dtype=torch.int16, device=dev)
    L.check(L.lib().ark_ce_fwd_bwd(L.ptr(lg), L.i64(ld), L.ptr(sq), L.i64(Lq + 1), L.ptr(hyper), L.ptr(row_loss),
                                   L.ptr(None if only16 else lg), L.ptr(d16), L.i32(L.PREC_BF16), L.i64(Vp), L.i32(B), L.i32(Lq),
                                   L.i32(V), L.cur_stream()), "ark_ce_fwd_bwd")
    torch.cuda.synchronize()
    assert abs(row_loss.double().sum().item() - ref.item()) <= 1e-5 * abs(ref.item())
    want = x.grad
    got16 = d16.view(torch.bfloat16).float().cpu().double()
    assert (got16[:, V:] == 0).all()
    assert (got16[:, :V] - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-12
    if not only16:
        assert (lg[:, :V].double().cpu() - want).abs().max().item() <= 1e-6 * want.abs().max().item() + 1e-12


def test_wgrad16_rows_guard():
    """ark_wgrad16_rows: A is column-padded to a tile multiple, C has only m_valid rows -- rows beyond are not touched"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    K, M, mv, N = 1024, 128, 100, 128
    A = torch.randn(K, M, generator=g)
    A[:, mv:] = 0
    X = torch.randn(K, N, generator=g)
    Ad, Xd = A.to(dev).to(torch.bfloat16), X.to(dev).to(torch.bfloat16)
    C = torch.full((M, N), 5.0, device=dev)          # rows >= mv are "somebody else's memory"
    L.check(L.lib().ark_wgrad16_rows(L.i32(L.PREC_BF16), L.ptr(Ad), L.i64(M), L.ptr(Xd), L.i64(N), L.ptr(C), L.i64(N), L.i32(M),
                                     L.i32(mv), L.i32(N), L.i32(K), None, L.cur_stream()), "ark_wgrad16_rows")
    torch.cuda.synchronize()
    ref = 5.0 + Ad.float().t().double().cpu() @ Xd.float().double().cpu()
    got = C.double().cpu()
    assert torch.equal(got[mv:], torch.full((M - mv, N), 5.0, dtype=torch.float64))
    assert (got[:mv] - ref[:mv]).abs().max().item() <= 1e-4 * ref.abs().max().item()


def _tile_native_index(rows, D):
    r = torch.arange(rows).view(-1, 1)
    c = torch.arange(D).view(1, -1)
    return ((r >> 4) * (D >> 4) + (c >> 4)) * 256 + ((((r >> 2) & 3) << 4) + (c & 15)) * 4 + (r & 3)


@pytest.mark.parametrize("prec,tol", [(L.PREC_F16, 3e-3), (L.PREC_BF16, 2e-2)])
@pytest.mark.parametrize("B,Lq,V,D", [(16, 5, 777, 128), (16, 9, 4099, 512), (32, 7, 24101, 128), (16, 3, 60943, 512), (48, 1, 64, 64),
                                       (16, 6, 2000, 256)])
def test_fused_vocabulary_cross_entropy(prec, tol, B, Lq, V, D):
    """ark_vocab_ce_fwd / ark_vocab_ce_dw (logits never materialised) against torch in fp64 on the same 16-bit-rounded
    operands: per-row loss, log-sum-exp, dY = dlogits W, dW = dlogits^T Y, db = colsum(dlogits), with PAD targets,
    a vocabulary that is not a multiple of the 64-token tile and a row count that is not a multiple of the 64-row tile"""
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(V + D)
    R = B * Lq
    dt = torch.float16 if prec == L.PREC_F16 else torch.bfloat16
    Y = (torch.randn(R, D, generator=g) * 0.5).to(dt)
    W = (torch.randn(V, D, generator=g) * (2.0 / D ** 0.5)).to(dt)
    bias = torch.randn(V, generator=g) * 0.3
    seq = torch.randint(1, V, (B, Lq + 1), generator=g)
    seq[1, 2:] = 0                       # PAD targets are ignored
    seq[0, 1] = V - 1                    # a target in the last (partial) tile
    tgt = seq[:, 1:].t().reshape(-1)     # rows are time-major: row (t, b) -> seq[b, t+1]
    count = int((tgt != 0).sum())
    Yd, Wd = Y.double().requires_grad_(True), W.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    logits = Yd @ Wd.t() + bd
    per_row = F.cross_entropy(logits, tgt, ignore_index=0, reduction="none")
    (per_row.sum() / count).backward()
    lse_ref = torch.logsumexp(logits, dim=1).detach()
    hyper = torch.zeros(16, device=dev)
    hyper[3], hyper[4] = 1.0 / count, count          # ARK_HP_CE_INV_COUNT, ARK_HP_CE_COUNT
    Yg, Wg, bg, sq = Y.to(dev), W.to(dev), bias.to(dev), seq.to(dev)
    row_loss, lse = torch.full((R,), 9.0, device=dev), torch.full((R,), 9.0, device=dev)
    dY_t = torch.full((R * D,), 9.0, device=dev)
    common = (L.i32(prec), L.ptr(Yg), L.ptr(Wg), L.ptr(bg), L.ptr(sq), L.i64(Lq + 1), L.ptr(hyper))
    tail = (L.i32(B), L.i32(Lq), L.i32(V), L.i32(D), L.cur_stream())
    # evaluation flavour (no dY) first, then the training flavour
    L.check(L.lib().ark_vocab_ce_fwd(*common, L.ptr(row_loss), L.ptr(lse), L.ptr(None), *tail), "ark_vocab_ce_fwd")
    torch.cuda.synchronize()
    ev = row_loss.double().cpu().clone()
    L.check(L.lib().ark_vocab_ce_fwd(*common, L.ptr(row_loss), L.ptr(lse), L.ptr(dY_t), *tail), "ark_vocab_ce_fwd")
    dW = torch.full((V, D), 0.25, device=dev)         # += semantics
    db = torch.full((V,), 0.5, device=dev)
    L.check(L.lib().ark_vocab_ce_dw(*common, L.ptr(lse), L.ptr(dW), L.ptr(db), *tail), "ark_vocab_ce_dw")
    torch.cuda.synchronize()
    got_loss = row_loss.double().cpu()
    assert (ev - got_loss).abs().max().item() <= 1e-5     # (two instantiations: same sums, different instruction order)
    assert (got_loss - per_row.detach()).abs().max().item() <= 2e-5 * per_row.max().item()
    assert (lse.double().cpu() - lse_ref).abs().max().item() <= 2e-5 * lse_ref.abs().max().item()
    assert (got_loss[tgt == 0] == 0).all()
    idx = _tile_native_index(R, D).reshape(-1)
    dY = dY_t.double().cpu()[idx].reshape(R, D)
    close = lambda got, want, t: (got - want).abs().max().item() <= t * want.abs().max().item() + 1e-12
    assert close(dY, Yd.grad, tol), (dY - Yd.grad).abs().max().item() / Yd.grad.abs().max().item()
    assert close(dW.double().cpu() - 0.25, Wd.grad, tol), ((dW.double().cpu() - 0.25) - Wd.grad).abs().max().item() / Wd.grad.abs().max().item()
    assert close(db.double().cpu() - 0.5, bd.grad, tol)
    # few rows x a wide model: the vocabulary-split forward (several workgroups per row block + a merging launch)
    nv0 = L.lib().ark_vocab_ce_fwd_splits(L.i32(R), L.i32(V), L.i32(D), L.i32(0))
    assert (nv0 > 1) == (V > 4096)
    # (cu_budget: the CUs a persistent sweep beside the launch leaves free -- fewer, larger splits, the same result)
    for budget in ((0, 160) if nv0 > 1 else ()):
        nv = L.lib().ark_vocab_ce_fwd_splits(L.i32(R), L.i32(V), L.i32(D), L.i32(budget))
        assert 1 <= nv <= nv0 and ((R + 63) // 64) * nv <= max(budget or 256, (R + 63) // 64)
        ws = torch.full((max(nv, 2) * (R * D + 4 * R),), float("nan"), device=dev)
        rl2, lse2, dY2 = torch.full((R,), 9.0, device=dev), torch.full((R,), 9.0, device=dev), torch.full((R * D,), 9.0, device=dev)
        L.check(L.lib().ark_vocab_ce_fwd_ws(*common, L.ptr(rl2), L.ptr(lse2), L.ptr(dY2), L.ptr(ws), L.i64(ws.numel()), *tail[:-1],
                                            L.i32(budget), tail[-1]), "ark_vocab_ce_fwd_ws")
        torch.cuda.synchronize()
        assert (rl2.double().cpu() - per_row.detach()).abs().max().item() <= 2e-5 * per_row.max().item()
        assert (lse2.double().cpu() - lse_ref).abs().max().item() <= 2e-5 * lse_ref.abs().max().item()
        assert (rl2.cpu()[tgt == 0] == 0).all()
        d2 = dY2.double().cpu()[idx].reshape(R, D)
        assert close(d2, Yd.grad, tol), (d2 - Yd.grad).abs().max().item() / Yd.grad.abs().max().item()


@pytest.mark.parametrize("B,nv,Z,D,n", [(32, 32, 10, 512, 3), (16, 7, 64, 384, 2), (48, 48, 128, 320, 1)])
def test_latent_zproj_fwd_and_rowwise_kl(B, nv, Z, D, n):
    """ark_latent_zproj_fwd (reparameterisation + z-projection in one launch) against torch: mu, clamped logv, z, the KL
    through ark_loss_finalize_rows, h0 = tanh(z Wz^T + bz) in the row-major, tile-native and 16-bit layouts of every
    layer; rows >= n_valid (padding of a ragged batch) get z = 0, are not written and carry no KL term"""
    import ctypes
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + Z)
    head = torch.randn(B, 2 * Z, generator=g) * 3.0
    head[0, Z] = 14.0          # clamped high
    head[1, Z + 1] = -12.0     # clamped low
    eps = torch.randn(B, Z, generator=g)
    Wz = torch.randn(D, Z, generator=g) * 0.3
    bz = torch.randn(D, generator=g) * 0.1
    mu_ref = head[:nv, :Z].double()
    lv_ref = head[:nv, Z:].double().clamp(-10, 10)
    z_ref = mu_ref + eps[:nv].double() * (0.5 * lv_ref).exp()
    kl_ref = (-0.5 * (1 + lv_ref - mu_ref ** 2 - lv_ref.exp())).mean().item()
    z_full = torch.zeros(B, Z, dtype=torch.float64)
    z_full[:nv] = z_ref
    h_ref = torch.tanh(z_full @ Wz.double().t() + bz.double())
    t = lambda x: x.to(dev)
    mu, logv, z = (torch.full((B, Z), 7.0, device=dev) for _ in range(3))
    kl_rows = torch.full((B,), 7.0, device=dev)
    h0 = torch.zeros(B, D, device=dev)
    yt = [torch.zeros(2 * B * D, device=dev) for _ in range(n)]
    ya = [torch.zeros(2 * B * D, device=dev, dtype=torch.float16) for _ in range(n)]
    yb = [torch.zeros(2 * B * D, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    arr = lambda ts: (ctypes.c_void_p * n)(*[x.data_ptr() for x in ts])
    hd, ed, Wd, bd = t(head), t(eps), t(Wz), t(bz)
    L.check(L.lib().ark_latent_zproj_fwd(L.i32(L.PREC_F16), L.i32(L.PREC_BF16), L.ptr(hd), L.ptr(ed), L.ptr(mu), L.ptr(logv),
                                         L.ptr(z), L.ptr(kl_rows), L.ptr(Wd), L.ptr(bd), L.ptr(h0), L.i32(n), arr(yt), arr(ya),
                                         arr(yb), L.i32(B), L.i32(nv), L.i32(Z), L.i32(D), L.cur_stream()), "ark_latent_zproj_fwd")
    row_loss = torch.tensor([1.0, 2.0, 3.5], device=dev)
    hyper = torch.zeros(16, device=dev)
    hyper[3], hyper[1] = 0.5, 0.25        # ARK_HP_CE_INV_COUNT, ARK_HP_BETA
    out4 = torch.zeros(4, device=dev)
    L.check(L.lib().ark_loss_finalize_rows(L.ptr(row_loss), L.i32(3), L.ptr(kl_rows), L.i32(nv), L.f32(-0.5 / (nv * Z)),
                                           L.ptr(hyper), L.ptr(out4), L.cur_stream()), "ark_loss_finalize_rows")
    torch.cuda.synchronize()
    assert (mu[:nv].double().cpu() - mu_ref).abs().max().item() == 0
    assert (logv[:nv].double().cpu() - lv_ref).abs().max().item() <= 1e-6
    assert (z[:nv].double().cpu() - z_ref).abs().max().item() <= 1e-4 * z_ref.abs().max().item()
    assert (mu[nv:] == 7.0).all() and (z[nv:] == 7.0).all() and (kl_rows[nv:] == 7.0).all()      # padding rows: untouched
    o = out4.cpu().double()
    assert abs(o[2].item() - kl_ref) <= 1e-5 * abs(kl_ref)
    assert abs(o[1].item() - 3.25) < 1e-6 and abs(o[0].item() - (3.25 + 0.25 * o[2].item())) < 1e-5
    assert (h0.double().cpu() - h_ref).abs().max().item() <= 2e-5
    idx = _tile_native_index(B, D).reshape(-1)
    for l in range(n):
        assert torch.equal(yt[l][:B * D].cpu()[idx].reshape(B, D), h0.cpu())
        assert torch.equal(ya[l][:B * D].cpu().reshape(B, D), h0.cpu().to(torch.float16))
        assert torch.equal(yb[l][:B * D].cpu().reshape(B, D), h0.cpu().to(torch.bfloat16))


