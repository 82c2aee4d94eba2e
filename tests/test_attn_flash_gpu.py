"""Matrix-core (flash-style) self-attention, csrc/attn_mfma.hip, through the C-ABI: against torch in fp64 (forward and
backward, causal / key-padding masks, every head width the kernels are instantiated for, ragged last tiles), against the
vector-unit kernels of txf.hip under the SAME dropout seed (the two paths must draw identical masks), and inside the
Transformer engines (t-ARK / t-SAIL train steps with `ark_txf_flash` on and off).
Reference op on both sides: F.scaled_dot_product_attention inside the stock Transformer layers,
kgvae/model/models.py:73-74, 104-105, 355-356."""
import math

import pytest
import torch

from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu

F16, BF16 = 2, 1   # ARK_PREC_F16 / ARK_PREC_BF16 (include/ark_amd.h)


def _call(name, *a):
    from ark_amd import _lib as L
    L.check(getattr(L.lib(), name)(*a), name)


def _stat(B, Lq, H):
    return torch.full((B * H * ((Lq + 63) // 64 * 64),), float("nan"), device="cuda")   # (the padded tail must never matter)


def _flash(prec_f, prec_b, qkv, dout, km, B, Lq, D, H, causal, p=0.0, seed=0, hyper=None):
    from ark_amd import _lib as L
    out, lse, delta = torch.empty(Lq * B, D, device="cuda"), _stat(B, Lq, H), _stat(B, Lq, H)
    dqkv = torch.full_like(qkv, float("nan"))
    _call("ark_attn_flash_fwd", L.i32(prec_f), L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(km), L.i32(B), L.i32(Lq), L.i32(D), L.i32(H),
          L.i32(causal), L.f32(p), L.u64(seed), L.ptr(hyper), L.cur_stream())
    _call("ark_attn_flash_bwd", L.i32(prec_b), L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(dout), L.ptr(delta), L.ptr(dqkv), L.ptr(km),
          L.i32(B), L.i32(Lq), L.i32(D), L.i32(H), L.i32(causal), L.f32(p), L.u64(seed), L.ptr(hyper), L.cur_stream())
    return out, dqkv, lse


def _inputs(B, Lq, D, masked, seed=1):
    torch.manual_seed(seed)
    qkv = torch.randn(Lq * B, 3 * D, device="cuda")
    dout = torch.randn(Lq * B, D, device="cuda")
    km = None
    if masked:   # 1 = key may be attended to; every batch keeps a ragged prefix of its keys plus a hole
        km = torch.zeros(B, Lq, dtype=torch.uint8, device="cuda")
        for b in range(B):
            km[b, : max(3, Lq - 7 * b - 5)] = 1
            km[b, 1] = 0
    return qkv, dout, km


@pytest.mark.parametrize("B,Lq,D,H,causal,masked", [
    (2, 70, 128, 4, 1, False),     # dh = 32: wd-movies decoder (zero-padded half image)
    (2, 130, 512, 4, 1, False),    # dh = 128: the wd-articles / syn decoders; three query tiles, ragged last one
    (2, 100, 384, 4, 0, True),     # dh = 96: wd-movies encoder width 3 * 128, key-padding mask
    (1, 65, 1536, 4, 0, True),     # dh = 384: the encoder of d_model 512 (six k-images)
    (3, 17, 256, 4, 1, False),     # dh = 64, the shortest sequence that takes this path
    (1, 212, 1024, 4, 0, True),    # dh = 256
    (1, 637, 512, 4, 1, False),    # the longest decoder sequence of BASELINE.json (wd-articles)
    (2, 64, 640, 4, 1, False),     # dh = 160: five 32-wide k-steps, three k-images; exactly one tile
])
@pytest.mark.parametrize("prec", ["mixed", "bf16"])
def test_flash_attention_matches_torch(B, Lq, D, H, causal, masked, prec):
    dh = D // H
    qkv, dout, km = _inputs(B, Lq, D, masked)
    pf, pb = (F16, BF16) if prec == "mixed" else (BF16, BF16)
    out, dqkv, lse = _flash(pf, pb, qkv, dout, km, B, Lq, D, H, causal)
    x = qkv.double().view(Lq, B, 3, H, dh).requires_grad_(True)
    q, k, v = (x[:, :, i].permute(1, 2, 0, 3) for i in range(3))          # [B, H, L, dh]
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(Lq, Lq, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    if km is not None:
        sc = sc.masked_fill((km == 0)[:, None, None, :], float("-inf"))
    pr = torch.softmax(sc, -1)
    o = (pr @ v).permute(2, 0, 1, 3).reshape(Lq * B, D)
    o.backward(dout.double())
    want_lse = torch.logsumexp(sc, -1) / math.log(2.0)                     # the kernels keep it in the log2 domain
    Lp = (Lq + 63) // 64 * 64
    got_lse = lse.view(B, H, Lp)[:, :, :Lq].double()
    tol_o, tol_g = (3e-3, 1.2e-2) if prec == "mixed" else (1.2e-2, 1.5e-2)
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    assert (got_lse - want_lse).abs().max().item() < (2e-2 if prec == "mixed" else 8e-2)
    assert rel_err_t(out.double(), o) < tol_o, rel_err_t(out.double(), o)
    want = x.grad.reshape(Lq * B, 3 * D)
    for i, nm in enumerate("qkv"):
        a, w = dqkv[:, i * D:(i + 1) * D].double(), want[:, i * D:(i + 1) * D]
        assert rel_err_t(a, w) < tol_g, (nm, rel_err_t(a, w))
    if km is not None:   # a masked key receives no gradient at all
        dead = (km == 0).t().reshape(-1)                                   # rows (t, b) -> key t of batch b
        assert dqkv[dead][:, D:].abs().max().item() == 0.0


def rel_err_t(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("prec", ["mixed", "bf16"])
def test_flash_attention_with_an_all_masked_graph(prec):
    """a batch row whose keys are ALL masked (an all-PAD graph in the encoder): context 0, gradients 0, nothing NaN -- for
    that row and, above all, for the others (round 4: out = 0 * inf, exp2(-inf - -inf) in the backward)"""
    B, Lq, D, H = 3, 70, 384, 4
    qkv, dout, km = _inputs(B, Lq, D, True)
    km[1, :] = 0                                        # graph 1: every key masked
    pf, pb = (F16, BF16) if prec == "mixed" else (BF16, BF16)
    out, dqkv, lse = _flash(pf, pb, qkv, dout, km, B, Lq, D, H, 0)
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    rows1 = torch.arange(Lq, device="cuda") * B + 1     # rows (t, b = 1)
    assert out[rows1].abs().max().item() == 0.0 and dqkv[rows1].abs().max().item() == 0.0
    # the other graphs are what they are without graph 1
    keep = torch.tensor([0, 2], device="cuda")
    sel = (torch.arange(Lq, device="cuda")[:, None] * B + keep[None, :]).reshape(-1)
    out2, dqkv2, _ = _flash(pf, pb, qkv[sel].contiguous(), dout[sel].contiguous(), km[keep].contiguous(), 2, Lq, D, H, 0)
    assert (out[sel] - out2).abs().max().item() < 1e-5 and (dqkv[sel] - dqkv2).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,Lq,D,H,causal,masked", [(2, 70, 128, 4, 1, False), (2, 130, 512, 4, 1, False), (2, 100, 384, 4, 0, True),
                                                    (1, 259, 512, 4, 1, False)])
def test_flash_attention_draws_the_masks_of_the_vector_kernels(B, Lq, D, H, causal, masked):
    """dropout 0.1 on the probabilities: same seed, same step counter -> the same keep decisions in both implementations (a
    different mask would move the context by O(1), not by rounding).  The sequence lengths are NOT multiples of four, so a
    query's row of the virtual [B, H, L, L] array starts inside a hash quad"""
    from ark_amd import _lib as L
    qkv, dout, km = _inputs(B, Lq, D, masked, seed=3)
    hyper = torch.zeros(64, device="cuda")
    seed, p = 0x1234_5678_9ABC, 0.1
    out_v, probs = torch.empty(Lq * B, D, device="cuda"), torch.empty(B * H * Lq * Lq, device="cuda")
    dsc, dq_v = torch.empty_like(probs), torch.empty_like(qkv)
    _call("ark_attn_fwd", L.ptr(qkv), L.ptr(out_v), L.ptr(probs), L.ptr(km), L.i32(B), L.i32(Lq), L.i32(D), L.i32(H), L.i32(causal),
          L.f32(p), L.u64(seed), L.ptr(hyper), L.cur_stream())
    _call("ark_attn_bwd", L.ptr(qkv), L.ptr(out_v), L.ptr(probs), L.ptr(dout), L.ptr(dsc), L.ptr(dq_v), L.ptr(km), L.i32(B), L.i32(Lq),
          L.i32(D), L.i32(H), L.i32(causal), L.f32(p), L.u64(seed), L.ptr(hyper), L.cur_stream())
    out_f, dq_f, _ = _flash(F16, BF16, qkv, dout, km, B, Lq, D, H, causal, p, seed, hyper)
    assert rel_err_t(out_f.double(), out_v.double()) < 3e-3
    assert rel_err_t(dq_f.double(), dq_v.double()) < 1.5e-2
    out_0, _, _ = _flash(F16, BF16, qkv, dout, km, B, Lq, D, H, causal)   # (and dropout does something)
    assert rel_err_t(out_0.double(), out_v.double()) > 0.1


def test_flash_attention_refuses_head_widths_it_has_no_kernel_for():
    from ark_amd import _lib as L
    qkv = torch.zeros(40 * 3 * 80, device="cuda")
    out, lse = torch.zeros(40 * 80, device="cuda"), _stat(1, 40, 4)
    for D in (80, 2048):   # dh = 20 (not a multiple of 32), dh = 512 (> 384)
        rc = L.lib().ark_attn_flash_fwd(L.i32(F16), L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(None), L.i32(1), L.i32(1), L.i32(D),
                                        L.i32(4), L.i32(1), L.f32(0.0), L.u64(0), L.ptr(None), L.cur_stream())
        assert rc < 0


@pytest.mark.parametrize("mt,B,T,D", [("t-ARK", 6, 13, 128), ("t-SAIL", 4, 30, 128)])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_transformer_train_step_with_and_without_flash_attention(mt, B, T, D, drop):
    """the engines with `ark_txf_flash` on (matrix-core attention wherever a sequence is longer than 16 positions: the decoder
    AND, for t-SAIL, the encoder over the triples at width 3 D -> dh = 96) against the same engines on the vector-unit
    kernels: loss and every gradient, dropout on (same masks) and off"""
    from oracle import sail_oracle as O
    from tests.test_configs_gpu import _cfg
    from tests.parity_util import synth_batch
    from ark_amd.txf_engine import TxfEngine
    cfg = dict(_cfg(D, 16, 300, 3, T, True), model_type=mt, dec_dropout=drop, ark_txf_dropout=drop)
    seq_len = cfg["seq_len"]
    P = O.init_params(cfg, 5)
    triples, seq = synth_batch(cfg, B, seed=11, padded=True)
    outs, grads = [], []
    for flash in (0, 1):
        eng = TxfEngine(dict(cfg, ark_txf_flash=flash), torch.device("cuda:0"), precision="mixed")
        eng.load_params(P)
        eng.set_hyper(lr=1e-3, beta=0.3)
        eng.drop_seed = 77
        tri = triples.cuda() if mt == "t-SAIL" else None
        out = eng.train_step(tri, seq.cuda()).cpu().numpy().copy()
        torch.cuda.synchronize()
        Lq = seq_len - 1
        assert eng._flash_ok(D, Lq) == bool(flash)
        if mt == "t-SAIL":
            assert eng._flash_ok(3 * D, triples.shape[1]) == bool(flash)
        outs.append(out)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 2e-4, outs
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert (a - b).norm().item() <= 2.5e-2 * a.norm().item() + 1e-7, (k, (a - b).norm().item(), a.norm().item())


@pytest.mark.parametrize("mt,B,T", [("t-ARK", 8, 13), ("t-SAIL", 16, 7)])
def test_transformer_fused_vocabulary_ce(mt, B, T):
    """V >= 2048: the output projection fused with the cross-entropy (no [B*L, V] logits; vocab_ce.hip, vocabulary-split
    forward for the few rows of these batches) against the logits path of the same engine: loss, the gradient of the output
    layer and every gradient below it; the logits API still works afterwards (lazily allocated buffer)"""
    from oracle import sail_oracle as O
    from tests.test_configs_gpu import _cfg
    from tests.parity_util import synth_batch
    from ark_amd.txf_engine import TxfEngine
    cfg = dict(_cfg(128, 16, 3000, 3, T, True), model_type=mt, dec_dropout=0.0, ark_txf_dropout=0.0)
    assert cfg["vocab_size"] >= 2048
    P = O.init_params(cfg, 6)
    triples, seq = synth_batch(cfg, B, seed=12, padded=True)
    outs, grads, evs = [], [], []
    for fused in (0, 1):
        eng = TxfEngine(dict(cfg, ark_fused_ce=fused), torch.device("cuda:0"), precision="mixed")
        eng.load_params(P)
        eng.set_hyper(lr=1e-3, beta=0.3)
        tri = triples.cuda() if mt == "t-SAIL" else None
        out = eng.train_step(tri, seq.cuda()).cpu().numpy().copy()
        torch.cuda.synchronize()
        assert eng._fused_step == bool(fused) and ("logits" in eng.ws) == (not fused)
        outs.append(out)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
        evs.append(eng.eval_loss(tri, seq.cuda()).cpu().numpy().copy())   # (no dY: the forward-only form of the kernel)
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 2e-4, outs
    assert rel_err(float(evs[1][0]), float(evs[0][0])) < 2e-3, evs           # (after one Adam step of each engine)
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert (a - b).norm().item() <= 3e-2 * a.norm().item() + 1e-7, (k, (a - b).norm().item(), a.norm().item())


def test_prefix_logits_with_flash_attention():
    """generation re-runs the prefix per token (TxfEngine.prefix_logits, ONE decode workspace at full length whose layers
    carry the log-sum-exp buffers): prefixes shorter and longer than 16 positions through the matrix-core kernels agree with
    the vector-unit kernels (mixed precision, logits of the next position)"""
    from oracle import sail_oracle as O
    from tests.test_configs_gpu import _cfg
    from tests.parity_util import synth_batch
    from ark_amd.txf_engine import TxfEngine
    cfg = dict(_cfg(128, 16, 60, 3, 13, True), model_type="t-ARK", dec_dropout=0.0)
    P = O.init_params(cfg, 7)
    _, seq = synth_batch(cfg, 5, seed=13, padded=True)
    got = []
    for flash in (0, 1):
        eng = TxfEngine(dict(cfg, ark_txf_flash=flash), torch.device("cuda:0"), precision="mixed")
        eng.load_params(P)
        got.append([eng.prefix_logits(seq[:, :t].cuda()).float().cpu().clone() for t in (3, 16, 17, 40)])
    for a, b in zip(*got):
        assert torch.isfinite(b).all()
        assert (a - b).abs().max().item() < 2e-2 * (a.abs().max().item() + 1.0), (a - b).abs().max().item()


@pytest.mark.parametrize("mt,B,T", [("t-ARK", 6, 13), ("t-SAIL", 5, 30)])
def test_row_counts_that_are_not_multiples_of_64_take_the_fast_products(mt, B, T):
    """rows = B * L = 240 (t-ARK) / 455 and 150 (t-SAIL decoder / encoder): the weight gradients walk the rows as their K
    dimension in stages of 64 -- the 16-bit operand copies are zero-padded -- against the register-staged products of the same
    engine (`ark_txf_fast_gemm: 0`: the same 16-bit operand values, formed on the fly); dropout on, so the fused
    dropout / column-sum / cast pass (`ark_prep16`) feeds the padded products too"""
    from oracle import sail_oracle as O
    from tests.test_configs_gpu import _cfg
    from tests.parity_util import synth_batch
    from ark_amd.txf_engine import TxfEngine
    cfg = dict(_cfg(128, 16, 300, 3, T, True), model_type=mt, dec_dropout=0.1, ark_txf_dropout=0.1)
    P = O.init_params(cfg, 8)
    triples, seq = synth_batch(cfg, B, seed=14, padded=True)
    assert (B * (cfg["seq_len"] - 1)) % 64 != 0
    outs, grads = [], []
    for fast in (0, 1):
        eng = TxfEngine(dict(cfg, ark_txf_fast_gemm=fast), torch.device("cuda:0"), precision="mixed")
        eng.load_params(P)
        eng.set_hyper(lr=1e-3, beta=0.3)
        eng.drop_seed = 21
        tri = triples.cuda() if mt == "t-SAIL" else None
        outs.append(eng.train_step(tri, seq.cuda()).cpu().numpy().copy())
        torch.cuda.synchronize()
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 5e-5, outs
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.isfinite(b).all(), k
        assert (a - b).norm().item() <= 5e-3 * a.norm().item() + 1e-7, (k, (a - b).norm().item(), a.norm().item())


@pytest.mark.parametrize("mt", ["t-SAIL", "t-ARK"])
def test_wd_movies_shape_through_the_train_entry_point(tmp_path, mt):
    """`python -m kgvae.experiments.train` on the wd-movies YAML (V = 24 101, 70 decoder positions, padded graphs) with a
    Transformer model type: flash attention in decoder (and encoder), fused vocabulary CE, captured steps, validation and
    the posterior-bits log of the loop -- two epochs, finite parameters, a falling training loss"""
    import os
    import yaml
    from kgvae.experiments import train as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "sail_wd-movies.yaml")))
    cfg.update(model_type=mt, n_layers=2, num_epochs=2, batch_size=32, save_every=2, compression_log_every=1, verify_every=100,
               learning_rate=1e-3, precision="mixed", synthetic_sizes={"n_train": 128, "n_val": 32, "n_test": 32},
               dump_final_params=str(tmp_path / "P"))
    cpath = tmp_path / "c.yaml"
    yaml.safe_dump(cfg, open(cpath, "w"))
    T.main(["--config", str(cpath), "--checkpoint-dir", str(tmp_path / "ck")])
    assert torch.isfinite(torch.load(str(tmp_path / "P.rank0.pt"), weights_only=True)).all()
