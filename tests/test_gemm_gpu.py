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
@pytest.mark.parametrize("tile,nbuf", [(64, 4), (64, 2), (128, 3), (128, 2)])
@pytest.mark.parametrize("shape", [(64, 64, 64), (192, 128, 640), (1536, 512, 2048), (128, 256, 10240)])
def test_wgrad16_tr_read_kernel(prec, tile, nbuf, shape):
    """C += A^T B with both operands reduction-major 16-bit (LDS-DMA + ds_read_b64_tr_b16)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16 if prec == L.PREC_BF16 else torch.float16
    A = torch.randn(K, M, generator=g).to(dt)
    B = torch.randn(K, N, generator=g).to(dt)
    C0 = torch.randn(M, N, generator=g)
    dev = torch.device("cuda:0")
    Ad, Bd, C = A.to(dev), B.to(dev), C0.to(dev).clone()
    L.check(L.lib().ark_set_wgrad16_tuning(L.i32(tile), L.i32(nbuf), L.i32(512)), "tune")
    L.check(L.lib().ark_wgrad16(L.i32(prec), L.ptr(Ad), L.i64(M), L.ptr(Bd), L.i64(N), L.ptr(C), L.i64(N), L.i32(M), L.i32(N),
                                L.i32(K), L.cur_stream()), "ark_wgrad16")
    torch.cuda.synchronize()
    L.check(L.lib().ark_set_wgrad16_tuning(L.i32(128), L.i32(2), L.i32(96)), "tune")
    ref = C0.double() + A.double().t() @ B.double()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 3e-5 * (K ** 0.5) + 1e-4, err
