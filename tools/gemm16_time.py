"""Same-process A/B of the ark_gemm16 engines (1 = shared ring, 2 = wave-private K-slices) at the encoder shapes:
graph-captured back-to-back launches on rotating buffers, event-timed.  python tools/gemm16_time.py [M N K]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_amd import _lib as L

_a = [x for x in sys.argv[1:] if not x.startswith("--")]
M, N, K = (int(x) for x in _a[:3]) if len(_a) >= 3 else (1024, 1536, 1536)
dev = torch.device("cuda:0")
NB = 8   # rotating operand sets (the step never re-runs a product on cache-hot operands)
A = [torch.randn(M, K, device=dev).half() for _ in range(NB)]
B = [torch.randn(N, K, device=dev).half() for _ in range(NB)]
C = [torch.zeros(M, N, device=dev) for _ in range(NB)]
c16a = [torch.zeros(M, N, device=dev, dtype=torch.float16) for _ in range(NB)]
c16b = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
bias = torch.randn(N, device=dev)
aux = torch.randn(M, N, device=dev)
cs = torch.zeros(N, device=dev)


def launch(engine, epi, i):
    rc = L.lib().ark_gemm16_engine(L.i32(engine), L.i32(L.PREC_F16), L.i32(epi), L.ptr(A[i]), L.i64(K), L.ptr(B[i]), L.i64(K),
                                   L.ptr(C[i]), L.i64(N), L.ptr(bias), L.ptr(aux), L.ptr(c16a[i]), L.ptr(c16b[i]),
                                   L.i32(L.PREC_BF16), L.ptr(cs if epi == L.EPI_MUL_DGELU else None), L.i32(M), L.i32(N), L.i32(K),
                                   L.cur_stream())
    L.check(rc, "ark_gemm16_engine")


def timed(engine, epi, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(NB):
            launch(engine, epi, i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(NB):
                launch(engine, epi, i)
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * NB)


if "--eager" in sys.argv:   # for rocprofv3 --pmc passes: plain launches, both engines, rotating operands
    for rnd in range(3):
        for engine in (1, 2):
            for i in range(NB):
                launch(engine, L.EPI_BIAS_GELU, i)
    torch.cuda.synchronize()
    sys.exit(0)

for rnd in range(3):
    for epi, name in ((L.EPI_BIAS_GELU, "bias+gelu"), (L.EPI_MUL_DGELU, "mul dgelu + colsum"), (L.EPI_NONE, "none")):
        t1, t2 = timed(1, epi), timed(2, epi)
        print(f"round {rnd} [{M}x{N}x{K}] {name:20s} ring {t1:6.2f} us   wpk {t2:6.2f} us   ({2.0 * M * N * K / t2 * 1e-6:.0f} TFLOP/s)", flush=True)
