"""Where a launch of the wave-private K-slice product goes (diagnostic build):

    bash tools/build_variant.sh wpkstamps gemm16.hip -DARK_STAMPS
    ARK_AMD_LIB=$PWD/ark_amd/lib/variants/wpkstamps/libark_amd.so python tools/wpk_stamps.py [M N K] [epi]

Per wave, 100-MHz real-time stamps: entry, ring primed, first slice landed, main loop done, rings free (barrier), tiles
summed, stores drained.  Prints medians / 90th percentiles over the waves of one launch on cold operands, relative to the
launch's earliest entry, in us."""
import ctypes
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_amd import _lib as L

warm = "--warm" in sys.argv   # no cache flush between the launches: operands L2-resident from the previous launch
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M, N, K = (int(x) for x in args[:3]) if len(args) >= 3 else (1024, 1536, 1536)
epi = int(args[3]) if len(args) >= 4 else L.EPI_BIAS_GELU
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).half()
B = torch.randn(N, K, device=dev).half()
C = torch.zeros(M, N, device=dev)
c16a = torch.zeros(M, N, device=dev, dtype=torch.float16)
c16b = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
aux = torch.randn(M, N, device=dev)
cs = torch.zeros(N, device=dev)
flush = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
NWAVES = int(os.environ.get("WPK_WAVES", "8"))
names = ["primed", "first slice", "loop done", "rings free", "summed", "stored"]
for rnd in range(4):
    if not warm:
        flush.add_(1)   # cold caches: 512 MiB written elsewhere
    torch.cuda.synchronize()
    rc = L.lib().ark_gemm16_engine(L.i32(2), L.i32(L.PREC_F16), L.i32(epi), L.ptr(A), L.i64(K), L.ptr(B), L.i64(K), L.ptr(C),
                                   L.i64(N), L.ptr(bias), L.ptr(aux), L.ptr(c16a), L.ptr(c16b), L.i32(L.PREC_BF16),
                                   L.ptr(cs if epi == L.EPI_MUL_DGELU else None), L.i32(M), L.i32(N), L.i32(K), L.cur_stream())
    L.check(rc, "ark_gemm16_engine")
    torch.cuda.synchronize()
    nb = (M // 64) * (N // 96)
    host = (ctypes.c_ulonglong * (64 * nb))()
    L.check(L.lib().ark_debug_wpk_stamps(host, L.i32(nb)), "stamps")
    s = np.array(host, dtype=np.float64).reshape(nb, 8, 8)[:, :NWAVES, :7] * 0.01   # us
    t0 = s[:, :, 0].min()
    rel = s - t0
    print(f"round {rnd}: entry spread {rel[:, :, 0].max():.2f} us; launch span {rel[:, :, 6].max():.2f} us")
    for i, n in enumerate(names, start=1):
        d = (s[:, :, i] - s[:, :, i - 1]).ravel()
        print(f"   {n:12s} +{np.median(d):6.2f} (p90 {np.percentile(d, 90):6.2f})   at {np.median(rel[:, :, i]):6.2f} (max {rel[:, :, i].max():6.2f})")
