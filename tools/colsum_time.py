"""Time ark_colsum16 on the gate-gradient panel of syn-paths (10 240 x 2 048, 16-bit): python tools/colsum_time.py (GPU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_amd import _lib as L
dev = torch.device("cuda:0")
M, N = 10240, 2048
Xs = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(8)]
out = torch.zeros(N, device=dev)
def run(i): L.check(L.lib().ark_colsum16(L.i32(L.PREC_BF16), L.ptr(Xs[i % 8]), L.i64(N), L.ptr(out), L.i32(M), L.i32(N), L.i32(1), L.cur_stream()), "colsum16")
for i in range(8): run(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(80): run(i)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 80 * 1e3
print(f"colsum16 {M} x {N}: {us:.2f} us = {M * N * 2 / us / 1e6:.2f} TB/s")
