"""Time ark_colsum16 (rocprofv3 --kernel-trace --stats gives the kernel time): python tools/colsum_time.py [M N ld] (GPU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_amd import _lib as L
dev = torch.device("cuda:0")
M, N, ld = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (10240, 45, 64)
Xs = [torch.randn(M, ld, device=dev).to(torch.bfloat16) for _ in range(8)]
out = torch.zeros(N, device=dev)
def run(i): L.check(L.lib().ark_colsum16(L.i32(L.PREC_BF16), L.ptr(Xs[i % 8]), L.i64(ld), L.ptr(out), L.i32(M), L.i32(N), L.i32(1), L.cur_stream()), "colsum16")
for i in range(88): run(i)
torch.cuda.synchronize()
print("done")
