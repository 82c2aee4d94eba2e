"""Debug aid (GPU): time the LDS-DMA forward GRU cell with parts ablated."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_amd import _lib as L
B, D = 1024, 512
dev = torch.device("cuda:0")
f = lambda *s: torch.randn(*s, device=dev)
hp, gi, ho, bh = f(B, D) * 0.1, f(B, 3 * D), f(B, D), f(3 * D)
hp16, ho16 = hp.half(), torch.empty(B, D, device=dev, dtype=torch.float16)
w16 = (f(3 * D, D) * 0.04).half()
sv = [f(B, D) for _ in range(4)]
st = torch.cuda.Stream()
def launch():
    L.check(L.lib().ark_gru_cell_fwd_dma(L.i32(2), L.ptr(hp16), L.ptr(w16), L.ptr(hp), L.ptr(bh), L.ptr(gi), L.ptr(ho), L.ptr(ho16),
            L.ptr(None), L.ptr(None), L.ptr(sv[0]), L.ptr(sv[1]), L.ptr(sv[2]), L.ptr(sv[3]), L.i32(B), L.i32(D), L.cur_stream()), "cell")
with torch.cuda.stream(st):
    for mask in (0, 1, 2, 4, 6, 7, 3):
        L.lib().ark_set_dma_debug(mask)
        for _ in range(3): launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): launch()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); e1.synchronize()
        print(f"dbg mask {mask}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us/launch", flush=True)
