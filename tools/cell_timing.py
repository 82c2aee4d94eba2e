"""Debug aid (GPU): time the LDS-DMA forward GRU cell with parts ablated, GPU-bound (graph replay)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_amd import _lib as L
B, D = 1024, 512
dev = torch.device("cuda:0")
f = lambda *s: torch.randn(*s, device=dev)
NB = 12   # rotate over distinct buffers so the data is not L2-hot (as in the real step)
yp = [f(B, D) * 0.1 for _ in range(NB)]; gi = [f(B, 3 * D) for _ in range(NB)]; yo = [f(B, D) for _ in range(NB)]
bh = f(3 * D)
h16 = [x.half() for x in yp]
o16a = [torch.empty(B, D, device=dev, dtype=torch.float16) for _ in range(NB)]
o16b = [torch.empty(B, D, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
w16 = (f(3 * D, D) * 0.04).half()
sv = [[torch.empty(B, D, device=dev, dtype=torch.float16) for _ in range(4)] for _ in range(NB)]
st = torch.cuda.Stream()
def launch(i):
    i %= NB
    L.check(L.lib().ark_gru_cell_fwd_dma(L.i32(2), L.i32(1), L.ptr(h16[i]), L.ptr(w16), L.ptr(yp[i]), L.ptr(bh), L.ptr(gi[i]), L.ptr(yo[i]),
            L.ptr(o16a[i]), L.ptr(o16b[i]), L.ptr(None), L.ptr(None), L.f32(0.0), L.u64(0), L.i64(0), L.ptr(None), L.ptr(sv[i][0]), L.ptr(sv[i][1]), L.ptr(sv[i][2]),
            L.ptr(sv[i][3]), L.i32(B), L.i32(D), L.cur_stream()), "cell")
with torch.cuda.stream(st):
    for mask in (0, 1, 2, 4, 3, 5, 6, 7):
        L.lib().ark_set_dma_debug(mask)
        for i in range(3): launch(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(48): launch(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); e1.synchronize()
        print(f"dbg mask {mask} (1=no mainloop 2=no stores 4=no loads): {e0.elapsed_time(e1) / 240 * 1e3:.2f} us/launch", flush=True)
    L.lib().ark_set_dma_debug(0)
