"""Standalone timing of the fused vocabulary projection + cross-entropy kernels (ark_vocab_ce_fwd / ark_vocab_ce_dw) at the
wd-movies shape (default) or the wd-articles one: python tools/vc_time.py [articles] (GPU)."""
import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_amd import _lib as L
dev = torch.device("cuda:0")
B, Lq, V, D = (16, 637, 60943, 512) if "articles" in sys.argv[1:] else (256, 70, 24101, 128)
R = B * Lq
g = torch.Generator().manual_seed(0)
Y = (torch.randn(R, D, generator=g) * 0.5).half().to(dev)
W = (torch.randn(V, D, generator=g) * 0.2).half().to(dev)
bias = torch.zeros(V, device=dev)
seq = torch.randint(1, V, (B, Lq + 1), generator=g).to(dev)
hyper = torch.zeros(16, device=dev); hyper[3] = 1.0 / R
rl, lse, dY = torch.zeros(R, device=dev), torch.zeros(R, device=dev), torch.zeros(R * D, device=dev)
dW, db = torch.zeros(V, D, device=dev), torch.zeros(V, device=dev)
common = (L.i32(L.PREC_F16), L.ptr(Y), L.ptr(W), L.ptr(bias), L.ptr(seq), L.i64(Lq + 1), L.ptr(hyper))
tail = (L.i32(B), L.i32(Lq), L.i32(V), L.i32(D), L.cur_stream())
nv = L.lib().ark_vocab_ce_fwd_splits(L.i32(R), L.i32(V), L.i32(D), L.i32(0))
ws = torch.empty(nv * (R * D + 4 * R), device=dev) if nv > 1 else None
print("vocabulary splits:", nv)
def fwd():
    if nv > 1: L.check(L.lib().ark_vocab_ce_fwd_ws(*common, L.ptr(rl), L.ptr(lse), L.ptr(dY), L.ptr(ws), L.i64(ws.numel()), *tail[:-1], L.i32(0), tail[-1]), "f")
    else: L.check(L.lib().ark_vocab_ce_fwd(*common, L.ptr(rl), L.ptr(lse), L.ptr(dY), *tail), "f")
def dw(): L.check(L.lib().ark_vocab_ce_dw(*common, L.ptr(lse), L.ptr(dW), L.ptr(db), *tail), "d")
def fwd1(): L.check(L.lib().ark_vocab_ce_fwd(*common, L.ptr(rl), L.ptr(lse), L.ptr(dY), *tail), "f")
for name, fn in (("fwd", fwd), ("fwd-nosplit", fwd1), ("dw", dw)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); e1.synchronize()
    print(os.environ.get("ARK_VC_RG"), os.environ.get("ARK_VC_VG"), name, e0.elapsed_time(e1) / 10 * 1e3, "us")
