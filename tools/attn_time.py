"""Self-attention forward + backward at the wd-* shapes of the Transformer variants: the vector-unit kernels of txf.hip
(probabilities [B, H, L, L] in memory) against the matrix-core flash kernels of attn_mfma.hip.  python tools/attn_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_amd import _lib as L

def call(name, *a):
    L.check(getattr(L.lib(), name)(*a), name)

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

H = 4
hyper = torch.zeros(64, device="cuda")
for name, B, Lq, D, causal in (("wd-articles decoder", 16, 637, 512, 1), ("wd-articles encoder", 16, 212, 1536, 0),
                               ("wd-movies decoder", 256, 70, 128, 1), ("wd-movies encoder", 256, 23, 384, 0),
                               ("syn-paths-long decoder", 64, 130, 512, 1), ("syn-paths decoder", 1024, 10, 512, 1),
                               ("syn-paths encoder", 1024, 3, 1536, 0), ("syn-types decoder", 256, 10, 1024, 1)):
    torch.manual_seed(0)
    qkv = torch.randn(Lq * B, 3 * D, device="cuda")
    dout = torch.randn(Lq * B, D, device="cuda")
    out, dqkv = torch.empty(Lq * B, D, device="cuda"), torch.empty_like(qkv)
    probs, dsc = torch.empty(B * H * Lq * Lq, device="cuda"), torch.empty(B * H * Lq * Lq, device="cuda")
    Lp = (Lq + 63) // 64 * 64
    lse, delta = torch.empty(B * H * Lp, device="cuda"), torch.empty(B * H * Lp, device="cuda")
    for p in (0.0, 0.1):
        args = (L.i32(B), L.i32(Lq), L.i32(D), L.i32(H), L.i32(causal), L.f32(p), L.u64(5), L.ptr(hyper), L.cur_stream())
        vf = timed(lambda: call("ark_attn_fwd", L.ptr(qkv), L.ptr(out), L.ptr(probs), L.ptr(None), *args))
        vb = timed(lambda: call("ark_attn_bwd", L.ptr(qkv), L.ptr(out), L.ptr(probs), L.ptr(dout), L.ptr(dsc), L.ptr(dqkv), L.ptr(None), *args))
        ff = timed(lambda: call("ark_attn_flash_fwd", L.i32(2), L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(None), *args))
        fb = timed(lambda: call("ark_attn_flash_bwd", L.i32(1), L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(dout), L.ptr(delta), L.ptr(dqkv), L.ptr(None), *args))
        flops = 4.0 * B * H * Lq * Lq * (D // H) * (0.5 if causal else 1.0)
        print(f"{name:24s} B={B:3d} L={Lq:3d} dh={D // H:3d} drop={p}: vector fwd {vf:8.1f} bwd {vb:8.1f} us | flash fwd {ff:7.1f} bwd {fb:7.1f} us"
              f" | fwd {flops / ff / 1e6:6.1f} TFLOP/s, bwd {2.5 * flops / fb / 1e6:6.1f}  | probs+dscore {2 * probs.numel() * 4 / 1e6:.0f} MB avoided", flush=True)
