"""Time the grouped weight-gradient launch of the decoder GRU (syn-paths shapes by default) for a list of
ArkWgradTuning settings in one process, interleaved rounds (standalone: nothing runs beside it).

    python tools/wgrad_time.py [--D 512] [--rows 10240] [--layers 3] "tile=128" "tile=256,nbuf=2" "tile=256,nbuf=4"
"""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ark_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--rows", type=int, default=10240)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("settings", nargs="*", default=[""])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D, R, n = a.D, a.rows, a.layers
    items = []
    keep = []
    for l in range(n):
        G = torch.randn(R, 4 * D, device=dev).to(torch.bfloat16)
        x = torch.randn(R, D, device=dev).to(torch.bfloat16)
        h = torch.randn(R, D, device=dev).to(torch.bfloat16)
        dWih, dWhh = torch.zeros(3 * D, D, device=dev), torch.zeros(3 * D, D, device=dev)
        keep += [G, x, h, dWih, dWhh]
        items.append((G, 4 * D, h, D, dWhh, D, 2 * D, D, R))
        items.append((G[:, 3 * D:], 4 * D, h, D, dWhh[2 * D:], D, D, D, R))
        if l > 0:
            items.append((G, 4 * D, x, D, dWih, D, 3 * D, D, R))
    k = len(items)
    vp = lambda j: (ctypes.c_void_p * k)(*[it[j].data_ptr() for it in items])
    i64 = lambda j: (ctypes.c_int64 * k)(*[it[j] for it in items])
    i32 = lambda j: (ctypes.c_int * k)(*[it[j] for it in items])
    flops = sum(2.0 * it[6] * it[7] * it[8] for it in items)
    res = {s: [] for s in a.settings}
    for _ in range(5):
        for s in a.settings:
            tn = L.wgrad_tuning(**{kk: int(v) for kk, v in (kv.split("=") for kv in s.split(",") if kv)})
            def run():
                L.check(L.lib().ark_wgrad16_group(L.i32(L.PREC_BF16), L.i32(k), vp(0), i64(1), vp(2), i64(3), vp(4), i64(5), i32(6), i32(7),
                                                  i32(8), tn, L.cur_stream()), "wgrad")
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            e1.synchronize()
            res[s].append(e0.elapsed_time(e1) / a.iters * 1e3)
    for s in a.settings:
        t = statistics.median(res[s])
        print(f"{s or '(default)':30s} {t:8.1f} us   {flops / t / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
