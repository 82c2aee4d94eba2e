"""Time the forward recurrence alone -- persistent sweep (csrc/gru_sweep.hip) against the layer-diagonal launches -- at a
BASELINE workload's shape, eager launches between HIP events.

    python tools/sweep_time.py [--workload wd-articles] [--batch 16] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ark_amd.engine import Engine
from ark_amd.initlib import init_state


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="wd-articles")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.1)
    args = ap.parse_args()
    cfg = bench.build_cfg(args.dropout, args.workload)
    B = args.batch or cfg["batch"]
    triples, seq = bench.synth_global_batch(cfg, B, 0)
    P = init_state(cfg, seed=0)
    for mode in (0, 1):
        eng = Engine(dict(cfg, ark_sweep=mode), "cuda:0", precision="mixed")
        eng.load_params(P)
        eng.set_hyper(beta=0.1)
        dev = eng.device
        eps = torch.randn(B, cfg["d_latent"], device=dev)
        eng.train_step(triples.to(dev), seq.to(dev), eps)
        torch.cuda.synchronize()
        w, Lq = eng.ws, cfg["seq_len"] - 1
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng._diag_sweep(w, B, Lq, True, True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        tb = []
        for _ in range(args.reps):
            eng._zero(w["dH0"])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng._diag_bwd_sweep(w, B, Lq, True, None)
            e1.record()
            torch.cuda.synchronize()
            tb.append(e0.elapsed_time(e1))
        steps = Lq + cfg["n_layers"] - 1
        print(f"   backward: {min(tb):.3f} ms  ({min(tb) * 1e3 / steps:.2f} us per diagonal)  all: {' '.join(f'{t:.2f}' for t in tb)}", flush=True)
        best = min(ts)
        print(f"{args.workload} B={B} L={Lq} {'sweep' if mode else 'diagonals'}: {best:.3f} ms  ({best * 1e3 / steps:.2f} us per diagonal)"
              f"  all: {' '.join(f'{t:.2f}' for t in ts)}  err={eng.sweep_error()}", flush=True)


if __name__ == "__main__":
    main()
