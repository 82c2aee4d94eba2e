"""Time the two layer-diagonal GRU kernels for a list of ArkDiagTuning settings in ONE process on one device
(cdna_hip_programming.md rule 24: rank variants by interleaved rounds, never across processes / devices).

    python tools/diag_sweep.py [--workload syn-paths] [--batch 1024] "fwd_rows=64,fwd_units=32" "bwd_rows=64,bwd_ki=1" ...

Prints, per setting, the median over rounds of the average launch time of each kernel (us) and the step time of a
captured train step (ms)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn-paths")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("settings", nargs="*", default=[""])
    args = ap.parse_args()
    from ark_amd import initlib
    from ark_amd.engine import Engine
    dev = torch.device("cuda", 0)
    cfg0 = bench.build_cfg(0.1, args.workload)
    B = args.batch or cfg0["batch"]
    tri, seq = bench.synth_global_batch(cfg0, B, 1)
    tri, seq = tri.to(dev), seq.to(dev)
    eps = torch.randn(B, cfg0["d_latent"], device=dev)
    cnt = float((seq[:, 1:] != 0).sum())
    engines = []
    for s in args.settings:
        cfg = dict(cfg0)
        if s:   # `ark_*` keys are engine options (e.g. ark_diag_chains=1), the rest are ArkDiagTuning fields
            kv = {k: int(v) for k, v in (kv.split("=") for kv in s.split(","))}
            cfg.update({k: v for k, v in kv.items() if k.startswith("ark_")})
            tun = {k: v for k, v in kv.items() if not k.startswith("ark_")}
            if tun:
                cfg["ark_diag_tuning"] = tun
        eng = Engine(cfg, dev, precision="mixed")
        eng.load_params(initlib.init_state(cfg, seed=0))
        eng.set_hyper(lr=cfg["learning_rate"], beta=cfg["beta"])
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            step = eng.capture_train_step(tri, seq, eps, ce_count=cnt)
        engines.append((s, eng, step, st))
    for s, eng, step, st in engines:   # settle: the first ~300 replays after a capture run slower
        with torch.cuda.stream(st):
            for _ in range(300):
                step()
    torch.cuda.synchronize()
    res = {s: {"fwd": [], "bwd": [], "step": []} for s in args.settings}
    for _ in range(args.rounds):
        for s, eng, step, st in engines:
            with torch.cuda.stream(st):
                for _ in range(10):
                    step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    step()
                e1.record()
                e1.synchronize()
                res[s]["step"].append(e0.elapsed_time(e1) / args.steps)
            t = bench.time_diag_kernels(eng, B)
            res[s]["fwd"].append(t["gru_diag_fwd_kernel"][0] * 1e6)
            res[s]["bwd"].append(t["gru_diag_bwd_kernel"][0] * 1e6)
    for s in args.settings:
        r = res[s]
        print(f"{s or '(default)':60s} fwd {statistics.median(r['fwd']):6.2f} us  bwd {statistics.median(r['bwd']):6.2f} us  "
              f"step {statistics.median(r['step']):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
