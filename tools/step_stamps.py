"""End time of every launch of one captured train step, without a profiler in the process.

    python tools/step_stamps.py [--workload syn-paths] [--cfg k=v,...] [--replays 200]

A one-thread `ark_stamp` launch is captured right behind every library call of the step, on the same queue; each writes
the device's 100-MHz real-time counter.  After N replays the stamps of the last replay give, per queue, when each launch
finished (us from the step's first stamp) and how long after its predecessor ON THE SAME QUEUE that was (an upper bound of
the launch's duration: it includes waiting for cross-queue dependencies).  The stamps cost one extra launch boundary each
(~1.5 us on the dependent chain): read the shape of the step, not its length (printed beside the unstamped step time)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn-paths")
    ap.add_argument("--cfg", default="")
    ap.add_argument("--replays", type=int, default=200)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--model", default="SAIL", help="SAIL | t-ARK | t-SAIL (the Transformer engines: ark_amd/txf_engine.py)")
    ap.add_argument("--by-call", action="store_true", help="also print the time per library call, summed over the step")
    a = ap.parse_args()
    from ark_amd import engine as E, initlib
    dev = torch.device("cuda", 0)
    cfg = bench.build_cfg(0.1, a.workload)
    if a.model != "SAIL":
        cfg.update(model_type=a.model, ark_txf_dropout=0.1)
    for kv in filter(None, a.cfg.split(",")):
        k, v = kv.split("=")
        cfg[k] = int(v)
    B = a.batch or bench.WORKLOADS[a.workload]["batch"]
    tri, seq = bench.synth_global_batch(cfg, B, 1)
    cnt = float((seq[:, 1:] != 0).sum())          # as bench.py / the train loop: target count from the host, noise as an input
    tri, seq = tri.to(dev), seq.to(dev)
    torch.manual_seed(1000)
    eps = torch.randn(B, cfg["d_latent"], device=dev)

    def run(stamped):
        if a.model != "SAIL":
            from ark_amd.txf_engine import TxfEngine
            eng = TxfEngine(cfg, dev, precision="mixed")
        else:
            eng = E.Engine(cfg, dev, precision="mixed")
        eng.load_params(initlib.init_state(cfg, seed=0))
        eng.set_hyper(lr=1e-4, beta=0.1)
        buf = torch.zeros(4096, dtype=torch.int64, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            if stamped:
                eng.train_step(tri, seq, eps, ce_count=cnt)   # (workspaces, shadows: outside the log)
                torch.cuda.synchronize()
                E._stamps = {"buf": buf, "log": []}
                replay = None
                try:
                    # capture_train_step runs one eager warm-up step first: its log entries are dropped below
                    replay = eng.capture_train_step(tri, seq, eps, ce_count=cnt)
                finally:
                    log = E._stamps["log"]
                    E._stamps = None
            else:
                replay = eng.capture_train_step(tri, seq, eps, ce_count=cnt)
            for _ in range(300):
                replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.replays):
                replay()
            e1.record(s)
            s.synchronize()
        ms = e0.elapsed_time(e1) / a.replays
        if not stamped:
            return ms, None, None
        return ms, buf.cpu().numpy(), log

    ms_plain, _, _ = run(False)
    ms, buf, log = run(True)
    n = len(log) // 2            # eager warm-up + capture issue the same sequence: the captured half wrote last
    log = log[n:]
    t = buf[n:2 * n].astype("float64") * 0.01
    t0 = t.min()
    print(f"{a.workload}: {ms_plain:.4f} ms per step unstamped, {ms:.4f} ms with {n} stamps")
    queues = {}
    for (name, st), ti in zip(log, t):
        queues.setdefault(st, []).append((ti - t0, name))
    order = sorted(queues, key=lambda q: min(x[0] for x in queues[q]))
    qname = {q: f"q{i}" for i, q in enumerate(order)}
    prev = {}
    rows = []
    for (name, st), ti in zip(log, t):
        rel = ti - t0
        d = rel - prev.get(st, 0.0)
        prev[st] = rel
        rows.append((rel, qname[st], d, name))
    for rel, q, d, name in sorted(rows):
        print(f"{rel:9.2f}  {q}  +{d:7.2f}  {name}")
    if a.by_call:
        tot = {}
        for rel, q, d, name in rows:
            c = tot.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += d
        print("per library call (us since the previous stamp on the same queue, summed):")
        for name, (k, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(f"  {us:9.1f} us  {k:4d} x  {name}")


if __name__ == "__main__":
    main()
