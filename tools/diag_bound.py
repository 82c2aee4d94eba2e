"""What could a persistent (dependency-driven) forward sweep of a FULL batch gain over the launch-per-diagonal chain?

    python tools/diag_bound.py [--workload syn-paths] [--batch 1024]

The gain of persistence is bounded by the difference between (a) the dependent chain as the step runs it (two row-block
chains of L + n - 1 launches each) and (b) the SAME launches with every dependency removed: each anti-diagonal on a queue
of its own inside one captured graph, so the chip overlaps ramps, tails, epilogues and boundaries of different diagonals as
freely as a perfect in-kernel scheduler with free hand-offs could.  (b) does the same work on the same operands (the buffers
hold a finished forward pass, so every launch even recomputes the right values).  Whatever (a) - (b) is, a persistent
kernel pays its hand-offs (write-through stores drained, counter, poll: ~2 us per cell and row block at the measured
prices of the small-batch sweeps) out of it."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, st, reps=30):
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn-paths")
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    from ark_amd import initlib
    from ark_amd.engine import Engine
    dev = torch.device("cuda", 0)
    cfg = bench.build_cfg(0.1, a.workload)
    B = a.batch or cfg["batch"]
    tri, seq = bench.synth_global_batch(cfg, B, 1)
    tri, seq = tri.to(dev), seq.to(dev)
    eps = torch.randn(B, cfg["d_latent"], device=dev)
    eng = Engine(cfg, dev, precision="mixed")
    eng.load_params(initlib.init_state(cfg, seed=0))
    eng.set_hyper(lr=cfg["learning_rate"], beta=cfg["beta"])
    eng.train_step(tri, seq, eps)
    torch.cuda.synchronize()
    w, Lq, n = eng.ws, eng.L, eng.n
    nd = Lq + n - 1
    st = torch.cuda.Stream()
    pool = [torch.cuda.Stream() for _ in range(nd)]

    def chain():            # (a) as in the step: row-block chains of dependent launches
        eng._diag_sweep(w, B, Lq, True, True)

    def one_chain():        # the same as ONE chain of full-batch launches
        eng._diag_chain(w, B, 0, B, Lq, True, True)

    def free():             # (b) every anti-diagonal on its own queue: no dependency between them
        main = torch.cuda.current_stream()
        for d in range(nd):
            pool[d].wait_stream(main)
            with torch.cuda.stream(pool[d]):
                eng._diag_chain(w, B, 0, B, Lq, True, True, diagonals=[d])
        for d in range(nd):
            main.wait_stream(pool[d])

    def free_halves():      # (b') the two 512-row halves of every diagonal, each on its own queue
        main = torch.cuda.current_stream()
        for d in range(nd):
            pool[d].wait_stream(main)
            with torch.cuda.stream(pool[d]):
                for b0 in (0, B // 2):
                    eng._diag_chain(w, B, b0, B // 2, Lq, True, True, diagonals=[d])
        for d in range(nd):
            main.wait_stream(pool[d])

    def bwd_chain():        # the backward diagonals as the step runs them (two row-block chains)
        eng._run_chains(B, lambda b0, Bc: eng._diag_bwd_chain(w, B, b0, Bc, Lq, True))

    def bwd_free():         # every backward anti-diagonal on its own queue
        main = torch.cuda.current_stream()
        for d in range(nd):
            pool[d].wait_stream(main)
            with torch.cuda.stream(pool[d]):
                eng._diag_bwd_chain(w, B, 0, B, Lq, True, diagonals=[d])
        for d in range(nd):
            main.wait_stream(pool[d])

    for rnd in range(3):
        ba, bb = timed(bwd_chain, st), timed(bwd_free, st)
        print(f"round {rnd}: BACKWARD recurrence: step's chains {ba:7.1f} us ({ba / nd:5.2f} per diagonal)   no dependencies {bb:7.1f} us "
              f"({bb / nd:5.2f} per diagonal)", flush=True)
        ta, t1, tb, tc = timed(chain, st), timed(one_chain, st), timed(free, st), timed(free_halves, st)
        print(f"round {rnd}: forward recurrence, {nd} anti-diagonals, B = {B}: step's chains {ta:7.1f} us ({ta / nd:5.2f} per diagonal)   "
              f"one chain {t1:7.1f}   no dependencies {tb:7.1f} us ({tb / nd:5.2f} per diagonal)   "
              f"no dependencies, half-batch launches {tc:7.1f}", flush=True)


if __name__ == "__main__":
    main()
