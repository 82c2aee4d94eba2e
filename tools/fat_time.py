"""Forward recurrence alone: the weights-stationary launch (ark_fat=1) against the layer-diagonal launches (ark_fat=0),
event-timed replays of a captured forward sweep.   python tools/fat_time.py [B] [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ark_amd.engine import Engine, CAPTURE_MODE
from ark_amd import initlib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "syn-paths"
dev = torch.device("cuda:0")
for fat, kern in ((0, "-"), (1, "ring"), (1, "regs"), (0, "-"), (1, "ring")):
    cfg = dict(bench.build_cfg(0.1, wl), ark_fat=fat, ark_fat_kernel=kern)
    eng = Engine(cfg, dev, precision="mixed")
    eng.load_params(initlib.init_state(cfg, seed=0))
    tr, sq = bench.synth_global_batch(cfg, B, seed=1)
    eps = torch.randn(B, cfg["d_latent"])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.train_step(tr.to(dev), sq.to(dev), eps.to(dev))
        torch.cuda.synchronize()
        w, Lq = eng.ws, eng.L
        fn = lambda: eng._diag_sweep(w, B, Lq, True, True)
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            fn()
        for _ in range(5):
            g.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{wl} B={B} fat={fat} kernel={kern} use_fat={eng._use_fat(B, Lq)}: forward recurrence {min(ts):.1f} us (runs: {' '.join(f'{t:.1f}' for t in ts)}) err={eng.sweep_error()}", flush=True)
    del eng
    torch.cuda.empty_cache()
