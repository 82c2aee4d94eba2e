"""Alone time of the fused Adam + 16-bit-shadow launches of a workload's parameter buffer (HBM-bound: 28 B per parameter
+ the shadows).  python tools/adam_time.py [--workload syn-paths]   (ARK_AMD_LIB=<variant> for an A/B of library builds)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="syn-paths")
a = ap.parse_args()
from ark_amd import initlib  # noqa: E402
from ark_amd.engine import Engine  # noqa: E402
dev = torch.device("cuda", 0)
cfg = bench.build_cfg(0.1, a.workload)
eng = Engine(cfg, dev, precision="mixed")
eng.load_params(initlib.init_state(cfg, seed=0))
eng.set_hyper(lr=1e-4, beta=0.1)
eng.G.normal_()
n = eng.layout.total
shadow = sum(t.numel() * t.element_size() for t in list(eng.wih16) + list(eng.whh16) + list(eng.wihT16) + list(eng.whhT16)
             + [eng.wtok16, eng.wtokT16] + list(getattr(eng, "wm16", [])) + list(getattr(eng, "wmT16", [])) + [eng.wh16])
nbytes = 28 * n + shadow
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for which in ("all", "dec", "mlp"):
        if which not in eng._adam_jobs:
            continue
        for _ in range(3):
            eng._adam_launch(which)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                eng._adam_launch(which)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            g.replay()
        e1.record(st)
        st.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        extra = f"  = {nbytes / us / 1e6:.2f} TB/s over {nbytes / 1e6:.0f} MB" if which == "all" else ""
        print(f"{a.workload} adam[{which}] {us:7.1f} us{extra}", flush=True)
