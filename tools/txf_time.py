"""Development timing of the Transformer variants' train step (eager launches, then the captured graph): python tools/txf_time.py [t-ARK|t-SAIL] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ark_amd import initlib
from ark_amd.txf_engine import TxfEngine

mt = sys.argv[1] if len(sys.argv) > 1 else "t-ARK"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = dict(bench.build_cfg(0.1, "syn-paths"), model_type=mt)
dev = torch.device("cuda", 0)
for prec in ("mixed", "f32"):
    eng = TxfEngine(cfg, dev, precision=prec)
    eng.load_params(initlib.init_state(cfg, seed=0))
    eng.set_hyper(lr=1e-4, beta=0.1)
    tri, seq = bench.synth_global_batch(cfg, B, 1)
    tri, seq = tri.to(dev), seq.to(dev)
    for _ in range(5):
        out = eng.train_step(tri, seq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = eng.train_step(tri, seq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mt} B={B} {prec}: {dt * 1e3:.2f} ms/step = {B / dt:.0f} graphs/s, loss {float(out[0]):.4f}", flush=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = eng.capture_train_step(tri, seq)
        for _ in range(5):
            out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{mt} B={B} {prec} captured: {dt * 1e3:.2f} ms/step = {B / dt:.0f} graphs/s, loss {float(out[0]):.4f}", flush=True)
    del eng
    torch.cuda.empty_cache()
