"""Train step of the Transformer variants at the wd-* shapes (YAML batch), matrix-core flash attention on / off:
python tools/txf_wd_time.py [t-SAIL|t-ARK] [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ark_amd import initlib
from ark_amd.txf_engine import TxfEngine

mt = sys.argv[1] if len(sys.argv) > 1 else "t-SAIL"
wls = sys.argv[2:] or ["wd-movies", "wd-articles"]
dev = torch.device("cuda", 0)
for wl in wls:
    for flash in (0, 1):
        cfg = dict(bench.build_cfg(0.1, wl), model_type=mt, ark_txf_flash=flash)
        B = cfg["batch"]
        eng = TxfEngine(cfg, dev, precision="mixed")
        eng.load_params(initlib.init_state(cfg, seed=0))
        eng.set_hyper(lr=1e-4, beta=0.1)
        tri, seq = bench.synth_global_batch(cfg, B, 1)
        tri, seq = tri.to(dev), seq.to(dev)
        for _ in range(3):
            out = eng.train_step(tri if mt == "t-SAIL" else None, seq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            out = eng.train_step(tri if mt == "t-SAIL" else None, seq)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        mem = torch.cuda.max_memory_allocated() / 2 ** 30
        print(f"{mt} {wl} B={B} L={cfg['seq_len'] - 1} flash={flash}: {dt * 1e3:.2f} ms/step eager, loss {float(out[0]):.4f}, peak {mem:.1f} GiB", flush=True)
        del eng
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
