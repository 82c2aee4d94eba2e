"""One t-ARK / t-SAIL train-step loop at a wd-* shape for rocprofv3 --kernel-trace --stats: python tools/txf_wd_prof.py t-ARK wd-articles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ark_amd import initlib
from ark_amd.txf_engine import TxfEngine
mt, wl = sys.argv[1], sys.argv[2]
dev = torch.device("cuda", 0)
cfg = dict(bench.build_cfg(0.1, wl), model_type=mt)
B = cfg["batch"]
eng = TxfEngine(cfg, dev, precision="mixed")
eng.load_params(initlib.init_state(cfg, seed=0))
eng.set_hyper(lr=1e-4, beta=0.1)
tri, seq = bench.synth_global_batch(cfg, B, 1)
tri, seq = tri.to(dev), seq.to(dev)
for _ in range(10):
    eng.train_step(tri if mt == "t-SAIL" else None, seq)
torch.cuda.synchronize()
