import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from ark_amd import initlib
from ark_amd.txf_engine import TxfEngine
mt = sys.argv[1]
cfg = dict(bench.build_cfg(0.1, "syn-paths"), model_type=mt)
dev = torch.device("cuda", 0)
eng = TxfEngine(cfg, dev, precision="mixed")
eng.load_params(initlib.init_state(cfg, seed=0))
eng.set_hyper(lr=1e-4, beta=0.1)
tri, seq = bench.synth_global_batch(cfg, 1024, 1)
tri, seq = tri.to(dev), seq.to(dev)
for _ in range(12):
    eng.train_step(tri, seq)
torch.cuda.synchronize()
