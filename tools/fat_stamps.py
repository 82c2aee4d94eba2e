"""Where a step of the weights-stationary forward goes (diagnostic build, -DARK_FAT_STAMPS):

    bash tools/build_variant.sh fatstamps gru_fat.hip -DARK_FAT_STAMPS
    ARK_AMD_LIB=$PWD/ark_amd/lib/variants/fatstamps/libark_amd.so python tools/fat_stamps.py [workload] [batch]

Thread 0 of every workgroup sums the 100-MHz ticks per phase over the launch; printed per layer as microseconds per
recurrence step (two subgroups) and per tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ark_amd import _lib as L
from ark_amd.engine import Engine
from ark_amd.initlib import init_state

wl = sys.argv[1] if len(sys.argv) > 1 else "syn-paths"
cfg = dict(bench.build_cfg(0.1, wl), ark_fat=1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
tri, seq = bench.synth_global_batch(cfg, B, 0)
eng = Engine(cfg, "cuda:0", precision="mixed")
eng.load_params(init_state(cfg, seed=0))
eng.set_hyper(beta=0.1)
dev = eng.device
eng.train_step(tri.to(dev), seq.to(dev), torch.randn(B, cfg["d_latent"], device=dev))
torch.cuda.synchronize()
Lq = cfg["seq_len"] - 1
assert eng._use_fat(B, Lq)
for _ in range(3):
    eng._diag_sweep(eng.ws, B, Lq, True, True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 8))()
L.check(L.lib().ark_debug_fat_stamps(buf, ctypes.c_int(512 * 8)), "stamps")
a = np.array(buf[:], dtype=np.float64).reshape(512, 8)
a = a[a[:, 4] > 0]
tiles = a[:, 4]
us = a / 100.0
names = ["wait counters", "first fragments + multiply", "tile loop", "last copies + drain + publish", "-", "loop overhead"]
print(f"{wl} B={B}: {len(a)} workgroups, {tiles.mean() / Lq:.1f} tiles per workgroup and step, err={eng.sweep_error()}")
tot = us[:, [0, 1, 2, 3, 5]].sum(1)
print(f"per recurrence step (mean over workgroups): " + "  ".join(f"{names[i]} {us[:, i].mean() / Lq:.2f}" for i in (0, 1, 2, 3, 5)) +
      f"  | sum {tot.mean() / Lq:.2f} us/step;  tile loop per tile {(us[:, 2] / np.maximum(tiles - 2 * Lq, 1)).mean():.3f} us")
