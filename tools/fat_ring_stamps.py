"""Iterations of the LDS-ring weights-stationary forward (diagnostic build, -DARK_FAT_STAMPS):

    bash tools/build_variant.sh fatstamps gru_fat.hip -DARK_FAT_STAMPS
    ARK_AMD_LIB=$PWD/ark_amd/lib/variants/fatstamps/libark_amd.so python tools/fat_ring_stamps.py [workload] [batch]

Thread 0 of every workgroup counts its loop iterations, the ones in which the issue cursor stood waiting for a counter
(bubbles), and the 100-MHz ticks of the whole launch; lane 0 of every helper wave sums the ticks of its phases.
Timing-only switches for the same build (results invalid, the hand-off still runs): -DARK_RING_NO_STORES (every store dropped
by its descriptor), -DARK_RING_DUMMY_SRC / -DARK_RING_DUMMY_SRC_EX (every fragment piece from one cache-hot tile, row-major /
exchange layout)."""
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
cfg = dict(bench.build_cfg(0.1, wl), ark_fat=1, ark_fat_kernel="ring")
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
buf = (ctypes.c_ulonglong * (512 * 8 * 5))()
L.check(L.lib().ark_debug_fat_stamps(buf, ctypes.c_int(512 * 8 * 5)), "stamps")
allw = np.array(buf[:], dtype=np.float64)
a = allw[:512 * 8].reshape(512, 8)
ph = allw[512 * 8:].reshape(512, 4, 8)
live = a[:, 4] > 0
ph = ph[live]
a = a[live]
print(f"{wl} B={B}: {len(a)} workgroups, err={eng.sweep_error()}")
print(f"tiles per workgroup {a[:, 4].mean():.1f}; iterations {a[:, 0].mean():.1f} (min {a[:, 0].min():.0f}, max {a[:, 0].max():.0f}); "
      f"stalled {a[:, 1].mean():.1f}; launch {a[:, 2].mean() / 100:.1f} us -> {a[:, 2].mean() / 100 / a[:, 0].mean():.3f} us per iteration, "
      f"{a[:, 2].mean() / 100 / a[:, 4].mean():.3f} us per tile")
names = ["issue", "gate math / copies", "counted wait", "verdict + barrier", "-", "-", "loop bookkeeping"]
for wv in range(4):   # (the helper waves; the matrix waves carry no stamps)
    it = a[:, 0].mean()
    print(f"wave {wv}: " + "  ".join(f"{names[i]} {ph[:, wv, i].mean() / 100 / it:.3f}" for i in (0, 1, 2, 3, 6)) + "  us per iteration")
