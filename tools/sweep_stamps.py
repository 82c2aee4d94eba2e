"""Where a step of the persistent forward sweep goes (diagnostic build, -DARK_SWEEP_STAMPS):

    bash tools/build_variant.sh swstamps gru_sweep.hip -DARK_SWEEP_STAMPS
    ARK_AMD_LIB=$PWD/ark_amd/lib/variants/swstamps/libark_amd.so python tools/sweep_stamps.py [workload]

Wave 0 of every forward workgroup sums the 100-MHz ticks it spends per phase over the whole sweep; printed per layer as
microseconds per step.  The stamped build waits for the handed-off fragments before the products (the real kernel lets
the first MFMAs start earlier): read the shares."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ark_amd import _lib as L  # noqa: E402
from ark_amd.engine import Engine  # noqa: E402
from ark_amd.initlib import init_state  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "wd-articles"
cfg = bench.build_cfg(0.1, wl)
B = cfg["batch"]
tri, seq = bench.synth_global_batch(cfg, B, 0)
eng = Engine(dict(cfg, ark_sweep=1), "cuda:0", precision="mixed")
eng.load_params(init_state(cfg, seed=0))
eng.set_hyper(beta=0.1)
dev = eng.device
eng.train_step(tri.to(dev), seq.to(dev), torch.randn(B, cfg["d_latent"], device=dev))
torch.cuda.synchronize()
Lq = cfg["seq_len"] - 1
for _ in range(2):
    eng._diag_sweep(eng.ws, B, Lq, True, True)
torch.cuda.synchronize()
nwg = eng._sweep_wgs(B)
buf = (ctypes.c_ulonglong * (nwg * 8))()
L.check(L.lib().ark_debug_sweep_stamps(buf, ctypes.c_int(nwg * 8)), "stamps")
a = np.array(buf[:], dtype=np.float64).reshape(nwg, 8) / 100.0 / Lq   # us per step
names = ["wait counters", "barrier", "fragments land", "MFMA + partials + barrier", "reduce + gate math + tiles", "store drain",
         "-", "atomic + bulk stores + loop"]
per_layer = nwg // eng.n
for l in range(eng.n):
    m = a[l * per_layer:(l + 1) * per_layer].mean(0)
    print(f"layer {l}: " + "  ".join(f"{names[i]} {m[i]:.2f}" for i in (7, 0, 1, 2, 3, 4, 5)) + f"  | sum {m.sum():.2f} us/step")
