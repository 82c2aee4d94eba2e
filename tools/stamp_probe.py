"""In-kernel phase timing of the forward diagonal kernel (diagnostic build, -DARK_STAMPS):

    bash tools/build_variant.sh stamps gru_diag.hip -DARK_STAMPS
    ARK_AMD_LIB=$PWD/ark_amd/lib/variants/stamps/libark_amd.so python tools/stamp_probe.py "fwd_rows=64,fwd_units=32" "fwd_rows=128,fwd_units=64,fwd_nbuf=4"

Re-issues ONE full (3-role) forward diagonal of the syn-paths B = 1024 step and prints, over its workgroups, the shader-clock
cycles between the stamps (entry -> main loop -> epilogue math -> stores -> end) and the launch's wall span from the
100-MHz real-time counter.  The stamped build forbids overlaps the real kernel has: read the SHARES, not the length."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from ark_amd import _lib as L, initlib
    from ark_amd.engine import Engine
    dev = torch.device("cuda", 0)
    cfg0 = bench.build_cfg(0.1, "syn-paths")
    B = int(os.environ.get("PROBE_B", "1024"))
    tri, seq = bench.synth_global_batch(cfg0, B, 1)
    tri, seq = tri.to(dev), seq.to(dev)
    eps = torch.randn(B, cfg0["d_latent"], device=dev)
    for s in (sys.argv[1:] or [""]):
        cfg = dict(cfg0, ark_diag_chains=1)
        if s:
            cfg["ark_diag_tuning"] = {k: int(v) for k, v in (kv.split("=") for kv in s.split(","))}
        eng = Engine(cfg, dev, precision="mixed")
        eng.load_params(initlib.init_state(cfg, seed=0))
        eng.set_hyper(lr=1e-4, beta=0.1)
        eng.train_step(tri, seq, eps)
        torch.cuda.synchronize()
        w = eng.ws
        rows = []
        for rep in range(6):
            for _ in range(3):   # neighbours in time, as in the real sweep
                eng._diag_chain(w, B, 0, B, eng.L, True, True, diagonals=[4, 5])
            torch.cuda.synchronize()
            nb = 8192
            buf = (ctypes.c_ulonglong * (8 * nb))()
            rc = L.lib().ark_debug_stamps(buf, nb)
            assert rc == 0, rc
            a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
            a = a[a[:, 0] != 0]
            a = a[a[:, 6] > a[:, 6].max() - 5000]   # this launch only (the buffer keeps entries of earlier, larger grids)
            rows.append(a)
        a = rows[-1]
        n = len(a)
        seg = np.stack([a[:, i + 1] - a[:, i] for i in range(4)], 1)
        rt = (a[:, 6].max() - a[:, 5].min()) * 0.01   # us
        start = (a[:, 5] - a[:, 5].min()) * 0.01
        life = (a[:, 6] - a[:, 5]) * 0.01
        names = ["prologue", "main loop", "epilogue math", "stores"]
        print(f"--- {s or '(default)'}: {n} workgroups, launch span {rt:.2f} us; workgroup start p50 {np.median(start):.2f} "
              f"p90 {np.percentile(start, 90):.2f} max {start.max():.2f} us; lifetime p50 {np.median(life):.2f} max {life.max():.2f} us")
        for i, nm in enumerate(names):
            print(f"    {nm:14s} cycles p10 {np.percentile(seg[:, i], 10):8.0f}  p50 {np.median(seg[:, i]):8.0f}  p90 {np.percentile(seg[:, i], 90):8.0f}")
        tot = (a[:, 4] - a[:, 0])
        print(f"    {'total':14s} cycles p50 {np.median(tot):8.0f}; clock ~ {np.median(tot / np.maximum(life, 1e-3)) / 1e3:.2f} GHz")
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
