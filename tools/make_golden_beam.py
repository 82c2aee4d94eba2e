"""Golden beam-search outputs from the REAL reference (build container only; same import recipe as make_golden.py):
for the trained weights and latent vectors already stored in tests/golden/<name>.npz, the triples returned by the
reference's SAIL.decode_latent(z, beam=b) for b in BEAMS  ->  tests/golden/beam_decode.npz.

    python tools/make_golden_beam.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tools.make_golden import OUT, import_reference
from tests.parity_util import load_golden

NAMES = ["sail_tiny", "sail_small", "sail_small_pad"]
BEAMS = [2, 3, 4]


def main():
    M, U = import_reference()
    out = {}
    for name in NAMES:
        z, cfg = load_golden(name)
        torch.manual_seed(int(z["seed"]))
        model = M.SAIL(cfg)
        steps = len(z["losses"])
        model.load_state_dict({f[len(f"w{steps}/"):]: torch.from_numpy(z[f].copy()) for f in z.files if f.startswith(f"w{steps}/")})
        model.eval()
        zs = torch.from_numpy(z["dec_z"])
        for b in BEAMS:
            with torch.no_grad():
                trip = model.decode_latent(zs, cfg["seq_len"], cfg["special_tokens"], U.seq_to_triples, cfg["ENT_BASE"],
                                           cfg["REL_BASE"], beam=b)
            L = max([len(t) for t in trip] + [1])
            arr = -np.ones((len(trip), L, 3), dtype=np.int64)
            for i, tl in enumerate(trip):
                for j, t3 in enumerate(tl):
                    arr[i, j] = t3
            out[f"{name}/beam{b}/triples"] = arr
            out[f"{name}/beam{b}/n"] = np.array([len(t) for t in trip])
            print(name, b, [len(t) for t in trip])
    np.savez_compressed(os.path.join(OUT, "beam_decode.npz"), **out)


if __name__ == "__main__":
    main()
