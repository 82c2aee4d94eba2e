"""Development probe: ELBO of the HIP forward (mixed / f32) against the CPU oracle at the YAML batch sizes of the other
BASELINE configurations (what tests/test_configs_gpu.py asserts).  python tools/elbo_probe.py [name ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import sail_oracle as O
from tests.parity_util import make_engine, rel_err, synth_batch
from tests.test_configs_gpu import SHAPES

torch.set_num_threads(16)
for name in (sys.argv[1:] or list(SHAPES)):
    cfg, B = SHAPES[name]
    P = O.init_params(cfg, 0)
    triples, seq = synth_batch(cfg, B, seed=11, padded=cfg["pad_rid"] is not None)
    torch.manual_seed(3)
    eps = torch.randn(B, cfg["d_latent"])
    t0 = time.time()
    with torch.no_grad():
        loss, ce, kl, *_ = O.sail_elbo(P, triples, seq, eps, 0.5, cfg)
    line = f"{name} B={B} oracle loss {float(loss):.7f} ce {float(ce):.7f} kl {float(kl):.3e} ({time.time() - t0:.1f}s)"
    for prec in ("f32", "mixed", "bf16"):
        eng = make_engine(cfg, P, prec)
        eng.set_hyper(beta=0.5)
        out4 = eng.eval_loss(triples.to(eng.device), seq.to(eng.device), eps.to(eng.device)).cpu().numpy()
        line += f" | {prec}: {rel_err(float(out4[0]), float(loss)):.2e}"
        del eng
        torch.cuda.empty_cache()
    print(line, flush=True)
