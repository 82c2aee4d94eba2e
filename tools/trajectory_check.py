"""Debug aid: loss trajectory of the captured train step under different speed knobs (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_cfg, synth_global_batch
from ark_amd.engine import Engine
from ark_amd import initlib, _lib as L

def run(tuning, dropout, graph, splitk, steps=40, prec="mixed"):
    for k, v in tuning:
        L.check(L.lib().ark_set_tuning(k, v), "tune")
    L.lib().ark_set_split_k(1 if splitk else 0)
    cfg = build_cfg(dropout)
    dev = torch.device("cuda:0")
    eng = Engine(cfg, dev, precision=prec)
    eng.load_params(initlib.init_state(cfg, seed=0))
    eng.set_hyper(lr=1e-4, beta=0.1)
    B = 1024
    ring = []
    for i in range(4):
        tr, sq = synth_global_batch(cfg, B, seed=1 + i)
        torch.manual_seed(1000 + i)
        ring.append((tr.to(dev), sq.to(dev), torch.randn(B, 10).to(dev)))
    tri, seq, eps = (x.clone() for x in ring[0])
    cnt = float(B * 10)
    step = eng.capture_train_step(tri, seq, eps, ce_count=cnt) if graph else (lambda: eng.train_step(tri, seq, eps, ce_count=cnt))
    out = []
    for i in range(steps):
        a, b, c = ring[i % 4]
        tri.copy_(a); seq.copy_(b); eps.copy_(c)
        o = step()
        if i % 5 == 4:
            out.append(round(float(o[0]), 4))
    return out

if __name__ == "__main__":
    for name, args in [("64/0 eager nodrop nosplit", ([(1, 64), (2, 0)], 0.0, False, False)),
                       ("64/0 eager nodrop nosplit again", ([(1, 64), (2, 0)], 0.0, False, False)),
                       ("64/0 eager nodrop split", ([(1, 64), (2, 0)], 0.0, False, True)),
                       ("32/0 eager nodrop nosplit", ([(1, 32), (2, 0)], 0.0, False, False)),
                       ("64/2 eager nodrop nosplit", ([(1, 64), (2, 2)], 0.0, False, False)),
                       ("64/1 eager nodrop nosplit", ([(1, 64), (2, 1)], 0.0, False, False)),
                       ("32/2 graph nodrop split", ([(1, 32), (2, 2)], 0.0, True, True)),
                       ("32/2 graph drop split", ([(1, 32), (2, 2)], 0.1, True, True)),
                       ("32/2 graph drop split again", ([(1, 32), (2, 2)], 0.1, True, True)),
                       ("64/0 f32 eager nodrop nosplit", ([(1, 64), (2, 0)], 0.0, False, False, 40, "f32")),
                       ]:
        print(f"{name:40s}", run(*args), flush=True)
