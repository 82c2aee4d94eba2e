"""Shared helpers for the GPU parity tests and __graft_entry__.smoke(): run the HIP engine and the
CPU oracle on identical weights / inputs / eps and compare."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return z, json.loads(str(z["cfg_json"]))


def weights_from(z, prefix):
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(z[k].copy())
    return out


def make_engine(cfg, weights, precision="f32", device="cuda:0", lr=None):
    from ark_amd.engine import Engine
    c = dict(cfg)
    if lr is not None:
        c["learning_rate"] = lr
    eng = Engine(c, device, precision=precision)
    eng.load_params(weights)
    if lr is not None:
        eng.set_hyper(lr=lr)
    return eng


def synth_batch(cfg, B, seed, padded=False):
    """IntelliGraphs-shaped synthetic batch (uniform ids), SURVEY.md section 8d."""
    g = torch.Generator().manual_seed(seed)
    T = cfg["max_triples"]
    nE, nR = cfg["nE"], cfg["nR"]
    eb, rb = cfg["ENT_BASE"], cfg["REL_BASE"]
    h = torch.randint(0, nE, (B, T), generator=g)
    r = torch.randint(0, nR, (B, T), generator=g)
    t = torch.randint(0, nE, (B, T), generator=g)
    triples = torch.stack([h, r, t], dim=-1)
    seq = torch.zeros(B, cfg["seq_len"], dtype=torch.long)
    seq[:, 0] = 1
    if padded:
        k = torch.randint(1, T + 1, (B,), generator=g)
    else:
        k = torch.full((B,), T)
    for b in range(B):
        kb = int(k[b])
        body = torch.stack([eb + h[b, :kb], rb + r[b, :kb], eb + t[b, :kb]], dim=-1).reshape(-1)
        seq[b, 1:1 + 3 * kb] = body
        seq[b, 1 + 3 * kb] = 2
        if padded:
            triples[b, kb:] = torch.tensor([cfg["pad_eid"], cfg["pad_rid"], cfg["pad_eid"]])
    return triples, seq


def rel_err(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def smoke_check(verbose=False):
    """tiny SAIL train step on cuda:0 vs the oracle (used by __graft_entry__.smoke)."""
    from oracle import sail_oracle as O
    z, cfg = load_golden("sail_tiny")
    W = weights_from(z, "w0/")
    eng = make_engine(cfg, W, "f32", lr=float(z["lr"]))
    dev = eng.device
    triples, seq = torch.from_numpy(z["triples"]).to(dev), torch.from_numpy(z["seq"]).to(dev)
    eps = torch.from_numpy(z["eps0"]).to(dev)
    eng.set_hyper(beta=float(z["betas"][0]))
    out4 = eng.train_step(triples, seq, eps).cpu().numpy()
    torch.cuda.synchronize()
    # oracle on the same inputs
    P = O.init_params(cfg, int(z["seed"]))
    st = O.adam_init(O.leaf_params(P))
    loss, ce, kl, grads = O.train_step(P, st, (torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])), cfg,
                                       float(z["lr"]), beta=float(z["betas"][0]), eps=torch.from_numpy(z["eps0"]))
    assert rel_err(float(out4[0]), loss) < 1e-5, (out4, loss)
    for (k, _), g in zip(O.leaf_params(P), grads):
        got = eng.g[k].cpu()
        assert torch.allclose(got, g, rtol=2e-3, atol=2e-6), (k, (got - g).abs().max())
    if verbose:
        print(f"smoke ok: loss {out4[0]:.6f} (oracle {loss:.6f}), ce {out4[1]:.6f}, kl {out4[2]:.6e}")
    return True
