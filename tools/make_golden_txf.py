"""Golden fixtures of the Transformer variants from the REAL reference (build container only; same import recipe as
make_golden.py):  tests/golden/tark_tiny.npz (full tensors: initial weights, logits, losses, every gradient, weights
after 1 and 3 Adam steps, greedy and sampled generations), tests/golden/tark_synpaths_b32_s0.npz (D = 512 scalars).
Dropout is switched off through the reference's own config key (`dec_dropout: 0.0`, models.py:389): train-mode numerics
are then deterministic, as for the GRU goldens.

    python tools/make_golden_txf.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tools.make_golden import OUT, REF, import_reference, make_batch, make_cfg
from tools.make_golden_r3 import SAMPLING, quiet_model


def dump_tark(M, U, name, D, heads, n, nE, nR, T, B, seed, lr=1e-3, steps=3, full=True):
    cfg = make_cfg("t-ARK", D, 4, n, nE, nR, T, False)
    cfg["n_heads"] = heads
    torch.manual_seed(seed)
    model = quiet_model(M.ARK, cfg)
    model.train()
    triples, seq, _ = make_batch(U, cfg, B, seed + 100, False)
    out = {"cfg_json": np.array(json.dumps(cfg)), "seed": np.array(seed), "lr": np.array(lr), "triples": triples.numpy(),
           "seq": seq.numpy()}
    for k, v in model.state_dict().items():   # initial weights: regenerated from the seed by the init-order-compatible code;
        v64 = v.detach().double()              # pinned here by exact float64 sums and squared norms
        out["w0sum/" + k], out["w0sq/" + k] = np.array(float(v64.sum())), np.array(float((v64 * v64).sum()))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for s in range(steps):
        opt.zero_grad()
        logits = model(seq[:, :-1])
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), seq[:, 1:].reshape(-1), ignore_index=0)
        ce.backward()
        losses.append([ce.item(), ce.item(), 0.0])
        if s == 0:
            if full:
                out["logits0"] = logits.detach().numpy()
            for k, p in model.named_parameters():
                gnp = p.grad.detach().numpy()
                if full:
                    out["g0/" + k] = gnp
                out["g0norm/" + k] = np.array(float(np.sqrt((gnp.astype(np.float64) ** 2).sum())))
        opt.step()
        if full and s == 0:
            for k, v in model.state_dict().items():
                out[f"w{s + 1}/" + k] = v.detach().clone().numpy()
        for k, v in model.state_dict().items():
            out[f"w{s + 1}sum/" + k] = np.array(float(v.detach().double().sum()))
    out["losses"] = np.array(losses, dtype=np.float64)
    # generation from the INITIAL weights (regenerated from the seed on the GPU box): greedy, and the sampling grid
    torch.manual_seed(seed)
    gen = quiet_model(M.ARK, cfg)
    gen.eval()
    st = cfg["special_tokens"]
    out["gen_greedy"] = gen.generate(cfg["seq_len"], st, device="cpu", batch_size=4, sample=False).numpy()
    out["gen_combos"] = np.array(SAMPLING, dtype=np.float64)
    for i, (temp, top_p, top_k) in enumerate(SAMPLING):
        torch.manual_seed(500 + i)
        out[f"gen_seq{i}"] = gen.generate(cfg["seq_len"], st, device="cpu", batch_size=4, sample=True, temperature=temp,
                                          top_p=top_p, top_k=int(top_k)).numpy()
    # teacher-forced bits of the first graphs (ARK.posterior_bits, models.py:473-520) from the initial weights
    from tools.make_golden_r3 import dataset_of
    import contextlib, io
    ds, _, _ = dataset_of(U, cfg, B, seed + 100, False)
    with contextlib.redirect_stderr(io.StringIO()):
        stats = gen.posterior_bits(ds, "cpu", pad_id=0, sample_frac=min(1.0, 8 / B))
    out["bits_ar"] = np.array([r["ar_bits"] for r in stats["records"]])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "losses", losses, "greedy", out["gen_greedy"][0].tolist())


def dump_tsail(M, U, name, D, Z, heads, n, nE, nR, T, B, padded, seed, lr=1e-3, steps=3, full=True):
    """t-SAIL (models.py:160-170, 187-195).  Its layers hard-code dropout 0.1 (models.py:73,104), so the deterministic
    goldens are taken in eval() mode WITH autograd enabled: dropout off, the stock (non-fused) layer code path."""
    cfg = make_cfg("t-SAIL", D, Z, n, nE, nR, T, padded)
    cfg["n_heads"] = heads
    torch.manual_seed(seed)
    model = quiet_model(M.SAIL, cfg)
    model.eval()
    triples, seq, _ = make_batch(U, cfg, B, seed + 100, padded)
    out = {"cfg_json": np.array(json.dumps(cfg)), "seed": np.array(seed), "lr": np.array(lr), "triples": triples.numpy(),
           "seq": seq.numpy()}
    for k, v in model.state_dict().items():
        v64 = v.detach().double()
        out["w0sum/" + k], out["w0sq/" + k] = np.array(float(v64.sum())), np.array(float((v64 * v64).sum()))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    betas = [0.1, 0.55, 1.0]
    losses = []
    for s in range(steps):
        opt.zero_grad()
        torch.manual_seed(1000 + s)
        eps = torch.randn(B, Z)
        torch.manual_seed(1000 + s)
        logits, mu, logv = model(triples, seq[:, :-1])
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), seq[:, 1:].reshape(-1), ignore_index=0)
        kl = model.kl_mean(mu, logv)
        loss = ce + betas[s] * kl
        out[f"eps{s}"] = eps.numpy()
        loss.backward()
        losses.append([loss.item(), ce.item(), kl.item()])
        if s == 0:
            if full:
                out["logits0"], out["mu0"], out["logv0"] = logits.detach().numpy(), mu.detach().numpy(), logv.detach().numpy()
            for k, p in model.named_parameters():
                gnp = p.grad.detach().numpy()
                if full:
                    out["g0/" + k] = gnp
                out["g0norm/" + k] = np.array(float(np.sqrt((gnp.astype(np.float64) ** 2).sum())))
        opt.step()
        for k, v in model.state_dict().items():
            out[f"w{s + 1}sum/" + k] = np.array(float(v.detach().double().sum()))
    out["losses"] = np.array(losses, dtype=np.float64)
    out["betas"] = np.array(betas[:steps])
    # decode from the INITIAL weights: greedy and beam 2 (decode_latent, models.py:262-300)
    torch.manual_seed(seed)
    gen = quiet_model(M.SAIL, cfg)
    gen.eval()
    zs = torch.randn(6, Z, generator=torch.Generator().manual_seed(seed + 7))
    out["dec_z"] = zs.numpy()
    from tools.make_golden_r3 import triples_array
    for b in (1, 2):
        with torch.no_grad():
            trip = gen.decode_latent(zs, cfg["seq_len"], cfg["special_tokens"], U.seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=b)
        out[f"beam{b}/triples"], out[f"beam{b}/n"] = triples_array(trip)
    # posterior_bits of the first graphs (models.py:202-260) from the initial weights, with the noise those calls drew
    # (DataLoader iterator: one int64 from the global generator first -- tools/make_golden_r3.py)
    from tools.make_golden_r3 import dataset_of
    import contextlib, io
    ds, _, _ = dataset_of(U, cfg, B, seed + 100, padded)
    torch.manual_seed(900 + seed)
    with contextlib.redirect_stderr(io.StringIO()):
        stats = gen.posterior_bits(ds, "cpu", pad_id=0, sample_frac=min(1.0, 6 / B))
    nb = len(stats["records"])
    out["bits_ar"] = np.array([r["ar_bits"] for r in stats["records"]])
    out["bits_kl"] = np.array([r["kl_bits"] for r in stats["records"]])
    torch.manual_seed(900 + seed)
    torch.empty((), dtype=torch.int64).random_()
    out["bits_eps"] = torch.cat([torch.randn(1, Z) for _ in range(nb)]).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "losses", losses)


def main():
    torch.set_num_threads(4)
    M, U = import_reference()
    dump_tsail(M, U, "tsail_tiny", D=16, Z=4, heads=4, n=2, nE=20, nR=3, T=3, B=4, padded=False, seed=7)
    dump_tsail(M, U, "tsail_tiny_pad", D=16, Z=6, heads=2, n=2, nE=30, nR=4, T=5, B=6, padded=True, seed=8, full=False)
    dump_tsail(M, U, "tsail_small", D=64, Z=10, heads=4, n=3, nE=49, nR=3, T=3, B=32, padded=False, seed=9, full=False)
    dump_tark(M, U, "tark_tiny", D=32, heads=4, n=2, nE=20, nR=3, T=3, B=4, seed=5)
    dump_tark(M, U, "tark_small", D=64, heads=4, n=3, nE=49, nR=3, T=5, B=24, seed=6, full=False)
    dump_tark(M, U, "tark_synpaths_b32_s0", D=512, heads=4, n=3, nE=49, nR=3, T=3, B=32, seed=0, lr=1e-4, full=False)
    assert not any(d == "__pycache__" for _, ds, _ in os.walk(REF) for d in ds), "bytecode leaked into reference"


if __name__ == "__main__":
    main()
