"""IntelliGraphs data access for the training entry point.

The reference loads data through the third-party `intelligraphs` package
(`intelligraphs.data_loaders.load_data_as_list(name)`, reference kgvae/experiments/train.py:328),
which downloads the datasets on first use.  This module exposes the same 7-tuple

    (train_g, val_g, test_g, (e2i, i2e), (r2i, i2r), (min_edges, max_edges), extra)

and uses the real package when it is importable (and its data present); otherwise it generates
IntelliGraphs-SHAPED synthetic graphs (uniform random ids) so the whole path can
be exercised offline.  The synthetic presets are parameters, not facts about the real datasets.
"""
import random

# (n_entities, n_relations, min_edges, max_edges) -- recalled from the IntelliGraphs paper, overridable
PRESETS = {
    "syn-paths": (49, 3, 3, 3),
    "syn-types": (30, 3, 3, 3),
    "syn-tipr": (130, 5, 5, 5),
    "wd-movies": (24093, 3, 2, 23),
    "wd-articles": (60932, 6, 4, 212),
}


def synthetic_as_list(name, n_train=4096, n_val=512, n_test=512, seed=1, shape=None):
    nE, nR, lo, hi = shape if shape is not None else PRESETS[name]
    rng = random.Random(seed)

    def graphs(n):
        out = []
        for _ in range(n):
            k = rng.randint(lo, hi)
            out.append([(rng.randrange(nE), rng.randrange(nR), rng.randrange(nE)) for _ in range(k)])
        return out

    e2i = {f"e{i}": i for i in range(nE)}
    r2i = {f"r{i}": i for i in range(nR)}
    i2e = {i: k for k, i in e2i.items()}
    i2r = {i: k for k, i in r2i.items()}
    return graphs(n_train), graphs(n_val), graphs(n_test), (e2i, i2e), (r2i, i2r), (lo, hi), None


def load_data_as_list(name, allow_synthetic=True, **synthetic_kwargs):
    """the reference's loader when available, else synthetic graphs of the same shape"""
    try:
        from intelligraphs.data_loaders import load_data_as_list as real
        return real(name)
    except Exception as exc:  # package or data absent (no network here)
        if not allow_synthetic:
            raise
        print(f"[data] intelligraphs unavailable ({type(exc).__name__}); using synthetic '{name}'-shaped graphs")
        return synthetic_as_list(name, **synthetic_kwargs)


def synthetic_batch(n_ent, n_rel, max_triples, batch, seed, padded=False, min_triples=None):
    """One IntelliGraphs-shaped batch as the tensors GraphSeqDataset would yield (vectorised):
    triples [B,T,3] and seq [B, 2+3T] int64 with uniform random ids; with `padded`, per-graph edge
    counts are uniform in [min_triples, max_triples] and the tail is filled with the pad ids
    (n_ent, n_rel), exactly as kgvae/model/utils.py does for use_padding datasets."""
    import torch
    g = torch.Generator().manual_seed(seed)
    T = max_triples
    ne2, nr2 = (n_ent + 1, n_rel + 1) if padded else (n_ent, n_rel)
    ent_base, rel_base = 3, 3 + ne2
    h = torch.randint(0, n_ent, (batch, T), generator=g)
    r = torch.randint(0, n_rel, (batch, T), generator=g)
    t = torch.randint(0, n_ent, (batch, T), generator=g)
    k = torch.randint(min_triples or 1, T + 1, (batch,), generator=g) if padded else torch.full((batch,), T)
    live = torch.arange(T).view(1, -1) < k.view(-1, 1)
    triples = torch.stack([h, r, t], -1)
    if padded:
        triples = torch.where(live.unsqueeze(-1), triples, torch.tensor([n_ent, n_rel, n_ent]))
    seq = torch.zeros(batch, 2 + 3 * T, dtype=torch.long)
    seq[:, 0] = 1
    body = torch.stack([ent_base + h, rel_base + r, ent_base + t], -1).reshape(batch, 3 * T)
    seq[:, 1:1 + 3 * T] = torch.where(live.repeat_interleave(3, dim=1), body, torch.zeros_like(body))
    seq[torch.arange(batch), 1 + 3 * k] = 2
    return triples, seq
