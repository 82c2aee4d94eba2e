"""Data adaptation for the ARK / SAIL training path (drop-in for the reference module of the same
import path: kgvae/model/utils.py).  Same public names and semantics:

    triples_to_seq      reference kgvae/model/utils.py:102-108
    seq_to_triples      reference kgvae/model/utils.py:70-78
    ints_to_labels      reference kgvae/model/utils.py:81-94
    canonicalize        reference kgvae/model/utils.py:96-99
    canonical_graph_string  reference kgvae/model/utils.py:66-67
    GraphSeqDataset     reference kgvae/model/utils.py:112-146

plus `GraphSeqDataset.tensorize()`, which builds the whole split once as two int64 arrays so a
GPU step measured in milliseconds is not starved by per-item Python.
Unlike the reference module, importing this file does not require `intelligraphs`.
"""
import random

import numpy as np
import torch
from torch.utils.data import Dataset

SPECIAL_TOKENS = {"PAD": 0, "BOS": 1, "EOS": 2}


def canonical_graph_string(graph):
    """order-independent text key of a graph (list of triples)"""
    return str(sorted(graph))


def seq_to_triples(seq, special_tokens, ENT_BASE, REL_BASE):
    """token sequence -> list of (h, r, t) ids.  Position 0 (BOS) is skipped; triples are read three
    tokens at a time and reading stops at the first EOS found at a triple boundary, or when fewer
    than three tokens remain."""
    toks = seq.tolist() if torch.is_tensor(seq) else list(seq)
    eos = special_tokens["EOS"]
    out = []
    pos = 1
    while pos + 2 < len(toks):
        if toks[pos] == eos:
            break
        out.append((toks[pos] - ENT_BASE, toks[pos + 1] - REL_BASE, toks[pos + 2] - ENT_BASE))
        pos += 3
    return out


def ints_to_labels(graphs, i2e, i2r):
    """id triples -> label triples; triples with an unknown id are dropped (and counted)"""
    labelled, dropped = [], 0
    for graph in graphs:
        keep = []
        for h, r, t in graph:
            if (h in i2e) and (r in i2r) and (t in i2e):
                keep.append((i2e[h], i2r[r], i2e[t]))
            else:
                dropped += 1
        labelled.append(keep)
    if dropped:
        print(f"[!] Skipped {dropped} invalid triples")
    return labelled


def canonicalize(triples, i2e=None, i2r=None, mode="keep"):
    """`keep` leaves the dataset order; any other mode sorts by (head, relation, tail) labels"""
    if mode == "keep":
        return triples
    return sorted(triples, key=lambda tr: (i2e[tr[0]], i2r[tr[1]], i2e[tr[2]]))


def triples_to_seq(triples, special_tokens, ENT_BASE, REL_BASE, seq_len):
    """[BOS, (ENT_BASE+h, REL_BASE+r, ENT_BASE+t) per triple, EOS, PAD ...] as a long tensor"""
    body = []
    for h, r, t in triples:
        body.extend((ENT_BASE + h, REL_BASE + r, ENT_BASE + t))
    toks = [special_tokens["BOS"], *body, special_tokens["EOS"]]
    toks.extend([special_tokens["PAD"]] * (seq_len - len(toks)))
    return torch.tensor(toks, dtype=torch.long)


def sample_perms(n, T):
    """n successive `random.sample(range(T), T)` draws from Python's global generator, bit for bit -- same
    permutations, same generator state afterwards -- computed in numpy instead of n interpreter-level calls
    (the per-epoch permutation redraw of GraphSeqDataset was the slowest part of an epoch on the GPU path).

    How CPython draws (Lib/random.py, Python 3.8-3.12): sample(pop, k=T) with T <= 21 + 4**ceil(log4(3T)) uses the
    pool method -- for i in range(T): j = _randbelow(T - i); result[i] = pool[j]; pool[j] = pool[T - i - 1] --
    and _randbelow(m) = getrandbits(m.bit_length()) until < m, getrandbits(k <= 32) = next 32-bit
    Mersenne-Twister output >> (32 - k).  numpy's MT19937 bit generator is the same generator, so its raw
    stream started from random.getstate() is the stream CPython would consume.  The only sequential part, "where
    does graph g start in the stream", is a function iteration x -> F[x]; its orbit is built by doubling."""
    if n == 0 or T <= 1:
        return np.zeros((n, max(T, 0)), dtype=np.int64)
    ver, internal, gauss = random.getstate()
    if ver != 3:
        return np.array([random.sample(range(T), T) for _ in range(n)], dtype=np.int64)
    key, pos0 = np.array(internal[:-1], dtype=np.uint32), int(internal[-1])
    needs = list(range(T, 0, -1))
    exp_draws = sum((1 << m.bit_length()) / m for m in needs)      # expected raw outputs per graph
    M = int(n * exp_draws * 1.10) + 4096
    while True:
        bg = np.random.MT19937()
        bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": pos0}}
        raw = bg.random_raw(M).astype(np.uint32)
        INF = M   # "no accepted draw left in this block"
        nxt, val = {}, {}
        for m in needs:
            r = raw >> np.uint32(32 - m.bit_length())
            ok = r < m
            idx = np.where(ok, np.arange(M), INF)
            nx = np.minimum.accumulate(idx[::-1])[::-1]             # first accepted position >= i
            nxt[m] = np.append(nx, INF)                              # (index M maps to INF)
            val[m] = r.astype(np.int64)
        # F[i] = stream position after one whole sample() that starts at position i
        F = np.arange(M + 1)
        for m in needs:
            a = nxt[m][np.minimum(F, M)]
            F = np.minimum(a + 1, M)
            F[a >= INF] = M
        # start position of every graph: orbit of 0 under F, by doubling
        starts = np.zeros(n, dtype=np.int64)
        have, Fp = 1, F
        while have < n:
            take = min(have, n - have)
            starts[have:have + take] = Fp[starts[:take]]
            have += take
            if have < n:
                Fp = Fp[Fp]
        end = F[starts[-1]]
        if end < M:   # every draw of every graph was found inside the block
            break
        M *= 2
    js = np.empty((n, T), dtype=np.int64)
    cur = starts.copy()
    for c, m in enumerate(needs):
        a = nxt[m][cur]
        js[:, c] = val[m][a]
        cur = a + 1
    pool = np.tile(np.arange(T, dtype=np.int64), (n, 1))
    out = np.empty((n, T), dtype=np.int64)
    rows = np.arange(n)
    for i in range(T):
        j = js[:, i]
        out[:, i] = pool[rows, j]
        pool[rows, j] = pool[:, T - i - 1]
    # leave Python's generator exactly where n sample() calls would have left it
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": pos0}}
    bg.random_raw(int(end))
    stt = bg.state["state"]
    random.setstate((3, tuple(int(x) for x in stt["key"]) + (int(stt["pos"]),), gauss))
    return out


class GraphSeqDataset(Dataset):
    """items are (triples long[T,3], seq long[seq_len]); same constructor as the reference."""

    def __init__(self, graphs, i2e, i2r, triple_order="keep", permute=False, use_padding=False, pad_eid=None,
                 pad_rid=None, max_triples=None, special_tokens=None, ent_base=None, rel_base=None, seq_len=None):
        self.graphs = [canonicalize(g, i2e, i2r, triple_order) for g in graphs]
        self.permute = permute
        self.use_padding = use_padding
        self.pad_eid, self.pad_rid = pad_eid, pad_rid
        self.max_triples = max_triples
        self.special_tokens = special_tokens if special_tokens is not None else dict(SPECIAL_TOKENS)
        self.ent_base, self.rel_base = ent_base, rel_base
        self.seq_len = seq_len
        self.fast_rng = None   # set to np.random.default_rng(seed) for vectorised permutations in tensorize()

    def __len__(self):
        return len(self.graphs)

    def _ordered(self, idx):
        triples = self.graphs[idx]
        # the per-item permutation draws from Python's `random` (not torch), and only without padding
        if self.permute and not self.use_padding:
            triples = random.sample(triples, k=len(triples))
        return triples

    def __getitem__(self, idx):
        triples = self._ordered(idx)
        if self.use_padding:
            fill = [(self.pad_eid, self.pad_rid, self.pad_eid)] * (self.max_triples - len(triples))
            t3 = torch.tensor(list(triples) + fill, dtype=torch.long)
        else:
            t3 = torch.tensor(triples, dtype=torch.long)
        return t3, triples_to_seq(triples, self.special_tokens, self.ent_base, self.rel_base, self.seq_len)

    def _base_arrays(self):
        """one-time numpy image of the split: triples [N,Tmax,3] (padded), lengths [N]"""
        if getattr(self, "_base", None) is None:
            n = len(self.graphs)
            lens = np.fromiter((len(g) for g in self.graphs), dtype=np.int64, count=n)
            tmax = int(self.max_triples) if self.use_padding else (int(lens.max()) if n else 0)
            base = np.zeros((n, tmax, 3), dtype=np.int64)
            if self.use_padding:
                base[:] = (self.pad_eid, self.pad_rid, self.pad_eid)
            for i, g in enumerate(self.graphs):
                if len(g):
                    base[i, :len(g)] = np.asarray(g, dtype=np.int64)
            self._base = (base, lens)
        return self._base

    def tensorize(self, indices=None):
        """(triples [N,T,3], seq [N,seq_len]) for `indices` (default: all) in one vectorised pass.
        Draws the same `random.sample` permutations, in the same order, as N successive __getitem__
        calls (random.sample picks POSITIONS, so sampling range(k) consumes the generator identically)."""
        base, lens = self._base_arrays()
        idx = np.arange(len(self.graphs)) if indices is None else np.asarray(list(indices), dtype=np.int64)
        n = len(idx)
        st = self.special_tokens
        tri = base[idx]
        ln = lens[idx]
        if not self.use_padding:
            if n and not (ln == ln[0]).all():
                raise ValueError("graphs of different sizes need use_padding=True to be batched")
            T = int(ln[0]) if n else 0
            tri = tri[:, :T]
            if self.permute and T > 1:
                if self.fast_rng is not None:   # opt-in: numpy generator, one vectorised draw (not Python-RNG identical)
                    perm = self.fast_rng.permuted(np.tile(np.arange(T), (n, 1)), axis=1)
                else:
                    perm = sample_perms(n, T)   # == [random.sample(range(T), T) for _ in range(n)], vectorised
                tri = np.take_along_axis(tri, perm[:, :, None], axis=1)
        T = tri.shape[1]
        seq = np.full((n, self.seq_len), st["PAD"], dtype=np.int64)
        seq[:, 0] = st["BOS"]
        body = (tri + np.array([self.ent_base, self.rel_base, self.ent_base], dtype=np.int64)).reshape(n, 3 * T)
        valid = np.arange(3 * T)[None, :] < (3 * ln)[:, None]
        seq[:, 1:1 + 3 * T] = np.where(valid, body, st["PAD"])
        seq[np.arange(n), 1 + 3 * ln] = st["EOS"]
        return torch.from_numpy(np.ascontiguousarray(tri)), torch.from_numpy(seq)
