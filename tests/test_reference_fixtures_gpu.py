"""Round-3 parity tests of the product path against outputs of the REAL reference (tests/golden/decode_init.npz,
ark_sampling.npz, posterior_bits.npz; written by tools/make_golden_r3.py in the build container): "bit-exact sampled
triple indices" at BASELINE sizes, ARK.generate's sampled tokens under the reference's seeds, and the reference's own
posterior_bits records.  Weights are regenerated from the seed on the GPU box (bit-identical initialisation, pinned in
tests/test_oracle_golden.py::test_init_order_reproduces_reference_weights)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _cfg(g, name, **over):
    return dict(json.loads(str(g[f"{name}/cfg_json"])), **over)


def _model(cfg, seed):
    from kgvae.model.models import ARK, SAIL
    torch.manual_seed(seed)
    return (SAIL if cfg["model_type"] == "SAIL" else ARK)(cfg).to("cuda")


@pytest.mark.parametrize("name,beams", [("synpaths_d512", (1, 2, 4)), ("wdmovies_pad", (1, 2))])
def test_decode_bit_exact_at_baseline_sizes(name, beams):
    """north_star's 'bit-exact on sampled triple indices' at BASELINE sizes: SAIL.decode_latent in exact-fp32 mode on the
    incremental per-beam GRU states, D = 512 syn-paths and the padded wd-movies shape (V = 24 101, T = 23, L = 70), equals
    the reference's prefix-re-running decode_latent triple for triple, greedy AND beam (greedy also token for token)"""
    from kgvae.model.utils import seq_to_triples
    g = _npz("decode_init.npz")
    cfg = _cfg(g, name, precision="f32")
    model = _model(cfg, int(g[f"{name}/seed"]))
    zs = torch.from_numpy(g[f"{name}/z"])
    eng = model.engine()
    toks = eng.greedy_decode(zs.to(eng.device), max_len=cfg["seq_len"] - 1).cpu().numpy()
    assert np.array_equal(toks, g[f"{name}/greedy_tokens"]), "greedy tokens differ from the reference's"
    st = cfg["special_tokens"]
    for b in beams:
        tri = model.decode_latent(zs, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=b)
        want, n = g[f"{name}/beam{b}/triples"], g[f"{name}/beam{b}/n"]
        for i, tl in enumerate(tri):
            assert [list(t) for t in tl] == want[i, :int(n[i])].tolist(), (b, i)
    assert float(g[f"{name}/greedy_margin"].min()) > 1e-4   # (the fixture's argmaxes are not near-ties)


@pytest.mark.parametrize("name", ["ark_tiny", "ark_synpaths"])
def test_ark_generate_host_draws_reproduce_the_reference_tokens(name):
    """ARK.generate(sample=True, host_draws=True): logits from the exact-fp32 one-token GRU step, the reference's filter
    and the reference's draw order on the host -> under the same torch.manual_seed the sampled sequences are the ones the
    REAL reference sampled on CPU, token for token, for every (temperature, top_p, top_k) of the fixture"""
    g = _npz("ark_sampling.npz")
    cfg = _cfg(g, name, precision="f32")
    model = _model(cfg, int(g[f"{name}/seed"]))
    model.eval()
    st = cfg["special_tokens"]
    B = g[f"{name}/greedy"].shape[0]
    assert np.array_equal(model.generate(cfg["seq_len"], st, batch_size=B).cpu().numpy(), g[f"{name}/greedy"])
    for i, (temp, top_p, top_k) in enumerate(g[f"{name}/combos"]):
        torch.manual_seed(500 + i)
        got = model.generate(cfg["seq_len"], st, batch_size=B, sample=True, temperature=float(temp), top_p=float(top_p),
                             top_k=int(top_k), host_draws=True)
        assert np.array_equal(got.cpu().numpy(), g[f"{name}/seq{i}"]), (name, i, temp, top_p, top_k)
    # device draws: same support as the filter allows (top-k = 3 -> at most 3 distinct continuations of BOS)
    torch.manual_seed(0)
    dev = model.generate(cfg["seq_len"], st, batch_size=64, sample=True, top_k=3).cpu()
    assert dev.shape == (64, cfg["seq_len"]) and len(set(dev[:, 1].tolist())) <= 3


def _dataset(cfg, triples, padded):
    from kgvae.model.utils import GraphSeqDataset
    pr = cfg["pad_rid"]
    graphs = [[tuple(int(x) for x in t) for t in gr if not padded or int(t[1]) != pr] for gr in triples]
    return GraphSeqDataset(graphs, None, None, use_padding=padded, pad_eid=cfg["pad_eid"], pad_rid=pr,
                           max_triples=triples.shape[1], special_tokens=cfg["special_tokens"], ent_base=cfg["ENT_BASE"],
                           rel_base=cfg["REL_BASE"], seq_len=cfg["seq_len"])


@pytest.mark.parametrize("name,padded", [("sail_small", False), ("sail_pad", True)])
def test_sail_posterior_bits_equal_the_reference_records(name, padded):
    """SAIL.posterior_bits (ONE teacher-forced pass per batch) against the per-item records of the reference's own
    posterior_bits (batch 1, one decoder run per prefix; models.py:202-260), with the noise the reference drew"""
    g = _npz("posterior_bits.npz")
    cfg = _cfg(g, name, precision="f32")
    model = _model(cfg, int(g[f"{name}/seed"]))
    ds = _dataset(cfg, g[f"{name}/triples"], padded)
    t2, s2 = ds.tensorize()
    assert np.array_equal(t2.numpy(), g[f"{name}/triples"]) and np.array_equal(s2.numpy(), g[f"{name}/seq"])
    stats = model.posterior_bits(ds, "cuda", sample_frac=float(g[f"{name}/frac"]), eps=torch.from_numpy(g[f"{name}/eps"]))
    np.testing.assert_allclose([r["ar_bits"] for r in stats["records"]], g[f"{name}/ar_bits"], rtol=3e-5)
    np.testing.assert_allclose([r["kl_bits"] for r in stats["records"]], g[f"{name}/kl_bits"], rtol=1e-4, atol=1e-7)
    got = [stats[k] for k in ("avg_total_bits", "avg_ar_bits", "avg_kl_bits", "min_total_bits", "max_total_bits")]
    np.testing.assert_allclose(got, g[f"{name}/summary"], rtol=3e-5)


def test_ark_posterior_bits_equal_the_reference_records():
    g = _npz("posterior_bits.npz")
    cfg = _cfg(g, "ark_tiny", precision="f32")
    model = _model(cfg, int(g["ark_tiny/seed"]))
    ds = _dataset(cfg, g["ark_tiny/triples"], False)
    stats = model.posterior_bits(ds, "cuda", sample_frac=1.0)
    np.testing.assert_allclose([r["ar_bits"] for r in stats["records"]], g["ark_tiny/ar_bits"], rtol=3e-5)
    assert stats["avg_kl_bits"] == 0.0


def test_decode_honours_special_token_ids():
    """BOS / EOS ids come from special_tokens (reference models.py:286,297), not from literals: with the roles of tokens
    1 and 2 swapped, greedy and beam decode start from token 2 and equal the oracle's decode with those ids"""
    from oracle import sail_oracle as O
    from tests.parity_util import load_golden, weights_from
    from kgvae.model.utils import seq_to_triples
    z, cfg = load_golden("sail_small")
    cfg = dict(cfg, precision="f32")
    model = _model(cfg, int(z["seed"]))
    W = weights_from(z, f"w{len(z['losses'])}/")
    model.load_state_dict(W)
    zs = torch.from_numpy(z["dec_z"])
    st = {"PAD": 0, "BOS": 2, "EOS": 1}
    for beam in (1, 3):
        want = O.greedy_decode(W, zs, cfg, bos=2, eos=1) if beam == 1 else O.beam_decode(W, zs, cfg, beam, bos=2, eos=1)
        tri = model.decode_latent(zs, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=beam)
        assert tri == [seq_to_triples(row, st, cfg["ENT_BASE"], cfg["REL_BASE"]) for row in want]
        assert int(want[0, 0]) == 2
