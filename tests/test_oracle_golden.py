"""Pin the CPU oracle (oracle/sail_oracle.py) to golden vectors produced by the REAL reference
(tools/make_golden.py imported /root/reference's kgvae.model.models in the build container)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import sail_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg_json"]))
    return z, cfg


FULL = ["sail_tiny", "sail_tiny_pad", "ark_tiny", "sail_small", "sail_small_pad"]
SCALAR = ["sail_synpaths_b32_s0", "sail_synpaths_b32_s1", "ark_synpaths_b32_s0"]


@pytest.mark.parametrize("name", FULL + SCALAR)
def test_init_order_reproduces_reference_weights(name):
    z, cfg = load(name)
    P = O.init_params(cfg, int(z["seed"]))
    if name in FULL:
        for k, v in P.items():
            assert np.array_equal(v.numpy(), z["w0/" + k]), k
    assert P["dec.out.weight"] is P["dec.tok_emb.weight"]


@pytest.mark.parametrize("name", FULL)
def test_forward_backward_adam_match_reference(name):
    z, cfg = load(name)
    P = OrderedDictFrom(z, "w0/")
    triples, seq = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])
    state = O.adam_init(O.leaf_params(P))
    lr = float(z["lr"])
    for s in range(len(z["losses"])):
        eps = torch.from_numpy(z[f"eps{s}"]) if cfg["model_type"] == "SAIL" else None
        if s == 0:
            with torch.no_grad():
                if cfg["model_type"] == "SAIL":
                    _, _, _, logits, mu, logv = O.sail_elbo(P, triples, seq, eps, float(z["betas"][0]), cfg)
                    np.testing.assert_allclose(mu.numpy(), z["mu0"], rtol=1e-5, atol=1e-6)
                    np.testing.assert_allclose(logv.numpy(), z["logv0"], rtol=1e-5, atol=1e-6)
                else:
                    _, logits = O.ark_loss(P, seq, cfg)
                np.testing.assert_allclose(logits.numpy(), z["logits0"], rtol=1e-4, atol=2e-5)
        loss, ce, kl, grads = O.train_step(P, state, (triples, seq), cfg, lr, beta=float(z["betas"][s]), eps=eps)
        ref = z["losses"][s]
        assert abs(loss - ref[0]) <= 2e-6 * abs(ref[0]) + 1e-6, (s, loss, ref)
        assert abs(ce - ref[1]) <= 2e-6 * abs(ref[1]) + 1e-6
        assert abs(kl - ref[2]) <= 1e-5 * abs(ref[2]) + 1e-7
        if s == 0:
            for (k, _), g in zip(O.leaf_params(P), grads):
                np.testing.assert_allclose(g.numpy(), z["g0/" + k], rtol=2e-4, atol=2e-6, err_msg=k)
        if f"w{s + 1}/dec.out.bias" in z:
            for k, v in P.items():
                assert_close_frac(v.numpy(), z[f"w{s + 1}/" + k], lr, f"{k}@{s + 1}")


@pytest.mark.parametrize("name", SCALAR)
def test_full_size_scalars(name):
    torch.set_num_threads(8)
    z, cfg = load(name)
    P = O.init_params(cfg, int(z["seed"]))
    triples, seq = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])
    state = O.adam_init(O.leaf_params(P))
    for s in range(len(z["losses"])):
        eps = torch.from_numpy(z[f"eps{s}"]) if cfg["model_type"] == "SAIL" else None
        loss, ce, kl, grads = O.train_step(P, state, (triples, seq), cfg, float(z["lr"]), beta=float(z["betas"][s]),
                                           eps=eps)
        ref = z["losses"][s]
        assert abs(loss - ref[0]) <= 1e-5 * abs(ref[0]), (s, loss, ref)
        if s == 0:
            for (k, _), g in zip(O.leaf_params(P), grads):
                n = float(np.sqrt((g.numpy().astype(np.float64) ** 2).sum()))
                assert abs(n - float(z["g0norm/" + k])) <= 1e-4 * float(z["g0norm/" + k]) + 1e-9, k
        for k, v in P.items():
            want = float(z[f"w{s + 1}sum/" + k])
            assert abs(float(v.double().sum()) - want) <= 1e-5 * abs(want) + 1e-3, (k, s)


@pytest.mark.parametrize("name", ["sail_tiny", "sail_tiny_pad", "sail_small", "sail_small_pad"])
def test_greedy_decode_bit_exact(name):
    z, cfg = load(name)
    P = OrderedDictFrom(z, f"w{len(z['losses'])}/")
    toks = O.greedy_decode(P, torch.from_numpy(z["dec_z"]), cfg)
    for i in range(toks.shape[0]):
        tr = O.seq_to_triples(toks[i].tolist(), cfg["ENT_BASE"], cfg["REL_BASE"])
        n = int(z["dec_ntriples"][i])
        assert len(tr) == n
        assert [list(t) for t in tr] == z["dec_triples"][i, :n].tolist()


def test_codec_vectors():
    cases = json.load(open(os.path.join(GOLD, "codec.json")))
    for c in cases:
        T = (len(c["seq"]) - 2) // 3
        if c["triples"] is not None:
            assert O.triples_to_seq([tuple(t) for t in c["triples"]], 3, 13, len(c["seq"])) == c["seq"]
        assert [list(t) for t in O.seq_to_triples(c["seq"], 3, 13)] == c["decoded"]


def test_adam_matches_torch_optim():
    torch.manual_seed(0)
    p = torch.randn(50)
    q = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([q], lr=3e-3)
    st = O.adam_init([("p", p)])
    for i in range(5):
        g = torch.randn(50)
        q.grad = g.clone()
        opt.step()
        O.adam_step([("p", p)], [g], st, 3e-3)
    np.testing.assert_allclose(p.numpy(), q.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_schedules():
    sched_p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([sched_p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=1e-6)
    for ep in range(20):
        assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(1e-3, ep, 20)) < 1e-9
        opt.step()
        sch.step()
    assert O.beta_schedule({"beta0": 0.1, "beta1": 1.0, "num_epochs": 100}, 50) == pytest.approx(0.55)


def assert_close_frac(got, want, lr, msg):
    """post-Adam weights: an element whose gradient is ~1e-8 (Adam's eps) legitimately moves by
    anything in [-lr, lr] depending on last-bit gradient noise, so allow <=0.1% of elements to
    differ by up to 2*lr*steps; everything else must agree to 1e-4 relative."""
    bad = np.abs(got - want) > (1e-4 * np.abs(want) + 3e-6)
    assert bad.mean() <= 1e-3, (msg, bad.mean())
    assert np.abs(got - want).max() <= 8 * lr, (msg, np.abs(got - want).max())


def OrderedDictFrom(z, prefix):
    from collections import OrderedDict
    P = OrderedDict()
    for k in z.files:
        if k.startswith(prefix):
            P[k[len(prefix):]] = torch.from_numpy(z[k].copy())
    if "dec.out.weight" in P and np.array_equal(P["dec.out.weight"].numpy(), P["dec.tok_emb.weight"].numpy()):
        P["dec.out.weight"] = P["dec.tok_emb.weight"]
    return P


def test_posterior_bits_restatement_equals_teacher_forced_nll():
    """the oracle's prefix-by-prefix restatement of bits_per_sequence (reference models.py:202-213) against ONE
    teacher-forced decoder pass on the reference-generated golden batch: a causal GRU makes them the same numbers"""
    import math
    import torch.nn.functional as F
    z, cfg = load("sail_tiny_pad")
    P = O.init_params(cfg, int(z["seed"]))
    tri, seq, eps = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"]), torch.from_numpy(z["eps0"])
    ar, kl = O.posterior_bits(P, tri, seq, eps, cfg)
    with torch.no_grad():
        zz, mu, logv = O.encoder_forward(P, tri, eps, cfg)
        logits = O.decoder_forward(P, zz, seq[:, :-1], cfg)
        nll = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), seq[:, 1:].reshape(-1), ignore_index=0,
                              reduction="none").reshape(seq.shape[0], -1).sum(1) / math.log(2)
    np.testing.assert_allclose(ar, nll.numpy(), rtol=1e-5)
    assert all(k >= 0 for k in kl)
    # the golden fixture pins mu / logv, hence the KL bits
    want_kl = (-0.5 * (1 + z["logv0"] - z["mu0"] ** 2 - np.exp(z["logv0"])).sum(1)) / math.log(2)
    np.testing.assert_allclose(kl, want_kl, rtol=1e-4, atol=1e-7)


def test_sampling_distribution_properties():
    """ARK.generate's sampling rules (reference models.py:431-456) are inline code, not callable: the restatement is
    pinned by the properties the reference's code implies"""
    torch.manual_seed(0)
    logits = torch.randn(5, 23) * 2
    base = torch.softmax(logits, -1)
    p = O.sampling_distribution(logits)
    assert torch.allclose(p, base, atol=1e-7)
    pk = O.sampling_distribution(logits, top_k=4)
    assert ((pk > 0).sum(-1) == 4).all() and torch.allclose(pk.sum(-1), torch.ones(5), atol=1e-6)
    for b in range(5):   # the survivors are the 4 most probable tokens, in their original proportions
        top = base[b].topk(4).indices
        assert set(top.tolist()) == set(torch.nonzero(pk[b]).flatten().tolist())
        assert torch.allclose(pk[b, top] / pk[b, top].sum(), base[b, top] / base[b, top].sum(), atol=1e-6)
    pp = O.sampling_distribution(logits, top_p=0.6)
    for b in range(5):   # smallest descending-probability prefix whose mass EXCEEDS top_p
        sp, si = base[b].sort(descending=True)
        n = int((sp.cumsum(0) > 0.6).nonzero()[0]) + 1
        assert set(si[:n].tolist()) == set(torch.nonzero(pp[b]).flatten().tolist())
        assert float(sp[:n - 1].sum()) <= 0.6 < float(sp[:n].sum())
    pt = O.sampling_distribution(logits, temperature=0.5)
    assert torch.allclose(pt, torch.softmax(logits / 0.5, -1), atol=1e-7)


@pytest.mark.parametrize("name", ["sail_tiny", "sail_small", "sail_small_pad"])
@pytest.mark.parametrize("beam", [2, 3, 4])
def test_beam_decode_matches_reference_goldens(name, beam):
    """the oracle's restatement of the batch-shared beam (models.py:282-300) against triples decoded by the REAL
    reference's SAIL.decode_latent(z, beam) on the stored trained weights (tests/golden/beam_decode.npz, written by
    tools/make_golden_beam.py)"""
    z, cfg = load(name)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "beam_decode.npz"))
    P = OrderedDictFrom(z, f"w{len(z['losses'])}/")
    seqs = O.beam_decode(P, torch.from_numpy(z["dec_z"]), cfg, beam)
    want, n = g[f"{name}/beam{beam}/triples"], g[f"{name}/beam{beam}/n"]
    for i in range(seqs.shape[0]):
        got = O.seq_to_triples(seqs[i].tolist(), cfg["ENT_BASE"], cfg["REL_BASE"])
        assert [list(t) for t in got] == want[i, :int(n[i])].tolist(), (i, got)


def _npz(name):
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), allow_pickle=False)


def _case_cfg(g, name):
    import json
    return json.loads(str(g[f"{name}/cfg_json"]))


@pytest.mark.parametrize("name,beams", [("synpaths_d512", (1, 2, 4)), ("wdmovies_pad", (1, 2))])
def test_decode_at_baseline_sizes_from_seed_weights(name, beams):
    """greedy and beam decode at BASELINE sizes (D = 512 syn-paths; padded V = 24 101 / T = 23 wd-movies shape) against the
    REAL reference's decode_latent on seed-initialised weights (tests/golden/decode_init.npz, tools/make_golden_r3.py)"""
    g = _npz("decode_init.npz")
    cfg = _case_cfg(g, name)
    torch.set_num_threads(8)
    P = O.init_params(cfg, int(g[f"{name}/seed"]))
    zs = torch.from_numpy(g[f"{name}/z"])
    toks = O.greedy_decode(P, zs, cfg)
    assert np.array_equal(toks.numpy(), g[f"{name}/greedy_tokens"])
    for b in beams:
        seqs = toks if b == 1 else O.beam_decode(P, zs, cfg, b)
        want, n = g[f"{name}/beam{b}/triples"], g[f"{name}/beam{b}/n"]
        for i in range(seqs.shape[0]):
            got = O.seq_to_triples(seqs[i].tolist(), cfg["ENT_BASE"], cfg["REL_BASE"])
            assert [list(t) for t in got] == want[i, :int(n[i])].tolist(), (b, i)


@pytest.mark.parametrize("name", ["ark_tiny", "ark_synpaths"])
def test_ark_generate_reproduces_the_reference_samples(name):
    """ARK.generate's filtering AND draw order (models.py:407-471), token for token: the oracle under the seeds the
    golden script used against the sequences the REAL reference sampled (tests/golden/ark_sampling.npz)"""
    g = _npz("ark_sampling.npz")
    cfg = _case_cfg(g, name)
    P = O.init_params(cfg, int(g[f"{name}/seed"]))
    B = g[f"{name}/greedy"].shape[0]
    assert np.array_equal(O.ark_generate(P, cfg, B).numpy(), g[f"{name}/greedy"])
    for i, (temp, top_p, top_k) in enumerate(g[f"{name}/combos"]):
        torch.manual_seed(500 + i)
        got = O.ark_generate(P, cfg, B, sample=True, temperature=float(temp), top_p=float(top_p), top_k=int(top_k))
        assert np.array_equal(got.numpy(), g[f"{name}/seq{i}"]), (name, i, temp, top_p, top_k)


def test_sampling_distribution_two_statements_agree():
    """the vectorised filter that ark_generate draws from (pinned by the reference's samples above) against the
    loop-by-loop second statement of the same rules"""
    torch.manual_seed(1)
    logits = torch.randn(7, 31) * 2.5
    for temp, top_p, top_k in [(1.0, 0.0, 0), (0.6, 0.0, 0), (1.0, 0.85, 0), (1.0, 0.0, 6), (0.9, 0.7, 9), (1.4, 0.4, 0)]:
        a = O.sampling_distribution(logits, temp, top_p, top_k)
        b = O.sampling_distribution_loops(logits, temp, top_p, top_k)
        assert torch.allclose(a, b, atol=1e-6), (temp, top_p, top_k)


@pytest.mark.parametrize("name", ["sail_small", "sail_pad"])
def test_sail_posterior_bits_match_the_reference_records(name):
    """the oracle's restatement of SAIL.posterior_bits / bits_per_sequence against the per-item records the REAL
    reference produced (models.py:202-260), fed the latent noise the reference drew (tests/golden/posterior_bits.npz)"""
    g = _npz("posterior_bits.npz")
    cfg = _case_cfg(g, name)
    P = O.init_params(cfg, int(g[f"{name}/seed"]))
    n = len(g[f"{name}/ar_bits"])
    tri, seq = torch.from_numpy(g[f"{name}/triples"])[:n], torch.from_numpy(g[f"{name}/seq"])[:n]
    ar, kl = O.posterior_bits(P, tri, seq, torch.from_numpy(g[f"{name}/eps"]), cfg)
    np.testing.assert_allclose(ar, g[f"{name}/ar_bits"], rtol=2e-5)
    np.testing.assert_allclose(kl, g[f"{name}/kl_bits"], rtol=1e-4, atol=1e-7)
    tot = np.array(ar) + np.array(kl)
    np.testing.assert_allclose([tot.mean(), np.mean(ar), np.mean(kl), tot.min(), tot.max()], g[f"{name}/summary"], rtol=2e-5)


def test_ark_posterior_bits_match_the_reference_records():
    g = _npz("posterior_bits.npz")
    cfg = _case_cfg(g, "ark_tiny")
    P = O.init_params(cfg, int(g["ark_tiny/seed"]))
    n = len(g["ark_tiny/ar_bits"])
    ar = O.ark_posterior_bits(P, torch.from_numpy(g["ark_tiny/seq"])[:n], cfg)
    np.testing.assert_allclose(ar, g["ark_tiny/ar_bits"], rtol=2e-5)
    assert np.all(g["ark_tiny/kl_bits"] == 0)


# ---- Transformer variant t-ARK (reference models.py:349-366): the oracle's explicit-math restatement of the stock
# nn.TransformerEncoderLayer stack against goldens of the REAL reference (tools/make_golden_txf.py)
@pytest.mark.parametrize("name", ["tark_tiny", "tark_small", "tark_synpaths_b32_s0"])
def test_tark_oracle_matches_reference(name):
    z, cfg = load(name)
    torch.set_num_threads(8)
    P = O.init_params(cfg, int(z["seed"]))
    for k in [f[len("w0sum/"):] for f in z.files if f.startswith("w0sum/")]:   # bit-identical initial weights
        v = P[k].double()
        # (float64 sums of the SAME float32 values: equal up to the summation order of the threaded reduction)
        assert abs(float(v.sum()) - float(z["w0sum/" + k])) <= 1e-12 * v.numel() and \
            abs(float((v * v).sum()) - float(z["w0sq/" + k])) <= 1e-12 * v.numel(), k
    seq = torch.from_numpy(z["seq"])
    st = O.adam_init(O.leaf_params(P))
    for s in range(len(z["losses"])):
        loss, ce, kl, grads = O.train_step(P, st, (None, seq), cfg, float(z["lr"]))
        assert abs(loss - z["losses"][s][0]) <= 2e-5 * abs(z["losses"][s][0]), (s, loss, z["losses"][s])
        if s == 0:
            for (k, _), g in zip(O.leaf_params(P), grads):
                want = float(z["g0norm/" + k])
                assert abs(float(g.double().norm()) - want) <= 2e-4 * want + 1e-7, k
                if "g0/" + k in z.files:
                    np.testing.assert_allclose(g.numpy(), z["g0/" + k], rtol=2e-3, atol=2e-5 * (np.abs(z["g0/" + k]).max() + 1e-12))
            if "w1/dec.tok_emb.weight" in z.files:
                # Adam's first step is lr * sign(g): where the true gradient is zero (the key bias of a softmax attention:
                # a shift of every score of a row) rounding noise decides the sign -- compare where |g| is not noise
                for k, v in O.leaf_params(P):
                    g0 = np.abs(z["g0/" + k])
                    live = g0 > 1e-5 * (g0.max() + 1e-30)
                    np.testing.assert_allclose(v.numpy()[live], z["w1/" + k][live], rtol=1e-4, atol=2e-6)
        for k, v in O.leaf_params(P):
            want = float(z[f"w{s + 1}sum/" + k])   # (sums: loose enough for the +-lr sign noise of zero-gradient elements)
            assert abs(float(v.double().sum()) - want) <= 5e-4 * abs(want) + 2e-3 * (s + 1) * max(1.0, v.numel() / 64), (s, k)
    if "logits0" in z.files:
        P0 = O.init_params(cfg, int(z["seed"]))
        with torch.no_grad():
            lg = O.ark_forward(P0, seq[:, :-1], cfg)
        np.testing.assert_allclose(lg.numpy(), z["logits0"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["tark_tiny", "tark_small"])
def test_tark_generation_and_bits_match_reference(name):
    """greedy AND sampled generations (the reference's draw order) and the teacher-forced AR bits of t-ARK from the seed's
    initial weights, token for token / to 2e-5"""
    z, cfg = load(name)
    P = O.init_params(cfg, int(z["seed"]))
    B = z["gen_greedy"].shape[0]
    assert np.array_equal(O.ark_generate(P, cfg, B).numpy(), z["gen_greedy"])
    for i, (temp, top_p, top_k) in enumerate(z["gen_combos"]):
        torch.manual_seed(500 + i)
        got = O.ark_generate(P, cfg, B, sample=True, temperature=float(temp), top_p=float(top_p), top_k=int(top_k))
        assert np.array_equal(got.numpy(), z[f"gen_seq{i}"]), (name, i)
    n = len(z["bits_ar"])
    np.testing.assert_allclose(O.ark_posterior_bits(P, torch.from_numpy(z["seq"])[:n], cfg), z["bits_ar"], rtol=3e-5)


# ---- Transformer VAE t-SAIL (reference models.py:66-114, 160-170, 187-195): oracle restatement vs reference goldens
@pytest.mark.parametrize("name", ["tsail_tiny", "tsail_tiny_pad", "tsail_small"])
def test_tsail_oracle_matches_reference(name):
    z, cfg = load(name)
    torch.set_num_threads(8)
    P = O.init_params(cfg, int(z["seed"]))
    for k in [f[len("w0sum/"):] for f in z.files if f.startswith("w0sum/")]:
        v = P[k].double()
        assert abs(float(v.sum()) - float(z["w0sum/" + k])) <= 1e-12 * v.numel() and \
            abs(float((v * v).sum()) - float(z["w0sq/" + k])) <= 1e-12 * v.numel(), k
    tri, seq = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])
    if "logits0" in z.files:
        with torch.no_grad():
            loss, ce, kl, logits, mu, logv = O.sail_elbo(P, tri, seq, torch.from_numpy(z["eps0"]), float(z["betas"][0]), cfg)
        np.testing.assert_allclose(logits.numpy(), z["logits0"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(mu.numpy(), z["mu0"], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(logv.numpy(), z["logv0"], rtol=2e-4, atol=2e-6)
    st = O.adam_init(O.leaf_params(P))
    for s in range(len(z["losses"])):
        loss, ce, kl, grads = O.train_step(P, st, (tri, seq), cfg, float(z["lr"]), beta=float(z["betas"][s]),
                                           eps=torch.from_numpy(z[f"eps{s}"]))
        want = z["losses"][s]
        assert abs(loss - want[0]) <= 3e-5 * abs(want[0]) * (1 + 3 * s), (s, loss, want)
        assert abs(kl - want[2]) <= 1e-4 * abs(want[2]) * (1 + 3 * s) + 1e-7, (s, kl, want)
        if s == 0:
            for (k, _), g in zip(O.leaf_params(P), grads):
                wn = float(z["g0norm/" + k])
                assert abs(float(g.double().norm()) - wn) <= 3e-4 * wn + 2e-7, (k, float(g.double().norm()), wn)
                if "g0/" + k in z.files:
                    g0 = z["g0/" + k]
                    np.testing.assert_allclose(g.numpy(), g0, rtol=3e-3, atol=3e-5 * (np.abs(g0).max() + 1e-12) + 1e-8)


@pytest.mark.parametrize("name", ["tsail_tiny", "tsail_tiny_pad", "tsail_small"])
def test_tsail_decode_matches_reference(name):
    """greedy and beam-2 decode_latent of t-SAIL from the seed's initial weights, triple for triple"""
    z, cfg = load(name)
    P = O.init_params(cfg, int(z["seed"]))
    zs = torch.from_numpy(z["dec_z"])
    for b in (1, 2):
        seqs = O.greedy_decode(P, zs, cfg) if b == 1 else O.beam_decode(P, zs, cfg, b)
        want, n = z[f"beam{b}/triples"], z[f"beam{b}/n"]
        for i in range(seqs.shape[0]):
            got = O.seq_to_triples(seqs[i].tolist(), cfg["ENT_BASE"], cfg["REL_BASE"])
            assert [list(t) for t in got] == want[i, :int(n[i])].tolist(), (b, i)


@pytest.mark.parametrize("name", ["tsail_tiny", "tsail_tiny_pad", "tsail_small"])
def test_tsail_posterior_bits_match_reference(name):
    """the oracle's posterior_bits on t-SAIL against the reference's per-item records (AR bits by prefix re-runs of the
    Transformer decoder, KL bits summed over the latent), fed the noise the reference drew"""
    z, cfg = load(name)
    P = O.init_params(cfg, int(z["seed"]))
    n = len(z["bits_ar"])
    ar, kl = O.posterior_bits(P, torch.from_numpy(z["triples"])[:n], torch.from_numpy(z["seq"])[:n], torch.from_numpy(z["bits_eps"]), cfg)
    np.testing.assert_allclose(ar, z["bits_ar"], rtol=5e-5)
    np.testing.assert_allclose(kl, z["bits_kl"], rtol=2e-4, atol=1e-7)
