"""GPU parity of the whole HIP train step (ark_amd.engine through the C-ABI) against
(a) golden vectors generated from the real reference and (b) the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from tests.parity_util import load_golden, weights_from, make_engine, synth_batch, rel_err, smoke_check

pytestmark = pytest.mark.gpu

FULL = ["sail_tiny", "sail_tiny_pad", "ark_tiny", "sail_small", "sail_small_pad"]


def test_smoke():
    assert smoke_check()


@pytest.mark.parametrize("name", FULL)
def test_train_steps_match_reference_golden(name):
    """fp32 MFMA mode: loss/ce/kl of 3 consecutive steps, every gradient of step 0, weights after
    step 1 and 3 -- all against tensors produced by the reference itself."""
    z, cfg = load_golden(name)
    lr = float(z["lr"])
    eng = make_engine(cfg, weights_from(z, "w0/"), "f32", lr=lr)
    dev = eng.device
    triples, seq = torch.from_numpy(z["triples"]).to(dev), torch.from_numpy(z["seq"]).to(dev)
    sail = cfg["model_type"] == "SAIL"
    for s in range(len(z["losses"])):
        eps = torch.from_numpy(z[f"eps{s}"]).to(dev) if sail else None
        eng.set_hyper(beta=float(z["betas"][s]))
        w = eng.forward(triples if sail else None, seq, eps)
        if s == 0:
            # logits before the CE kernel overwrote them are not kept; check mu/logv and the loss
            if sail:
                np.testing.assert_allclose(w["mu"].cpu().numpy(), z["mu0"], rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(w["logv"].cpu().numpy(), z["logv0"], rtol=1e-4, atol=1e-6)
        eng._default_norms(seq.shape[0])
        eng.backward()
        out4 = w["out4"].cpu().numpy()
        ref = z["losses"][s]
        # tolerance: ELBO within 1e-5 relative in exact-fp32 mode (north_star asks 1e-4)
        assert rel_err(float(out4[0]), ref[0]) < 1e-5, (s, out4, ref)
        assert rel_err(float(out4[1]), ref[1]) < 1e-5
        if sail:
            assert rel_err(float(out4[2]), ref[2]) < 1e-4
        if s == 0:
            for k in [f[3:] for f in z.files if f.startswith("g0/")]:
                got = eng.g[k].cpu().numpy()
                want = z["g0/" + k]
                scale = np.abs(want).max() + 1e-12
                assert np.abs(got - want).max() <= 2e-4 * scale + 1e-7, (k, np.abs(got - want).max(), scale)
        eng.adam()
        if f"w{s + 1}/dec.out.bias" in z.files:
            for k, v in eng.p.items():
                want = z[f"w{s + 1}/" + k]
                got = v.cpu().numpy()
                bad = np.abs(got - want) > (1e-4 * np.abs(want) + 5e-6)
                assert bad.mean() <= 2e-3, (k, s, bad.mean())
                assert np.abs(got - want).max() <= 8 * lr, (k, s)


def test_kl_norm_default():
    z, cfg = load_golden("sail_tiny")
    eng = make_engine(cfg, weights_from(z, "w0/"), "f32")
    eng._default_norms(4)
    assert eng._hp["KL_NORM"] == pytest.approx(1.0 / (4 * cfg["d_latent"]))


@pytest.mark.parametrize("name", ["sail_synpaths_b32_s0", "sail_synpaths_b32_s1", "ark_synpaths_b32_s0"])
@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("mixed", 3e-4), ("bf16", 2e-3)])
def test_fullsize_synpaths_scalars(name, precision, tol):
    """syn-paths (D=512) B=32: weights re-created from the seed by the init-order-compatible
    oracle code; 3 steps of loss + per-parameter gradient norms against reference goldens.
    bf16 tolerance on this 320-token batch is statistical: 2e-3."""
    from oracle import sail_oracle as O
    z, cfg = load_golden(name)
    P = O.init_params(cfg, int(z["seed"]))
    eng = make_engine(cfg, P, precision, lr=float(z["lr"]))
    dev = eng.device
    sail = cfg["model_type"] == "SAIL"
    triples, seq = torch.from_numpy(z["triples"]).to(dev), torch.from_numpy(z["seq"]).to(dev)
    for s in range(len(z["losses"])):
        eps = torch.from_numpy(z[f"eps{s}"]).to(dev) if sail else None
        eng.set_hyper(beta=float(z["betas"][s]))
        out4 = eng.train_step(triples if sail else None, seq, eps).cpu().numpy()
        # every step in every precision (steps 1, 2 follow Adam updates made in that precision)
        assert rel_err(float(out4[0]), z["losses"][s][0]) < tol * (1 + 4 * s), (s, out4, z["losses"][s])
        if s == 0 and precision == "f32":
            for k in [f[7:] for f in z.files if f.startswith("g0norm/")]:
                n = float(eng.g[k].double().norm())
                assert rel_err(n, float(z["g0norm/" + k])) < 2e-4, k


@pytest.mark.parametrize("name", ["sail_tiny", "sail_tiny_pad", "sail_small", "sail_small_pad"])
def test_greedy_decode_bit_exact(name):
    """'bit-exact sampled triple indices': greedy decode (beam=1) in exact-fp32 MFMA mode equals the
    reference's decode_latent output token for token."""
    from oracle import sail_oracle as O
    z, cfg = load_golden(name)
    eng = make_engine(cfg, weights_from(z, f"w{len(z['losses'])}/"), "f32")
    toks = eng.greedy_decode(torch.from_numpy(z["dec_z"])).cpu()
    for i in range(toks.shape[0]):
        tr = O.seq_to_triples(toks[i].tolist(), cfg["ENT_BASE"], cfg["REL_BASE"])
        n = int(z["dec_ntriples"][i])
        assert len(tr) == n
        assert [list(t) for t in tr] == z["dec_triples"][i, :n].tolist()


def _big_cfg():
    _, cfg = load_golden("sail_synpaths_b32_s0")
    return cfg


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("mixed", 1e-4), ("bf16", 1e-3)])
def test_elbo_parity_b1024(precision, tol):
    """BASELINE config 1 (syn-paths, B=1024): ELBO of the HIP forward vs the CPU oracle on the same
    weights / batch / eps.  Tolerance = north_star's 1e-4 relative for the shipped fast mode
    ("mixed": fp16 forward operands), 1e-5 for exact fp32; pure-bf16 forward operands measure
    ~2.5e-4 here and are kept only as a documented comparison (tol 1e-3)."""
    from oracle import sail_oracle as O
    cfg = _big_cfg()
    P = O.init_params(cfg, 0)
    triples, seq = synth_batch(cfg, 1024, seed=1)
    torch.manual_seed(1000)
    eps = torch.randn(1024, cfg["d_latent"])
    with torch.no_grad():
        loss, ce, kl, *_ = O.sail_elbo(P, triples, seq, eps, 0.1, cfg)
    eng = make_engine(cfg, P, precision)
    eng.set_hyper(beta=0.1)
    out4 = eng.eval_loss(triples.to(eng.device), seq.to(eng.device), eps.to(eng.device)).cpu().numpy()
    assert rel_err(float(out4[0]), float(loss)) < tol, (out4, float(loss), float(ce), float(kl))


@pytest.mark.parametrize("precision,rel,cos", [("mixed", 1.5e-2, 0.9998), ("f32", 3e-4, 0.999999)])
def test_gradients_match_oracle_fast_path(precision, rel, cos):
    """every parameter's gradient of the shipped fast path (layer-diagonal kernels, 16-bit operands, token
    reduction on MFMA, fused latent backward, two queues) against the CPU oracle's autograd on the same
    weights / batch / eps at the syn-paths model size: norm-relative error and cosine per tensor.  The
    tolerance of the mixed mode is the bf16 operand rounding of the backward products (2^-9 per operand)."""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.0)
    P = O.init_params(cfg, 0)
    B = 128
    triples, seq = synth_batch(cfg, B, seed=6)
    torch.manual_seed(12)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, precision)
    dev = eng.device
    eng.set_hyper(beta=0.5)
    eng._default_norms(B)
    eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    got = {k: v.detach().double().cpu().clone() for k, v in eng.g.items()}
    Pc = O._detach_tied(P, True)
    leaves = O.leaf_params(Pc)
    for _, t in leaves:
        t.requires_grad_(True)
    loss, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.5, cfg)
    loss.backward()
    checked = 0
    for k, t in leaves:
        want = t.grad.double()
        g = got[k]
        nw = want.norm().item()
        if nw < 1e-12:
            assert g.norm().item() < 1e-9, k
            continue
        assert (g - want).norm().item() <= rel * nw, (k, (g - want).norm().item() / nw)
        assert torch.dot(g.flatten(), want.flatten()).item() / (g.norm().item() * nw) >= cos, k
        checked += 1
    assert checked >= 20


def _trained_weights(cfg, steps, lr, scale):
    """weights off the initialisation: `steps` Adam steps of the CPU oracle (B=128, the usual synthetic batches), then every
    GRU / MLP / projection matrix multiplied by `scale` (pre-activations well outside the init's +-1: fp16 operand
    range, saturating gates)"""
    from oracle import sail_oracle as O
    P = O._detach_tied(O.init_params(cfg, 0), True)
    state = O.adam_init(O.leaf_params(P))
    for s_ in range(steps):
        tri, seq = synth_batch(cfg, 128, seed=200 + s_ % 4)
        torch.manual_seed(300 + s_)
        O.train_step(P, state, (tri, seq), cfg, lr, beta=0.3, eps=torch.randn(128, cfg["d_latent"]))
    P = {k: v.detach().clone() for k, v in P.items()}
    if scale != 1.0:
        for k in P:
            if P[k].dim() == 2 and ("gru.weight" in k or "mlp" in k or "z_proj" in k):
                P[k] = P[k] * scale
    return P


@pytest.mark.parametrize("steps,lr,scale,tol,grel", [(40, 1e-3, 1.0, 1e-4, 1.5e-2), (10, 1e-3, 4.0, 2e-4, 2.5e-2)])
def test_mixed_precision_parity_on_trained_and_scaled_weights(steps, lr, scale, tol, grel):
    """the shipped mixed mode (fp16 forward / bf16 backward operands) away from the initialisation: weights after 40
    oracle Adam steps at lr 1e-3 (the loss has dropped by > 0.5), and weights with every hidden matrix scaled x4 (gate
    pre-activations of +-10 and more).  ELBO within north_star's 1e-4 (2e-4 for the scaled case) of the fp32 oracle and
    per-tensor gradients within the bf16 operand rounding, dropout off (same batch / eps)."""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.0)
    P0 = O.init_params(cfg, 0)
    P = _trained_weights(cfg, steps, lr, scale)
    moved = max((P[k] - P0[k]).abs().max().item() for k in P if k in P0 and P[k].shape == P0[k].shape)
    assert moved > 5e-3
    B = 256
    triples, seq = synth_batch(cfg, B, seed=9)
    torch.manual_seed(19)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, "mixed")
    dev = eng.device
    eng.set_hyper(beta=0.3)
    eng._default_norms(B)
    eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    val = float(eng.ws["out4"].cpu()[0])
    got = {k: v.detach().double().cpu().clone() for k, v in eng.g.items()}
    Pc = O._detach_tied(P, True)
    leaves = O.leaf_params(Pc)
    for _, t in leaves:
        t.requires_grad_(True)
    loss, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.3, cfg)
    loss.backward()
    assert rel_err(val, float(loss)) < tol, (val, float(loss))
    worst = 0.0
    for k, t in leaves:
        want = t.grad.double()
        nw = want.norm().item()
        if nw < 1e-12:
            continue
        e = (got[k] - want).norm().item() / nw
        worst = max(worst, e)
        assert e <= grel, (k, e)
    assert worst > 0


@pytest.mark.parametrize("B", [100, 7])
def test_ragged_batch_is_padded_onto_the_fast_path(B):
    """a batch that is not a multiple of 16 rows (the last batch of an epoch) runs the layer-diagonal fast path on
    all-PAD padding rows instead of the register-staged path: ELBO and every gradient against the CPU oracle on the
    real rows only (padding rows carry no target and no KL term)"""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.0)
    P = O.init_params(cfg, 0)
    triples, seq = synth_batch(cfg, B, seed=11)
    torch.manual_seed(13)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, "mixed")
    dev = eng.device
    eng.set_hyper(beta=0.5)
    eng._default_norms(B)
    eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    assert eng.ws["v2"] and eng._B % 16 == 0 and eng._n_valid == B
    eng.backward()
    torch.cuda.synchronize()
    val = eng.ws["out4"].cpu().numpy().copy()
    got = {k: v.detach().double().cpu().clone() for k, v in eng.g.items()}
    Pc = O._detach_tied(P, True)
    leaves = O.leaf_params(Pc)
    for _, t in leaves:
        t.requires_grad_(True)
    loss, ce, kl, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.5, cfg)
    loss.backward()
    assert rel_err(float(val[0]), float(loss)) < 1e-4, (val, float(loss))
    assert rel_err(float(val[2]), float(kl)) < 1e-3, (val, float(kl))   # (a KL of 3e-4 nats: fp16 operand noise)
    for k, t in leaves:
        want = t.grad.double()
        nw = want.norm().item()
        if nw < 1e-12:
            continue
        assert (got[k] - want).norm().item() <= 1.5e-2 * nw, (k, (got[k] - want).norm().item() / nw)
    # evaluation entry point on the same ragged batch
    out = eng.eval_loss(triples.to(dev), seq.to(dev), eps.to(dev)).cpu().numpy()
    assert rel_err(float(out[0]), float(loss)) < 1e-4


def test_loss_trajectory_matches_oracle_fast_path():
    """12 consecutive optimiser steps of the shipped fast path (captured hipGraph, mixed precision, diagonal
    kernels, 16-bit shadows refreshed after every Adam) against the fp32 CPU oracle on the same batches and
    eps: the ELBO of every step stays within 2e-4 relative (first step 1e-4; measured <= 1.2e-4 throughout), i.e. gradients, Adam state and the
    shadow refresh carry over correctly from step to step."""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.0, learning_rate=2e-4)
    steps, B = 12, 128
    P = O.init_params(cfg, 0)
    eng = make_engine(cfg, P, "mixed", lr=2e-4)
    eng.set_hyper(beta=0.3)
    dev = eng.device
    Pc = O._detach_tied(P, True)
    state = O.adam_init(O.leaf_params(Pc))
    batches = []
    for s_ in range(steps):
        tri, seq = synth_batch(cfg, B, seed=30 + s_ % 3)
        torch.manual_seed(70 + s_)
        batches.append((tri, seq, torch.randn(B, cfg["d_latent"])))
    tri_in, seq_in, eps_in = (x.clone().to(dev) for x in batches[0])
    st = torch.cuda.Stream()
    got, want = [], []
    with torch.cuda.stream(st):
        step = eng.capture_train_step(tri_in, seq_in, eps_in)
        eng.load_params(P)          # the capture's warm-up step moved the weights
        eng.reset_optimizer()
        eng.refresh_shadows()
        for (tri, seq, eps) in batches:
            tri_in.copy_(tri.to(dev)); seq_in.copy_(seq.to(dev)); eps_in.copy_(eps.to(dev))
            got.append(step().cpu().numpy().copy())
    for (tri, seq, eps) in batches:
        loss, ce, kl, _ = O.train_step(Pc, state, (tri, seq), cfg, 2e-4, beta=0.3, eps=eps)
        want.append(loss)
    assert want[-1] < want[0] - 0.05         # it trains
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert rel_err(float(g_[0]), w_) < (1e-4 if i == 0 else 2e-4), (i, float(g_[0]), w_)


def test_shard_gradients_sum_to_full_batch():
    """size-independent DP property at B=1024: gradients of 4 shards (global CE count and global
    KL normaliser) sum to the full-batch gradient."""
    from oracle import sail_oracle as O
    cfg = _big_cfg()
    P = O.init_params(cfg, 0)
    B = 1024
    triples, seq = synth_batch(cfg, B, seed=2)
    torch.manual_seed(7)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, "f32")
    dev = eng.device
    triples, seq, eps = triples.to(dev), seq.to(dev), eps.to(dev)
    eng._default_norms(B)
    eng.forward(triples, seq, eps)
    eng.backward()
    torch.cuda.synchronize()
    full = eng.G.clone()
    count = float(eng.hyper[4])
    acc = torch.zeros_like(full)
    for k in range(4):
        sl = slice(k * 256, (k + 1) * 256)
        eng.set_hyper(kl_norm=1.0 / (B * cfg["d_latent"]), ce_count=count)
        eng.forward(triples[sl].contiguous(), seq[sl].contiguous(), eps[sl].contiguous(), ce_count=count)
        eng.backward()
        acc += eng.G
    scale = full.abs().max()
    assert (acc - full).abs().max() <= 2e-4 * scale, ((acc - full).abs().max(), scale)


@pytest.mark.parametrize("dropout", [0.0])
def test_dp_schedule_two_virtual_ranks(dropout):
    """Engine._dp_steps (the bucket-by-bucket data-parallel schedule) on two virtual ranks in one process: the
    ranks walk the schedule in lockstep, every yielded bucket is summed across them (what the all-reduce does),
    then both run Adam.  The buckets must tile the gradient buffer, the ranks must stay identical, and the
    result must equal the single-process step on the full batch."""
    from oracle import sail_oracle as O
    from ark_amd.engine import Engine
    cfg = dict(_big_cfg(), dec_dropout=dropout, learning_rate=1e-3)
    P = O.init_params(cfg, 0)
    B = 256
    triples, seq = synth_batch(cfg, B, seed=4)
    torch.manual_seed(9)
    eps = torch.randn(B, cfg["d_latent"])
    count = int((seq[:, 1:] != 0).sum())
    full = make_engine(cfg, P, "mixed", lr=1e-3)
    dev = full.device
    tri_d, seq_d, eps_d = triples.to(dev), seq.to(dev), eps.to(dev)
    full.train_step(tri_d, seq_d, eps_d, ce_count=count)
    ranks = []
    for k in range(2):
        e = Engine(dict(cfg, ark_dp_bf16=False), dev, precision="mixed", world_size=2)   # (the test sums the fp32 buckets itself)
        e.load_params(P)
        e.set_hyper(lr=1e-3)
        ranks.append(e)
    sl = [slice(0, B // 2), slice(B // 2, B)]
    gens = []
    for e, s_ in zip(ranks, sl):
        e._default_norms(B // 2)
        e.set_hyper(ce_count=count)
        gens.append(e._dp_steps(tri_d[s_].contiguous(), seq_d[s_].contiguous(), eps_d[s_].contiguous(), count))
    covered = []
    for spans in zip(*gens):
        assert spans[0] == spans[1]
        if spans[0] == "seam":   # between the encoder and decoder halves of the forward pass
            continue
        lo, hi = spans[0]
        torch.cuda.synchronize()
        tot = ranks[0].G[lo:hi] + ranks[1].G[lo:hi]
        ranks[0].G[lo:hi] = tot
        ranks[1].G[lo:hi] = tot
        covered.append((lo, hi))
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == full.layout.total
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
    assert len(covered) == 2
    gd = (ranks[0].G - full.G).abs().max().item()
    assert gd <= 3e-3 * full.G.abs().max().item(), gd
    ranks[0].adam()
    ranks[1]._dp_tick()          # the pipelined step's tick + two half updates must equal one adam()
    ranks[1]._adam_part("enc")
    ranks[1]._adam_part("dec")
    torch.cuda.synchronize()
    assert ranks[0].adam_steps == ranks[1].adam_steps == 1
    for a_, b_ in zip(ranks[0].wih16 + ranks[0].wm16, ranks[1].wih16 + ranks[1].wm16):
        assert torch.equal(a_, b_)   # 16-bit weight shadows refreshed by both routes
    assert torch.equal(ranks[0].P, ranks[1].P)
    assert (ranks[0].P - full.P).abs().max().item() <= 2.5e-3   # Adam's first step moves every weight by ~lr


def test_graph_replay_matches_eager():
    from oracle import sail_oracle as O
    cfg = _big_cfg()
    P = O.init_params(cfg, 0)
    B = 256
    triples, seq = synth_batch(cfg, B, seed=3)
    torch.manual_seed(8)
    eps = torch.randn(B, cfg["d_latent"])
    cfg = dict(cfg, dec_dropout=0.0)
    a = make_engine(cfg, P, "f32", lr=1e-4)
    b = make_engine(cfg, P, "f32", lr=1e-4)
    dev = a.device
    triples, seq, eps = triples.to(dev), seq.to(dev), eps.to(dev)
    la = [float(a.train_step(triples, seq, eps)[0]) for _ in range(4)]
    replay = b.capture_train_step(triples, seq, eps)  # performs 1 eager warm-up step itself
    lb = [la[0]] + [float(replay()[0]) for _ in range(3)]
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    assert la[-1] < la[0]


def _tile_native_index(rows, D):
    """row-major (row, col) -> offset in the MFMA-tile-native order (include/ark_amd.h)"""
    r = torch.arange(rows).view(-1, 1)
    c = torch.arange(D).view(1, -1)
    return ((r >> 4) * (D >> 4) + (c >> 4)) * 256 + ((((r >> 2) & 3) << 4) + (c & 15)) * 4 + (r & 3)


def _row_major_masks(eng, B, Lq):
    """the dropout masks of the current draw, as the oracle wants them: [B, L, D] per inter-layer gap, scaled by
    1/(1-p).  ark_dropout_mask materialises the counter hash in the engine's element order (time-major rows, MFMA
    tile-native inside a timestep block)."""
    from ark_amd import _lib as L
    dev, D = eng.device, eng.D
    idx = _tile_native_index(B, D).to(dev)
    out = []
    for l in range(eng.n - 1):
        m = torch.empty(Lq * B * D, device=dev)
        L.check(L.lib().ark_dropout_mask(L.ptr(m), L.i64(Lq * B * D), L.f32(eng.p_drop), L.u64(eng._layer_seed(l)),
                                         L.ptr(eng.hyper), L.cur_stream()), "mask")
        m = m.view(Lq, B * D)[:, idx.reshape(-1)].view(Lq, B, D)      # (t, b, d), row-major inside the block
        out.append(m.permute(1, 0, 2).contiguous().cpu())
    return out


def test_in_kernel_dropout_matches_the_materialised_mask():
    """fast path: the forward cells drop exactly the elements ark_dropout_mask marks for the same draw
    (seed, draw counter, element index), and every training forward draws a NEW mask"""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.5)
    P = O.init_params(cfg, 0)
    B = 64
    triples, seq = synth_batch(cfg, B, seed=4)
    eng = make_engine(cfg, P, "mixed")
    dev = eng.device
    eng.training = True
    eng._default_norms(B)
    args = (triples.to(dev), seq.to(dev), torch.randn(B, cfg["d_latent"]).to(dev))
    w = eng.forward(*args)
    assert w["v2"] and eng.dropout_draws() == 1
    masks = _row_major_masks(eng, B, eng.L)
    kept = []
    for l in range(eng.n - 1):
        y = w["Y16a"][l][B:B + eng.L * B].view(torch.float16).float().view(eng.L, B, eng.D).permute(1, 0, 2).cpu()
        yd = w["Yd16a"][l][:eng.L * B].view(torch.float16).float().view(eng.L, B, eng.D).permute(1, 0, 2).cpu()
        m = masks[l]
        assert set(m.unique().tolist()) == {0.0, 2.0} and 0.45 < (m == 0).float().mean().item() < 0.55
        assert torch.allclose(yd, y * m, rtol=2e-3, atol=1e-4)
        kept.append(m != 0)
    eng.forward(*args)            # the next training forward must not reuse the mask
    assert eng.dropout_draws() == 2
    for l, m in enumerate(_row_major_masks(eng, B, eng.L)):
        assert ((m != 0) != kept[l]).float().mean().item() > 0.3
    eng.training = False
    eng.forward(*args)            # evaluation draws nothing
    assert eng.dropout_draws() == 2


@pytest.mark.parametrize("B", [128, 1024])
def test_dropout_on_fast_path_matches_oracle_with_the_same_masks(B):
    """THE benchmarked configuration (syn-paths, mixed precision, dec_dropout 0.1, in-kernel counter-hash masks):
    ELBO and every parameter's gradient against the CPU oracle fed the SAME masks (reference semantics:
    nn.GRU(dropout=p), kgvae/model/models.py:121-127,184; loss kgvae/experiments/ablation_study.py:59-73)."""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=0.1)
    P = O.init_params(cfg, 0)
    triples, seq = synth_batch(cfg, B, seed=11)
    torch.manual_seed(13)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, "mixed")
    dev = eng.device
    eng.training = True
    eng.set_hyper(beta=0.1)
    eng._default_norms(B)
    eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    out4 = eng.ws["out4"].cpu().numpy()
    got = {k: v.detach().double().cpu().clone() for k, v in eng.g.items()}
    masks = _row_major_masks(eng, B, eng.L)
    Pc = O._detach_tied(P, True)
    leaves = O.leaf_params(Pc)
    for _, t in leaves:
        t.requires_grad_(True)
    loss, ce, kl, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.1, cfg, drop_masks=masks)
    with torch.no_grad():
        plain, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.1, cfg)
    assert abs(float(plain) - float(loss)) > 1e-3 * float(loss)          # the masks matter
    assert rel_err(float(out4[0]), float(loss)) < 1e-4, (out4, float(loss), float(ce), float(kl))   # north_star's bar
    loss.backward()
    checked = 0
    for k, t in leaves:
        want = t.grad.double()
        g = got[k]
        nw = want.norm().item()
        if nw < 1e-12:
            assert g.norm().item() < 1e-9, k
            continue
        assert (g - want).norm().item() <= 1.5e-2 * nw, (k, (g - want).norm().item() / nw)
        assert torch.dot(g.flatten(), want.flatten()).item() / (g.norm().item() * nw) >= 0.9998, k
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("rows,ki,nbuf,xcd,units,bcols", [(32, 2, 2, 0, 32, 0), (64, 1, 2, 0, 32, 64), (64, 2, 2, 1, 32, 0),
                                                         (32, 1, 4, 1, 32, 64), (64, 1, 2, 0, 64, 0),
                                                         (64, 2, 2, 1, 64, 0),
                                                         (32, 2, 2, 1, 32, 32), (32, 1, 2, 0, 32, 32), (32, 2, 2, 1, 16, 32), (32, 1, 2, 0, 16, 0)])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_diagonal_tilings_agree(rows, ki, nbuf, xcd, units, bcols, drop):
    """every tile / ring configuration of the two diagonal kernels (ArkDiagTuning, passed per call) computes the
    same states, losses and gradients as the default one, and the same ELBO as the CPU oracle within north_star's
    tolerance"""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=drop)
    P = O.init_params(cfg, 0)
    B = 256
    triples, seq = synth_batch(cfg, B, seed=3)
    torch.manual_seed(5)
    eps = torch.randn(B, cfg["d_latent"])
    tun = dict(fwd_rows=rows, fwd_ki=ki, fwd_nbuf=nbuf, fwd_xcd=xcd, fwd_units=units, bwd_rows=rows,
               bwd_ki=ki, bwd_nbuf=nbuf, bwd_xcd_rows=4 if xcd else 1, bwd_cols=bcols)
    a = make_engine(cfg, P, "mixed")
    b = make_engine(dict(cfg, ark_diag_tuning=tun), P, "mixed")
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    for eng in (a, b):
        eng.set_hyper(beta=0.1)
    a.drop_seed = b.drop_seed = 1234
    if drop == 0.0:   # before any optimiser step: the oracle sees the same weights
        with torch.no_grad():
            loss, *_ = O.sail_elbo(P, triples, seq, eps, 0.1, cfg)
        out = b.eval_loss(*args).cpu().numpy()
        assert rel_err(float(out[0]), float(loss)) < 1e-4, (out, float(loss))
    oa = a.train_step(*args).cpu().numpy()
    ob = b.train_step(*args).cpu().numpy()
    torch.cuda.synchronize()
    assert rel_err(float(ob[0]), float(oa[0])) < 2e-5, (oa, ob)
    n = cfg["n_layers"]
    for l in range(n):
        ya, yb = a.ws["Y"][l], b.ws["Y"][l]
        assert (ya - yb).abs().max().item() < 2e-3, l
        assert torch.equal(a.ws["Y16a"][l][:B], b.ws["Y16a"][l][:B])
    for k in a.g:   # every parameter's gradient, relative to its own norm
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 3e-3 * da.norm().item() + 1e-9, k


@pytest.mark.parametrize("chains", [2, 4])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_row_block_chains_agree_with_one_chain(chains, drop):
    """the diagonal sweeps split into independent row-block chains on parallel queues (Engine._chains) compute exactly what
    the single chain of dependent launches computes: same states bit for bit (the per-row arithmetic does not change),
    same loss, same gradients up to the order of the bias-gradient atomics; eager and inside a captured graph"""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=drop)
    P = O.init_params(cfg, 0)
    B = 1024
    triples, seq = synth_batch(cfg, B, seed=3)
    torch.manual_seed(5)
    eps = torch.randn(B, cfg["d_latent"])
    a = make_engine(dict(cfg, ark_diag_chains=1), P, "mixed")
    b = make_engine(dict(cfg, ark_diag_chains=chains), P, "mixed")
    assert len(a._chains(B)) == 1 and len(b._chains(B)) == chains
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    for eng in (a, b):
        eng.set_hyper(beta=0.1)
    a.drop_seed = b.drop_seed = 1234
    oa = a.train_step(*args).cpu().numpy()
    ob = b.train_step(*args).cpu().numpy()
    torch.cuda.synchronize()
    assert rel_err(float(ob[0]), float(oa[0])) < 1e-6, (oa, ob)
    for l in range(cfg["n_layers"]):
        assert torch.equal(a.ws["Y"][l], b.ws["Y"][l]), l
        assert torch.equal(a.ws["dG16"][l][:B * (cfg["seq_len"] - 1)], b.ws["dG16"][l][:B * (cfg["seq_len"] - 1)]), l
    for k in a.g:
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 1e-4 * da.norm().item() + 1e-9, k
    # captured: replays of the forked / joined graph keep reproducing the eager step
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = b.capture_train_step(*args)
        b.load_params(P)
        b.reset_optimizer()
        b.refresh_shadows()
        b.set_dropout_draws(0)
        og = step().cpu().numpy()
    torch.cuda.synchronize()
    assert rel_err(float(og[0]), float(oa[0])) < 1e-6, (oa, og)


@pytest.mark.parametrize("B,D,n_roles", [(64, 128, 1), (48, 256, 3)])
def test_gru_diag_fwd_matches_torch_gru_cell(B, D, n_roles):
    """ark_gru_diag_fwd through the C-ABI against torch.nn.functional GRU-cell math in fp32 on the same fp16-rounded
    operands: every role of one launch, new state (tile-native fp32 + row-major 16-bit copies) and the gate saves"""
    from ark_amd import _lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + D)
    idx = _tile_native_index(B, D).reshape(-1)
    roles = (L.GruDiagRole * L.DIAG_MAX_ROLES)()
    keep, refs = [], []
    for k in range(n_roles):
        x = (torch.randn(B, D, generator=g) * 0.5).half()
        h = torch.tanh(torch.randn(B, D, generator=g))
        wih = (torch.randn(3 * D, D, generator=g) * 0.1).half()
        whh = (torch.randn(3 * D, D, generator=g) * 0.1).half()
        bih, bhh = torch.randn(3 * D, generator=g) * 0.1, torch.randn(3 * D, generator=g) * 0.1
        h16 = h.half()
        gi = x.float() @ wih.float().t() + bih
        gh = h16.float() @ whh.float().t() + bhh
        r = torch.sigmoid(gi[:, :D] + gh[:, :D])
        z = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
        n = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
        hn = (1 - z) * n + z * h
        refs.append((hn, r, z, n, gh[:, 2 * D:]))
        y_prev_t = torch.zeros(B * D)
        y_prev_t[idx] = h.reshape(-1)
        bufs = dict(x=x.to(dev), h16=h16.to(dev), wih=wih.to(dev), whh=whh.to(dev), bih=bih.to(dev), bhh=bhh.to(dev),
                    yp=y_prev_t.to(dev), yo=torch.zeros(B * D, device=dev), ya=torch.zeros(B, D, dtype=torch.float16, device=dev),
                    yb=torch.zeros(B, D, dtype=torch.bfloat16, device=dev),
                    sv=[torch.zeros(B * D, dtype=torch.float16, device=dev) for _ in range(4)])
        keep.append(bufs)
        ro = roles[k]
        ro.x16, ro.h_prev16, ro.w_ih16, ro.w_hh16 = (L.dptr(bufs[n_]) for n_ in ("x", "h16", "wih", "whh"))
        ro.b_ih, ro.b_hh, ro.y_prev_t, ro.y_out_t = (L.dptr(bufs[n_]) for n_ in ("bih", "bhh", "yp", "yo"))
        ro.y16a, ro.y16b = L.dptr(bufs["ya"]), L.dptr(bufs["yb"])
        ro.save_r, ro.save_z, ro.save_n, ro.save_hn = (L.dptr(t) for t in bufs["sv"])
        ro.drop_p = 0.0
    L.check(L.lib().ark_gru_diag_fwd(L.i32(L.PREC_F16), L.i32(L.PREC_BF16), L.i32(n_roles), roles, L.ptr(None), L.i32(B), L.i32(D),
                                     L.ptr(None), L.cur_stream()), "ark_gru_diag_fwd")
    torch.cuda.synchronize()
    for bufs, (hn, r, z, n, ghn) in zip(keep, refs):
        got = bufs["yo"].cpu()[idx].reshape(B, D)
        assert (got - hn).abs().max().item() < 2e-5
        assert (bufs["ya"].float().cpu() - hn).abs().max().item() < 1e-3       # fp16 copy
        assert (bufs["yb"].float().cpu() - hn).abs().max().item() < 8e-3       # bf16 copy
        for sv, want in zip(bufs["sv"], (r, z, n, ghn)):
            assert (sv.float().cpu()[idx].reshape(B, D) - want).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,D,V", [(64, 128, 55), (48, 256, 36)])
def test_gru_diag_fwd_token_table_role(B, D, V):
    """layer 0 of a small vocabulary through the C-ABI: a role with x_tab / x_tok (rows of W_tok16 W_ih16^T by token id,
    formed by ark_gemm16 as the engine does) beside an ordinary role in ONE launch, both against fp32 GRU-cell math on the
    same fp16-rounded operands.  Reference: the input half of nn.GRU, kgvae/model/models.py:121-127, on nn.Embedding rows
    (:119, :137)."""
    from ark_amd import _lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + D + V)
    idx = _tile_native_index(B, D).reshape(-1)
    wtok = (torch.randn(V, D, generator=g) * 0.5).half()
    tok = torch.randint(0, V, (B,), generator=g)
    roles = (L.GruDiagRole * L.DIAG_MAX_ROLES)()
    keep, refs = [], []
    for k in range(2):
        x = wtok[tok] if k == 0 else (torch.randn(B, D, generator=g) * 0.5).half()
        h = torch.tanh(torch.randn(B, D, generator=g))
        wih = (torch.randn(3 * D, D, generator=g) * 0.1).half()
        whh = (torch.randn(3 * D, D, generator=g) * 0.1).half()
        bih, bhh = torch.randn(3 * D, generator=g) * 0.1, torch.randn(3 * D, generator=g) * 0.1
        h16 = h.half()
        gi = x.float() @ wih.float().t() + bih
        gh = h16.float() @ whh.float().t() + bhh
        r = torch.sigmoid(gi[:, :D] + gh[:, :D])
        z = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
        n = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
        refs.append(((1 - z) * n + z * h, r, z, n, gh[:, 2 * D:]))
        y_prev_t = torch.zeros(B * D)
        y_prev_t[idx] = h.reshape(-1)
        bufs = dict(x=x.to(dev), h16=h16.to(dev), wih=wih.to(dev), whh=whh.to(dev), bih=bih.to(dev), bhh=bhh.to(dev),
                    yp=y_prev_t.to(dev), yo=torch.zeros(B * D, device=dev), ya=torch.zeros(B, D, dtype=torch.float16, device=dev),
                    sv=[torch.zeros(B * D, dtype=torch.float16, device=dev) for _ in range(4)])
        ro = roles[k]
        if k == 0:
            bufs["wtok"], bufs["tok"] = wtok.to(dev), tok.to(dev, dtype=torch.int32)
            bufs["tab"] = torch.full((V, 3 * D), float("nan"), device=dev)
            L.check(L.lib().ark_gemm16(L.i32(L.PREC_F16), L.i32(L.EPI_NONE), L.ptr(bufs["wtok"]), L.i64(D), L.ptr(bufs["wih"]), L.i64(D),
                                       L.ptr(bufs["tab"]), L.i64(3 * D), L.ptr(None), L.ptr(None), L.i32(V), L.i32(3 * D), L.i32(D),
                                       L.i32(0), L.cur_stream()), "ark_gemm16")
            ro.x_tab, ro.x_tok = L.dptr(bufs["tab"]), L.dptr(bufs["tok"])      # (x16 / w_ih16 stay NULL)
        else:
            ro.x16, ro.w_ih16 = L.dptr(bufs["x"]), L.dptr(bufs["wih"])
        keep.append(bufs)
        ro.h_prev16, ro.w_hh16 = L.dptr(bufs["h16"]), L.dptr(bufs["whh"])
        ro.b_ih, ro.b_hh, ro.y_prev_t, ro.y_out_t = (L.dptr(bufs[n_]) for n_ in ("bih", "bhh", "yp", "yo"))
        ro.y16a = L.dptr(bufs["ya"])
        ro.save_r, ro.save_z, ro.save_n, ro.save_hn = (L.dptr(t) for t in bufs["sv"])
        ro.drop_p = 0.0
    L.check(L.lib().ark_gru_diag_fwd(L.i32(L.PREC_F16), L.i32(L.PREC_F16), L.i32(2), roles, L.ptr(None), L.i32(B), L.i32(D),
                                     L.ptr(None), L.cur_stream()), "ark_gru_diag_fwd")
    torch.cuda.synchronize()
    tab = keep[0]["tab"].cpu()
    assert (tab - wtok.float() @ keep[0]["wih"].float().cpu().t()).abs().max().item() < 1e-4
    for bufs, (hn, r, z, n, ghn) in zip(keep, refs):
        got = bufs["yo"].cpu()[idx].reshape(B, D)
        assert (got - hn).abs().max().item() < 2e-5
        assert (bufs["ya"].float().cpu() - hn).abs().max().item() < 1e-3
        for sv, want in zip(bufs["sv"], (r, z, n, ghn)):
            assert (sv.float().cpu()[idx].reshape(B, D) - want).abs().max().item() < 2e-3
    # a role must name its input one way or the other
    roles[1].x16 = 0
    rc = L.lib().ark_gru_diag_fwd(L.i32(L.PREC_F16), L.i32(L.PREC_F16), L.i32(2), roles, L.ptr(None), L.i32(B), L.i32(D),
                                  L.ptr(None), L.cur_stream())
    assert rc < 0


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_token_table_forward_agrees_with_the_streamed_input_projection(drop):
    """`ark_fwd_tab` (default on for V <= 256): layer 0's roles add rows of x_tab = W_tok W_ih0^T instead of streaming
    x_t W_ih0^T, and the 16-bit embedding rows X0 are not gathered at all.  Same states, loss, gradients as the streamed
    path (fp32 sums in another order), over three optimiser steps (the table is refreshed with the weights)."""
    from oracle import sail_oracle as O
    cfg = dict(_big_cfg(), dec_dropout=drop)
    P = O.init_params(cfg, 0)
    B = 256
    triples, seq = synth_batch(cfg, B, seed=3)
    torch.manual_seed(5)
    eps = torch.randn(B, cfg["d_latent"])
    a = make_engine(dict(cfg, ark_fwd_tab=0), P, "mixed", lr=1e-3)
    b = make_engine(cfg, P, "mixed", lr=1e-3)
    assert a.xtab is None and b.xtab is not None
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    for eng in (a, b):
        eng.set_hyper(beta=0.1)
    a.drop_seed = b.drop_seed = 1234
    for step in range(3):
        oa = a.train_step(*args).cpu().numpy()
        ob = b.train_step(*args).cpu().numpy()
        torch.cuda.synchronize()
        assert b._skip_x0(b.ws, B, b.L) and not a._skip_x0(a.ws, B, a.L)
        # (after an Adam step the two runs' weights differ where a near-zero gradient's rounded sign decides a full lr step)
        assert rel_err(float(ob[0]), float(oa[0])) < (2e-5 if step == 0 else 3e-4), (step, oa, ob)
        for l in range(cfg["n_layers"]):
            assert (a.ws["Y"][l] - b.ws["Y"][l]).abs().max().item() < (2e-3 if step == 0 else 2e-2), (step, l)
        if step == 0:
            for k in a.g:
                da, db = a.g[k].float(), b.g[k].float()
                assert (da - db).norm().item() <= 5e-3 * da.norm().item() + 1e-9, k
    want = b.wtok16.view(torch.float16).float() @ b.wih16[0].view(torch.float16).float().t()
    assert (b.xtab - want).abs().max().item() < 1e-4      # the table follows the updated shadows
    # (three Adam steps of lr = 1e-3 in opposite directions where a near-zero gradient's sign differs, |m_hat / sqrt(v_hat)| a
    # little over 1 in the first steps)
    assert (a.P - b.P).abs().max().item() <= 3 * 1e-3 * 1.25 and (a.P - b.P).abs().mean().item() < 2e-5


@pytest.mark.parametrize("top", [False, True])
def test_gru_diag_bwd_matches_autograd(top):
    """ark_gru_diag_bwd through the C-ABI against torch autograd of one GRU cell: dh = carry + dgh_next W_hh + dy with
    dy given (top layer) or formed in the kernel as dgi_above W_ih_above; outputs the [dr|dz|dn|dn*r] panel, carry, bias sums"""
    from ark_amd import _lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5 + top)
    B, D = 64, 128
    idx = _tile_native_index(B, D).reshape(-1)
    bf = lambda t: t.to(torch.bfloat16)
    tn = lambda t, dt=torch.float32: (lambda o: (o.__setitem__(idx, t.reshape(-1).to(dt)), o)[1])(torch.zeros(B * D, dtype=dt))
    gi = torch.randn(B, 3 * D, generator=g).double().requires_grad_(True)
    gh = torch.randn(B, 3 * D, generator=g).double().requires_grad_(True)
    h_prev = torch.tanh(torch.randn(B, D, generator=g)).double()
    r = torch.sigmoid(gi[:, :D] + gh[:, :D])
    z = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
    n = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
    h = (1 - z) * n + z * h_prev
    # saves exactly as the forward pass stores them (fp16): the reference differentiates at those values
    carry = torch.randn(B, D, generator=g) * 0.1
    dgh_next = bf(torch.randn(B, 3 * D, generator=g) * 0.1)
    whh = bf(torch.randn(3 * D, D, generator=g) * 0.1)           # W_hh [3D, D]
    dgi_up = bf(torch.randn(B, 3 * D, generator=g) * 0.1)
    wih_up = bf(torch.randn(3 * D, D, generator=g) * 0.1)        # W_ih of the layer above [3D, D]
    dy_top = torch.randn(B, D, generator=g) * 0.1
    dy = dy_top if top else dgi_up.float() @ wih_up.float()
    dh = (carry + dgh_next.float() @ whh.float() + dy).double()
    (dh * h).sum().backward()
    roles = (L.GruDiagBwdRole * L.DIAG_MAX_ROLES)()
    ro = roles[0]
    junk = bf(torch.randn(B, D, generator=g))   # the dn column block of a panel is not part of dgh: must be ignored
    panel_next = torch.cat([dgh_next[:, :2 * D], junk, dgh_next[:, 2 * D:]], dim=1).contiguous()
    panel_up = torch.cat([dgi_up, junk], dim=1).contiguous()
    bufs = dict(dgn=panel_next.to(dev), whhT=whh.t().contiguous().to(dev), carry=tn(carry).to(dev),
                sr=tn(r.detach(), torch.float16).to(dev), sz=tn(z.detach(), torch.float16).to(dev),
                sn=tn(n.detach(), torch.float16).to(dev), shn=tn(gh[:, 2 * D:].detach(), torch.float16).to(dev),
                yp=tn(h_prev).to(dev), dg=torch.zeros(B, 4 * D, dtype=torch.bfloat16, device=dev), dbi=torch.zeros(3 * D, device=dev),
                dbh=torch.zeros(3 * D, device=dev), dy=tn(dy_top).to(dev), up=panel_up.to(dev),
                wupT=wih_up.t().contiguous().to(dev))
    if top:
        ro.dy_t = L.dptr(bufs["dy"])
    else:
        ro.dgi_up16, ro.w_ihT_up16 = L.dptr(bufs["up"]), L.dptr(bufs["wupT"])
    ro.dg_next16, ro.w_hhT16, ro.carry_t = L.dptr(bufs["dgn"]), L.dptr(bufs["whhT"]), L.dptr(bufs["carry"])
    ro.save_r, ro.save_z, ro.save_n, ro.save_hn = (L.dptr(bufs[k]) for k in ("sr", "sz", "sn", "shn"))
    ro.y_prev_t, ro.dg16 = L.dptr(bufs["yp"]), L.dptr(bufs["dg"])
    ro.db_ih, ro.db_hh = L.dptr(bufs["dbi"]), L.dptr(bufs["dbh"])
    ro.first, ro.drop_p = 0, 0.0
    L.check(L.lib().ark_gru_diag_bwd(L.i32(L.PREC_BF16), L.i32(1), roles, L.ptr(None), L.i32(B), L.i32(D), L.ptr(None),
                                     L.cur_stream()), "ark_gru_diag_bwd")
    torch.cuda.synchronize()
    close = lambda got, want, tol: (got.double().cpu() - want).abs().max().item() <= tol * want.abs().max().item()
    out = bufs["dg"].float()
    assert close(out[:, :3 * D], gi.grad, 1.5e-2)      # bf16 outputs, fp16 saves
    assert close(torch.cat([out[:, :2 * D], out[:, 3 * D:]], dim=1), gh.grad, 1.5e-2)
    assert close(bufs["carry"].cpu()[idx].reshape(B, D), (dh * z).detach(), 2e-3)
    assert close(bufs["dbi"], gi.grad.sum(0), 1.5e-2) and close(bufs["dbh"], gh.grad.sum(0), 1.5e-2)
