"""The library's own latent-noise generator (ark_normal_fill; replaces torch.randn_like(mu), reference models.py:63):
distribution, freshness per launch / graph replay, per-rank streams, and the checkpointable draw counter."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(n, seed, hyper):
    from ark_amd import _lib as L
    out = torch.empty(n, device="cuda")
    L.check(L.lib().ark_normal_fill(L.ptr(out), L.i64(n), L.u64(seed), L.ptr(hyper), L.cur_stream()), "ark_normal_fill")
    return out


def test_normal_fill_is_standard_normal_and_counter_based():
    hyper = torch.zeros(16, device="cuda")
    x = _fill(1 << 20, 1234, hyper).double().cpu().numpy()
    assert abs(x.mean()) < 4e-3 and abs(x.std() - 1.0) < 4e-3
    assert abs(((x - x.mean()) ** 3).mean()) < 1.5e-2                      # skewness 0
    assert abs(((x - x.mean()) ** 4).mean() - 3.0) < 5e-2                  # kurtosis 3
    for q, want in ((0.5, 0.0), (0.8413447, 1.0), (0.9772499, 2.0), (0.0013499, -3.0)):
        assert abs(np.quantile(x, q) - want) < (3.5e-2 if abs(want) >= 3 else 2e-2), q   # (the 3-sigma quantile of 1M draws: +-0.008)
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 4e-3 and abs(np.corrcoef(x[0::2], x[1::2])[0, 1]) < 4e-3
    assert int(hyper.view(torch.int32)[13]) == 1                           # the launch bumped the draw counter
    y = _fill(1 << 20, 1234, hyper).double().cpu().numpy()                 # next draw: fresh values
    assert abs(np.corrcoef(x, y)[0, 1]) < 4e-3
    hyper.view(torch.int32)[13] = 0                                        # same (seed, draw) -> the same values
    assert np.array_equal(_fill(1 << 20, 1234, hyper).double().cpu().numpy(), x)
    hyper.view(torch.int32)[13] = 0
    z = _fill(1 << 20, 1235, hyper).double().cpu().numpy()                 # another seed: another stream
    assert abs(np.corrcoef(x, z)[0, 1]) < 4e-3
    odd = _fill(7, 5, hyper).cpu()
    assert odd.shape == (7,) and torch.isfinite(odd).all()


def test_no_draw_replays_a_shifted_copy_of_another_draw():
    """the first generator mixed (index, draw) LINEARLY -- i * 0x9E3779B1 + draw * 0x85EBCA77 -- so draw
    s + 433 983 was draw s shifted by 1 511 pairs (433 983 * 0x85EBCA77 + 1 511 * 0x9E3779B1 = 0 mod 2^32), 151 183 draws
    apart for 16 k pairs.  The draw counter is now hashed on its own first: the two draws (and a few neighbours of that
    lattice) share no shifted run, and the dropout masks -- same construction -- do not either."""
    from ark_amd import _lib as L
    hyper = torch.zeros(16, device="cuda")
    n = 1 << 15

    def draw(k):
        hyper.view(torch.int32)[13] = k
        return _fill(n, 99, hyper).cpu().numpy()
    assert (433983 * 0x85EBCA77 + 1511 * 0x9E3779B1) % (1 << 32) == 0       # the relation the old hash fell to
    for s, ds, dp in ((5, 433983, 1511), (0, 151183, 12417 - 1511), (17, 2 * 433983, 2 * 1511)):
        a, b = draw(s), draw(s + ds)
        best = 0.0
        for shift in {2 * dp, 2 * dp + 1, 2 * dp - 1, 0} | set(range(0, 64)):
            if 0 <= shift < n - 4096:
                best = max(best, float(np.mean(a[shift:shift + 4096] == b[:4096])), float(np.mean(b[shift:shift + 4096] == a[:4096])))
        assert best < 0.01, (s, ds, best)
    # dropout masks: ark_dropout_mask of two draws on that lattice
    m = []
    for k in (3, 3 + 433983):
        hyper.view(torch.int32)[12] = k
        out = torch.empty(n, device="cuda")
        L.check(L.lib().ark_dropout_mask(L.ptr(out), L.i64(n), L.f32(0.5), L.u64(4242), L.ptr(hyper), L.cur_stream()), "mask")
        m.append((out.cpu().numpy() != 0))
    for shift in (0, 4 * 1511, 4 * 1511 - 4, 4 * 1511 + 4):
        agree = float(np.mean(m[0][shift:shift + 8192] == m[1][:8192]))
        assert 0.45 < agree < 0.55, (shift, agree)


def test_engine_noise_is_fresh_per_step_per_rank_and_resumable():
    from ark_amd.engine import Engine
    from oracle import sail_oracle as O
    from tests.parity_util import load_golden, synth_batch
    _, cfg = load_golden("sail_synpaths_b32_s0")
    cfg = dict(cfg, seed=7)
    P = O.init_params(cfg, 0)
    tri, seq = synth_batch(cfg, 64, seed=3)
    tri, seq = tri.cuda(), seq.cuda()
    engs = [Engine(cfg, "cuda:0", precision="mixed", world_size=2, rank=r) for r in (0, 1)]
    eps = []
    for e in engs:
        e.load_params(P)
        e.forward(tri, seq)
        eps.append(e.ws["eps0"].clone())
    assert not torch.equal(eps[0], eps[1]) and abs(float(eps[0].std()) - 1.0) < 0.2      # ranks draw different noise
    e = engs[0]
    assert e.noise_draws() == 1
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = e.capture_train_step(tri, seq)       # eps drawn INSIDE the captured step
        a = e.ws["eps0"].clone()
        step()
        b = e.ws["eps0"].clone()
        step()
        c = e.ws["eps0"].clone()
    torch.cuda.synchronize()
    assert not torch.equal(a, b) and not torch.equal(b, c)                                 # every replay draws afresh
    n = e.noise_draws()
    e.set_noise_draws(n - 1)                                                               # resume: rewind one draw
    with torch.cuda.stream(st):
        step()
    torch.cuda.synchronize()
    assert torch.equal(e.ws["eps0"], c)
