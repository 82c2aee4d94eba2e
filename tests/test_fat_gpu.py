"""The weights-stationary forward (csrc/gru_fat.hip: the whole forward recurrence of a FULL batch as one persistent launch,
W_ih / W_hh rows resident in registers, hidden state handed between workgroups inside the launch) against the layer-diagonal
launches it replaces and against the CPU oracle: same states, saves, 16-bit copies, dropout masks, loss and gradients; no
workgroup gives up waiting; eager, captured, many consecutive steps.
Reference op on both sides: torch.nn.GRU, kgvae/model/models.py:121-127 (the diagonal path itself is pinned to the oracle by
tests/test_engine_gpu.py with `ark_fat: 0`)."""
import pytest
import torch

from tests.parity_util import make_engine, rel_err, synth_batch
from tests.test_configs_gpu import _cfg

pytestmark = pytest.mark.gpu


def _pair(cfg, B, P, drop, kernel="regs"):
    a = make_engine(dict(cfg, dec_dropout=drop, ark_fat=0, ark_sweep=0), P, "mixed")
    b = make_engine(dict(cfg, dec_dropout=drop, ark_fat=1, ark_fat_kernel=kernel, ark_sweep=0), P, "mixed")
    Lq = cfg["seq_len"] - 1
    assert not a._use_fat(B, Lq) and b._use_fat(B, Lq)
    for eng in (a, b):
        eng.set_hyper(beta=0.3)
    a.drop_seed = b.drop_seed = 4321
    return a, b


@pytest.mark.parametrize("D,Z,B,T,padded", [(512, 10, 1024, 3, False),    # syn-paths at the benchmarked batch: 5 row groups
                                            (512, 10, 256, 3, False),     # 16 row tiles over 10 subgroups: 1-2 tiles each
                                            (1024, 24, 256, 3, False),    # syn-types: 16-unit slices, one row group
                                            (512, 16, 48, 5, True),       # 3 row tiles: one row group, padded graphs
                                            (512, 32, 528, 4, True)])     # 33 row tiles: uneven subgroups
@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("kernel", ["regs", "ring"])
def test_fat_forward_matches_diagonal_launches(D, Z, B, T, padded, drop, kernel):
    """kernel: "regs" = fragments straight into registers; "ring" = the D = 512 build with an LDS-DMA landing ring, counted
    waits and role-specialised waves (eight waves; three subgroups per row group; a continuous tile stream with bubbles)"""
    from oracle import sail_oracle as O
    if kernel == "ring" and D != 512:
        pytest.skip("the LDS-ring kernel is a D = 512 build")
    cfg = _cfg(D, Z, 49, 3, T, padded)
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=padded)
    torch.manual_seed(9)
    eps = torch.randn(B, Z)
    a, b = _pair(cfg, B, P, drop, kernel)
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    oa = a.train_step(*args).cpu().numpy()
    ob = b.train_step(*args).cpu().numpy()
    torch.cuda.synchronize()
    assert b.sweep_error() == (0, 0)
    Lq = cfg["seq_len"] - 1
    R = B * Lq
    for l in range(cfg["n_layers"]):
        ya, yb = a.ws["Y"][l], b.ws["Y"][l]
        assert torch.isfinite(yb).all()
        # four waves split K here, the diagonal tiles do not: fp32 sums in another order, a 16-bit state that rounds the
        # other way now and then
        assert (ya - yb).abs().max().item() < 2e-3, l
        for nm in ("SR", "SZ", "SN", "SHN"):
            sa, sb = a.ws[nm][l].view(torch.float16)[:R].float(), b.ws[nm][l].view(torch.float16)[:R].float()
            assert (sa - sb).abs().max().item() < 4e-3, (nm, l)
        fa = a.ws["Y16a"][l].view(torch.float16)[:R + B].float()
        fb = b.ws["Y16a"][l].view(torch.float16)[:R + B].float()
        assert (fa - fb).abs().max().item() < 3e-3, l
        if a.ws["Y16b"][l] is not None:
            fa = a.ws["Y16b"][l].view(torch.bfloat16)[:R + B].float()
            fb = b.ws["Y16b"][l].view(torch.bfloat16)[:R + B].float()
            assert (fa - fb).abs().max().item() < 2e-2, l
        if drop > 0 and l < cfg["n_layers"] - 1:
            da = a.ws["Yd16a"][l].view(torch.float16)[:R].float()
            db = b.ws["Yd16a"][l].view(torch.float16)[:R].float()
            assert ((da == 0) == (db == 0)).float().mean().item() > 0.9999, l   # the same counter-hash masks
            assert (da - db).abs().max().item() < 4e-3, l
            if a.ws["Yd16b"][l] is not None:
                da = a.ws["Yd16b"][l].view(torch.bfloat16)[:R].float()
                db = b.ws["Yd16b"][l].view(torch.bfloat16)[:R].float()
                assert (da - db).abs().max().item() < 2e-2, l
    assert rel_err(float(ob[0]), float(oa[0])) < 2e-5, (oa, ob)
    for k in a.g:
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 5e-3 * da.norm().item() + 1e-9, k


@pytest.mark.parametrize("kernel", ["regs", "ring"])
def test_fat_forward_elbo_is_the_oracles_and_survives_capture(kernel):
    """the benchmarked shape: ELBO within north_star's 1e-4 of the CPU oracle; replays of a captured step that contains the
    persistent launch reproduce the eager step; twelve consecutive steps stay on the diagonal path's trajectory (monotone
    epoch counters: nothing is zeroed between the launches)"""
    from oracle import sail_oracle as O
    cfg = _cfg(512, 10, 49, 3, 3, False)
    B = 1024
    P = O.init_params(cfg, 2)
    triples, seq = synth_batch(cfg, B, seed=8)
    torch.manual_seed(4)
    eps = torch.randn(B, cfg["d_latent"])
    want = float(O.sail_elbo(P, triples, seq, eps, 0.3, cfg)[0])
    a, b = _pair(cfg, B, P, 0.0, kernel)
    dev = b.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    out = b.eval_loss(*args).cpu().numpy()
    assert b.sweep_error() == (0, 0)
    assert rel_err(float(out[0]), want) < 1e-4, (out, want)
    la = [float(a.train_step(*args)[0]) for _ in range(12)]
    lb = [float(b.train_step(*args)[0]) for _ in range(6)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = b.capture_train_step(*args)   # (its eager warm-up is step 7)
        lb.append(float(b.ws["out4"][0]))
        lb += [float(step()[0]) for _ in range(5)]
    torch.cuda.synchronize()
    assert b.sweep_error() == (0, 0)
    assert la[-1] < la[0]
    for x, y in zip(la, lb):
        assert rel_err(y, x) < 2e-3, (la, lb)


def test_fat_forward_decoder_only_model():
    """ARK (zero initial state, position embeddings) through the weights-stationary forward"""
    from oracle import sail_oracle as O
    cfg = dict(_cfg(512, 4, 49, 3, 3, False), model_type="ARK")
    B = 256
    P = O.init_params(cfg, 4)
    _, seq = synth_batch(cfg, B, seed=7)
    outs, grads = [], []
    for fat in (0, 1):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_fat=fat, ark_sweep=0), P, "mixed")
        eng.drop_seed = 5
        outs.append(eng.train_step(None, seq.to(eng.device)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0) and eng._use_fat(B, cfg["seq_len"] - 1) == bool(fat)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 2e-5, outs
    for k in grads[0]:
        assert (grads[0][k] - grads[1][k]).norm().item() <= 5e-3 * grads[0][k].norm().item() + 1e-9, k


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_fat_forward_in_the_single_type_precisions(precision):
    from oracle import sail_oracle as O
    cfg = _cfg(512, 16, 49, 3, 3, False)
    B = 256
    P = O.init_params(cfg, 3)
    triples, seq = synth_batch(cfg, B, seed=6)
    torch.manual_seed(2)
    eps = torch.randn(B, 16)
    outs, grads = [], []
    for fat in (0, 1):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_fat=fat, ark_sweep=0), P, precision)
        eng.set_hyper(beta=0.2)
        eng.drop_seed = 99
        dev = eng.device
        outs.append(eng.train_step(triples.to(dev), seq.to(dev), eps.to(dev)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    tol = 3e-2 if precision == "bf16" else 5e-3
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < (2e-3 if precision == "bf16" else 2e-5), outs
    for k in grads[0]:
        assert (grads[0][k] - grads[1][k]).norm().item() <= tol * grads[0][k].norm().item() + 1e-9, k


def test_fat_forward_refuses_what_does_not_fit():
    from oracle import sail_oracle as O
    from ark_amd import _lib as L
    cfg = _cfg(256, 8, 49, 3, 3, False)                # D = 256: not a shape of this kernel
    eng = make_engine(dict(cfg, ark_fat="auto"), O.init_params(cfg, 0), "mixed")
    assert not eng._use_fat(1024, 10)
    cfg = _cfg(512, 8, 49, 3, 3, False)
    eng = make_engine(dict(cfg, ark_fat=1), O.init_params(cfg, 0), "mixed")
    with pytest.raises(L.ArkError):
        eng._use_fat(8192, 10)                          # 512 row tiles: more than 16 per workgroup
    auto = make_engine(cfg, O.init_params(cfg, 0), "mixed")
    assert not auto._use_fat(1024, 10)                  # opt-in: measured slower than the diagonal launches
    on = make_engine(dict(cfg, ark_fat=1), O.init_params(cfg, 0), "mixed")
    assert on._use_fat(1024, 10) and on._use_fat(64, 10)
    with pytest.raises(L.ArkError):
        on._use_fat(16, 100)                            # one row tile: no two subgroups to walk
