"""The persistent GRU sweep (csrc/gru_sweep.hip: the whole forward recurrence of a small batch of long sequences as ONE
launch, hidden state handed between workgroups inside the launch) against the layer-diagonal launches it replaces: same
states, saves, loss and gradients; no workgroup ever gives up waiting; eager and inside a captured graph.
Reference op on both sides: torch.nn.GRU, kgvae/model/models.py:121-127 (the diagonal path is pinned to the oracle by
tests/test_engine_gpu.py and tests/test_configs_gpu.py, whose wd-articles cases run THROUGH the sweep)."""
import pytest
import torch

from tests.parity_util import make_engine, rel_err, synth_batch
from tests.test_configs_gpu import _cfg

pytestmark = pytest.mark.gpu


def _pair(cfg, B, P, drop):
    a = make_engine(dict(cfg, dec_dropout=drop, ark_sweep=0), P, "mixed")
    b = make_engine(dict(cfg, dec_dropout=drop, ark_sweep=1), P, "mixed")
    Lq = cfg["seq_len"] - 1
    assert not a._use_sweep(B, Lq) and b._use_sweep(B, Lq)
    for eng in (a, b):
        eng.set_hyper(beta=0.3)
    a.drop_seed = b.drop_seed = 4321
    return a, b


@pytest.mark.parametrize("D,Z,B,T,padded", [(128, 16, 32, 12, True), (512, 32, 16, 10, True), (256, 8, 48, 6, False),
                                            (512, 10, 32, 3, False),
                                            (128, 64, 256, 9, True),    # 384 cells' worth of workgroups: two row tiles each
                                            (512, 16, 64, 4, True)])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_sweep_matches_diagonal_launches(D, Z, B, T, padded, drop):
    from oracle import sail_oracle as O
    cfg = _cfg(D, Z, 300, 5, T, padded)
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=padded)
    torch.manual_seed(9)
    eps = torch.randn(B, Z)
    a, b = _pair(cfg, B, P, drop)
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
        # the four waves of a sweep workgroup split K, the diagonal tiles do not: fp32 sums in another order, and a 16-bit
        # state that rounds the other way now and then
        assert (ya - yb).abs().max().item() < 2e-3, l
        for nm in ("SR", "SZ", "SN", "SHN"):
            sa, sb = a.ws[nm][l].view(torch.float16)[:R].float(), b.ws[nm][l].view(torch.float16)[:R].float()
            assert (sa - sb).abs().max().item() < 4e-3, (nm, l)
        fa = a.ws["Y16a"][l].view(torch.float16)[:R + B].float()
        fb = b.ws["Y16a"][l].view(torch.float16)[:R + B].float()
        assert (fa - fb).abs().max().item() < 3e-3, l
        if drop > 0 and l < cfg["n_layers"] - 1:
            da = a.ws["Yd16a"][l].view(torch.float16)[:R].float()
            db = b.ws["Yd16a"][l].view(torch.float16)[:R].float()
            assert ((da == 0) == (db == 0)).float().mean().item() > 0.9999, l   # the same counter-hash masks
            assert (da - db).abs().max().item() < 4e-3, l
            if a.ws["Yd16b"][l] is not None:
                da = a.ws["Yd16b"][l].view(torch.bfloat16)[:R].float()
                db = b.ws["Yd16b"][l].view(torch.bfloat16)[:R].float()
                assert (da - db).abs().max().item() < 2e-2, l
        if a.ws["Y16b"][l] is not None:
            fa = a.ws["Y16b"][l].view(torch.bfloat16)[:R + B].float()
            fb = b.ws["Y16b"][l].view(torch.bfloat16)[:R + B].float()
            assert (fa - fb).abs().max().item() < 2e-2, l
        # backward: the gate-gradient panels [dr | dz | dn | dn*r] of the persistent BPTT launch
        bt = torch.bfloat16 if b.prec_bwd == 1 else torch.float16
        pa, pb = a.ws["dG16"][l].view(bt)[:R].float(), b.ws["dG16"][l].view(bt)[:R].float()
        assert torch.isfinite(pb).all()
        assert (pa - pb).norm().item() <= 1e-2 * pa.norm().item() + 1e-12, l
        assert (pa - pb).abs().max().item() <= 4e-2 * pa.abs().max().item() + 1e-12, l
    ha, hb = a.ws["dH0"], b.ws["dH0"]
    assert (ha - hb).norm().item() <= 5e-3 * ha.norm().item() + 1e-12
    assert rel_err(float(ob[0]), float(oa[0])) < 2e-5, (oa, ob)
    for k in a.g:
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 5e-3 * da.norm().item() + 1e-9, k


def test_forward_sweep_with_diagonal_backward():
    """ark_sweep_bwd=0: the persistent forward feeds the diagonal backward launches (same buffers either way)"""
    from oracle import sail_oracle as O
    cfg = _cfg(128, 16, 300, 5, 12, True)
    B = 32
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=True)
    torch.manual_seed(9)
    eps = torch.randn(B, 16)
    a, b = _pair(cfg, B, P, 0.1)
    c = make_engine(dict(cfg, dec_dropout=0.1, ark_sweep=1, ark_sweep_bwd=0), P, "mixed")
    c.set_hyper(beta=0.3)
    c.drop_seed = 4321
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    ob = b.train_step(*args).cpu().numpy()
    oc = c.train_step(*args).cpu().numpy()
    torch.cuda.synchronize()
    assert c.sweep_error() == (0, 0) and b.sweep_error() == (0, 0)
    assert rel_err(float(oc[0]), float(ob[0])) < 1e-6
    for l in range(cfg["n_layers"]):
        assert torch.equal(b.ws["Y"][l], c.ws["Y"][l])   # the same forward launch
    for k in b.g:
        db, dc = b.g[k].float(), c.g[k].float()
        assert (db - dc).norm().item() <= 5e-3 * db.norm().item() + 1e-9, k


def test_sweep_inside_a_captured_graph_and_oracle_elbo():
    """replays of a captured step that contains the persistent launch reproduce the eager step; the ELBO is the oracle's"""
    from oracle import sail_oracle as O
    cfg = _cfg(512, 32, 500, 6, 14, True)
    B = 16
    P = O.init_params(cfg, 2)
    triples, seq = synth_batch(cfg, B, seed=8, padded=True)
    torch.manual_seed(4)
    eps = torch.randn(B, cfg["d_latent"])
    want = float(O.sail_elbo(P, triples, seq, eps, 0.3, cfg)[0])
    eng = make_engine(dict(cfg, ark_sweep=1), P, "mixed")
    eng.set_hyper(beta=0.3)
    dev = eng.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    oe = eng.train_step(*args).cpu().numpy()
    assert eng.sweep_error() == (0, 0)
    assert rel_err(float(oe[0]), want) < 1e-4, (oe, want)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = eng.capture_train_step(*args)
        for _ in range(3):
            eng.load_params(P)
            eng.reset_optimizer()
            eng.refresh_shadows()
            og = step().cpu().numpy()
            assert eng.sweep_error() == (0, 0)
            assert rel_err(float(og[0]), float(oe[0])) < 1e-6, (oe, og)


def test_sweep_refuses_a_grid_that_cannot_be_resident():
    from oracle import sail_oracle as O
    from ark_amd import _lib as L
    cfg = _cfg(512, 8, 50, 3, 3, False)
    P = O.init_params(cfg, 0)
    eng = make_engine(dict(cfg, ark_sweep=1), P, "mixed")
    with pytest.raises(L.ArkError):
        eng._use_sweep(256, 10)   # 3 x 16 x 32 workgroups
    auto = make_engine(dict(cfg), P, "mixed")
    assert not auto._use_sweep(256, 100) and not auto._use_sweep(16, 10) and auto._use_sweep(16, 100)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_sweep_in_the_single_type_precisions(precision):
    """the bf16 / bf16 and f16 / f16 instantiations of both sweeps against the diagonal launches of the same precision"""
    from oracle import sail_oracle as O
    cfg = _cfg(256, 16, 200, 4, 11, True)
    B = 32
    P = O.init_params(cfg, 3)
    triples, seq = synth_batch(cfg, B, seed=6, padded=True)
    torch.manual_seed(2)
    eps = torch.randn(B, 16)
    outs, grads = [], []
    for sw in (0, 1):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_sweep=sw), P, precision)
        eng.set_hyper(beta=0.2)
        eng.drop_seed = 99
        dev = eng.device
        outs.append(eng.train_step(triples.to(dev), seq.to(dev), eps.to(dev)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    tol = 3e-2 if precision == "bf16" else 5e-3   # (bf16 states: a rounding that flips is 2^-8 of the value)
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < (2e-3 if precision == "bf16" else 2e-5), outs
    for k in grads[0]:
        assert (grads[0][k] - grads[1][k]).norm().item() <= tol * grads[0][k].norm().item() + 1e-9, k


def test_sweep_decoder_only_model():
    """ARK (no encoder, zero initial state, position embeddings, no initial-state gradient): sweeps against diagonals"""
    from oracle import sail_oracle as O
    cfg = dict(_cfg(128, 4, 120, 4, 14, True), model_type="ARK")
    B = 32
    P = O.init_params(cfg, 4)
    _, seq = synth_batch(cfg, B, seed=7, padded=True)
    outs, grads = [], []
    for sw in (0, 1):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_sweep=sw), P, "mixed")
        eng.drop_seed = 5
        outs.append(eng.train_step(None, seq.to(eng.device)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0) and eng._use_sweep(B, cfg["seq_len"] - 1) == bool(sw)
        grads.append({k: v.float().clone() for k, v in eng.g.items()})
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 2e-5, outs
    for k in grads[0]:
        assert (grads[0][k] - grads[1][k]).norm().item() <= 5e-3 * grads[0][k].norm().item() + 1e-9, k


@pytest.mark.parametrize("nc", [2, 3])
def test_chunked_sweep_with_the_fused_ce_beside_it(nc):
    """long sequences: the forward sweep in time chunks (each starting from the state its predecessor left, dropout masks
    indexed by the absolute step) with the fused vocabulary CE of a chunk running beside the next chunk's sweep -- same
    states, per-row losses, dY and gradients as one sweep followed by one CE"""
    from oracle import sail_oracle as O
    cfg = _cfg(128, 16, 3000, 5, 64, True)   # L = 193 steps, V = 3 010: fused CE
    B = 32
    P = O.init_params(cfg, 5)
    triples, seq = synth_batch(cfg, B, seed=9, padded=True)
    torch.manual_seed(1)
    eps = torch.randn(B, 16)
    outs, engs = [], []
    for chunks in (1, nc):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_sweep=1, ark_ce_chunks=chunks), P, "mixed")
        eng.set_hyper(beta=0.3)
        eng.drop_seed = 77
        dev = eng.device
        Lq = cfg["seq_len"] - 1
        assert eng.fused_ce and (eng._ce_chunks(B, Lq) is not None) == (chunks > 1)
        outs.append(eng.train_step(triples.to(dev), seq.to(dev), eps.to(dev)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0)
        engs.append(eng)
    a, b = engs
    assert len(b._live_sweep_syncs()) == nc + 1   # one counter workspace per forward chunk + the backward sweep's
    for l in range(cfg["n_layers"]):
        assert torch.equal(a.ws["Y"][l], b.ws["Y"][l]), l            # the same arithmetic, launch boundaries apart
        assert torch.equal(a.ws["SR"][l], b.ws["SR"][l]), l
        if l < cfg["n_layers"] - 1:
            assert torch.equal(a.ws["Yd16a"][l], b.ws["Yd16a"][l]), l  # masks indexed by the absolute step
    assert torch.equal(a.ws["row_loss"], b.ws["row_loss"])
    assert (a.ws["dYa"] - b.ws["dYa"]).abs().max().item() <= 1e-6 * a.ws["dYa"].abs().max().item()
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 1e-6, outs
    for k in a.g:
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 2e-3 * da.norm().item() + 1e-9, k


@pytest.mark.parametrize("shape", ["wd-movies-like", "wd-articles-like"])
def test_sweep_with_dropout_matches_oracle_with_the_same_masks(shape):
    """DIRECT pin of the persistent sweeps with dropout on (round 3 had it only through the diagonal path): a padded
    wd-shaped batch (long sequences, V >= 2 048 so the fused / vocabulary-split CE runs, mixed precision, dec_dropout
    0.1) through the forward AND backward sweep; the counter-hash masks of that draw are materialised with
    ark_dropout_mask and handed to the CPU oracle: ELBO within north_star's 1e-4, every gradient.
    Reference semantics: nn.GRU(dropout=p) kgvae/model/models.py:121-127,184; loss ablation_study.py:59-73."""
    from oracle import sail_oracle as O
    from tests.test_engine_gpu import _row_major_masks
    # (D, Z, nE, nR, T, padded), batch: wd-movies is D = 128 / Z = 64 / L = 70, wd-articles D = 512 / Z = 128 / B = 16
    cfg, B = {"wd-movies-like": (_cfg(128, 64, 3000, 3, 23, True), 64),
              "wd-articles-like": (_cfg(512, 128, 2500, 6, 24, True), 16)}[shape]
    cfg = dict(cfg, dec_dropout=0.1)
    P = O.init_params(cfg, 3)
    triples, seq = synth_batch(cfg, B, seed=21, padded=True)
    torch.manual_seed(17)
    eps = torch.randn(B, cfg["d_latent"])
    eng = make_engine(cfg, P, "mixed")
    Lq = cfg["seq_len"] - 1
    assert eng._use_sweep(B, Lq) and eng.sweep_bwd and eng.fused_ce      # the default configuration takes these paths
    dev = eng.device
    eng.training = True
    eng.set_hyper(beta=0.4)
    eng._default_norms(B)
    eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    assert eng.sweep_error() == (0, 0)
    out4 = eng.ws["out4"].cpu().numpy()
    got = {k: v.detach().double().cpu().clone() for k, v in eng.g.items()}
    masks = _row_major_masks(eng, B, Lq)
    Pc = O._detach_tied(P, True)
    leaves = O.leaf_params(Pc)
    for _, t in leaves:
        t.requires_grad_(True)
    loss, ce, kl, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.4, cfg, drop_masks=masks)
    with torch.no_grad():
        plain, *_ = O.sail_elbo(Pc, triples, seq, eps, 0.4, cfg)
    assert abs(float(plain) - float(loss)) > 1e-4 * float(loss)          # the masks matter
    assert rel_err(float(out4[0]), float(loss)) < 1e-4, (out4, float(loss), float(ce), float(kl))
    loss.backward()
    checked = 0
    for k, t in leaves:
        want = t.grad.double()
        if cfg["pad_eid"] is not None and k in ("enc.e_emb.weight", "enc.r_emb.weight"):
            want = want.clone()
            want[cfg["pad_eid"] if k == "enc.e_emb.weight" else cfg["pad_rid"]] = 0   # padding_idx rows get no gradient
        g = got[k]
        nw = want.norm().item()
        if nw < 1e-12:
            assert g.norm().item() < 1e-9, k
            continue
        assert (g - want).norm().item() <= 2e-2 * nw, (k, (g - want).norm().item() / nw)
        assert torch.dot(g.flatten(), want.flatten()).item() / (g.norm().item() * nw) >= 0.9997, k
        checked += 1
    assert checked >= 20


def test_many_consecutive_sweep_steps_track_the_diagonal_path():
    """monotone epoch counters: no launch zeroes the sweep's counters, step k waits for counter >= (k + 1) * NS.  Twelve
    consecutive optimiser steps (eager, then replays of one captured graph) stay on the diagonal path's trajectory."""
    from oracle import sail_oracle as O
    cfg = _cfg(128, 16, 2500, 5, 12, True)
    B = 32
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=True)
    torch.manual_seed(9)
    eps = torch.randn(B, 16)
    a, b = _pair(cfg, B, P, 0.1)
    dev = a.device
    args = (triples.to(dev), seq.to(dev), eps.to(dev))
    la = [float(a.train_step(*args)[0]) for _ in range(12)]
    lb = [float(b.train_step(*args)[0]) for _ in range(6)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step = b.capture_train_step(*args)   # (its eager warm-up is step 7)
        lb.append(float(b.ws["out4"][0]))
        lb += [float(step()[0]) for _ in range(5)]
    torch.cuda.synchronize()
    assert b.sweep_error() == (0, 0)
    epochs = [int(sy[2]) for sy in b._live_sweep_syncs()]
    assert min(epochs) >= 12 and all(int(sy[3]) == 0 for sy in b._live_sweep_syncs()), epochs
    assert la[-1] < la[0]
    for x, y in zip(la, lb):
        assert rel_err(y, x) < 2e-3, (la, lb)


def test_a_failed_sweep_is_remembered_until_the_host_reads_it():
    """sticky error word (round 3: the next step's zero launch erased it).  One counter of the forward sweep is poisoned
    so that step 2 of 4 times out; steps 3 and 4 leave at once; the failure is still there after step 4, raise_on_sweep_error
    raises -- every time it is asked, until the host resets the workspaces: the failure is FATAL for the parameters (Adam
    kept stepping on invalid activations), so recovery = reload the weights, clear_sweep_error(), go on.  The record says who
    waited for what: here a producer whose counter never moved in this launch."""
    from oracle import sail_oracle as O
    from ark_amd import _lib as L
    cfg = _cfg(128, 16, 300, 5, 12, True)
    B = 32
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=True)
    eng = make_engine(dict(cfg, ark_sweep=1), P, "mixed")
    dev = eng.device
    torch.manual_seed(3)
    args = (triples.to(dev), seq.to(dev), torch.randn(B, 16).to(dev))
    good = float(eng.train_step(*args)[0])
    torch.cuda.synchronize()
    assert eng.sweep_error() == (0, 0)
    fwd_sync = eng._live_sweep_syncs()[0]
    fwd_sync[32 + 32 * 3].fill_(-1000)       # counter of (layer 0, step 0, row block 3): never reaches its target again
    for _ in range(3):
        eng.train_step(*args)
    torch.cuda.synchronize()
    e, d = eng.sweep_error()
    assert e != 0
    rec = eng.sweep_error_record()
    assert rec["signals_per_counter"] > 0 and rec["need"] > 0 and "never signalled" in rec["verdict"], rec
    assert float(eng.sweep_error_flag()) == 1.0
    for _ in range(2):                        # sticky on the host side too: asking does not clear
        with pytest.raises(L.ArkError, match="never signalled"):
            eng.raise_on_sweep_error()
    with pytest.raises(L.ArkError, match="ANOTHER rank"):
        type(eng).raise_on_sweep_error(make_engine(dict(cfg, ark_sweep=0), P, "mixed"), flag=1.0)   # a healthy rank of the same run
    eng.clear_sweep_error()
    assert eng.sweep_error() == (0, 0) and float(eng.sweep_error_flag()) == 0.0
    eng.load_params(P)
    eng.reset_optimizer()
    again = float(eng.train_step(*args)[0])
    torch.cuda.synchronize()
    assert eng.sweep_error() == (0, 0)
    assert rel_err(again, good) < 1e-5, (good, again)


def test_deferred_ce_weight_gradient_reaches_adam():
    """wd-movies kind of step (a sweep that fills most of the chip: the fused CE's weight gradient is queued behind the
    latent kernel): dW_tok / db_out must be in the gradient buffer BEFORE the decoder bucket's Adam.  Rounds 3-4 queued that
    Adam (early, on the side queue) in front of them: `dec.out.bias` never moved in single-process training at this shape.
    One step of the two-queue schedule against the same step with everything on one queue."""
    from oracle import sail_oracle as O
    cfg = _cfg(128, 64, 3000, 3, 12, True)
    B = 256                                    # 3 x 8 x 8 = 192 workgroups per sweep (> 128: the deferral)
    P = O.init_params(cfg, 0)
    triples, seq = synth_batch(cfg, B, seed=7, padded=True)
    torch.manual_seed(9)
    eps = torch.randn(B, cfg["d_latent"])
    outs = {}
    for name, extra in (("two-queue", {}), ("serial", {"ark_overlap_wgrad": 0})):
        eng = make_engine(dict(cfg, **extra), P, "mixed", lr=1e-3)
        eng.set_hyper(beta=0.5)
        assert eng._use_sweep(B, cfg["seq_len"] - 1) and eng._sweep_wgs(B, backward=True) > 128
        dev = eng.device
        eng.train_step(triples.to(dev), seq.to(dev), eps.to(dev), ce_count=int((seq[:, 1:] != 0).sum()))
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0)
        outs[name] = {k: (v.detach().cpu() - P[k]) for k, v in eng.p.items() if k in ("dec.out.bias", "dec.tok_emb.weight")}
    for k in ("dec.out.bias", "dec.tok_emb.weight"):
        a, b = outs["two-queue"][k], outs["serial"][k]
        assert b.abs().max().item() > 5e-4, k                      # one Adam step of lr = 1e-3 moved it
        assert (a - b).abs().mean().item() <= 0.05 * b.abs().mean().item(), (k, (a - b).abs().mean().item(), b.abs().mean().item())


def test_sweeps_of_co_located_ranks_must_fit_the_chip_together():
    """a sweep makes progress only with ALL its workgroups resident (one per CU); when several ranks share a device (the
    two-process tests) each holds such a grid: the host admits the sweep only if they fit together, `ark_sweep: 1` refuses
    loudly, and a step that could run a sweep beside a pending all-reduce or an over-sized side launch is refused too"""
    from oracle import sail_oracle as O
    from ark_amd import _lib as L
    cfg = _cfg(512, 128, 300, 5, 40, True)      # wd-articles-like: 3 x 1 x 32 = 96 workgroups per sweep
    P = O.init_params(cfg, 1)
    one = make_engine(cfg, P, "mixed")
    assert one.ranks_per_device == 1 and one._use_sweep(16, 121)
    two = make_engine(dict(cfg, ark_ranks_per_device=2), P, "mixed")
    assert two._sweep_wgs(16) == 96 and two._use_sweep(16, 121)             # 2 x 96 <= 256
    three = make_engine(dict(cfg, ark_ranks_per_device=3), P, "mixed")
    assert not three._use_sweep(16, 121)                                     # 3 x 96 > 256: layer-diagonal launches
    forced = make_engine(dict(cfg, ark_ranks_per_device=3, ark_sweep=1), P, "mixed")
    with pytest.raises(L.ArkError, match="share this device"):
        forced._use_sweep(16, 121)
    # beside a sweep: a pending collective, or more one-per-CU workgroups than the sweeps of all co-located ranks leave
    two._B = 16
    two._check_beside_sweep(side_wgs=32)                                     # (96 + 32) x 2 = 256: fits
    with pytest.raises(L.ArkError, match="exceed"):
        two._check_beside_sweep(side_wgs=64)
    two._dp_pending = torch.cuda.Event()
    with pytest.raises(L.ArkError, match="all-reduce"):
        two._check_beside_sweep()
    two._dp_pending = None


@pytest.mark.parametrize("D,B,T", [(128, 64, 12), (512, 16, 12), (256, 48, 11)])
def test_two_slices_per_workgroup_change_nothing(D, B, T):
    """`ark_sweep_wg_slices` (default 2: two unit slices share a 512-thread workgroup, the sweep holds half the CUs) is a
    placement choice only: the same arithmetic per slice, so states, saves and panels are bit-identical to one slice per
    workgroup (the bias-gradient and dh0 atomics land in another order)"""
    from oracle import sail_oracle as O
    cfg = _cfg(D, 16, 300, 5, T, True)
    P = O.init_params(cfg, 1)
    triples, seq = synth_batch(cfg, B, seed=5, padded=True)
    torch.manual_seed(9)
    eps = torch.randn(B, 16)
    engs, outs = [], []
    for ws in (1, 2):
        eng = make_engine(dict(cfg, dec_dropout=0.1, ark_sweep=1, ark_sweep_wg_slices=ws), P, "mixed")
        eng.set_hyper(beta=0.3)
        eng.drop_seed = 4321
        dev = eng.device
        outs.append(eng.train_step(triples.to(dev), seq.to(dev), eps.to(dev)).cpu().numpy().copy())
        torch.cuda.synchronize()
        assert eng.sweep_error() == (0, 0)
        engs.append(eng)
    a, b = engs
    assert b._sweep_wgs(B) * 2 == a._sweep_wgs(B)
    for l in range(cfg["n_layers"]):
        assert torch.equal(a.ws["Y"][l], b.ws["Y"][l]) and torch.equal(a.ws["SR"][l], b.ws["SR"][l]), l
        assert torch.equal(a.ws["dG16"][l], b.ws["dG16"][l]), l
    assert rel_err(float(outs[1][0]), float(outs[0][0])) < 1e-6
    for k in a.g:
        da, db = a.g[k].float(), b.g[k].float()
        assert (da - db).norm().item() <= 1e-4 * da.norm().item() + 1e-9, k
