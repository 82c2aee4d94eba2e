"""GPU tests of the drop-in module API (kgvae.model.models, kgvae.experiments.train)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import yaml

from tests.parity_util import load_golden, weights_from, rel_err

pytestmark = pytest.mark.gpu


def _model(name, precision="f32"):
    from kgvae.model.models import ARK, SAIL
    z, cfg = load_golden(name)
    cfg = dict(cfg, precision=precision)
    torch.manual_seed(int(z["seed"]))
    m = (SAIL if cfg["model_type"] == "SAIL" else ARK)(cfg).to("cuda")
    return m, z, cfg


@pytest.mark.parametrize("name", ["sail_tiny", "sail_small_pad"])
def test_reference_style_training_loop(name):
    """the reference's own loop -- model(...), F.cross_entropy, kl_mean, loss.backward(), optim.Adam --
    runs unchanged on the engine-backed module and reproduces the reference's numbers."""
    model, z, cfg = _model(name)
    dev = "cuda"
    triples, seq = torch.from_numpy(z["triples"]).to(dev), torch.from_numpy(z["seq"]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=float(z["lr"]))
    model.train()
    for s in range(len(z["losses"])):
        opt.zero_grad()
        logits, mu, logv = model(triples, seq[:, :-1], eps=torch.from_numpy(z[f"eps{s}"]).to(dev))
        ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)), seq[:, 1:].reshape(-1), ignore_index=0)
        kl = model.kl_mean(mu, logv)
        loss = ce + float(z["betas"][s]) * kl
        loss.backward()
        if s == 0:
            np.testing.assert_allclose(logits.detach().cpu().numpy(), z["logits0"], rtol=2e-4, atol=5e-5)
            for k, p in model.named_parameters():
                want = z["g0/" + k]
                assert np.abs(p.grad.cpu().numpy() - want).max() <= 2e-4 * np.abs(want).max() + 1e-7, k
        opt.step()
        assert rel_err(loss.item(), z["losses"][s][0]) < 2e-5, (s, loss.item(), z["losses"][s])
    assert model.state_dict()["dec.out.weight"].data_ptr() == model.state_dict()["dec.tok_emb.weight"].data_ptr()


def test_ark_reference_style_loop():
    model, z, cfg = _model("ark_tiny")
    seq = torch.from_numpy(z["seq"]).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=float(z["lr"]))
    for s in range(len(z["losses"])):
        opt.zero_grad()
        logits = model(seq[:, :-1])
        ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)), seq[:, 1:].reshape(-1), ignore_index=0)
        ce.backward()
        opt.step()
        assert rel_err(ce.item(), z["losses"][s][0]) < 2e-5
    assert model(torch.zeros(4, 3, 3, dtype=torch.long).cuda(), seq[:, :-1]).shape == (4, cfg["seq_len"] - 1, cfg["vocab_size"])


@pytest.mark.parametrize("name", ["sail_small", "ark_tiny"])
def test_fused_train_step_matches_golden(name):
    model, z, cfg = _model(name)
    triples, seq = torch.from_numpy(z["triples"]).cuda(), torch.from_numpy(z["seq"]).cuda()
    for s in range(len(z["losses"])):
        eps = torch.from_numpy(z[f"eps{s}"]).cuda() if cfg["model_type"] == "SAIL" else None
        out4 = model.train_step(triples, seq, beta=float(z["betas"][s]), lr=float(z["lr"]), eps=eps)
        assert rel_err(float(out4[0]), z["losses"][s][0]) < 2e-5, (s, out4, z["losses"][s])
    sd = model.state_dict()
    for k in sd:   # parameters alias the engine buffer, so state_dict() is live
        want = z[f"w{len(z['losses'])}/" + k]
        bad = np.abs(sd[k].cpu().numpy() - want) > (1e-4 * np.abs(want) + 5e-6)
        assert bad.mean() <= 2e-3, k


def test_decode_latent_and_beam():
    from kgvae.model.utils import seq_to_triples
    model, z, cfg = _model("sail_small")
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                           ((f[3:], z[f]) for f in z.files if f.startswith(f"w{len(z['losses'])}/"))})
    zs = torch.from_numpy(z["dec_z"])
    st = cfg["special_tokens"]
    got = model.decode_latent(zs, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=1)
    for i, tr in enumerate(got):
        n = int(z["dec_ntriples"][i])
        assert [list(t) for t in tr] == z["dec_triples"][i, :n].tolist()
    wide = model.decode_latent(zs, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=3)
    assert len(wide) == zs.shape[0]
    uniq = model.count_unique_graphs(cfg["d_latent"], lambda zz, beam: model.decode_latent(
        zz, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=beam), num_samples=32, beam=1)
    assert 1 <= len(uniq) <= 32


@pytest.mark.parametrize("name,beam", [("sail_small", 3), ("sail_small_pad", 2), ("sail_tiny", 4)])
def test_beam_decode_matches_the_reference_beam(name, beam):
    """SAIL.decode_latent(beam > 1) on per-beam incremental GRU states (Engine.beam_decode) against the oracle's
    restatement of the reference's prefix-re-running, batch-shared beam (models.py:282-300): identical token sequences"""
    from oracle import sail_oracle as O
    from kgvae.model.utils import seq_to_triples
    model, z, cfg = _model(name)
    sd = {k: torch.from_numpy(v.copy()) for k, v in ((f[3:], z[f]) for f in z.files if f.startswith(f"w{len(z['losses'])}/"))}
    model.load_state_dict(sd)
    zs = torch.from_numpy(z["dec_z"])
    want = O.beam_decode(sd, zs, cfg, beam)
    eng = model.engine()
    got = eng.beam_decode(zs.to(eng.device), beam, max_len=cfg["seq_len"] - 1).cpu()
    assert got.shape == want.shape and torch.equal(got, want), (got, want)
    st = cfg["special_tokens"]
    tri = model.decode_latent(zs, cfg["seq_len"], st, seq_to_triples, cfg["ENT_BASE"], cfg["REL_BASE"], beam=beam)
    assert tri == [seq_to_triples(row, st, cfg["ENT_BASE"], cfg["REL_BASE"]) for row in want]


def test_posterior_bits_is_the_teacher_forced_nll():
    """ARK.posterior_bits (single pass) equals the oracle's per-sequence token NLL in bits"""
    from oracle import sail_oracle as O
    from kgvae.model.utils import GraphSeqDataset
    model, z, cfg = _model("ark_tiny")
    seq = torch.from_numpy(z["seq"])
    graphs = [[tuple(int(x) for x in t) for t in g] for g in z["triples"]]
    ds = GraphSeqDataset(graphs, None, None, special_tokens=cfg["special_tokens"], ent_base=cfg["ENT_BASE"],
                         rel_base=cfg["REL_BASE"], seq_len=cfg["seq_len"])
    stats = model.posterior_bits(ds, "cuda", sample_frac=1.0)
    P = O.init_params(cfg, int(z["seed"]))
    with torch.no_grad():
        logits = O.ark_forward(P, seq[:, :-1], cfg)
        nll = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), seq[:, 1:].reshape(-1), ignore_index=0,
                              reduction="none").reshape(seq.shape[0], -1).sum(1) / math.log(2)
    got = np.array([r["ar_bits"] for r in stats["records"]])
    np.testing.assert_allclose(got, nll.numpy(), rtol=2e-5)
    assert stats["avg_kl_bits"] == 0.0


def test_ark_generate_shapes_and_sampling():
    model, z, cfg = _model("ark_tiny")
    st = cfg["special_tokens"]
    g = model.generate(cfg["seq_len"], st, batch_size=5)
    assert g.shape == (5, cfg["seq_len"]) and bool((g[:, 0] == st["BOS"]).all())
    torch.manual_seed(0)
    s = model.generate(cfg["seq_len"], st, batch_size=6, sample=True, temperature=0.8, top_p=0.9, top_k=5)
    assert s.shape == (6, cfg["seq_len"]) and int(s.max()) < cfg["vocab_size"]


@pytest.mark.parametrize("model_type", ["SAIL", "ARK"])
def test_train_entry_point_end_to_end(tmp_path, model_type):
    """python -m kgvae.experiments.train on synthetic syn-paths-shaped data: 2 epochs, loss goes down,
    checkpoints are written in the reference's format."""
    from kgvae.experiments import train as T
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "sail_syn-paths.yaml")))
    cfg.update(model_type=model_type, d_model=64, num_epochs=2, batch_size=64, save_every=2, compression_log_every=2,
               learning_rate=1e-3, synthetic_sizes={"n_train": 512, "n_val": 128, "n_test": 64})
    cpath = tmp_path / "c.yaml"
    yaml.safe_dump(cfg, open(cpath, "w"))
    T.main(["--config", str(cpath), "--checkpoint-dir", str(tmp_path / "ck")])
    run = os.listdir(tmp_path / "ck")[0]
    files = os.listdir(tmp_path / "ck" / run)
    assert f"syn-paths_{model_type}_best_model.pt" in files and f"syn-paths_{model_type}_checkpoint_epoch_2.pt" in files
    assert "effective_config.yaml" in files
    ck = torch.load(tmp_path / "ck" / run / f"syn-paths_{model_type}_best_model.pt", weights_only=False)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "val_loss", "config", "vocabs",
            "dataset_meta"} <= set(ck)
    assert "dec.out.weight" in ck["model_state_dict"] and "dec.gru.weight_hh_l2" in ck["model_state_dict"]
    import json
    rows = [json.loads(l) for l in open(tmp_path / "ck" / run / "metrics.jsonl")]
    ep = [r for r in rows if "train/loss" in r]
    assert len(ep) == 2 and ep[1]["train/loss"] < ep[0]["train/loss"]


def test_graphed_train_step_matches_eager_ark():
    """ARK has no latent noise (and the golden config no dropout), so the cached-hipGraph step must
    reproduce the eager step bit for bit -- including across an eval at another batch size."""
    ma, z, cfg = _model("ark_tiny")
    mb, _, _ = _model("ark_tiny")
    seq = torch.from_numpy(z["seq"]).cuda()
    lr = float(z["lr"])
    for s in range(5):
        oa = ma.train_step(None, seq, lr=lr).clone()
        ob = mb.train_step(None, seq, lr=lr, graph=True).clone()
        torch.testing.assert_close(oa, ob, rtol=1e-6, atol=1e-7)
        if s == 2:   # another batch size in between must not disturb the captured workspace
            mb.eval(); mb.eval_loss(None, seq[: seq.shape[0] // 2].contiguous()); mb.train()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        # float atomics (embedding scatter, bias column sums) make the last ulp order-dependent
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7, msg=k)


def test_train_step_samples_latent_noise():
    """without an explicit eps the engine draws z = mu + N(0,1)*std like the reference (models.py:63):
    two evaluations of the same batch differ, two with the same eps do not; eager and graphed steps
    train to the same loss level."""
    model, z, cfg = _model("sail_tiny")
    tri, seq = torch.from_numpy(z["triples"]).cuda(), torch.from_numpy(z["seq"]).cuda()
    model.eval()
    a, b = model.eval_loss(tri, seq).clone(), model.eval_loss(tri, seq).clone()
    assert a[2] != b[2] or a[1] != b[1]
    eps = torch.from_numpy(z["eps0"]).cuda()
    a, b = model.eval_loss(tri, seq, eps=eps).clone(), model.eval_loss(tri, seq, eps=eps).clone()
    assert torch.equal(a, b)
    finals = []
    for graph in (False, True):
        m, _, _ = _model("sail_tiny")
        m.train()
        torch.manual_seed(3)
        for s in range(60):
            out = m.train_step(tri, seq, lr=3e-3, beta=0.1, graph=graph)
        finals.append(out[0].item())
    assert all(math.isfinite(v) for v in finals)
    assert abs(finals[0] - finals[1]) < 0.15 * abs(finals[0]) + 0.05, finals


def test_sail_posterior_bits_matches_oracle():
    """SAIL.posterior_bits (one teacher-forced pass per batch) against the oracle's prefix-by-prefix restatement of
    the reference (models.py:202-260) with the same reparameterisation noise"""
    from oracle import sail_oracle as O
    from kgvae.model.utils import GraphSeqDataset
    model, z, cfg = _model("sail_tiny_pad")
    tri, seq = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])
    pe, pr = cfg["pad_eid"], cfg["pad_rid"]
    graphs = [[tuple(int(x) for x in t) for t in g if int(t[1]) != pr] for g in z["triples"]]
    ds = GraphSeqDataset(graphs, None, None, use_padding=True, pad_eid=pe, pad_rid=pr, max_triples=tri.shape[1],
                         special_tokens=cfg["special_tokens"], ent_base=cfg["ENT_BASE"], rel_base=cfg["REL_BASE"],
                         seq_len=cfg["seq_len"])
    t2, s2 = ds.tensorize()
    assert torch.equal(t2, tri) and torch.equal(s2, seq)      # the dataset reproduces the golden batch
    eps = torch.from_numpy(z["eps0"])
    stats = model.posterior_bits(ds, "cuda", sample_frac=1.0, eps=eps)
    P = O.init_params(cfg, int(z["seed"]))
    ar, kl = O.posterior_bits(P, tri, seq, eps, cfg)
    np.testing.assert_allclose([r["ar_bits"] for r in stats["records"]], ar, rtol=3e-5)
    np.testing.assert_allclose([r["kl_bits"] for r in stats["records"]], kl, rtol=3e-5, atol=1e-7)
    assert abs(stats["avg_total_bits"] - (np.mean(ar) + np.mean(kl))) < 1e-3


@pytest.mark.parametrize("temperature,top_p,top_k", [(1.0, 0.0, 0), (0.7, 0.9, 0), (1.3, 0.0, 5), (0.8, 0.5, 7), (1.0, 0.95, 3)])
def test_ark_generate_distribution_and_incremental_step(temperature, top_p, top_k):
    """ARK.generate: (a) the one-token GRU step reproduces the logits of the full-prefix decoder run the reference
    performs per token (models.py:427), (b) the filtered next-token distribution equals the oracle's restatement of
    models.py:431-456, (c) greedy generation equals the oracle's argmax loop token for token"""
    from oracle import sail_oracle as O
    from kgvae.model.models import ARK
    model, z, cfg = _model("ark_tiny")
    P = O.init_params(cfg, int(z["seed"]))
    seq = torch.from_numpy(z["seq"])
    B, Lp = seq.shape[0], 6
    eng = model.engine()
    d = eng.decode_begin(B)
    with torch.no_grad():
        ref = O.ark_forward(P, seq[:, :Lp], cfg)                       # [B, Lp, V]: logits after each prefix
    for t in range(Lp):
        got = eng.decode_step(d, seq[:, t].contiguous().cuda(), t).cpu()
        np.testing.assert_allclose(got.numpy(), ref[:, t].numpy(), rtol=2e-4, atol=2e-5)
    dense = ARK.filtered_probs(ref[:, -1].cuda(), temperature, top_p, top_k).cpu()
    want = O.sampling_distribution(ref[:, -1], temperature, top_p, top_k)
    np.testing.assert_allclose(dense.numpy(), want.numpy(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(dense.sum(-1), torch.ones(B), atol=1e-5)
    g = model.generate(cfg["seq_len"], cfg["special_tokens"], batch_size=3).cpu()
    s = torch.full((3, 1), 1, dtype=torch.long)
    with torch.no_grad():
        for _ in range(cfg["seq_len"] - 1):
            s = torch.cat([s, O.ark_forward(P, s, cfg)[:, -1].argmax(-1, keepdim=True)], 1)
            if bool((s[:, -1] == 2).all()):
                break
    assert torch.equal(g[:, :s.shape[1]], s) and bool((g[:, s.shape[1]:] == 2).all())


def test_load_state_dict_refreshes_the_weight_shadows():
    """load_state_dict copies in place (same storage), so the engine must notice and rebuild its 16-bit
    weight shadows -- evaluation after loading other weights has to match the oracle on THOSE weights"""
    from oracle import sail_oracle as O
    _, cfg = load_golden("sail_synpaths_b32_s0")
    cfg = dict(cfg, precision="mixed", dec_dropout=0.0)
    from kgvae.model.models import SAIL
    from tests.parity_util import synth_batch
    torch.manual_seed(0)
    model = SAIL(cfg).to("cuda")
    tri, seq = synth_batch(cfg, 64, seed=2)
    eps = torch.randn(64, cfg["d_latent"])
    model.eval()
    model.eval_loss(tri.cuda(), seq.cuda(), beta=0.5, eps=eps.cuda())          # engine + shadows exist now
    P1 = O.init_params(cfg, 1)                                                  # other weights
    model.load_state_dict({k: v for k, v in P1.items()})
    out = model.eval_loss(tri.cuda(), seq.cuda(), beta=0.5, eps=eps.cuda()).cpu().numpy()
    with torch.no_grad():
        want, *_ = O.sail_elbo(P1, tri, seq, eps, 0.5, cfg)
    assert rel_err(float(out[0]), float(want)) < 3e-4, (out, float(want))
    with torch.no_grad():                                                       # an in-place edit is seen as well
        for p in model.parameters():
            p.mul_(0.5)
    P2 = {k: v * 0.5 for k, v in P1.items()}
    P2["dec.out.weight"] = P2["dec.tok_emb.weight"]
    out = model.eval_loss(tri.cuda(), seq.cuda(), beta=0.5, eps=eps.cuda()).cpu().numpy()
    with torch.no_grad():
        want, *_ = O.sail_elbo(P2, tri, seq, eps, 0.5, cfg)
    assert rel_err(float(out[0]), float(want)) < 3e-4, (out, float(want))


def test_ce_normaliser_survives_an_evaluation_between_train_steps():
    """an evaluation on a smaller batch lets the device count its own CE targets; the next train step with
    the same host-side count as before must re-issue it (the host mirror used to go stale: CE scaled by B/B_last)"""
    from oracle import sail_oracle as O
    from tests.parity_util import synth_batch, make_engine
    _, cfg = load_golden("sail_synpaths_b32_s0")
    cfg = dict(cfg, dec_dropout=0.0)
    P = O.init_params(cfg, 0)
    B = 64
    tri, seq = synth_batch(cfg, B, seed=5)
    eps = torch.randn(B, cfg["d_latent"])
    cnt = int((seq[:, 1:] != 0).sum())
    for graph in (False, True):
        eng = make_engine(cfg, P, "f32", lr=1e-4)
        dev = eng.device
        a = (tri.to(dev), seq.to(dev), eps.to(dev))
        if graph:
            step = eng.capture_train_step(*a, ce_count=cnt)
            eng.load_params(P); eng.reset_optimizer()
            run = lambda: (eng.set_hyper(ce_count=cnt), step())[1]
        else:
            run = lambda: eng.train_step(*a, ce_count=cnt)
        run()
        eng.eval_loss(a[0][:16].contiguous(), a[1][:16].contiguous(), a[2][:16].contiguous())   # device-side count of 16 rows
        eng.load_params(P); eng.reset_optimizer()
        out = run().cpu().numpy()
        with torch.no_grad():
            want, ce, kl, *_ = O.sail_elbo(P, tri, seq, eps, 1.0, cfg)
        assert rel_err(float(out[1]), float(ce)) < 2e-5, (graph, out, float(ce))


def test_backward_of_a_stale_forward_raises():
    from ark_amd._lib import ArkError
    model, z, cfg = _model("sail_tiny")
    tri, seq = torch.from_numpy(z["triples"]).cuda(), torch.from_numpy(z["seq"]).cuda()
    model.train()
    logits, mu, logv = model(tri, seq[:, :-1])
    model(tri, seq[:, :-1])                       # a second forward overwrites the saved activations
    with pytest.raises(ArkError):
        logits.sum().backward()


def test_checkpoint_format_and_resume(tmp_path):
    """(a) the checkpoint loads into the reference's own optimizer / scheduler objects (train.py:566-591: written
    after scheduler.step()); (b) resume_from_checkpoint: 1 epoch + resume 1 epoch == 2 epochs in f32 -- same data
    order, same latent noise, same dropout draws, same Adam state; what is left is the last-ulp order dependence of
    the float atomics in the embedding scatters and bias sums (two identical uninterrupted runs differ as much)"""
    from kgvae.experiments import train as T
    from kgvae.model.models import SAIL
    root = os.path.dirname(os.path.dirname(__file__))
    base = yaml.safe_load(open(os.path.join(root, "configs", "sail_syn-paths.yaml")))
    base.update(model_type="SAIL", d_model=64, batch_size=64, save_every=1, compression_log_every=100, verify_every=100,
                learning_rate=1e-3, precision="f32", seed=7, shuffle_train=True, use_hip_graph=False,
                synthetic_sizes={"n_train": 256, "n_val": 64, "n_test": 64})

    def run(name, **over):
        cfg = dict(base, **over)
        cpath = tmp_path / f"{name}.yaml"
        yaml.safe_dump(cfg, open(cpath, "w"))
        T.main(["--config", str(cpath), "--checkpoint-dir", str(tmp_path / name)])
        rd = tmp_path / name / os.listdir(tmp_path / name)[0]
        return rd

    full = run("full", num_epochs=2)
    ck2 = torch.load(full / "syn-paths_SAIL_checkpoint_epoch_2.pt", weights_only=True)
    ck1 = torch.load(full / "syn-paths_SAIL_checkpoint_epoch_1.pt", weights_only=True)
    # (a) reference-side consumers
    cfg = ck1["config"]
    m = SAIL(dict(cfg, precision="f32"))
    m.load_state_dict(ck1["model_state_dict"])
    opt = torch.optim.Adam(m.parameters(), lr=1.0)
    opt.load_state_dict(ck1["optimizer_state_dict"])
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=cfg["num_epochs"], eta_min=cfg.get("eta_min", 1e-6))
    sch.load_state_dict(ck1["scheduler_state_dict"])
    assert sch.last_epoch == 1 and ck1["epoch"] == 1
    want_lr = T.cosine_lr(cfg["learning_rate"], 1, cfg["num_epochs"], cfg.get("eta_min", 1e-6))
    assert abs(opt.param_groups[0]["lr"] - want_lr) < 1e-12 and abs(sch.get_last_lr()[0] - want_lr) < 1e-12
    st = opt.state_dict()["state"]
    assert len(st) == len(list(m.parameters())) and float(st[0]["step"]) == 256 // 64
    # (b) resume: the run is split after epoch 1
    rest = run("rest", num_epochs=2, resume_from_checkpoint=True,
               checkpoint_path=str(full / "syn-paths_SAIL_checkpoint_epoch_1.pt"))
    ckr = torch.load(rest / "syn-paths_SAIL_checkpoint_epoch_2.pt", weights_only=True)
    moved = 0.0
    for k, v in ck2["model_state_dict"].items():
        d = (v - ckr["model_state_dict"][k]).abs().max().item()
        moved = max(moved, (v - ck1["model_state_dict"][k]).abs().max().item())
        assert d <= 2e-6, (k, d)
    assert moved > 1e-3      # the second epoch did train: the resumed run is not trivially equal
    for i, s2 in ck2["optimizer_state_dict"]["state"].items():
        sr = ckr["optimizer_state_dict"]["state"][i]
        # (fp32 atomics of the split-K weight gradients land in another order on the resumed run: ~1e-7 of the gradient)
        torch.testing.assert_close(s2["exp_avg"], sr["exp_avg"], rtol=1e-4, atol=1e-8)
        torch.testing.assert_close(s2["exp_avg_sq"], sr["exp_avg_sq"], rtol=1e-4, atol=1e-12)
        assert float(s2["step"]) == float(sr["step"]) == 8.0
    assert ck2["ark_amd"]["dropout_draws"] == ckr["ark_amd"]["dropout_draws"] == 8


REFERENCE_YAML_KEYS = ["batch_size", "beam_width", "beta0", "beta1", "checkpoint_path", "compression_log_every", "d_latent",
                       "d_model", "dataset", "experiment_name", "learning_rate", "lr_scheduler", "model_type", "n_heads",
                       "n_layers", "num_diversity_samples", "num_epochs", "num_generated_latent_graphs",
                       "num_generated_test_graphs", "permute_triples", "resume_from_checkpoint", "sample_frac", "save_every",
                       "shuffle_train", "triple_order", "use_padding", "use_test_for_final_eval", "verify_every"]


def test_reference_yaml_key_set_runs_verbatim(tmp_path):
    """exactly the key set of the reference's configs/autoreg_syn-paths.yaml (:1-46) with its values -- model_type ARK,
    d_model 512, batch 256, lr 1e-4, cosine schedule, ... -- goes through train.main(); only the run length is cut
    (synthetic_sizes is this repo's offline data switch: no IntelliGraphs files exist here).  The reference's YAML
    itself is not copied into this repository; configs/sail_syn-paths.yaml differs from it in model_type, the two
    path/name strings and the extra `precision` key only."""
    from kgvae.experiments import train as T
    root = os.path.dirname(os.path.dirname(__file__))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "sail_syn-paths.yaml")))
    cfg.pop("precision")
    cfg.update(model_type="ARK", experiment_name="autoreg_vae_syn_paths_dec_only_TRF", checkpoint_path="checkpoints2/autoreg_syn_paths.pt")
    assert sorted(cfg) == REFERENCE_YAML_KEYS
    assert (cfg["d_model"], cfg["d_latent"], cfg["n_layers"], cfg["batch_size"], cfg["learning_rate"], cfg["num_epochs"]) == \
        (512, 10, 3, 256, 1e-4, 100)
    cfg.update(num_epochs=1, synthetic_sizes={"n_train": 1024, "n_val": 256, "n_test": 64})
    cpath = tmp_path / "c.yaml"
    yaml.safe_dump(cfg, open(cpath, "w"))
    T.main(["--config", str(cpath), "--checkpoint-dir", str(tmp_path / "ck")])
    run = os.listdir(tmp_path / "ck")[0]
    ck = torch.load(tmp_path / "ck" / run / "syn-paths_ARK_best_model.pt", weights_only=True)
    assert ck["epoch"] == 1 and ck["model_state_dict"]["dec.gru.weight_ih_l0"].shape == (1536, 512)
    assert ck["scheduler_state_dict"]["last_epoch"] == 1 and math.isfinite(ck["val_loss"])


def test_reference_style_loop_draws_fresh_dropout_masks_every_forward():
    """nn.GRU(dropout=p) draws a new mask per training forward; the engine-backed module does too in the reference-style
    loop (model(...) twice without an engine Adam step in between), on different ranks, and not at all in eval mode"""
    from kgvae.model.models import SAIL
    z, cfg = load_golden("sail_small")
    cfg = dict(cfg, dec_dropout=0.3, precision="mixed")
    torch.manual_seed(0)
    model = SAIL(cfg).to("cuda")
    triples, seq = torch.from_numpy(z["triples"]).to("cuda"), torch.from_numpy(z["seq"]).to("cuda")
    eps = torch.from_numpy(z["eps0"]).to("cuda")
    model.train()
    with torch.no_grad():
        a = model(triples, seq[:, :-1], eps=eps)[0].clone()
        b = model(triples, seq[:, :-1], eps=eps)[0].clone()
    assert (a - b).abs().max().item() > 1e-3          # a different mask
    model.eval()
    with torch.no_grad():
        c = model(triples, seq[:, :-1], eps=eps)[0].clone()
        d = model(triples, seq[:, :-1], eps=eps)[0].clone()
    assert torch.equal(c, d)
    eng = model.engine()
    s0 = eng._layer_seed(0)
    eng.rank = 1
    assert eng._layer_seed(0) != s0                    # shards of a data-parallel batch draw different masks
    eng.rank = 0
