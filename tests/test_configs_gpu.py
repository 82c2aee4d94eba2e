"""Parity at the shapes of the other BASELINE.json configs (syn-types, wd-movies, wd-articles):
larger d_model, padded variable-length graphs, vocabularies of 24 k / 61 k tokens (large-table
scatter path, un-fused [B*L, V] logits), long sequences.  HIP engine vs the CPU oracle on the same
seeded weights / inputs / eps: ELBO and every gradient."""
import numpy as np
import pytest
import torch

from tests.parity_util import make_engine, rel_err

pytestmark = pytest.mark.gpu


def _cfg(D, Z, nE, nR, T, padded, n_layers=3):
    nE2, nR2 = (nE + 1, nR + 1) if padded else (nE, nR)
    return dict(model_type="SAIL", d_model=D, d_latent=Z, n_layers=n_layers, n_heads=4, n_entities=nE2, n_relations=nR2,
                pad_eid=nE if padded else None, pad_rid=nR if padded else None, seq_len=2 + 3 * T,
                vocab_size=3 + nE2 + nR2, dec_dropout=0.0, ENT_BASE=3, REL_BASE=3 + nE2,
                special_tokens={"PAD": 0, "BOS": 1, "EOS": 2}, max_triples=T, nE=nE, nR=nR)


SHAPES = {
    # name: (cfg, batch)            reference YAML                     what it exercises
    "syn-types": (_cfg(1024, 24, 30, 3, 3, False), 256),         # configs/autoreg_syn-types.yaml: D=1024 tiles, Z=24, B=256
    "wd-movies": (_cfg(128, 64, 24093, 3, 23, True), 256),        # autoreg_wd-movies.yaml: V=24101, padding, L=70, B=256
    "wd-articles": (_cfg(512, 128, 60932, 6, 40, True), 16),      # autoreg_wd-articles.yaml: V=60943, B=16 (T cut to 40)
    # the same at the dataset's REAL length: T = 212 padded triples, L = 637 decoder steps, [B*L, V] = 10 192 x 60 943
    "wd-articles-full": (_cfg(512, 128, 60932, 6, 212, True), 16),
}


_ORACLE = {}


def _oracle(name):
    """oracle ELBO + gradients of one shape, computed once for both precisions (the full-length case is ~30 s of CPU)"""
    if name not in _ORACLE:
        from oracle import sail_oracle as O
        from tests.parity_util import synth_batch
        cfg, B = SHAPES[name]
        torch.set_num_threads(16)
        P = O.init_params(cfg, 0)
        triples, seq = synth_batch(cfg, B, seed=11, padded=cfg["pad_rid"] is not None)
        torch.manual_seed(3)
        eps = torch.randn(B, cfg["d_latent"])
        leaves = O.leaf_params(P)
        for _, p in leaves:
            p.requires_grad_(True)
        loss, ce, kl, *_ = O.sail_elbo(P, triples, seq, eps, 0.5, cfg)
        loss.backward()
        want = {k: p.grad.detach().clone() for k, p in leaves}
        with torch.no_grad():
            for _, p in leaves:
                p.requires_grad_(False)
                p.grad = None
        _ORACLE.clear()   # keep one shape at a time (the full-length logits graph is GBs)
        _ORACLE[name] = (P, triples, seq, eps, float(loss), float(kl), want)
    return _ORACLE[name]


# mixed (the shipped mode: fp16 forward / bf16 backward MFMA operands) is held to north_star's 1e-4 on the ELBO at the YAML
# batch sizes (measured on MI355X: syn-types 3.2e-5, wd-movies 2.6e-6, wd-articles 1.5e-6 / 1.1e-6)
# relative L2 error of every gradient tensor (measured on MI355X: f32 <= 1.4e-5, mixed <= 4.5e-3 over the four shapes)
L2TOL = {"f32": 1e-4, "mixed": 1.2e-2}


@pytest.mark.parametrize("precision,ltol,gtol", [("f32", 2e-5, 2e-3), ("mixed", 1e-4, 6e-2)])
@pytest.mark.parametrize("name", list(SHAPES))
def test_other_configs_match_oracle(name, precision, ltol, gtol):
    cfg, B = SHAPES[name]
    P, triples, seq, eps, loss, kl, want = _oracle(name)
    eng = make_engine(cfg, P, precision)
    dev = eng.device
    eng.set_hyper(beta=0.5)
    eng._default_norms(B)
    w = eng.forward(triples.to(dev), seq.to(dev), eps.to(dev))
    eng.backward()
    out4 = w["out4"].cpu().numpy()
    assert rel_err(float(out4[0]), loss) < ltol, (out4, loss)
    assert rel_err(float(out4[2]), kl) < 10 * ltol
    want = dict(want)
    if cfg["pad_eid"] is not None:   # padding_idx rows never receive gradient
        want["enc.e_emb.weight"] = want["enc.e_emb.weight"].clone()
        want["enc.r_emb.weight"] = want["enc.r_emb.weight"].clone()
        want["enc.e_emb.weight"][cfg["pad_eid"]] = 0
        want["enc.r_emb.weight"][cfg["pad_rid"]] = 0
    worst = 0.0
    for k, gw in want.items():
        got = eng.g[k].cpu()
        scale = gw.abs().max().item() + 1e-12
        err = (got - gw).abs().max().item()
        assert err <= gtol * scale, (k, err, scale)
        # the max-norm bound above is dominated by single 16-bit roundings; over the whole tensor the error is held tighter
        l2 = ((got - gw).norm() / (gw.norm() + 1e-30)).item()
        worst = max(worst, l2)
        assert l2 <= (L2TOL[precision] if gw.norm().item() > 1e-6 else 1.0), (k, l2)
    print(f"[{name} {precision}] worst relative L2 gradient error {worst:.2e}")
    del eng
    torch.cuda.empty_cache()
