"""CPU oracle for the ARK / SAIL training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (PyTorch-CPU fp32, explicit math) of what
the reference computes on the path named in BASELINE.json `north_star`.  It is imported only by
tests/, by `__graft_entry__.smoke()` and by bench.py's `cpu_baseline` leg — never by the product
path (ark_amd/, kgvae/), which must fail loudly when the HIP extension is missing.

Parity pin: every function here is checked in tests/test_oracle_golden.py against golden vectors
that tools/make_golden.py produced by importing the real reference (`/root/reference`,
kgvae.model.models.SAIL / ARK) in the build container.  The reference ships no tests or golden
vectors of its own, so those generated fixtures are the pin.

Reference call sites restated (paths relative to the reference repo):
  encoder gather / masked mean pool ........ kgvae/model/models.py:46-58
  encoder MLP (Linear+GELU)^n .............. kgvae/model/models.py:32-41,60
  mu / logv heads, clamp, reparameterise ... kgvae/model/models.py:43-44,61-63
  kl_mean .................................. kgvae/model/models.py:199-200
  GRU decoder (tok_emb, z_proj, GRU, out) .. kgvae/model/models.py:116-142
  decoder-only ARK (tok+pos emb, GRU, out) . kgvae/model/models.py:323-345, 395-405
  t-SAIL (Transformer VAE) ................. kgvae/model/models.py:66-114 (AutoRegEncoder -- no logv clamp --, AutoRegDecoder
                                             over a memory of z_proj(z) repeated L times, stock post-norm layers)
  decoder-only t-ARK (Transformer) ......... kgvae/model/models.py:349-366 (stock nn.TransformerEncoderLayer: post-norm,
                                             ReLU feed-forward 2048, causal mask; restated as explicit math)
  ELBO assembly (ce + b*kl) ................ kgvae/experiments/ablation_study.py:59-73
  Adam step ................................ kgvae/experiments/ablation_study.py:571,76
  greedy decode (beam=1) ................... kgvae/model/models.py:262-266, 282-300
  beam decode (batch-shared beam) .......... kgvae/model/models.py:282-300
  compression bits (AR + KL) ............... kgvae/model/models.py:202-260 (SAIL), 473-520 (ARK)
  ARK.generate (greedy / temperature / top-k / nucleus sampling, draw order) ... kgvae/model/models.py:407-471
                                             (pinned by token sequences the reference sampled under fixed seeds)
  sequence codec ........................... kgvae/model/utils.py:70-78, 102-108
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

PAD, BOS, EOS = 0, 1, 2


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------
def init_params(cfg, seed):
    """Create the state-dict the reference would create under ``torch.manual_seed(seed)``.

    The reference builds stock torch modules in constructor order (models.py:26-44 for the
    encoder, :119-128 for the decoder, :149-186 for SAIL; :327-338 for ARK); stock initialisers
    consume the global CPU generator in that order, so re-creating the same stock modules in the
    same order reproduces every tensor bit-for-bit.  ``dec.out.weight`` is drawn
    and then discarded by the weight tie (models.py:130-134).
    """
    import torch.nn as nn

    torch.manual_seed(seed)
    D, n = cfg["d_model"], cfg["n_layers"]
    V = cfg["vocab_size"]
    P = OrderedDict()
    mt = cfg["model_type"]
    if mt == "SAIL":
        Z = cfg["d_latent"]
        e = nn.Embedding(cfg["n_entities"], D, padding_idx=cfg.get("pad_eid"))
        r = nn.Embedding(cfg["n_relations"], D, padding_idx=cfg.get("pad_rid"))
        P["enc.e_emb.weight"], P["enc.r_emb.weight"] = e.weight, r.weight
        hid = max(3 * D, 2 * D)
        in_dim = 3 * D
        for i in range(n):
            lin = nn.Linear(in_dim, hid)
            P[f"enc.mlp.{2 * i}.weight"], P[f"enc.mlp.{2 * i}.bias"] = lin.weight, lin.bias
            in_dim = hid
        mu, lv = nn.Linear(hid, Z), nn.Linear(hid, Z)
        P["enc.mu.weight"], P["enc.mu.bias"] = mu.weight, mu.bias
        P["enc.logv.weight"], P["enc.logv.bias"] = lv.weight, lv.bias
        tok = nn.Embedding(V, D)
        zp = nn.Linear(Z, D)
        P["dec.tok_emb.weight"] = tok.weight
        P["dec.z_proj.weight"], P["dec.z_proj.bias"] = zp.weight, zp.bias
    elif mt == "ARK":
        tok = nn.Embedding(V, D)
        pos = nn.Embedding(cfg["seq_len"], D)
        P["dec.tok_emb.weight"], P["dec.pos_emb.weight"] = tok.weight, pos.weight
    elif mt == "t-SAIL":
        # AutoRegEncoder (models.py:66-76) then AutoRegDecoder (models.py:98-106), constructor order; each stock layer is
        # initialised ONCE and deep-copied n times by nn.TransformerEncoder / nn.TransformerDecoder; `out` is NOT tied
        Z, H3 = cfg["d_latent"], 3 * D
        e = nn.Embedding(cfg["n_entities"], D, padding_idx=cfg.get("pad_eid"))
        r = nn.Embedding(cfg["n_relations"], D, padding_idx=cfg.get("pad_rid"))
        P["enc.e_emb.weight"], P["enc.r_emb.weight"] = e.weight, r.weight
        elayer = nn.TransformerEncoderLayer(H3, cfg["n_heads"], batch_first=True, dropout=0.0)
        for i in range(cfg.get("n_layers", 2)):
            for k, v in elayer.state_dict().items():
                P[f"enc.txf.layers.{i}.{k}"] = v.clone()
        mu, lv = nn.Linear(H3, Z), nn.Linear(H3, Z)
        P["enc.mu.weight"], P["enc.mu.bias"] = mu.weight, mu.bias
        P["enc.logv.weight"], P["enc.logv.bias"] = lv.weight, lv.bias
        tok, pos, zp = nn.Embedding(V, D), nn.Embedding(cfg["seq_len"], D), nn.Linear(Z, D)
        P["dec.tok_emb.weight"], P["dec.pos_emb.weight"] = tok.weight, pos.weight
        P["dec.z_proj.weight"], P["dec.z_proj.bias"] = zp.weight, zp.bias
        dlayer = nn.TransformerDecoderLayer(D, cfg["n_heads"], batch_first=True, dropout=0.0)
        for i in range(n):
            for k, v in dlayer.state_dict().items():
                P[f"dec.txf.layers.{i}.{k}"] = v.clone()
        out = nn.Linear(D, V)
        P["dec.out.weight"], P["dec.out.bias"] = out.weight, out.bias
        return _detach_tied(P, False)
    elif mt == "t-ARK":
        # DecoderOnlyTransformer (models.py:349-359): tok_emb, pos_emb, ONE stock TransformerEncoderLayer that
        # nn.TransformerEncoder deep-copies n times (every layer starts from the same tensors), out (tied)
        tok = nn.Embedding(V, D)
        pos = nn.Embedding(cfg["seq_len"], D)
        P["dec.tok_emb.weight"], P["dec.pos_emb.weight"] = tok.weight, pos.weight
        layer = nn.TransformerEncoderLayer(D, cfg["n_heads"], batch_first=True, dropout=0.0)
        lsd = layer.state_dict()
        for i in range(n):
            for k, v in lsd.items():
                P[f"dec.txf.layers.{i}.{k}"] = v.clone()
        out = nn.Linear(D, V)
        tied = cfg.get("tie_weights", True) and out.weight.shape == tok.weight.shape
        P["dec.out.weight"] = tok.weight if tied else out.weight
        P["dec.out.bias"] = out.bias
        return _detach_tied(P, tied)
    else:
        raise NotImplementedError(f"Unknown model_type: {mt}")
    gru = nn.GRU(D, D, n, batch_first=True, dropout=0.0)
    for l in range(n):
        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            P[f"dec.gru.{nm}_l{l}"] = getattr(gru, f"{nm}_l{l}")
    out = nn.Linear(D, V)
    tied = cfg.get("tie_weights", True) and out.weight.shape == tok.weight.shape
    P["dec.out.weight"] = tok.weight if tied else out.weight
    P["dec.out.bias"] = out.bias
    return _detach_tied(P, tied)


def _detach_tied(P, tied):
    out = OrderedDict()
    for k, v in P.items():
        out[k] = v.detach().clone()
    if tied:
        out["dec.out.weight"] = out["dec.tok_emb.weight"]  # same tensor object, as in the reference
    return out


def leaf_params(P):
    """unique trainable tensors in state-dict order (tied weight appears once)"""
    seen, out = set(), []
    for k, v in P.items():
        if id(v) not in seen:
            seen.add(id(v))
            out.append((k, v))
    return out


# --------------------------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------------------------
def encoder_pool(P, triples, pad_rid=None):
    """g[b] = masked mean over triples of [E[h] | R[r] | E[t]]   (models.py:47-58)"""
    E, R = P["enc.e_emb.weight"], P["enc.r_emb.weight"]
    x = torch.cat([E[triples[:, :, 0]], R[triples[:, :, 1]], E[triples[:, :, 2]]], dim=-1)
    if pad_rid is not None:
        m = (triples[:, :, 1] != pad_rid)
        cnt = m.sum(dim=1, keepdim=True).clamp(min=1)
        return (x * m.unsqueeze(-1)).sum(dim=1) / cnt
    return x.mean(dim=1)


def encoder_forward(P, triples, eps, cfg):
    """-> z, mu, logv   (models.py:46-64); eps replaces torch.randn_like(mu).  model_type t-SAIL: tsail_encoder_forward"""
    if cfg["model_type"] == "t-SAIL":
        return tsail_encoder_forward(P, triples, eps, cfg)
    g = encoder_pool(P, triples, cfg.get("pad_rid"))
    for i in range(cfg["n_layers"]):
        g = F.gelu(g @ P[f"enc.mlp.{2 * i}.weight"].t() + P[f"enc.mlp.{2 * i}.bias"])
    mu = g @ P["enc.mu.weight"].t() + P["enc.mu.bias"]
    logv = (g @ P["enc.logv.weight"].t() + P["enc.logv.bias"]).clamp(-10, 10)
    z = mu + eps * torch.exp(0.5 * logv)
    return z, mu, logv


def gru_stack(P, x, h0, n_layers, drop_masks=None):
    """Multi-layer GRU, batch_first, gate row order (r, z, n) as torch.nn.GRU (models.py:121-127).

    x [B,L,D]; h0 [n,B,D].  drop_masks: optional list of n-1 tensors [B,L,D] already scaled by
    1/(1-p), applied to the output of layers 0..n-2 (torch's inter-layer dropout)."""
    B, L, D = x.shape
    inp = x
    for l in range(n_layers):
        Wih, Whh = P[f"dec.gru.weight_ih_l{l}"], P[f"dec.gru.weight_hh_l{l}"]
        bih, bhh = P[f"dec.gru.bias_ih_l{l}"], P[f"dec.gru.bias_hh_l{l}"]
        gi_all = inp @ Wih.t() + bih  # time-batched input projection
        h = h0[l]
        outs = []
        for t in range(L):
            gi = gi_all[:, t]
            gh = h @ Whh.t() + bhh
            r = torch.sigmoid(gi[:, :D] + gh[:, :D])
            zt = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
            nt = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
            h = (1.0 - zt) * nt + zt * h
            outs.append(h)
        inp = torch.stack(outs, dim=1)
        if drop_masks is not None and l < n_layers - 1:
            inp = inp * drop_masks[l]
    return inp


def decoder_forward(P, z, seq_in, cfg, drop_masks=None):
    """SAIL decoder: logits[B,L,V]   (models.py:136-142).  model_type t-SAIL: tsail_decoder_forward"""
    if cfg["model_type"] == "t-SAIL":
        return tsail_decoder_forward(P, z, seq_in, cfg)
    n = cfg["n_layers"]
    x = P["dec.tok_emb.weight"][seq_in]
    h0 = torch.tanh(z @ P["dec.z_proj.weight"].t() + P["dec.z_proj.bias"])
    y = gru_stack(P, x, h0.unsqueeze(0).repeat(n, 1, 1), n, drop_masks)
    return y @ P["dec.out.weight"].t() + P["dec.out.bias"]


def ark_forward(P, seq_in, cfg, drop_masks=None):
    """decoder-only ARK: logits[B,L,V]   (models.py:340-345); model_type t-ARK: the Transformer of tark_forward"""
    if cfg["model_type"] == "t-ARK":
        return tark_forward(P, seq_in, cfg, drop_masks)
    n = cfg["n_layers"]
    B, L = seq_in.shape
    x = P["dec.tok_emb.weight"][seq_in] + P["dec.pos_emb.weight"][torch.arange(L)].unsqueeze(0)
    h0 = torch.zeros(n, B, cfg["d_model"], dtype=x.dtype)
    y = gru_stack(P, x, h0, n, drop_masks)
    return y @ P["dec.out.weight"].t() + P["dec.out.bias"]


def layer_norm(x, g, b, eps=1e-5):
    """nn.LayerNorm over the last dimension (biased variance)"""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def self_attention(x, w_in, b_in, w_out, b_out, n_heads, causal, drop_p=None, mem=None, key_pad=None):
    """nn.MultiheadAttention(batch_first=True): queries from x [B, L, D], keys / values from `mem` [B, S, D] (default x
    itself); packed in-projection (rows q | k | v), heads of D / n_heads, scores / sqrt(head dim), -inf above the diagonal
    (causal: the reference's boolean triu mask, models.py:113,364) and on padded keys (key_pad [B, S] True = ignore,
    models.py:87), softmax, optional dropout keep-scales ON THE PROBABILITIES, out-projection"""
    B, L, D = x.shape
    src = x if mem is None else mem
    S = src.shape[1]
    dh = D // n_heads
    q = (x @ w_in[:D].t() + b_in[:D]).reshape(B, L, n_heads, dh).transpose(1, 2)
    k = (src @ w_in[D:2 * D].t() + b_in[D:2 * D]).reshape(B, S, n_heads, dh).transpose(1, 2)
    v = (src @ w_in[2 * D:].t() + b_in[2 * D:]).reshape(B, S, n_heads, dh).transpose(1, 2)
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(L, S, dtype=torch.bool), 1), float("-inf"))
    if key_pad is not None:
        sc = sc.masked_fill(key_pad[:, None, None, :], float("-inf"))
    pr = torch.softmax(sc, dim=-1)
    if drop_p is not None:
        pr = pr * drop_p
    o = (pr @ v).transpose(1, 2).reshape(B, L, D)
    return o @ w_out.t() + b_out


def txf_decoder_layer(P, pre, x, mem, n_heads):
    """nn.TransformerDecoderLayer, stock defaults (post-norm, ReLU): x = LN1(x + SA_causal(x)); x = LN2(x + MHA(x, mem));
    x = LN3(x + FF(x))   (dropout-free: the parity / evaluation numerics)"""
    g = lambda k: P[pre + k]
    x = layer_norm(x + self_attention(x, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), g("self_attn.out_proj.weight"),
                                      g("self_attn.out_proj.bias"), n_heads, True), g("norm1.weight"), g("norm1.bias"))
    x = layer_norm(x + self_attention(x, g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias"),
                                      g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"), n_heads, False, mem=mem),
                   g("norm2.weight"), g("norm2.bias"))
    ff = torch.relu(x @ g("linear1.weight").t() + g("linear1.bias")) @ g("linear2.weight").t() + g("linear2.bias")
    return layer_norm(x + ff, g("norm3.weight"), g("norm3.bias"))


def tsail_encoder_forward(P, triples, eps, cfg):
    """AutoRegEncoder.forward (models.py:78-95): [E[h] | R[r] | E[t]] per triple -> Transformer encoder over the triples
    (padded triples masked as keys) -> masked mean -> mu / logv heads -- NO clamp on logv (models.py:93) -> z"""
    E, R = P["enc.e_emb.weight"], P["enc.r_emb.weight"]
    x = torch.cat([E[triples[:, :, 0]], R[triples[:, :, 1]], E[triples[:, :, 2]]], dim=-1)
    pad_rid = cfg.get("pad_rid")
    valid = (triples[:, :, 1] != pad_rid) if pad_rid is not None else None
    for i in range(cfg.get("n_layers", 2)):
        x = txf_encoder_layer(P, f"enc.txf.layers.{i}.", x, cfg["n_heads"], False, key_pad=None if valid is None else ~valid)
    if valid is not None:
        g = (x * valid.unsqueeze(-1)).sum(1) / valid.sum(1, keepdim=True).clamp(min=1)
    else:
        g = x.mean(1)
    mu = g @ P["enc.mu.weight"].t() + P["enc.mu.bias"]
    logv = g @ P["enc.logv.weight"].t() + P["enc.logv.bias"]
    return mu + eps * torch.exp(0.5 * logv), mu, logv


def tsail_decoder_forward(P, z, seq_in, cfg):
    """AutoRegDecoder.forward (models.py:108-114): tok + pos embeddings; the memory is z_proj(z) repeated L times; causal
    Transformer decoder; untied output projection"""
    B, L = seq_in.shape
    x = P["dec.tok_emb.weight"][seq_in] + P["dec.pos_emb.weight"][torch.arange(L)].unsqueeze(0)
    mem = (z @ P["dec.z_proj.weight"].t() + P["dec.z_proj.bias"]).unsqueeze(1).repeat(1, L, 1)
    for i in range(cfg["n_layers"]):
        x = txf_decoder_layer(P, f"dec.txf.layers.{i}.", x, mem, cfg["n_heads"])
    return x @ P["dec.out.weight"].t() + P["dec.out.bias"]


def txf_encoder_layer(P, pre, x, n_heads, causal, masks=None, key_pad=None):
    """nn.TransformerEncoderLayer, stock defaults (post-norm, ReLU feed-forward): x = LN1(x + drop1(SA(x)));
    x = LN2(x + drop2(W2 drop(relu(W1 x)))).  masks (optional): dict of dropout keep-scales for the four dropout sites
    ("attn" [B,h,L,L], "sa" / "ff2" [B,L,D], "ff1" [B,L,F]); None = no dropout (eval, or p = 0)."""
    m = masks or {}
    sa = self_attention(x, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"],
                        P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"], n_heads, causal, m.get("attn"),
                        key_pad=key_pad)
    if "sa" in m:
        sa = sa * m["sa"]
    x = layer_norm(x + sa, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    h = torch.relu(x @ P[pre + "linear1.weight"].t() + P[pre + "linear1.bias"])
    if "ff1" in m:
        h = h * m["ff1"]
    ff = h @ P[pre + "linear2.weight"].t() + P[pre + "linear2.bias"]
    if "ff2" in m:
        ff = ff * m["ff2"]
    return layer_norm(x + ff, P[pre + "norm2.weight"], P[pre + "norm2.bias"])


def tark_forward(P, seq_in, cfg, masks=None):
    """decoder-only Transformer t-ARK: logits [B, L, V]   (DecoderOnlyTransformer.forward, models.py:361-366)"""
    B, L = seq_in.shape
    x = P["dec.tok_emb.weight"][seq_in] + P["dec.pos_emb.weight"][torch.arange(L)].unsqueeze(0)
    for i in range(cfg["n_layers"]):
        x = txf_encoder_layer(P, f"dec.txf.layers.{i}.", x, cfg["n_heads"], True, None if masks is None else masks[i])
    return x @ P["dec.out.weight"].t() + P["dec.out.bias"]


def kl_mean(mu, logv):
    """mean over ALL B*Z elements (models.py:199-200)"""
    return -0.5 * torch.mean(1 + logv - mu.pow(2) - logv.exp())


def token_ce(logits, targets):
    """mean over non-PAD targets (ablation_study.py:65-69)"""
    V = logits.shape[-1]
    return F.cross_entropy(logits.reshape(-1, V), targets.reshape(-1), ignore_index=PAD)


def sail_elbo(P, triples, seq, eps, beta, cfg, drop_masks=None):
    """one SAIL training forward: loss = ce + beta*kl  (ablation_study.py:59-73)"""
    z, mu, logv = encoder_forward(P, triples, eps, cfg)
    logits = decoder_forward(P, z, seq[:, :-1], cfg, drop_masks)
    ce = token_ce(logits, seq[:, 1:])
    kl = kl_mean(mu, logv)
    return ce + beta * kl, ce, kl, logits, mu, logv


def ark_loss(P, seq, cfg, drop_masks=None):
    """one ARK training forward (train.py:42-58)"""
    logits = ark_forward(P, seq[:, :-1], cfg, drop_masks)
    ce = token_ce(logits, seq[:, 1:])
    return ce, logits


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
def adam_init(leaves):
    return {"step": 0, "m": [torch.zeros_like(p) for _, p in leaves], "v": [torch.zeros_like(p) for _, p in leaves]}


def adam_step(leaves, grads, state, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad), in place."""
    state["step"] += 1
    t = state["step"]
    bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
    for (_, p), g, m, v in zip(leaves, grads, state["m"], state["v"]):
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_lr(base_lr, epoch, t_max, eta_min=1e-6):
    """closed form of CosineAnnealingLR stepped once per epoch (ablation_study.py:577-581)"""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def beta_schedule(cfg, epoch):
    """ablation_study.py:590-591"""
    return cfg["beta0"] + (cfg["beta1"] - cfg["beta0"]) * epoch / cfg["num_epochs"]


def train_step(P, state, batch, cfg, lr, beta=1.0, eps=None, drop_masks=None):
    """forward + backward + Adam on the oracle's own tensors; returns (loss, ce, kl, grads)"""
    leaves = leaf_params(P)
    for _, p in leaves:
        p.requires_grad_(True)
        p.grad = None
    triples, seq = batch
    if cfg["model_type"] in ("SAIL", "t-SAIL"):
        loss, ce, kl, *_ = sail_elbo(P, triples, seq, eps, beta, cfg, drop_masks)
    else:
        ce, _ = ark_loss(P, seq, cfg, drop_masks)
        loss, kl = ce, torch.zeros(())
    loss.backward()
    grads = [p.grad.detach().clone() for _, p in leaves]
    # nn.Embedding(padding_idx=...) rows never receive gradient
    if cfg["model_type"] in ("SAIL", "t-SAIL"):
        names = [k for k, _ in leaves]
        if cfg.get("pad_eid") is not None:
            grads[names.index("enc.e_emb.weight")][cfg["pad_eid"]] = 0
        if cfg.get("pad_rid") is not None:
            grads[names.index("enc.r_emb.weight")][cfg["pad_rid"]] = 0
    with torch.no_grad():
        for _, p in leaves:
            p.requires_grad_(False)
        adam_step(leaves, grads, state, lr)
    return loss.item(), ce.item(), float(kl.detach()), grads


# --------------------------------------------------------------------------------------------
# decode + codec
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_decode(P, z, cfg, bos=BOS, eos=EOS):
    """token sequences of SAIL.decode_latent(z, beam=1): start at BOS, re-run the decoder on the
    whole prefix, take argmax of the last position, stop when every row ends in EOS
    (models.py:282-300 with beam=1; log_softmax is monotone so argmax(logits) == topk(1))."""
    B = z.shape[0]
    s = torch.full((B, 1), bos, dtype=torch.long)
    for _ in range(cfg["seq_len"] - 1):
        nxt = decoder_forward(P, z, s, cfg)[:, -1].argmax(dim=-1, keepdim=True)
        s = torch.cat([s, nxt], dim=1)
        if bool((s[:, -1] == eos).all()):
            break
    return s


@torch.no_grad()
def beam_decode(P, z, cfg, beam, bos=BOS, eos=EOS):
    """token sequences of SAIL.decode_latent(z, beam > 1), restated from models.py:282-300: ONE beam shared by the whole
    batch; every step re-runs the decoder on each beam's prefix, extends it by its `beam` best tokens per row, ranks
    the candidates by the MEAN over the batch of their accumulated log-probabilities (stable descending sort) and
    keeps the first `beam`; stops when every row of every kept beam ends in EOS; returns the best beam's sequences."""
    B = z.shape[0]
    beams = [(torch.full((B, 1), bos, dtype=torch.long), torch.zeros(B))]
    for _ in range(cfg["seq_len"] - 1):
        cand = []
        for s, lp in beams:
            logp = F.log_softmax(decoder_forward(P, z, s, cfg)[:, -1], dim=-1)
            top_lp, ids = logp.topk(beam, dim=-1)
            for k in range(beam):
                cand.append((torch.cat([s, ids[:, k:k + 1]], 1), lp + top_lp[:, k]))
        cand.sort(key=lambda c: c[1].mean().item(), reverse=True)
        beams = cand[:beam]
        if all(bool((s[:, -1] == eos).all()) for s, _ in beams):
            break
    return beams[0][0]


@torch.no_grad()
def posterior_bits(P, triples, seq, eps, cfg):
    """per-graph (ar_bits, kl_bits) as SAIL.posterior_bits / bits_per_sequence compute them (models.py:202-260):
    z ~ q(z|x) with the given eps; AR bits = sum over target positions t >= 1, up to the first PAD target, of
    -log2 softmax(dec(z, seq[:t])[-1])[seq[t]], one decoder run per prefix exactly as the reference does;
    KL bits = KL summed over the latent dimension / ln 2."""
    ln2 = math.log(2)
    ar, kl = [], []
    for b in range(seq.shape[0]):
        z, mu, logv = encoder_forward(P, triples[b:b + 1], eps[b:b + 1], cfg)
        total = 0.0
        for t in range(1, seq.shape[1]):
            tgt = int(seq[b, t])
            if tgt == PAD:
                break
            logits = decoder_forward(P, z, seq[b:b + 1, :t], cfg)[:, -1]
            total += -float(F.log_softmax(logits, dim=-1)[0, tgt]) / ln2
        ar.append(total)
        kl.append(float(-0.5 * torch.sum(1 + logv - mu.pow(2) - logv.exp(), dim=1)) / ln2)
    return ar, kl


def next_token_filter(logits, temperature=1.0, top_p=0.0, top_k=0):
    """the filtering half of ARK.generate(sample=True) (models.py:431-449) on a batch of last-position logits:
    temperature, softmax, top-k mask + renormalise; with a nucleus: descending sort, every sorted position whose
    PREDECESSOR's cumulative mass already exceeds top_p is zeroed (so the token that crosses top_p stays in), renormalise.
    -> (probs, sorted_probs, sorted_idx); the last two are None without a nucleus.  Pinned through ark_generate by the
    token sequences the reference sampled (tests/golden/ark_sampling.npz)."""
    if temperature and temperature != 1.0:
        logits = logits / float(temperature)
    probs = F.softmax(logits, dim=-1)
    if top_k and top_k > 0:
        keep = probs.topk(top_k, dim=-1).indices
        probs = probs * torch.zeros_like(probs).scatter_(-1, keep, 1.0)
        probs = probs / probs.sum(dim=-1, keepdim=True).clamp_min(1e-12)
    if not (top_p and 0.0 < top_p < 1.0):
        return probs, None, None
    sp, si = probs.sort(dim=-1, descending=True)
    over = sp.cumsum(dim=-1) > top_p
    cut = torch.zeros_like(over)
    cut[..., 1:] = over[..., :-1]
    sp = sp.masked_fill(cut, 0.0)
    sp = sp / sp.sum(dim=-1, keepdim=True).clamp_min(1e-12)
    return probs, sp, si


@torch.no_grad()
def sampling_distribution(logits, temperature=1.0, top_p=0.0, top_k=0):
    """dense next-token probabilities of ARK.generate(sample=True): next_token_filter scattered back to vocabulary order"""
    probs, sp, si = next_token_filter(logits, temperature, top_p, top_k)
    return probs if sp is None else torch.zeros_like(probs).scatter_(-1, si, sp)


@torch.no_grad()
def ark_generate(P, cfg, batch_size, sample=False, temperature=1.0, top_p=0.0, top_k=0, bos=BOS, eos=EOS):
    """token sequences of ARK.generate (models.py:407-471): the decoder is re-run on the whole prefix, the next token is
    the argmax or a draw from next_token_filter's distribution, generation stops once EVERY row's last token is EOS and
    the result is padded with EOS to seq_len.  The draws consume torch's GLOBAL generator in the reference's pattern:
    with a nucleus ONE torch.multinomial PER ROW over the sorted distribution (the drawn position is mapped back
    through the sort), otherwise one batched torch.multinomial over the dense distribution."""
    B, seq_len = batch_size, cfg["seq_len"]
    s = torch.full((B, 1), bos, dtype=torch.long)
    for _ in range(seq_len - 1):
        logits = ark_forward(P, s, cfg)[:, -1]
        if not sample:
            nxt = logits.argmax(dim=-1, keepdim=True)
        else:
            probs, sp, si = next_token_filter(logits, temperature, top_p, top_k)
            if sp is None:
                nxt = torch.multinomial(probs, 1)
            else:
                nxt = torch.stack([si[b, torch.multinomial(sp[b], 1)] for b in range(B)])
        s = torch.cat([s, nxt], dim=1)
        if bool((s[:, -1] == eos).all()):
            break
    if s.shape[1] < seq_len:
        s = torch.cat([s, torch.full((B, seq_len - s.shape[1]), eos, dtype=torch.long)], dim=1)
    return s[:, :seq_len]


@torch.no_grad()
def ark_posterior_bits(P, seq, cfg):
    """per-sequence AR bits of ARK.posterior_bits / bits_per_sequence (models.py:473-520): sum over target positions
    t >= 1, up to the first PAD target, of -log2 softmax(dec(seq[:t])[-1])[seq[t]], one decoder run per prefix."""
    ln2 = math.log(2)
    out = []
    for b in range(seq.shape[0]):
        total = 0.0
        for t in range(1, seq.shape[1]):
            tgt = int(seq[b, t])
            if tgt == PAD:
                break
            logits = ark_forward(P, seq[b:b + 1, :t], cfg)[:, -1]
            total += -float(F.log_softmax(logits, dim=-1)[0, tgt]) / ln2
        out.append(total)
    return out


@torch.no_grad()
def sampling_distribution_loops(logits, temperature=1.0, top_p=0.0, top_k=0):
    """the same distribution as sampling_distribution, written row by row with explicit Python loops (an independent
    second statement of models.py:431-449 the vectorised one is cross-checked against)"""
    if temperature and temperature != 1.0:
        logits = logits / float(temperature)
    probs = F.softmax(logits, dim=-1)
    out = torch.zeros_like(probs)
    for b in range(probs.shape[0]):
        p = probs[b].clone()
        if top_k and top_k > 0:
            keep = sorted(range(p.numel()), key=lambda i: (-float(p[i]), i))[:top_k]
            m = torch.zeros_like(p)
            m[keep] = 1.0
            p = p * m
            p = p / p.sum().clamp_min(1e-12)
        if top_p and 0.0 < top_p < 1.0:
            order = sorted(range(p.numel()), key=lambda i: (-float(p[i]), i))
            acc, kept = 0.0, []
            for i in order:
                kept.append(i)          # the token that takes the cumulative mass past top_p is still kept ...
                acc += float(p[i])
                if acc > top_p:
                    break               # ... everything after it is cut
            q = torch.zeros_like(p)
            q[kept] = p[kept]
            p = q / q.sum().clamp_min(1e-12)
        out[b] = p
    return out


def triples_to_seq(triples, ent_base, rel_base, seq_len):
    """[BOS, (ent_base+h, rel_base+r, ent_base+t)*, EOS, PAD...]  (utils.py:102-108)"""
    s = [BOS]
    for h, r, t in triples:
        s += [ent_base + h, rel_base + r, ent_base + t]
    s.append(EOS)
    return s + [PAD] * (seq_len - len(s))


def seq_to_triples(seq, ent_base, rel_base):
    """inverse codec: skip BOS, read 3 tokens at a time until EOS at a triple boundary or fewer
    than 3 tokens remain (utils.py:70-78)"""
    seq = [int(x) for x in seq]
    out, i = [], 1
    while i + 2 < len(seq) and seq[i] != EOS:
        out.append((seq[i] - ent_base, seq[i + 1] - rel_base, seq[i + 2] - ent_base))
        i += 3
    return out
