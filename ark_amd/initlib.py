"""Parameter initialisation that is draw-for-draw compatible with the reference.

The reference creates stock torch modules in constructor order (kgvae/model/models.py:26-44 encoder,
:119-128 GRU decoder, :327-338 decoder-only); their default initialisers consume the global CPU
generator in that order.  Building the same stock modules in the same order under the same
``torch.manual_seed`` therefore yields bit-identical tensors -- including the draw for
``dec.out.weight`` that the weight tie then discards (models.py:130-134).
"""
from collections import OrderedDict

import torch
import torch.nn as nn


def build_modules(cfg):
    """-> (enc_parts or None, dec_parts) as dicts of freshly initialised stock modules"""
    D, n, V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
    mt = cfg["model_type"]
    enc = None
    dec = OrderedDict()
    if mt == "SAIL":
        Z = cfg["d_latent"]
        enc = OrderedDict()
        enc["e_emb"] = nn.Embedding(cfg["n_entities"], D, padding_idx=cfg.get("pad_eid"))
        enc["r_emb"] = nn.Embedding(cfg["n_relations"], D, padding_idx=cfg.get("pad_rid"))
        hid = max(3 * D, 2 * D)
        layers, in_dim = [], 3 * D
        for _ in range(n):
            layers += [nn.Linear(in_dim, hid), nn.GELU()]
            in_dim = hid
        enc["mlp"] = nn.Sequential(*layers)
        enc["mu"] = nn.Linear(hid, Z)
        enc["logv"] = nn.Linear(hid, Z)
        dec["tok_emb"] = nn.Embedding(V, D)
        dec["z_proj"] = nn.Linear(Z, D)
    elif mt == "ARK":
        dec["tok_emb"] = nn.Embedding(V, D)
        dec["pos_emb"] = nn.Embedding(cfg["seq_len"], D)
    elif mt == "t-SAIL":
        # AutoRegEncoder (reference models.py:66-76) and AutoRegDecoder (models.py:98-106), constructor order; the stock layers
        # are built with their DEFAULT dropout (0.1: the reference passes none) and deep-copied n times; `out` is not tied
        Z, W = cfg["d_latent"], 3 * D
        enc = OrderedDict()
        enc["e_emb"] = nn.Embedding(cfg["n_entities"], D, padding_idx=cfg.get("pad_eid"))
        enc["r_emb"] = nn.Embedding(cfg["n_relations"], D, padding_idx=cfg.get("pad_rid"))
        enc["txf"] = nn.TransformerEncoder(nn.TransformerEncoderLayer(W, cfg["n_heads"], batch_first=True), cfg.get("n_layers", 2))
        enc["mu"] = nn.Linear(W, Z)
        enc["logv"] = nn.Linear(W, Z)
        dec["tok_emb"] = nn.Embedding(V, D)
        dec["pos_emb"] = nn.Embedding(cfg["seq_len"], D)
        dec["z_proj"] = nn.Linear(Z, D)
        dec["txf"] = nn.TransformerDecoder(nn.TransformerDecoderLayer(D, cfg["n_heads"], batch_first=True), n)
        dec["out"] = nn.Linear(D, V)
        return enc, dec
    elif mt == "t-ARK":
        # DecoderOnlyTransformer (reference models.py:349-359): ONE stock encoder layer is initialised and
        # nn.TransformerEncoder deep-copies it n times -- every layer starts from the same tensors
        drop = cfg.get("dec_dropout", 0.1)
        dec["tok_emb"] = nn.Embedding(V, D)
        dec["pos_emb"] = nn.Embedding(cfg["seq_len"], D)
        layer = nn.TransformerEncoderLayer(D, cfg["n_heads"], batch_first=True, dropout=drop)
        dec["txf"] = nn.TransformerEncoder(layer, n)
        dec["out"] = nn.Linear(D, V)
        if cfg.get("tie_weights", True) and dec["out"].weight.shape == dec["tok_emb"].weight.shape:
            dec["out"].weight = dec["tok_emb"].weight
        return enc, dec
    else:
        raise NotImplementedError(f"Unknown model_type: {mt}")
    drop = cfg.get("dec_dropout", 0.1)
    dec["gru"] = nn.GRU(input_size=D, hidden_size=D, num_layers=n, batch_first=True, dropout=drop if n > 1 else 0.0)
    dec["out"] = nn.Linear(D, V)
    if cfg.get("tie_weights", True) and dec["out"].weight.shape == dec["tok_emb"].weight.shape:
        dec["out"].weight = dec["tok_emb"].weight
    return enc, dec


def init_state(cfg, seed=None):
    """state-dict-shaped OrderedDict of fresh parameters (optionally under torch.manual_seed(seed))"""
    if seed is not None:
        torch.manual_seed(seed)
    enc, dec = build_modules(cfg)
    sd = OrderedDict()
    for prefix, parts in (("enc", enc), ("dec", dec)):
        if parts is None:
            continue
        for name, mod in parts.items():
            for k, v in mod.state_dict().items():
                sd[f"{prefix}.{name}.{k}"] = v.detach().clone()
    if cfg.get("tie_weights", True) and "dec.out.weight" in sd and sd["dec.out.weight"].shape == sd["dec.tok_emb.weight"].shape:
        sd["dec.out.weight"] = sd["dec.tok_emb.weight"]
    return sd
