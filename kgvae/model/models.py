"""SAIL (VAE: MLP encoder + GRU decoder) and ARK (decoder-only GRU) on the MI355X engine.

Drop-in for the reference classes of the same import path (kgvae/model/models.py:144-320 SAIL,
:368-520 ARK): same constructor (`config` dict), same state-dict keys and parameter-creation
order (so `torch.manual_seed(s)` yields the same initial weights), same `forward` signatures and
return values.  What is different is where the arithmetic runs: every op of forward, backward and
the optimiser step is a hand-written gfx950 kernel reached through the C-ABI in
include/ark_amd.h (ark_amd.engine.Engine).  There is no torch fallback: calling forward on a
CPU-resident model raises.

Two ways to train:
  * reference style -- `logits, mu, logv = model(triples, seq[:, :-1])`, `F.cross_entropy`,
    `loss.backward()`, `torch.optim.Adam`: forward/backward run on the engine through a
    torch.autograd.Function (logits are materialised, as in the reference);
  * fused -- `model.train_step(triples, seq, beta=b, lr=lr)`: ELBO, backward and Adam stay on the
    device (what kgvae.experiments.train and bench.py use).

The Transformer variants `t-ARK` (decoder-only) and `t-SAIL` (VAE) run on ark_amd.txf_engine.TxfEngine; unknown model
types raise NotImplementedError as in the reference (models.py:172,197,393).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ark_amd import initlib
from ark_amd._lib import ArkError
from kgvae.model.utils import canonical_graph_string


class _Holder(nn.Module):
    """named container of stock modules (keeps the reference's state-dict key layout)"""

    def __init__(self, parts):
        super().__init__()
        for name, mod in parts.items():
            self.add_module(name, mod)


class _EngineModel(nn.Module):
    _kinds = ()

    def __init__(self, config):
        super().__init__()
        self.config = config
        if config["model_type"] not in self._kinds:
            raise NotImplementedError(f"Unknown model_type: {config['model_type']}")
        enc, dec = initlib.build_modules(config)
        if enc is not None:
            self.enc = _Holder(enc)
        self.dec = _Holder(dec)
        self._eng = None
        self._versions = None
        self.precision = config.get("precision", "mixed")
        # load_state_dict copies IN PLACE into the aliased flat buffer: the engine's 16-bit weight shadows are stale
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._params_written())

    # -- engine binding ---------------------------------------------------------------------
    def engine(self):
        """the MI355X engine bound to this module's parameters (created on first use)"""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise ArkError("this model runs on hand-written gfx950 kernels only: move it to a GPU "
                           "(model.to('cuda')); there is no CPU fallback")
        named = dict(self.named_parameters())
        if self._eng is None or self._eng.device != dev:
            from ark_amd.engine import Engine
            from ark_amd.txf_engine import TxfEngine
            import torch.distributed as dist
            live = dist.is_available() and dist.is_initialized()
            cls = TxfEngine if self.config["model_type"] in ("t-ARK", "t-SAIL") else Engine
            self._eng = cls(self.config, dev, precision=self.precision, world_size=dist.get_world_size() if live else 1,
                            rank=dist.get_rank() if live else 0)
            self._adopt(named)
        else:
            for k, prm in named.items():  # .to() re-allocates storage
                if prm.data_ptr() != self._eng.p[k].data_ptr():
                    self._adopt(named)
                    break
            else:
                # in-place writers (torch.optim, load_state_dict, p.copy_) keep the storage and bump the version
                ver = tuple(prm._version for prm in named.values())
                if ver != self._versions:
                    self._versions = ver
                    self._eng.mark_params_dirty()
        self._eng.training = self.training
        return self._eng

    def _params_written(self):
        if self._eng is not None:
            self._eng.mark_params_dirty()

    def _adopt(self, named):
        eng = self._eng
        with torch.no_grad():
            for k, prm in named.items():
                eng.p[k].copy_(prm.data)
                prm.data = eng.p[k]      # parameters now alias the engine's flat buffer
        eng.mark_params_dirty()
        self._versions = tuple(prm._version for prm in named.values())

    def _params_in_order(self):
        return list(self.named_parameters())

    # -- fused training step ----------------------------------------------------------------
    def train_step(self, triples, seq, beta=1.0, lr=None, eps=None, grad_sync=None, ce_count=None, dp=False, graph=False):
        """forward + ELBO + backward + Adam on the device; returns the device tensor
        [loss, ce, kl, token-loss sum] without synchronising.  `graph=True` replays the step from a
        cached hipGraph (captured on the first batch of each shape; needs eps and grad_sync unset)."""
        eng = self.engine()
        eng.set_hyper(lr=lr, beta=beta)
        if graph and eps is None and grad_sync is None:
            return eng.graphed_train_step(self._tri(triples), seq, ce_count=ce_count, dp=dp)
        return eng.train_step(self._tri(triples), seq, eps, grad_sync=grad_sync, ce_count=ce_count, dp=dp)

    @torch.no_grad()
    def eval_loss(self, triples, seq, beta=1.0, eps=None):
        eng = self.engine()
        eng.set_hyper(beta=beta)
        return eng.eval_loss(self._tri(triples), seq, eps)

    def _tri(self, triples):
        return triples


def _check_generation(eng, ctx):
    """the engine keeps ONE set of saved activations: backward must belong to its most recent forward"""
    if eng.fwd_gen != ctx.gen:
        raise ArkError("backward() of an earlier forward: the engine has run another forward (training, evaluation or "
                       "generation) since, and its saved activations are gone -- call loss.backward() before the next "
                       "forward, as the reference's train_epoch does (kgvae/experiments/ablation_study.py:63-76)")


class _SailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, triples, seq_in, eps, *params):
        eng = model.engine()   # (notices in-place writes of an external optimiser through the parameter versions)
        B, Lq = seq_in.shape
        w = eng.forward(triples, seq_in, eps, with_loss=False, L_run=Lq)
        V = eng.V
        logits = eng._logits(w)[:Lq * B, :V].reshape(Lq, B, V).permute(1, 0, 2).contiguous()
        ctx.model, ctx.shape, ctx.gen = model, (B, Lq, V), eng.fwd_gen
        return logits, w["mu"].clone(), w["logv"].clone()

    @staticmethod
    def backward(ctx, dlogits, dmu, dlogv):
        model = ctx.model
        eng = model._eng
        _check_generation(eng, ctx)
        B, Lq, V = ctx.shape
        w = eng.ws
        buf = eng._logits(w)
        buf.zero_()
        if dlogits is not None:
            buf[:Lq * B, :V].copy_(dlogits.permute(1, 0, 2).reshape(Lq * B, V))
        Z = eng.Z
        ext = torch.zeros(B, 2 * Z, device=buf.device)
        if dmu is not None:
            ext[:, :Z] = dmu
        if dlogv is not None:
            raw = w["head"][:, Z:]
            if model.config["model_type"] == "t-SAIL":              # (no clamp in the Transformer encoder, reference models.py:93)
                ext[:, Z:] = dlogv
            else:
                ext[:, Z:] = dlogv * ((raw >= -10) & (raw <= 10))   # clamp(-10, 10) passes gradient inside only
        eng.set_hyper(kl_norm=0.0)   # the KL gradient arrives through dmu / dlogv here
        eng.backward(ext_dhead=ext)
        grads = tuple(eng.g[k].clone() for k, _ in model._params_in_order())
        return (None, None, None, None) + grads


class _ArkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, seq_in, *params):
        eng = model.engine()
        B, Lq = seq_in.shape
        w = eng.forward(None, seq_in, None, with_loss=False, L_run=Lq)
        V = eng.V
        ctx.model, ctx.shape, ctx.gen = model, (B, Lq, V), eng.fwd_gen
        return eng._logits(w)[:Lq * B, :V].reshape(Lq, B, V).permute(1, 0, 2).contiguous()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        eng = model._eng
        _check_generation(eng, ctx)
        B, Lq, V = ctx.shape
        buf = eng._logits(eng.ws)
        buf.zero_()
        buf[:Lq * B, :V].copy_(dlogits.permute(1, 0, 2).reshape(Lq * B, V))
        eng.backward()
        return (None, None) + tuple(eng.g[k].clone() for k, _ in model._params_in_order())


class SAIL(_EngineModel):
    """VAEs (reference models.py:144-320): `SAIL` = masked-mean-pooled triple embeddings -> (Linear+GELU)^n -> mu/logv -> z ->
    GRU decoder; `t-SAIL` = Transformer encoder over the triples -> mu/logv (no clamp) -> z -> Transformer decoder over a
    memory of z_proj(z) (AutoRegEncoder / AutoRegDecoder, models.py:66-114; engine: ark_amd.txf_engine.TxfEngine)."""
    _kinds = ("SAIL", "t-SAIL")

    def __init__(self, config):
        super().__init__(config)
        if config["model_type"] == "SAIL":
            print("Using MLP encoder")
            print("Using GRU Decoder")

    def forward(self, triples, seq_in, eps=None):
        """-> (logits [B,L,V], mu [B,Z], logv [B,Z]); eps defaults to N(0,1) drawn on the device"""
        seq_in = seq_in.contiguous()
        triples = triples.contiguous()
        if eps is None:
            eps = torch.randn(seq_in.shape[0], self.config["d_latent"], device=seq_in.device)
        prm = [p for _, p in self._params_in_order()]
        return _SailFn.apply(self, triples, seq_in, eps, *prm)

    def kl_mean(self, mu, logv):
        # mean over ALL B*Z elements (reference models.py:199-200)
        return -0.5 * torch.mean(1 + logv - mu.pow(2) - logv.exp())

    @torch.no_grad()
    def encode(self, triples, eps=None):
        """-> (z, mu, logv), the reference's `self.enc(triples)`"""
        eng = self.engine()
        B = triples.shape[0]
        seq = torch.zeros(B, eng.seq_len, dtype=torch.long, device=triples.device)
        if eps is None:
            eps = torch.randn(B, eng.Z, device=triples.device)
        w = eng.forward(triples.contiguous(), seq, eps, with_loss=False, L_run=1)
        return w["z"].clone(), w["mu"].clone(), w["logv"].clone()

    # -- generation (reference models.py:262-315) ------------------------------------------
    @torch.no_grad()
    def decode_latent(self, z, seq_len, special_tokens, seq_to_triples, ent_base, rel_base, beam=4):
        self.eval()
        z = z.to(next(self.parameters()).device, dtype=torch.float32)
        return self.beam_generate(seq_len, special_tokens, seq_to_triples, z, ent_base, rel_base, beam=beam)

    @torch.no_grad()
    def beam_generate(self, seq_len, special_tokens, seq_to_triples, z, ent_base, rel_base, beam=4):
        eng = self.engine()
        if beam == 1:
            # beam 1 == greedy argmax per row; the causal GRU is advanced one token per step on
            # the device instead of re-running the whole prefix
            best = eng.greedy_decode(z, max_len=seq_len - 1, bos=special_tokens["BOS"], eos=special_tokens["EOS"]).cpu()
            return [seq_to_triples(row, special_tokens, ent_base, rel_base) for row in best]
        # beam > 1: the reference's batch-shared beam (candidates ranked by the batch-MEAN log-prob); every beam keeps its
        # GRU state on the device and advances one token per step (Engine.beam_decode)
        best = eng.beam_decode(z, beam, max_len=seq_len - 1, bos=special_tokens["BOS"], eos=special_tokens["EOS"]).cpu()
        return [seq_to_triples(row, special_tokens, ent_base, rel_base) for row in best]

    @torch.no_grad()
    def count_unique_graphs(self, latent_dim, decode_latent_fn, num_samples=1000, beam=1):
        self.eval()
        zs = torch.randn((num_samples, latent_dim), device=next(self.parameters()).device)
        graphs = decode_latent_fn(zs, beam=beam)
        uniq = {canonical_graph_string(g) for g in graphs}
        print(f"\n[Graph Diversity from {num_samples} Random Latents]")
        print(f"  Unique graphs generated: {len(uniq)}")
        print(f"  Diversity ratio: {len(uniq) / num_samples:.3f}")
        return uniq

    @torch.no_grad()
    def generate_test_graphs(self, test_loader, seq_len, special_tokens, seq_to_triples, ent_base, rel_base,
                             beam_width=4, num_generated_test_graphs=1000, device="cuda"):
        out = []
        for triples, _ in test_loader:
            z, _, _ = self.encode(triples.to(device))
            out.extend(self.beam_generate(seq_len, special_tokens, seq_to_triples, z, ent_base, rel_base, beam=beam_width))
            if len(out) >= num_generated_test_graphs:
                return out[:num_generated_test_graphs]
        return out

    # -- compression bits (reference models.py:202-260), one teacher-forced pass per batch -----
    @torch.no_grad()
    def posterior_bits(self, dataset, device, pad_id=0, sample_frac=0.1, desc="posterior bits", batch_size=256, eps=None):
        """Same statistics as the reference (AR bits of the sequence given z ~ q(z|x), KL bits summed
        over the latent dimension; models.py:202-260), computed with ONE decoder pass per batch: for a causal GRU
        the reference's prefix-by-prefix loop re-computes exactly these per-token log-probabilities.
        `eps` ([n, Z], optional): the reparameterisation noise per item instead of fresh N(0,1) draws."""
        eng = self.engine()
        n = max(1, int(sample_frac * len(dataset)))
        ln2 = math.log(2)
        records = []
        was = eng.training
        eng.training = False
        try:
            for i0 in range(0, n, batch_size):
                idx = list(range(i0, min(n, i0 + batch_size)))
                if hasattr(dataset, "tensorize"):
                    tri, seq = dataset.tensorize(idx)
                else:
                    items = [dataset[i] for i in idx]
                    tri, seq = torch.stack([a for a, _ in items]), torch.stack([b for _, b in items])
                tri, seq = tri.to(device), seq.to(device)
                B = seq.shape[0]
                e = torch.randn(B, eng.Z, device=device) if eps is None else eps[i0:i0 + B].to(device, dtype=torch.float32).contiguous()
                w = eng.forward(tri, seq, e, with_dlogits=False)
                Bp = eng._B   # (a ragged batch is padded to 16 rows inside the engine)
                ar = (w["row_loss"][:eng.L * Bp].reshape(eng.L, Bp)[:, :B].sum(0) / ln2).cpu().numpy()
                mu, logv = w["mu"][:B], w["logv"][:B]
                kl = (-0.5 * torch.sum(1 + logv - mu.pow(2) - logv.exp(), dim=1) / ln2).cpu().numpy()
                records += [{"ar_bits": float(a), "kl_bits": float(k), "total_bits": float(a + k)} for a, k in zip(ar, kl)]
        finally:
            eng.training = was
        total = np.array([r["total_bits"] for r in records])
        return {"avg_total_bits": float(total.mean()), "avg_ar_bits": float(np.mean([r["ar_bits"] for r in records])),
                "avg_kl_bits": float(np.mean([r["kl_bits"] for r in records])), "min_total_bits": float(total.min()),
                "max_total_bits": float(total.max()), "records": records}


class ARK(_EngineModel):
    """decoder-only models (reference models.py:368-405): `ARK` = tok_emb + pos_emb -> GRU (h0 = 0) -> tied vocabulary
    projection; `t-ARK` = the same embeddings -> causal stack of stock Transformer encoder layers -> tied projection
    (DecoderOnlyTransformer, models.py:349-366; engine: ark_amd.txf_engine.TxfEngine)."""
    _kinds = ("ARK", "t-ARK")

    def __init__(self, config):
        super().__init__(config)
        if config["model_type"] == "ARK":
            print("Using GRU Decoder")

    def _tri(self, triples):
        return None

    def forward(self, triples_or_seq, seq_in=None):
        """forward(seq) or forward(triples, seq): triples are ignored (reference models.py:395-405)"""
        seq = triples_or_seq if seq_in is None else seq_in
        prm = [p for _, p in self._params_in_order()]
        return _ArkFn.apply(self, seq.contiguous(), *prm)

    @staticmethod
    def _filter(logits, temperature=1.0, top_p=0.0, top_k=0):
        """next-token filtering of the reference's sampler (models.py:431-449): softmax(logits / temperature) -> keep the
        top_k -> nucleus in SORTED space (the sorted prefix up to AND including the token that crosses top_p), each cut
        followed by a renormalisation.  -> (probs, sorted_probs, sorted_idx), the last two None without a nucleus."""
        if temperature and temperature != 1.0:
            logits = logits / float(temperature)
        probs = torch.softmax(logits, dim=-1)
        if top_k and top_k > 0:
            _, keep = probs.topk(top_k, dim=-1)
            probs = probs * torch.zeros_like(probs).scatter_(-1, keep, 1.0)
            probs = probs / probs.sum(dim=-1, keepdim=True).clamp_min(1e-12)
        if not (top_p and 0.0 < top_p < 1.0):
            return probs, None, None
        sp, si = probs.sort(dim=-1, descending=True)
        cut = sp.cumsum(dim=-1) > top_p
        cut[..., 1:] = cut[..., :-1].clone()   # always keep the token that crosses top_p
        cut[..., 0] = False
        sp = sp.masked_fill(cut, 0.0)
        sp = sp / sp.sum(dim=-1, keepdim=True).clamp_min(1e-12)
        return probs, sp, si

    @staticmethod
    def filtered_probs(logits, temperature=1.0, top_p=0.0, top_k=0):
        """the distribution the next token is sampled from, as dense probabilities over the vocabulary"""
        probs, sp, si = ARK._filter(logits, temperature, top_p, top_k)
        return probs if sp is None else torch.zeros_like(probs).scatter_(-1, si, sp)

    @torch.no_grad()
    def generate(self, seq_len, special_tokens, device=None, batch_size=1, beam=1, sample=False, temperature=1.0,
                 top_p=0.0, top_k=0, host_draws=False):
        """autoregressive generation with the reference's sampling rules (models.py:407-471): greedy, or
        temperature / top-k / nucleus sampling.  The causal GRU advances ONE token per step on the engine
        (Engine.decode_step, exact-fp32 kernels) instead of re-running the whole prefix.

        Draws: with a nucleus the reference draws in SORTED space, one torch.multinomial per row, and maps the drawn
        position back through the sort; otherwise one batched multinomial over the dense distribution.  By default the
        draws are made on the device (one batched multinomial per step, in sorted space under a nucleus: the same
        distribution, the device generator's stream).  `host_draws=True` copies each step's logits to the host and makes
        the reference's draws there, call for call, from torch's global CPU generator: under the same torch.manual_seed
        the sampled tokens are then the reference's CPU path's, token for token (tests/golden/ark_sampling.npz)."""
        device = device or next(self.parameters()).device
        B = batch_size
        bos, eos = special_tokens["BOS"], special_tokens["EOS"]
        eng = self.engine()
        txf = self.config["model_type"] == "t-ARK"   # (no recurrent state: the prefix is re-run, as the reference does)
        d = None if txf else eng.decode_begin(B)
        seq = torch.full((B, 1), bos, dtype=torch.long, device=device)
        for t in range(seq_len - 1):
            logits = eng.prefix_logits(seq) if txf else eng.decode_step(d, seq[:, -1].contiguous(), t)
            if not sample:
                nxt = logits.argmax(dim=-1, keepdim=True)
            else:
                probs, sp, si = self._filter(logits.cpu() if host_draws else logits, temperature, top_p, top_k)
                if sp is None:
                    nxt = torch.multinomial(probs, 1)
                elif host_draws:
                    nxt = torch.stack([si[b, torch.multinomial(sp[b], 1)] for b in range(B)])
                else:
                    nxt = si.gather(-1, torch.multinomial(sp, 1))
                nxt = nxt.to(device)
            seq = torch.cat([seq, nxt], dim=1)
            if bool((seq[:, -1] == eos).all()):
                break
        if seq.size(1) < seq_len:
            fill = torch.full((B, seq_len - seq.size(1)), eos, dtype=torch.long, device=device)
            seq = torch.cat([seq, fill], dim=1)
        return seq[:, :seq_len]

    @torch.no_grad()
    def posterior_bits(self, dataset, device, pad_id=0, sample_frac=0.1, desc="Posterior compression", batch_size=256):
        """decoder-only: KL = 0, total = AR bits (reference models.py:488-520), one pass per batch"""
        eng = self.engine()
        n = max(1, int(sample_frac * len(dataset)))
        ln2 = math.log(2)
        records = []
        was = eng.training
        eng.training = False
        try:
            for i0 in range(0, n, batch_size):
                idx = list(range(i0, min(n, i0 + batch_size)))
                if hasattr(dataset, "tensorize"):
                    _, seq = dataset.tensorize(idx)
                else:
                    seq = torch.stack([dataset[i][1] for i in idx])
                seq = seq.to(device)
                B = seq.shape[0]
                w = eng.forward(None, seq, None, with_dlogits=False)
                Bp = eng._B
                ar = (w["row_loss"][:eng.L * Bp].reshape(eng.L, Bp)[:, :B].sum(0) / ln2).cpu().numpy()
                records += [{"ar_bits": float(a), "kl_bits": 0.0, "total_bits": float(a)} for a in ar]
        finally:
            eng.training = was
        total = np.array([r["total_bits"] for r in records]) if records else np.zeros(0)
        g = lambda f: float(f(total)) if len(total) else 0.0
        return {"avg_total_bits": g(np.mean), "avg_ar_bits": g(np.mean), "avg_kl_bits": 0.0, "min_total_bits": g(np.min),
                "max_total_bits": g(np.max), "records": records}
