"""Host-side engine of the decoder-only Transformer variant t-ARK on MI355X.

Reference: DecoderOnlyTransformer (kgvae/model/models.py:349-366) -- token + learned position embeddings, a stack of stock
nn.TransformerEncoderLayer (post-norm, ReLU feed-forward of width 2048, dropout `dec_dropout` at its four sites, causal
boolean mask), tied output projection -- trained by the reference's ARK loop (kgvae/experiments/train.py:42-58: token
cross-entropy, Adam).

Same division of labour as ark_amd.engine.Engine, whose optimiser / step-scalar / gradient-buffer plumbing this class
inherits: flat fp32 parameter, gradient and Adam-moment buffers, time-major activations (row (t, b) = t*B + b, so the token
gather, vocabulary projection and cross-entropy kernels are the GRU models'), every op a hand-written gfx950 kernel
behind the C-ABI (csrc/txf.hip: residual + LayerNorm, causal attention, counter-hash dropout; csrc/gemm.hip: the dense
products, exact fp32 or 16-bit MFMA operands).  No torch arithmetic, no CPU fallback.
"""
from collections import OrderedDict

import torch

from . import _lib as L
from .engine import Engine, HP, PREC, _call, _rup

FF = 2048          # nn.TransformerEncoderLayer default dim_feedforward (the reference never passes another)
LN_EPS = 1e-5      # nn.LayerNorm default


class TxfLayout:
    """name -> (offset, shape, numel) in state-dict order; 16-byte aligned blocks"""

    def __init__(self, cfg):
        D, n, V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        ents = [("dec.tok_emb.weight", (V, D)), ("dec.pos_emb.weight", (cfg["seq_len"], D))]
        for i in range(n):
            pre = f"dec.txf.layers.{i}."
            ents += [(pre + "self_attn.in_proj_weight", (3 * D, D)), (pre + "self_attn.in_proj_bias", (3 * D,)),
                     (pre + "self_attn.out_proj.weight", (D, D)), (pre + "self_attn.out_proj.bias", (D,)),
                     (pre + "linear1.weight", (FF, D)), (pre + "linear1.bias", (FF,)),
                     (pre + "linear2.weight", (D, FF)), (pre + "linear2.bias", (D,)),
                     (pre + "norm1.weight", (D,)), (pre + "norm1.bias", (D,)), (pre + "norm2.weight", (D,)), (pre + "norm2.bias", (D,))]
        self.tied = bool(cfg.get("tie_weights", True))
        if not self.tied:
            ents.append(("dec.out.weight", (V, D)))
        ents.append(("dec.out.bias", (V,)))
        self.entries = OrderedDict()
        off = 0
        for name, shape in ents:
            off = _rup(off, 4)
            numel = 1
            for s in shape:
                numel *= s
            self.entries[name] = (off, tuple(shape), numel)
            off += numel
        self.total = _rup(off, 4)
        self.dec_grad_offset = 0


class TxfEngine(Engine):
    def __init__(self, cfg, device, precision="f32", world_size=1, rank=0):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.ArkError("ark_amd.TxfEngine needs a GPU device (no CPU fallback exists)")
        L.lib()
        if cfg["model_type"] != "t-ARK":
            raise NotImplementedError(f"Unknown model_type: {cfg['model_type']}")
        self.prec_fwd, self.prec_bwd = PREC[precision]
        self.prec = self.prec_fwd
        self.precision = precision
        self.mt = "t-ARK"
        self.D, self.n, self.V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        self.H = cfg["n_heads"]
        if self.D % self.H != 0 or (self.D // self.H) % 4 != 0 or self.D // self.H > 256:
            raise L.ArkError("t-ARK: d_model / n_heads must be a multiple of 4 and at most 256")
        self.Z = 0
        self.seq_len = cfg["seq_len"]
        self.L = self.seq_len - 1
        if self.L > 640:
            raise L.ArkError("t-ARK: sequences longer than 640 tokens are not supported by the attention kernels")
        self.p_drop = float(cfg.get("dec_dropout", 0.1))
        self.world_size, self.rank = world_size, rank
        self.layout = TxfLayout(cfg)
        n = self.layout.total
        dev = self.device
        self.P = torch.zeros(n, device=dev, dtype=torch.float32)
        self.G = torch.zeros(n, device=dev, dtype=torch.float32)
        self.M = torch.zeros(n, device=dev, dtype=torch.float32)
        self.Vv = torch.zeros(n, device=dev, dtype=torch.float32)
        self.p, self.g = OrderedDict(), OrderedDict()
        for name, (off, shape, numel) in self.layout.entries.items():
            self.p[name] = self.P[off:off + numel].view(shape)
            self.g[name] = self.G[off:off + numel].view(shape)
        if self.layout.tied:
            self.p["dec.out.weight"] = self.p["dec.tok_emb.weight"]
            self.g["dec.out.weight"] = self.g["dec.tok_emb.weight"]
        init = torch.zeros(HP["COUNT"], dtype=torch.float32)
        init[HP["ADAM_B1"]], init[HP["ADAM_B2"]], init[HP["ADAM_EPS"]] = 0.9, 0.999, 1e-8
        init[HP["GRAD_SCALE"]], init[HP["BETA"]] = 1.0, 1.0
        init[HP["LR"]] = float(cfg.get("learning_rate", 1e-3))
        self.hyper = init.to(dev)
        self._hp = {"LR": float(cfg.get("learning_rate", 1e-3)), "BETA": 1.0, "GRAD_SCALE": 1.0}
        self.adam_steps = 0
        self.ws, self.ws_key, self._ws_cache = None, None, {}
        self.training = True
        self.drop_seed = int(cfg.get("dropout_seed", 0x5A11))
        self.noise_seed = int(cfg.get("noise_seed", cfg.get("seed", 0)))
        self.fwd_gen = 0
        self.use_dma = False
        self._shadow_ok = True
        self._dp_pending, self._dp_flush_graph = None, None
        self.dp_bf16 = False
        self._graph_steps = {}
        self.ldl = _rup(self.V, 4)

    # ------------------------------------------------------------------ plumbing the base class expects
    def refresh_shadows(self):
        self._shadow_ok = True

    def _site_seed(self, layer, site):
        """dropout stream of one of a layer's four dropout sites (0 attention probabilities, 1 attention output,
        2 feed-forward activation, 3 feed-forward output) on this rank"""
        return (self.drop_seed + 7919 * (4 * layer + site) + 104729 * self.rank) & 0xFFFFFFFFFFFFFFFF

    def _workspace(self, B, Lq):
        key = (B, Lq)
        if self.ws_key == key:
            return self.ws
        if key in self._ws_cache:
            self.ws, self.ws_key = self._ws_cache[key], key
            return self.ws
        dev, D, n, H = self.device, self.D, self.n, self.H
        R = Lq * B
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        w = {"v2": False, "_R": R, "X0": f(R, D)}
        for nm, cols in (("qkv", 3 * D), ("att", D), ("sa", D), ("s1", D), ("x1", D), ("f", FF), ("g2", D), ("s2", D), ("x2", D)):
            w[nm] = [f(R, cols) for _ in range(n)]
        w["st1"] = [f(R, 2) for _ in range(n)]
        w["st2"] = [f(R, 2) for _ in range(n)]
        w["probs"] = [f(B * H * Lq * Lq) for _ in range(n)]
        w["dscore"] = f(B * H * Lq * Lq)
        w["dA"], w["dB"], w["dC"] = f(R, D), f(R, D), f(R, D)
        w["dqkv"], w["df"] = f(R, 3 * D), f(R, FF)
        w["logits"] = torch.zeros(R, self.ldl, device=dev)
        w["row_loss"] = f(R)
        w["out4"] = torch.zeros(4, device=dev)
        if len(self._ws_cache) >= 8:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = w
        self.ws, self.ws_key = w, key
        return w

    def _logits(self, w):
        return w["logits"]

    def _copy(self, dst, src):
        _call("ark_copy", L.ptr(dst), L.ptr(src), L.i64(src.numel() * src.element_size()), L.cur_stream())

    def _drop(self, x, layer, site):
        _call("ark_dropout_apply", L.ptr(x), L.i64(x.numel()), L.f32(self.p_drop), L.u64(self._site_seed(layer, site)),
              L.ptr(self.hyper), L.cur_stream())

    # ------------------------------------------------------------------ forward
    def forward(self, triples, seq, eps=None, with_loss=True, with_dlogits=True, L_run=None, ce_count=None):
        """tok + pos embedding -> n x [self-attention, add & norm, feed-forward, add & norm] -> tied logits
        (-> cross-entropy + its gradient w.r.t. the logits).  seq [B, >= L_run] int64 on the device."""
        self.prec = self.prec_fwd
        self.fwd_gen += 1
        B = seq.shape[0]
        Lq = self.L if L_run is None else L_run
        assert seq.dtype == torch.int64 and seq.is_contiguous() and seq.device == self.device and seq.shape[1] >= Lq
        w = self._workspace(B, Lq)
        D, n, V, H = self.D, self.n, self.V, self.H
        R = Lq * B
        st = L.cur_stream()
        KM = L.LAY_KMAJ
        p = self.p
        ld_seq = seq.shape[1]
        self._seq, self._B, self._Lrun = seq, B, Lq
        use_drop = self.training and self.p_drop > 0
        self._used_drop = use_drop
        _call("ark_tok_gather", L.ptr(seq), L.i64(ld_seq), L.ptr(p["dec.tok_emb.weight"]), L.ptr(p["dec.pos_emb.weight"]),
              L.ptr(w["X0"]), L.i32(B), L.i32(Lq), L.i32(D), L.ptr(self.hyper if use_drop else None), st)
        x = w["X0"]
        for l in range(n):
            pre = f"dec.txf.layers.{l}."
            self._gemm(KM, KM, L.EPI_BIAS, x, D, p[pre + "self_attn.in_proj_weight"], D, w["qkv"][l], 3 * D, R, 3 * D, D,
                       bias=p[pre + "self_attn.in_proj_bias"])
            _call("ark_attn_fwd", L.ptr(w["qkv"][l]), L.ptr(w["att"][l]), L.ptr(w["probs"][l]), L.i32(B), L.i32(Lq), L.i32(D), L.i32(H),
                  L.i32(1), L.f32(self.p_drop if use_drop else 0.0), L.u64(self._site_seed(l, 0)), L.ptr(self.hyper), st)
            self._gemm(KM, KM, L.EPI_BIAS, w["att"][l], D, p[pre + "self_attn.out_proj.weight"], D, w["sa"][l], D, R, D, D,
                       bias=p[pre + "self_attn.out_proj.bias"])
            if use_drop:
                self._drop(w["sa"][l], l, 1)
            _call("ark_layernorm_fwd", L.ptr(x), L.ptr(w["sa"][l]), L.ptr(p[pre + "norm1.weight"]), L.ptr(p[pre + "norm1.bias"]),
                  L.ptr(w["s1"][l]), L.ptr(w["x1"][l]), L.ptr(w["st1"][l]), L.i32(R), L.i32(D), L.f32(LN_EPS), st)
            self._gemm(KM, KM, L.EPI_BIAS_RELU, w["x1"][l], D, p[pre + "linear1.weight"], D, w["f"][l], FF, R, FF, D,
                       bias=p[pre + "linear1.bias"])
            if use_drop:
                self._drop(w["f"][l], l, 2)   # in place: the dropped activation is positive exactly where ReLU fired AND the mask kept
            self._gemm(KM, KM, L.EPI_BIAS, w["f"][l], FF, p[pre + "linear2.weight"], FF, w["g2"][l], D, R, D, FF,
                       bias=p[pre + "linear2.bias"])
            if use_drop:
                self._drop(w["g2"][l], l, 3)
            _call("ark_layernorm_fwd", L.ptr(w["x1"][l]), L.ptr(w["g2"][l]), L.ptr(p[pre + "norm2.weight"]), L.ptr(p[pre + "norm2.bias"]),
                  L.ptr(w["s2"][l]), L.ptr(w["x2"][l]), L.ptr(w["st2"][l]), L.i32(R), L.i32(D), L.f32(LN_EPS), st)
            x = w["x2"][l]
        self._gemm(KM, KM, L.EPI_BIAS, x, D, p["dec.out.weight"], D, w["logits"], self.ldl, R, V, D, bias=p["dec.out.bias"])
        if with_loss:
            if ce_count is None:
                _call("ark_count_targets", L.ptr(seq), L.i64(ld_seq), L.i32(B), L.i32(Lq), L.ptr(self.hyper), st)
                self._hp.pop("CE_COUNT", None)
            _call("ark_ce_fwd_bwd", L.ptr(w["logits"]), L.i64(self.ldl), L.ptr(seq), L.i64(ld_seq), L.ptr(self.hyper),
                  L.ptr(w["row_loss"]), L.ptr(w["logits"] if with_dlogits else None), L.ptr(None), L.i32(0), L.i64(0), L.i32(B),
                  L.i32(Lq), L.i32(V), st)
            _call("ark_loss_finalize", L.ptr(w["row_loss"]), L.i32(R), L.ptr(None), L.ptr(self.hyper), L.ptr(w["out4"]), st)
        return w

    # ------------------------------------------------------------------ backward
    def backward(self, ext_dhead=None):
        """backward of the last forward; the gradient w.r.t. the logits sits in ws['logits']"""
        self.prec = self.prec_bwd
        w, B, Lq = self.ws, self._B, self._Lrun
        D, n, V, H = self.D, self.n, self.V, self.H
        R = Lq * B
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p, g = self.p, self.g
        seq = self._seq
        ld_seq = seq.shape[1]
        use_drop = self._used_drop
        self._zero(self.G)
        dlog = w["logits"]
        top = w["x2"][n - 1]
        self._colsum(dlog, self.ldl, g["dec.out.bias"], R, V)
        self._gemm(MM, MM, L.EPI_NONE, dlog, self.ldl, top, D, g["dec.out.weight"], D, V, D, R, acc=1)
        dx, other, third = w["dA"], w["dB"], w["dC"]
        self._gemm(KM, MM, L.EPI_NONE, dlog, self.ldl, p["dec.out.weight"], D, dx, D, R, D, V)
        for l in range(n - 1, -1, -1):
            pre = f"dec.txf.layers.{l}."
            xin = w["x2"][l - 1] if l > 0 else w["X0"]
            # x2 = LN2(x1 + drop(ff)): ds2 feeds both the feed-forward branch and the residual
            _call("ark_layernorm_bwd", L.ptr(dx), L.ptr(w["s2"][l]), L.ptr(w["st2"][l]), L.ptr(p[pre + "norm2.weight"]), L.ptr(other),
                  L.ptr(g[pre + "norm2.weight"]), L.ptr(g[pre + "norm2.bias"]), L.i32(R), L.i32(D), st)
            ds2 = other
            dg2 = ds2
            if use_drop:
                dg2 = third
                self._copy(dg2, ds2)
                self._drop(dg2, l, 3)
            self._colsum(dg2, D, g[pre + "linear2.bias"], R, D)
            self._gemm(MM, MM, L.EPI_NONE, dg2, D, w["f"][l], FF, g[pre + "linear2.weight"], FF, D, FF, R, acc=1)
            # df = (dg2 W2) masked by ReLU (and by the feed-forward dropout: f is positive only where both let it through)
            self._gemm(KM, MM, L.EPI_MUL_RELU, dg2, D, p[pre + "linear2.weight"], FF, w["df"], FF, R, FF, D, aux=w["f"][l])
            if use_drop:
                self._drop(w["df"], l, 2)
            self._colsum(w["df"], FF, g[pre + "linear1.bias"], R, FF)
            self._gemm(MM, MM, L.EPI_NONE, w["df"], FF, w["x1"][l], D, g[pre + "linear1.weight"], D, FF, D, R, acc=1)
            self._gemm(KM, MM, L.EPI_NONE, w["df"], FF, p[pre + "linear1.weight"], D, ds2, D, R, D, FF, acc=1)   # dx1 = ds2 + df W1
            # x1 = LN1(x_in + drop(sa))
            _call("ark_layernorm_bwd", L.ptr(ds2), L.ptr(w["s1"][l]), L.ptr(w["st1"][l]), L.ptr(p[pre + "norm1.weight"]), L.ptr(dx),
                  L.ptr(g[pre + "norm1.weight"]), L.ptr(g[pre + "norm1.bias"]), L.i32(R), L.i32(D), st)
            ds1 = dx
            dsa = ds1
            if use_drop:
                dsa = third
                self._copy(dsa, ds1)
                self._drop(dsa, l, 1)
            self._colsum(dsa, D, g[pre + "self_attn.out_proj.bias"], R, D)
            self._gemm(MM, MM, L.EPI_NONE, dsa, D, w["att"][l], D, g[pre + "self_attn.out_proj.weight"], D, D, D, R, acc=1)
            self._gemm(KM, MM, L.EPI_NONE, dsa, D, p[pre + "self_attn.out_proj.weight"], D, ds2, D, R, D, D)   # d(att) -> ds2 buffer
            _call("ark_attn_bwd", L.ptr(w["qkv"][l]), L.ptr(w["att"][l]), L.ptr(w["probs"][l]), L.ptr(ds2), L.ptr(w["dscore"]),
                  L.ptr(w["dqkv"]), L.i32(B), L.i32(Lq), L.i32(D), L.i32(H), L.i32(1), L.f32(self.p_drop if use_drop else 0.0),
                  L.u64(self._site_seed(l, 0)), L.ptr(self.hyper), st)
            self._colsum(w["dqkv"], 3 * D, g[pre + "self_attn.in_proj_bias"], R, 3 * D)
            self._gemm(MM, MM, L.EPI_NONE, w["dqkv"], 3 * D, xin, D, g[pre + "self_attn.in_proj_weight"], D, 3 * D, D, R, acc=1)
            self._gemm(KM, MM, L.EPI_NONE, w["dqkv"], 3 * D, p[pre + "self_attn.in_proj_weight"], D, ds1, D, R, D, 3 * D, acc=1)
            dx, other = ds1, ds2
        _call("ark_tok_scatter", L.ptr(seq), L.i64(ld_seq), L.ptr(dx), L.ptr(g["dec.tok_emb.weight"]), L.i32(B), L.i32(Lq), L.i32(D),
              L.i32(V), st)
        self._colsum(dx, D, g["dec.pos_emb.weight"], B, D, n_batch=Lq, bs_in=B * D, bs_out=D)

    # ------------------------------------------------------------------ whole step
    def train_step(self, triples, seq, eps=None, grad_sync=None, ce_count=None, dp=False):
        """forward + cross-entropy + backward (+ gradient all-reduce) + Adam; returns out4 on the device"""
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        self.forward(None, seq, None, ce_count=ce_count)
        self.backward()
        if dp:   # one bucket: the whole flat gradient buffer, summed over the ranks
            import torch.distributed as dist
            dist.all_reduce(self.G, op=dist.ReduceOp.SUM)
        if grad_sync is not None:
            grad_sync(self.G)
        self.adam()
        return self.ws["out4"]

    def graphed_train_step(self, triples, seq, ce_count=None, dp=False):
        """(the Transformer variant's step is launched eagerly: no captured graph yet)"""
        return self.train_step(None, seq, ce_count=ce_count, dp=dp)

    def capture_train_step(self, *a, **k):
        raise L.ArkError("t-ARK: hipGraph capture of the train step is not implemented; use train_step()")

    def eval_loss(self, triples, seq, eps=None):
        was = self.training
        self.training = False
        try:
            w = self.forward(None, seq, None, with_dlogits=False)
        finally:
            self.training = was
        return w["out4"]

    @torch.no_grad()
    def prefix_logits(self, prefix):
        """logits [B, V] of the position after `prefix` [B, t] (the reference re-runs the whole prefix per generated
        token, models.py:430; so does this: the Transformer has no recurrent state to carry)"""
        was = self.training
        self.training = False
        try:
            t = prefix.shape[1]
            B = prefix.shape[0]
            w = self.forward(None, prefix.contiguous(), None, with_loss=False, L_run=t)
        finally:
            self.training = was
        return w["logits"][(t - 1) * B:t * B, :self.V]

    def decode_begin(self, *a, **k):
        raise L.ArkError("t-ARK has no incremental decoder state: use prefix_logits()")
