"""Host-side engine of the Transformer variants t-ARK and t-SAIL on MI355X.

Reference (kgvae/model/models.py):
  t-ARK   DecoderOnlyTransformer :349-366 -- token + learned position embeddings, a causal stack of stock
          nn.TransformerEncoderLayer (post-norm, ReLU feed-forward of width 2048, dropout `dec_dropout`), tied projection;
          trained by the ARK loop (kgvae/experiments/train.py:42-58).
  t-SAIL  AutoRegEncoder :66-95 -- [E[h] | R[r] | E[t]] per triple, Transformer encoder over the triples (padded triples
          masked as keys), masked mean, mu / logv heads WITHOUT the logv clamp, reparameterisation -- and AutoRegDecoder
          :98-114 -- token + position embeddings, causal stock nn.TransformerDecoderLayer stack whose memory is z_proj(z)
          repeated L times, UNTIED projection; layer dropout hard-coded 0.1; trained by the VAE loop
          (kgvae/experiments/ablation_study.py:59-73: ce + b * kl).

Same division of labour as ark_amd.engine.Engine, whose optimiser / step-scalar / gradient-buffer plumbing this class
inherits: flat fp32 parameter, gradient and Adam-moment buffers, time-major activations (row (t, b) = t*B + b, so the token
gather, vocabulary projection, cross-entropy and latent kernels are the GRU models'), every op a hand-written gfx950
kernel behind the C-ABI (csrc/txf.hip: residual + LayerNorm, masked / causal attention, the uniform cross-attention of the
repeated memory, counter-hash dropout; csrc/gemm.hip: the dense products, exact fp32 or 16-bit MFMA operands).  No torch
arithmetic, no CPU fallback.
"""
from collections import OrderedDict

import torch

from . import _lib as L
from .engine import CAPTURE_MODE, Engine, HP, PREC, _call, _rup
from .engine import _calls as _launches   # library calls made so far (TxfEngine._cast: "nothing ran since the previous product")

FF = 2048          # nn.Transformer*Layer default dim_feedforward (the reference never passes another)
LN_EPS = 1e-5      # nn.LayerNorm default
TXF_DROPOUT = 0.1  # nn.Transformer*Layer default dropout: what t-SAIL's layers are built with (models.py:73,104)


def _attn_entries(pre, W):
    return [(pre + "in_proj_weight", (3 * W, W)), (pre + "in_proj_bias", (3 * W,)), (pre + "out_proj.weight", (W, W)),
            (pre + "out_proj.bias", (W,))]


def _ff_entries(pre, W):
    return [(pre + "linear1.weight", (FF, W)), (pre + "linear1.bias", (FF,)), (pre + "linear2.weight", (W, FF)),
            (pre + "linear2.bias", (W,))]


def _norm_entries(pre, W, names):
    out = []
    for nm in names:
        out += [(pre + nm + ".weight", (W,)), (pre + nm + ".bias", (W,))]
    return out


class TxfLayout:
    """name -> (offset, shape, numel) in the reference's state-dict order; 16-byte aligned blocks"""

    def __init__(self, cfg):
        mt = cfg["model_type"]
        D, n, V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        ents = []
        if mt == "t-SAIL":
            W, Z = 3 * D, cfg["d_latent"]
            ents += [("enc.e_emb.weight", (cfg["n_entities"], D)), ("enc.r_emb.weight", (cfg["n_relations"], D))]
            for i in range(cfg.get("n_layers", 2)):
                pre = f"enc.txf.layers.{i}."
                ents += _attn_entries(pre + "self_attn.", W) + _ff_entries(pre, W) + _norm_entries(pre, W, ("norm1", "norm2"))
            # mu and logv are adjacent (weights, then biases): the two heads run as ONE [2Z, 3D] product
            ents += [("enc.mu.weight", (Z, W)), ("enc.logv.weight", (Z, W)), ("enc.mu.bias", (Z,)), ("enc.logv.bias", (Z,))]
            ents += [("dec.tok_emb.weight", (V, D)), ("dec.pos_emb.weight", (cfg["seq_len"], D)), ("dec.z_proj.weight", (D, Z)),
                     ("dec.z_proj.bias", (D,))]
            for i in range(n):
                pre = f"dec.txf.layers.{i}."
                ents += _attn_entries(pre + "self_attn.", D) + _attn_entries(pre + "multihead_attn.", D) + _ff_entries(pre, D) + \
                    _norm_entries(pre, D, ("norm1", "norm2", "norm3"))
            self.tied = False
            ents += [("dec.out.weight", (V, D)), ("dec.out.bias", (V,))]
        elif mt == "t-ARK":
            ents += [("dec.tok_emb.weight", (V, D)), ("dec.pos_emb.weight", (cfg["seq_len"], D))]
            for i in range(n):
                pre = f"dec.txf.layers.{i}."
                ents += _attn_entries(pre + "self_attn.", D) + _ff_entries(pre, D) + _norm_entries(pre, D, ("norm1", "norm2"))
            self.tied = bool(cfg.get("tie_weights", True))
            if not self.tied:
                ents.append(("dec.out.weight", (V, D)))
            ents.append(("dec.out.bias", (V,)))
        else:
            raise NotImplementedError(f"Unknown model_type: {mt}")
        self.entries = OrderedDict()
        off = 0
        prev_packed = False
        for name, shape in ents:
            packed = name in ("enc.logv.weight", "enc.logv.bias")   # directly behind enc.mu.*: no alignment gap
            if not packed:
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
        if cfg["model_type"] not in ("t-ARK", "t-SAIL"):
            raise NotImplementedError(f"Unknown model_type: {cfg['model_type']}")
        self.prec_fwd, self.prec_bwd = PREC[precision]
        self.prec = self.prec_fwd
        self.precision = precision
        self.mt = cfg["model_type"]
        self.vae = self.mt == "t-SAIL"
        self.D, self.n, self.V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        self.n_enc = cfg.get("n_layers", 2) if self.vae else 0
        self.H = cfg["n_heads"]
        widths = [self.D] + ([3 * self.D] if self.vae else [])
        for W in widths:
            if W % self.H != 0 or (W // self.H) % 4 != 0 or W // self.H > 768 or W > 3072:
                raise L.ArkError(f"{self.mt}: layer width {W} / n_heads {self.H}: head widths must be multiples of 4 up to 768, "
                                 "layer widths at most 3072")
        self.Z = cfg.get("d_latent", 0) if self.vae else 0
        self.seq_len = cfg["seq_len"]
        self.L = self.seq_len - 1
        if self.L > 640:
            raise L.ArkError(f"{self.mt}: sequences longer than 640 tokens are not supported by the attention kernels")
        self.pad_eid, self.pad_rid = cfg.get("pad_eid"), cfg.get("pad_rid")
        # t-ARK's layers take `dec_dropout` (models.py:389); t-SAIL's are built with the stock default (models.py:73,104)
        self.p_drop = float(cfg.get("ark_txf_dropout", TXF_DROPOUT)) if self.vae else float(cfg.get("dec_dropout", 0.1))
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
        self.dp_bf16 = bool(cfg.get("ark_dp_bf16", True))   # bf16 transport of the gradient buckets (fp32 master gradients)
        self._graph_steps = {}
        # 16-bit precisions: the dense products run on the LDS-DMA engines (ark_gemm16 / ark_wgrad16) from 16-bit copies of
        # their operands cast per call -- the register-staged ark_gemm converts the same fp32 operands on the fly, to the same
        # 16-bit values, several times slower
        self.fast_gemm = bool(cfg.get("ark_txf_fast_gemm", True)) and self.prec_fwd != L.PREC_F32
        # self-attention longer than 16 positions on the matrix cores, flash style (no [B, H, L, L] arrays): csrc/attn_mfma.hip
        self.flash = bool(cfg.get("ark_txf_flash", True))
        # large vocabularies: output projection FUSED with the cross-entropy (csrc/vocab_ce.hip, as in the GRU engine): the
        # [B*L, V] logits and their gradient never exist in the train / eval-loss steps (2.5 GB at wd-articles)
        self.fused_ce = bool(cfg.get("ark_fused_ce", self.prec_fwd != L.PREC_F32 and self.V >= 2048 and self.D in (64, 128, 256, 512)))
        self._fused_step = False
        self._scratch = {}
        self._last_cast = None   # (only consecutive _gemm calls may share an operand copy: anything else resets it)
        self._graphs = {}
        self._pinned = set()   # workspace keys a captured graph points at: never evicted
        self.ldl = _rup(self.V, 4)

    # ------------------------------------------------------------------ plumbing the base class expects
    def refresh_shadows(self):
        self._shadow_ok = True

    def _default_norms(self, B):
        if self.vae:
            self.set_hyper(kl_norm=1.0 / (B * self.world_size * self.Z))

    def _seed(self, stack, layer, site):
        """dropout stream of one dropout site of a layer on this rank.  stack 0 = decoder, 1 = encoder; sites: 0 self-attention
        probabilities, 1 self-attention output, 2 feed-forward activation, 3 feed-forward output, 4 cross-attention
        probabilities, 5 cross-attention output"""
        return (self.drop_seed + 7919 * (8 * (2 * layer + stack) + site) + 104729 * self.rank) & 0xFFFFFFFFFFFFFFFF

    def _flash_ok(self, W, Ls):
        """self-attention of width W over Ls positions on the matrix cores (csrc/attn_mfma.hip)?  16-bit products only (the
        exact-fp32 mode keeps the vector-unit kernels); short sequences keep the one-wave-per-head kernels"""
        dh = W // self.H
        return bool(self.flash) and self.prec_fwd != L.PREC_F32 and Ls > 16 and dh % 32 == 0 and dh <= 384

    def _layer_bufs(self, R, W, B, Ls, cross):
        dev = self.device
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        d = {k: f(R, c) for k, c in (("qkv", 3 * W), ("att", W), ("sa", W), ("s1", W), ("x1", W), ("f", FF), ("g2", W), ("s2", W),
                                    ("x2", W))}
        d["st1"], d["st2"] = f(R, 2), f(R, 2)
        if self._flash_ok(W, Ls):   # matrix-core attention: one log-sum-exp per (batch, head, query) instead of [B, H, L, L]
            d["lse"] = f(B * self.H * ((Ls + 63) // 64 * 64))
        else:
            d["probs"] = f(B * self.H * Ls * Ls)
        if cross:
            d.update({"ctx": f(R, W), "cs": f(R, self.H), "ca": f(R, W), "s3": f(R, W), "x3": f(R, W), "st3": f(R, 2),
                      "vmem": f(B, W)})
        return d

    def _workspace(self, B, Lq, T=0, decode=False):
        """buffers of a (batch, decoder length, triples) shape.  decode=True: the ONE workspace generation uses for every
        prefix length (allocated at the full length; buffers are time-major, so a prefix of t steps is a leading slice and
        the kernels take the run length as an argument) -- kept apart from the training workspaces, which a stream of
        per-token shapes used to evict"""
        if decode:
            cache = self.__dict__.setdefault("_dec_ws_cache", {})
            cap, w = cache.get((B, T), (0, None))
            if cap < Lq:
                cap = max(self.L, Lq)
                keep = (self.ws, self.ws_key)
                had = (B, cap, T) in self._ws_cache
                self.ws_key = None
                w = self._workspace(B, cap, T)
                if had:   # a training workspace of that very shape exists: decode gets buffers of its own
                    self._ws_cache.pop((B, cap, T))
                    self.ws_key = None
                    w2 = self._workspace(B, cap, T)
                    self._ws_cache[(B, cap, T)] = w
                    w = w2
                else:
                    self._ws_cache.pop((B, cap, T), None)   # (built by the ordinary path below, owned by the decode cache)
                self.ws, self.ws_key = keep
                cache.pop((B, T), None)
                if len(cache) >= 2:
                    cache.pop(next(iter(cache)))
                cache[(B, T)] = (cap, w)
            return w
        key = (B, Lq, T)
        if self.ws_key == key:
            return self.ws
        if key in self._ws_cache:
            self.ws, self.ws_key = self._ws_cache[key], key
            return self.ws
        dev, D, n, H = self.device, self.D, self.n, self.H
        R = Lq * B
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        w = {"v2": False, "_R": R, "X0": f(R, D)}
        w["dec"] = [self._layer_bufs(R, D, B, Lq, self.vae) for _ in range(n)]
        w["dscore"] = None if self._flash_ok(D, Lq) else f(B * H * Lq * Lq)
        w["delta"] = f(B * H * ((max(Lq, T) + 63) // 64 * 64)) if (self._flash_ok(D, Lq) or (self.vae and self._flash_ok(3 * D, T))) else None
        w["dA"], w["dB"], w["dC"] = f(R, D), f(R, D), f(R, D)
        w["dqkv"], w["df"] = f(R, 3 * D), f(R, FF)
        if self.vae:
            W3, Z, Re = 3 * D, self.Z, T * B
            w["E0"] = f(Re, W3)
            w["kmask"] = torch.ones(B, T, device=dev, dtype=torch.uint8)
            w["enc"] = [self._layer_bufs(Re, W3, B, T, False) for _ in range(self.n_enc)]
            w["e_dscore"] = None if self._flash_ok(W3, T) else f(B * H * T * T)
            w["eA"], w["eB"], w["eC"] = f(Re, W3), f(Re, W3), f(Re, W3)
            w["e_dqkv"], w["e_df"] = f(Re, 3 * W3), f(Re, FF)
            w["g"], w["inv_cnt"], w["dg"] = f(B, W3), f(B), f(B, W3)
            w["head"], w["dhead"] = f(B, 2 * Z), f(B, 2 * Z)
            z0 = lambda *s: torch.zeros(*s, device=dev)
            w["mu"], w["logv"], w["z"], w["dz"] = z0(B, Z), z0(B, Z), z0(B, Z), z0(B, Z)
            w["kl"] = torch.zeros(1, device=dev)
            w["eps0"] = torch.zeros(B, Z, device=dev)
            w["mem"], w["dmem"], w["dv"] = f(B, D), f(B, D), f(B, D)
        if not (self.fused_ce and R % 16 == 0):   # (the fused path allocates them on first use: model(seq), generation)
            w["logits"] = torch.zeros(R, self.ldl, device=dev)
        w["row_loss"] = f(R)
        w["out4"] = torch.zeros(4, device=dev)
        if len(self._ws_cache) >= 8:
            for k in [k for k in self._ws_cache if k not in self._pinned][:1]:
                self._ws_cache.pop(k)
        self._ws_cache[key] = w
        self.ws, self.ws_key = w, key
        return w

    def _logits(self, w):
        if "logits" not in w:
            w["logits"] = torch.zeros(w["_R"], self.ldl, device=self.device)
        return w["logits"]

    def _fused_ce_bufs(self, w, R):
        """operands and outputs of the fused vocabulary CE: 16-bit copies of the top activations and of the output weight
        (forward type), log-sum-exp, tile-native dY, the vocabulary-split scratch"""
        if "ce_y16" not in w:
            D, V, dev = self.D, self.V, self.device
            w["ce_y16"] = torch.empty(R * D, device=dev, dtype=torch.int16)
            w["ce_lse"] = torch.empty(R, device=dev)
            w["ce_dY"] = torch.empty(R * D, device=dev)
            ns = int(L.lib().ark_vocab_ce_fwd_splits(L.i32(R), L.i32(V), L.i32(D), L.i32(0)))
            w["ce_ws"] = torch.empty(ns * (R * D + 4 * R), device=dev) if ns > 1 else None
        if getattr(self, "_ce_w16", None) is None:
            self._ce_w16 = torch.empty(self.V * self.D, device=self.device, dtype=torch.int16)
        return w["ce_y16"], self._ce_w16, w["ce_lse"], w["ce_dY"], w["ce_ws"]

    # ------------------------------------------------------------------ dense products, 16-bit fast path
    def _buf(self, slot, nbytes):
        """scratch of at least nbytes (uint8), one per slot; grown on demand by the eager warm-up step, never inside a capture"""
        t = self._scratch.get(slot)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * 1.25) + 256, device=self.device, dtype=torch.uint8)
            self._scratch[slot] = t
        return t

    def _cast(self, slot, x, n, prec):
        out = self._buf(slot, 2 * n)
        # the backward pass multiplies the same gradient twice in a row (its weight gradient, then its input gradient):
        # the second product finds the 16-bit copy of the first still in the slot
        key = (x.data_ptr(), n, prec, out.data_ptr())
        if slot == "a" and self._last_cast is not None and self._last_cast[0] == key and self._last_cast[1] == _launches[0]:
            return out   # (valid only while NO other library call has run since that product)
        if slot == "a" and getattr(self, "_prepared", None) == x.data_ptr():
            # _prep() left the dropout-applied gradient ONLY in this 16-bit slot (the fp32 tensor is un-dropped): re-casting the
            # raw tensor here would silently lose the mask from the backward pass
            raise L.ArkError("the prepared (dropout-applied) 16-bit operand is no longer in its slot: a library call ran between "
                             "_prep() and the products that consume it, or the slot was reallocated")
        _call("ark_cast16", L.i32(prec), L.ptr(x), L.ptr(out), L.i64(n), L.cur_stream())
        if slot == "a":
            self._last_cast = [key, None]
        return out

    def _cast_t(self, slot, x, R, C, prec):
        """16-bit copy of x[R, C]^T: [C, R] dense"""
        import ctypes
        out = self._buf(slot, 2 * R * C)
        one = lambda v, t: (t * 1)(v)
        _call("ark_weight_shadows", L.i32(1), one(x.data_ptr(), ctypes.c_void_p), one(0, ctypes.c_void_p),
              one(out.data_ptr(), ctypes.c_void_p), one(R, ctypes.c_int), one(C, ctypes.c_int), one(prec, ctypes.c_int),
              one(prec, ctypes.c_int), one(R, ctypes.c_int), L.cur_stream())
        return out

    def _gemm(self, a_lay, b_lay, epi, A, lda, Bm, ldb, C, ldc, M, N, K, C2=None, bias=None, aux=None, acc=0):
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        pr = self.prec
        if getattr(self, "_prepared", None) is not None and self._prepared != A.data_ptr():
            self._prepared = None
        if not self.fast_gemm or pr == L.PREC_F32 or C2 is not None:
            if getattr(self, "_prepared", None) is not None:
                raise L.ArkError("a prepared (dropout-applied, 16-bit) operand reached the register-staged product")
            return super()._gemm(a_lay, b_lay, epi, A, lda, Bm, ldb, C, ldc, M, N, K, C2=C2, bias=bias, aux=aux, acc=acc)
        st = L.cur_stream()
        if a_lay == KM and K % 64 == 0 and lda == K and ((b_lay == KM and ldb == K) or (b_lay == MM and ldb == N)):
            # C[M,N] (+)= A[M,K] op(B): B is [N,K] (KM) or [K,N] (MM: its transposed 16-bit copy is made)
            A16 = self._cast("a", A, M * K, pr)
            B16 = self._cast("b", Bm, N * K, pr) if b_lay == KM else self._cast_t("b", Bm, K, N, pr)
            if acc and epi == L.EPI_NONE and ldc == N:   # C += A op(B) in the product's own epilogue
                _call("ark_gemm16", L.i32(pr), L.i32(L.EPI_ADD), L.ptr(A16), L.i64(K), L.ptr(B16), L.i64(K), L.ptr(C), L.i64(ldc),
                      L.ptr(None), L.ptr(None), L.i32(M), L.i32(N), L.i32(K), L.i32(0), st)
            elif acc:
                tmp = self._buf("c", 4 * M * N)
                _call("ark_gemm16", L.i32(pr), L.i32(epi), L.ptr(A16), L.i64(K), L.ptr(B16), L.i64(K), L.ptr(tmp), L.i64(N),
                      L.ptr(bias), L.ptr(aux), L.i32(M), L.i32(N), L.i32(K), L.i32(0), st)
                if ldc != N:
                    raise L.ArkError("accumulating fast product needs a dense output")
                _call("ark_axpy", L.ptr(C), L.ptr(tmp), L.i64(M * N), L.f32(1.0), st)
            else:
                _call("ark_gemm16", L.i32(pr), L.i32(epi), L.ptr(A16), L.i64(K), L.ptr(B16), L.i64(K), L.ptr(C), L.i64(ldc),
                      L.ptr(bias), L.ptr(aux), L.i32(M), L.i32(N), L.i32(K), L.i32(0), st)
            if self._last_cast is not None:
                self._last_cast[1] = _launches[0]
            return
        if a_lay == MM and b_lay == MM and acc and epi == L.EPI_NONE and M % 64 == 0 and N % 64 == 0 and lda == M and ldb == N:
            # weight gradient C[M,N] += A[K,M]^T B[K,N] (K = rows of the batch).  The LDS-DMA engine walks K in stages of 64:
            # a row count that is not a multiple (wd-articles: 16 x 637 = 10 192) gets zero rows behind its 16-bit copies
            Kp = (K + 63) // 64 * 64
            self._buf("a", 2 * Kp * M)
            self._buf("b", 2 * Kp * N)
            A16 = self._cast("a", A, K * M, pr)
            B16 = self._cast("b", Bm, K * N, pr)
            if Kp != K:
                self._zero(A16[2 * K * M:2 * Kp * M])
                self._zero(B16[2 * K * N:2 * Kp * N])
            _call("ark_wgrad16", L.i32(pr), L.ptr(A16), L.i64(M), L.ptr(B16), L.i64(N), L.ptr(C), L.i64(ldc), L.i32(M), L.i32(N),
                  L.i32(Kp), None, st)
            if self._last_cast is not None:
                self._last_cast[1] = _launches[0]
            return
        if getattr(self, "_prepared", None) is not None:
            raise L.ArkError("a prepared (dropout-applied, 16-bit) operand reached the register-staged product")
        return super()._gemm(a_lay, b_lay, epi, A, lda, Bm, ldb, C, ldc, M, N, K, C2=C2, bias=bias, aux=aux, acc=acc)

    def _fast_ok(self, rows, *dims):
        """the products around a [rows, N] gradient take the 16-bit LDS-DMA engines (whole 64-wide stages in every feature
        dimension; any row count: as the M dimension rows are clamped, as the K dimension of a weight gradient they are
        zero-padded in the 16-bit copies, see _gemm)"""
        return bool(self.fast_gemm and self.prec != L.PREC_F32 and all(d % 64 == 0 for d in dims))

    def _prep(self, x, R, N, bias_grad, seed=None, x_out=None):
        """ONE pass (ark_prep16) for what used to be copy + dropout + bias column sum + 16-bit cast: the 16-bit copy of
        dropout(x) lands in the product-operand slot, so the products that follow take `x` as their A operand and find it
        there; bias_grad (nullable) += column sums; x_out (nullable, may be x): the fp32 dropped values"""
        out = self._buf("a", 2 * ((R + 63) // 64 * 64) * N)   # (room for the zero rows a weight gradient pads K with)
        _call("ark_prep16", L.i32(self.prec), L.ptr(x), L.ptr(x_out), L.ptr(out), L.ptr(bias_grad), L.i32(R), L.i32(N),
              L.f32(self.p_drop if seed is not None else 0.0), L.u64(seed or 0), L.ptr(self.hyper), L.cur_stream())
        self._last_cast = [(x.data_ptr(), R * N, self.prec, out.data_ptr()), _launches[0]]
        # (without x_out = x and with dropout on, the fp32 tensor itself is NOT what the products must read: see _gemm / _cast)
        self._prepared = x.data_ptr() if not (x_out is x or seed is None) else None

    def _copy(self, dst, src):
        _call("ark_copy", L.ptr(dst), L.ptr(src), L.i64(src.numel() * src.element_size()), L.cur_stream())

    def _drop(self, x, seed):
        _call("ark_dropout_apply", L.ptr(x), L.i64(x.numel()), L.f32(self.p_drop), L.u64(seed), L.ptr(self.hyper), L.cur_stream())

    def _ln_fwd(self, x, sub, pre, s, y, st, R, W, drop_seed=None):
        """y = LN(x + sub); drop_seed: `sub` has NOT been through its dropout yet -- the kernel applies that mask on the way in"""
        _call("ark_layernorm_fwd_drop", L.ptr(x), L.ptr(sub), L.ptr(self.p[pre + ".weight"]), L.ptr(self.p[pre + ".bias"]), L.ptr(s),
              L.ptr(y), L.ptr(st), L.i32(R), L.i32(W), L.f32(LN_EPS), L.f32(self.p_drop if drop_seed is not None else 0.0),
              L.u64(drop_seed or 0), L.ptr(self.hyper), L.cur_stream())

    def _ln_bwd(self, dy, s, st, pre, ds, R, W):
        _call("ark_layernorm_bwd", L.ptr(dy), L.ptr(s), L.ptr(st), L.ptr(self.p[pre + ".weight"]), L.ptr(ds), L.ptr(self.g[pre + ".weight"]),
              L.ptr(self.g[pre + ".bias"]), L.i32(R), L.i32(W), L.cur_stream())

    # ------------------------------------------------------------------ sublayers, forward
    def _self_attn_fwd(self, d, x, pre, R, W, B, Ls, causal, kmask, drop, stack, l):
        """x -> LN1(x + drop(SA(x))): d['x1']"""
        KM, p = L.LAY_KMAJ, self.p
        a = pre + "self_attn."
        self._gemm(KM, KM, L.EPI_BIAS, x, W, p[a + "in_proj_weight"], W, d["qkv"], 3 * W, R, 3 * W, W, bias=p[a + "in_proj_bias"])
        if "lse" in d:
            _call("ark_attn_flash_fwd", L.i32(self.prec_fwd), L.ptr(d["qkv"]), L.ptr(d["att"]), L.ptr(d["lse"]), L.ptr(kmask), L.i32(B),
                  L.i32(Ls), L.i32(W), L.i32(self.H), L.i32(1 if causal else 0), L.f32(self.p_drop if drop else 0.0),
                  L.u64(self._seed(stack, l, 0)), L.ptr(self.hyper), L.cur_stream())
        else:
            _call("ark_attn_fwd", L.ptr(d["qkv"]), L.ptr(d["att"]), L.ptr(d["probs"]), L.ptr(kmask), L.i32(B), L.i32(Ls), L.i32(W),
                  L.i32(self.H), L.i32(1 if causal else 0), L.f32(self.p_drop if drop else 0.0), L.u64(self._seed(stack, l, 0)),
                  L.ptr(self.hyper), L.cur_stream())
        self._gemm(KM, KM, L.EPI_BIAS, d["att"], W, p[a + "out_proj.weight"], W, d["sa"], W, R, W, W, bias=p[a + "out_proj.bias"])
        self._ln_fwd(x, d["sa"], pre + "norm1", d["s1"], d["x1"], d["st1"], R, W, self._seed(stack, l, 1) if drop else None)
        return d["x1"]

    def _ff_fwd(self, d, x, pre, norm, R, W, drop, stack, l, s_key, x_key, st_key):
        """x -> LN(x + drop(W2 drop(relu(W1 x))))"""
        KM, p = L.LAY_KMAJ, self.p
        self._gemm(KM, KM, L.EPI_BIAS_RELU, x, W, p[pre + "linear1.weight"], W, d["f"], FF, R, FF, W, bias=p[pre + "linear1.bias"])
        if drop and self._fast_ok(R, FF, W):
            # in place (positive exactly where ReLU fired AND the mask kept) AND as the 16-bit operand of linear2: one pass
            self._prep(d["f"], R, FF, None, self._seed(stack, l, 2), x_out=d["f"])
        elif drop:
            self._drop(d["f"], self._seed(stack, l, 2))   # in place: positive exactly where ReLU fired AND the mask kept
        self._gemm(KM, KM, L.EPI_BIAS, d["f"], FF, p[pre + "linear2.weight"], FF, d["g2"], W, R, W, FF, bias=p[pre + "linear2.bias"])
        self._ln_fwd(x, d["g2"], pre + norm, d[s_key], d[x_key], d[st_key], R, W, self._seed(stack, l, 3) if drop else None)
        return d[x_key]

    def _cross_attn_fwd(self, d, x, mem, pre, R, W, B, Ls, drop, l):
        """x -> LN2(x + drop(MHA(x, mem repeated Ls times))): the softmax over Ls identical keys is uniform, so the context is
        the value row of the memory (times the attention-dropout factor), whatever the queries are"""
        KM, p = L.LAY_KMAJ, self.p
        a = pre + "multihead_attn."
        self._gemm(KM, KM, L.EPI_BIAS, mem, W, p[a + "in_proj_weight"][2 * W:], W, d["vmem"], W, B, W, W, bias=p[a + "in_proj_bias"][2 * W:])
        _call("ark_xattn_bcast_fwd", L.ptr(d["vmem"]), L.ptr(d["ctx"]), L.ptr(d["cs"]), L.i32(B), L.i32(Ls), L.i32(W), L.i32(self.H),
              L.f32(self.p_drop if drop else 0.0), L.u64(self._seed(0, l, 4)), L.ptr(self.hyper), L.cur_stream())
        self._gemm(KM, KM, L.EPI_BIAS, d["ctx"], W, p[a + "out_proj.weight"], W, d["ca"], W, R, W, W, bias=p[a + "out_proj.bias"])
        self._ln_fwd(x, d["ca"], pre + "norm2", d["s2"], d["x2"], d["st2"], R, W, self._seed(0, l, 5) if drop else None)
        return d["x2"]

    # ------------------------------------------------------------------ forward
    def forward(self, triples, seq, eps=None, with_loss=True, with_dlogits=True, L_run=None, ce_count=None, z_given=None,
                decode_ws=False):
        """t-ARK: forward(None, seq).  t-SAIL: forward(triples, seq, eps) (eps None: drawn on the device), or
        forward(None, seq, z_given=z) to run the decoder alone on given latents (generation).
        seq [B, >= L_run] int64 on the device; results stay in the workspace (out4 = loss, ce, kl, token-loss sum)."""
        self.prec = self.prec_fwd
        self.fwd_gen += 1
        B = seq.shape[0]
        Lq = self.L if L_run is None else L_run
        assert seq.dtype == torch.int64 and seq.is_contiguous() and seq.device == self.device and seq.shape[1] >= Lq
        enc_on = self.vae and z_given is None
        if enc_on and triples is None:
            raise L.ArkError("t-SAIL forward needs `triples` (encoder input) or `z_given` (decoder alone)")
        T = triples.shape[1] if enc_on else 0
        w = self._workspace(B, Lq, T, decode=decode_ws)
        D, n, V, H = self.D, self.n, self.V, self.H
        R = Lq * B
        st = L.cur_stream()
        KM = L.LAY_KMAJ
        p = self.p
        ld_seq = seq.shape[1]
        self._seq, self._triples, self._B, self._Lrun, self._T = seq, triples, B, Lq, T
        use_drop = self.training and self.p_drop > 0
        self._used_drop = use_drop
        self._enc_on = enc_on
        _call("ark_tok_gather", L.ptr(seq), L.i64(ld_seq), L.ptr(p["dec.tok_emb.weight"]), L.ptr(p["dec.pos_emb.weight"]),
              L.ptr(w["X0"]), L.i32(B), L.i32(Lq), L.i32(D), L.ptr(self.hyper if use_drop else None), st)
        if self.vae:
            Z = self.Z
            if enc_on:
                assert triples.dtype == torch.int64 and triples.is_contiguous() and triples.device == self.device
                W3, Re = 3 * D, T * B
                masked = self.pad_rid is not None
                _call("ark_triple_gather", L.ptr(triples), L.ptr(p["enc.e_emb.weight"]), L.ptr(p["enc.r_emb.weight"]), L.ptr(w["E0"]),
                      L.ptr(w["kmask"] if masked else None), L.i32(B), L.i32(T), L.i32(D), L.i64(self.pad_rid if masked else -1), st)
                km = w["kmask"] if masked else None
                x = w["E0"]
                for l in range(self.n_enc):
                    pre = f"enc.txf.layers.{l}."
                    x = self._self_attn_fwd(w["enc"][l], x, pre, Re, W3, B, T, False, km, use_drop, 1, l)
                    x = self._ff_fwd(w["enc"][l], x, pre, "norm2", Re, W3, use_drop, 1, l, "s2", "x2", "st2")
                _call("ark_seq_pool_fwd", L.ptr(x), L.ptr(km), L.ptr(w["g"]), L.ptr(w["inv_cnt"]), L.i32(B), L.i32(T), L.i32(W3), st)
                self._gemm(KM, KM, L.EPI_BIAS, w["g"], W3, p["enc.mu.weight"], W3, w["head"], 2 * Z, B, 2 * Z, W3, bias=p["enc.mu.bias"])
                if eps is None:
                    eps = w["eps0"]
                    _call("ark_normal_fill", L.ptr(eps), L.i64(eps.numel()), L.u64(self._noise_seed()), L.ptr(self.hyper), st)
                self._eps = eps
                _call("ark_latent_fwd_ex", L.ptr(w["head"]), L.ptr(eps), L.ptr(w["mu"]), L.ptr(w["logv"]), L.ptr(w["z"]), L.ptr(w["kl"]),
                      L.i32(B), L.i32(Z), L.i32(0), st)
                zt = w["z"]
            else:
                zt = z_given.to(self.device, dtype=torch.float32).contiguous()
            self._gemm(KM, KM, L.EPI_BIAS, zt, Z, p["dec.z_proj.weight"], Z, w["mem"], D, B, D, Z, bias=p["dec.z_proj.bias"])
        x = w["X0"]
        for l in range(n):
            pre = f"dec.txf.layers.{l}."
            d = w["dec"][l]
            x = self._self_attn_fwd(d, x, pre, R, D, B, Lq, True, None, use_drop, 0, l)
            if self.vae:
                x = self._cross_attn_fwd(d, x, w["mem"], pre, R, D, B, Lq, use_drop, l)
                x = self._ff_fwd(d, x, pre, "norm3", R, D, use_drop, 0, l, "s3", "x3", "st3")
            else:
                x = self._ff_fwd(d, x, pre, "norm2", R, D, use_drop, 0, l, "s2", "x2", "st2")
        self._top = x
        fused = self.fused_ce and with_loss and R % 16 == 0
        self._fused_step = bool(fused and with_dlogits)
        if not fused:
            logits = self._logits(w)
            self._gemm(KM, KM, L.EPI_BIAS, x, D, p["dec.out.weight"], D, logits, self.ldl, R, V, D, bias=p["dec.out.bias"])
        if with_loss:
            if ce_count is None:
                _call("ark_count_targets", L.ptr(seq), L.i64(ld_seq), L.i32(B), L.i32(Lq), L.ptr(self.hyper), st)
                self._hp.pop("CE_COUNT", None)
            if fused:
                y16, w16, lse, dY, ws = self._fused_ce_bufs(w, R)
                _call("ark_cast16", L.i32(self.prec_fwd), L.ptr(x), L.ptr(y16), L.i64(R * D), st)
                _call("ark_cast16", L.i32(self.prec_fwd), L.ptr(p["dec.out.weight"]), L.ptr(w16), L.i64(V * D), st)
                if with_dlogits and ws is not None:   # few rows: the vocabulary sweep split over workgroups + one merging launch
                    _call("ark_vocab_ce_fwd_ws", L.i32(self.prec_fwd), L.ptr(y16), L.ptr(w16), L.ptr(p["dec.out.bias"]), L.ptr(seq),
                          L.i64(ld_seq), L.ptr(self.hyper), L.ptr(w["row_loss"]), L.ptr(lse), L.ptr(dY), L.ptr(ws), L.i64(ws.numel()),
                          L.i32(B), L.i32(Lq), L.i32(V), L.i32(D), L.i32(0), st)
                else:
                    _call("ark_vocab_ce_fwd", L.i32(self.prec_fwd), L.ptr(y16), L.ptr(w16), L.ptr(p["dec.out.bias"]), L.ptr(seq),
                          L.i64(ld_seq), L.ptr(self.hyper), L.ptr(w["row_loss"]), L.ptr(lse), L.ptr(dY if with_dlogits else None),
                          L.i32(B), L.i32(Lq), L.i32(V), L.i32(D), st)
            else:
                _call("ark_ce_fwd_bwd", L.ptr(logits), L.i64(self.ldl), L.ptr(seq), L.i64(ld_seq), L.ptr(self.hyper),
                      L.ptr(w["row_loss"]), L.ptr(logits if with_dlogits else None), L.ptr(None), L.i32(0), L.i64(0), L.i32(B),
                      L.i32(Lq), L.i32(V), st)
            _call("ark_loss_finalize", L.ptr(w["row_loss"]), L.i32(R), L.ptr(w["kl"] if enc_on else None), L.ptr(self.hyper),
                  L.ptr(w["out4"]), st)
        return w

    # ------------------------------------------------------------------ sublayers, backward
    def _ff_bwd(self, d, dy, x_in, pre, norm, s_key, st_key, R, W, drop, stack, l, ds, tmp, df):
        """dy = gradient of LN(x_in + drop(FF(x_in))) -> gradient w.r.t. x_in, returned in `ds`"""
        KM, MM, p, g = L.LAY_KMAJ, L.LAY_MMAJ, self.p, self.g
        self._ln_bwd(dy, d[s_key], d[st_key], pre + norm, ds, R, W)
        fast = self._fast_ok(R, W, FF)
        dg2 = ds
        if fast:   # dropout + bias gradient + the products' 16-bit operand in ONE pass over ds (which stays the residual's)
            self._prep(ds, R, W, g[pre + "linear2.bias"], self._seed(stack, l, 3) if drop else None)
        else:
            if drop:
                dg2 = tmp
                self._copy(dg2, ds)
                self._drop(dg2, self._seed(stack, l, 3))
            self._colsum(dg2, W, g[pre + "linear2.bias"], R, W)
        self._gemm(MM, MM, L.EPI_NONE, dg2, W, d["f"], FF, g[pre + "linear2.weight"], FF, W, FF, R, acc=1)
        # df = (dg2 W2) masked by ReLU (and by the feed-forward dropout: f is positive only where both let it through)
        self._gemm(KM, MM, L.EPI_MUL_RELU, dg2, W, p[pre + "linear2.weight"], FF, df, FF, R, FF, W, aux=d["f"])
        if fast:
            self._prep(df, R, FF, g[pre + "linear1.bias"], self._seed(stack, l, 2) if drop else None)
        else:
            if drop:
                self._drop(df, self._seed(stack, l, 2))
            self._colsum(df, FF, g[pre + "linear1.bias"], R, FF)
        self._gemm(MM, MM, L.EPI_NONE, df, FF, x_in, W, g[pre + "linear1.weight"], W, FF, W, R, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, df, FF, p[pre + "linear1.weight"], W, ds, W, R, W, FF, acc=1)   # + df W1 (residual already there)
        return ds

    def _self_attn_bwd(self, d, dy, x_in, pre, R, W, B, Ls, causal, kmask, drop, stack, l, ds, tmp, scratch, dqkv, dscore):
        """dy = gradient of LN1(x_in + drop(SA(x_in))) -> gradient w.r.t. x_in, returned in `ds`"""
        KM, MM, p, g = L.LAY_KMAJ, L.LAY_MMAJ, self.p, self.g
        a = pre + "self_attn."
        self._ln_bwd(dy, d["s1"], d["st1"], pre + "norm1", ds, R, W)
        fast = self._fast_ok(R, W)
        dsa = ds
        if fast:
            self._prep(ds, R, W, g[a + "out_proj.bias"], self._seed(stack, l, 1) if drop else None)
        else:
            if drop:
                dsa = tmp
                self._copy(dsa, ds)
                self._drop(dsa, self._seed(stack, l, 1))
            self._colsum(dsa, W, g[a + "out_proj.bias"], R, W)
        self._gemm(MM, MM, L.EPI_NONE, dsa, W, d["att"], W, g[a + "out_proj.weight"], W, W, W, R, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, dsa, W, p[a + "out_proj.weight"], W, scratch, W, R, W, W)            # d(att)
        if "lse" in d:
            _call("ark_attn_flash_bwd", L.i32(self.prec_bwd), L.ptr(d["qkv"]), L.ptr(d["att"]), L.ptr(d["lse"]), L.ptr(scratch),
                  L.ptr(self.ws["delta"]), L.ptr(dqkv), L.ptr(kmask), L.i32(B), L.i32(Ls), L.i32(W), L.i32(self.H),
                  L.i32(1 if causal else 0), L.f32(self.p_drop if drop else 0.0), L.u64(self._seed(stack, l, 0)), L.ptr(self.hyper),
                  L.cur_stream())
        else:
            _call("ark_attn_bwd", L.ptr(d["qkv"]), L.ptr(d["att"]), L.ptr(d["probs"]), L.ptr(scratch), L.ptr(dscore), L.ptr(dqkv),
                  L.ptr(kmask), L.i32(B), L.i32(Ls), L.i32(W), L.i32(self.H), L.i32(1 if causal else 0),
                  L.f32(self.p_drop if drop else 0.0), L.u64(self._seed(stack, l, 0)), L.ptr(self.hyper), L.cur_stream())
        if fast:
            self._prep(dqkv, R, 3 * W, g[a + "in_proj_bias"])
        else:
            self._colsum(dqkv, 3 * W, g[a + "in_proj_bias"], R, 3 * W)
        self._gemm(MM, MM, L.EPI_NONE, dqkv, 3 * W, x_in, W, g[a + "in_proj_weight"], W, 3 * W, W, R, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, dqkv, 3 * W, p[a + "in_proj_weight"], W, ds, W, R, W, 3 * W, acc=1)
        return ds

    def _cross_attn_bwd(self, d, dy, mem, pre, R, W, B, Ls, drop, l, ds, tmp, scratch, dmem, dv):
        """dy = gradient of LN2(x + drop(CA)) -> gradient w.r.t. x in `ds` (the residual only: the queries do not reach the
        output); the memory's gradient ACCUMULATES into dmem [B, W].  Query / key projection gradients are exactly zero."""
        KM, MM, p, g = L.LAY_KMAJ, L.LAY_MMAJ, self.p, self.g
        a = pre + "multihead_attn."
        self._ln_bwd(dy, d["s2"], d["st2"], pre + "norm2", ds, R, W)
        dca = ds
        if self._fast_ok(R, W):
            self._prep(ds, R, W, g[a + "out_proj.bias"], self._seed(0, l, 5) if drop else None)
        else:
            if drop:
                dca = tmp
                self._copy(dca, ds)
                self._drop(dca, self._seed(0, l, 5))
            self._colsum(dca, W, g[a + "out_proj.bias"], R, W)
        self._gemm(MM, MM, L.EPI_NONE, dca, W, d["ctx"], W, g[a + "out_proj.weight"], W, W, W, R, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, dca, W, p[a + "out_proj.weight"], W, scratch, W, R, W, W)            # d(ctx)
        _call("ark_xattn_bcast_bwd", L.ptr(scratch), L.ptr(d["cs"]), L.ptr(dv), L.i32(B), L.i32(Ls), L.i32(W), L.i32(self.H), L.cur_stream())
        self._colsum(dv, W, g[a + "in_proj_bias"][2 * W:], B, W)
        self._gemm(MM, MM, L.EPI_NONE, dv, W, mem, W, g[a + "in_proj_weight"][2 * W:], W, W, W, B, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, dv, W, p[a + "in_proj_weight"][2 * W:], W, dmem, W, B, W, W, acc=1)
        return ds

    # ------------------------------------------------------------------ backward
    def backward(self, ext_dhead=None):
        """backward of the last forward; the gradient w.r.t. the logits sits in ws['logits'].  t-SAIL: the KL gradient is
        added inside the latent kernel (hyper BETA x KL_NORM); ext_dhead [B, 2Z] (optional) adds an external gradient
        w.r.t. (mu | logv), as the reference-style autograd path needs."""
        for _ in self._backward_steps(ext_dhead):
            pass

    def dp_split(self):
        """offset that splits the flat gradient buffer into the two data-parallel buckets of t-SAIL: [split, total) -- the
        decoder layers and the output projection -- is complete after the decoder half of the backward pass, [0, split) --
        encoder, heads, embeddings, z-projection -- after the latent / encoder half.  t-ARK: one bucket (0)."""
        return self.layout.entries["dec.txf.layers.0.self_attn.in_proj_weight"][0] if self.vae else 0

    def _backward_steps(self, ext_dhead=None):
        """backward() as a generator with ONE seam: it yields after the decoder half when a latent / encoder half follows
        (the data-parallel step starts the decoder bucket's all-reduce there)"""
        self.prec = self.prec_bwd
        w, B, Lq = self.ws, self._B, self._Lrun
        D, n, V, H = self.D, self.n, self.V, self.H
        R = Lq * B
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p, g = self.p, self.g
        seq = self._seq
        ld_seq = seq.shape[1]
        drop = self._used_drop
        self._zero(self.G)
        bufs = [w["dA"], w["dB"], w["dC"]]
        dy = bufs[0]
        if self._fused_step:
            # dW_out += dlogits^T y and db_out += colsum(dlogits) with dlogits recomputed from the log-sum-exp; dY came out of
            # the forward sweep (tile-native: one small pass turns it row-major)
            y16, w16, lse, dY, _ = self._fused_ce_bufs(w, R)
            _call("ark_vocab_ce_dw", L.i32(self.prec_fwd), L.ptr(y16), L.ptr(w16), L.ptr(p["dec.out.bias"]), L.ptr(seq), L.i64(ld_seq),
                  L.ptr(self.hyper), L.ptr(lse), L.ptr(g["dec.out.weight"]), L.ptr(g["dec.out.bias"]), L.i32(B), L.i32(Lq), L.i32(V),
                  L.i32(D), st)
            _call("ark_from_tiled", L.ptr(dY), L.ptr(dy), L.i32(R), L.i32(D), st)
        else:
            dlog = self._logits(w)
            self._colsum(dlog, self.ldl, g["dec.out.bias"], R, V)
            self._gemm(MM, MM, L.EPI_NONE, dlog, self.ldl, self._top, D, g["dec.out.weight"], D, V, D, R, acc=1)
            self._gemm(KM, MM, L.EPI_NONE, dlog, self.ldl, p["dec.out.weight"], D, dy, D, R, D, V)
        if self.vae:
            self._zero(w["dmem"])

        def others(cur):
            """the two rotating [R, W] buffers that are not `cur`"""
            o = [b for b in bufs if b is not cur]
            return o[0], o[1]

        for l in range(n - 1, -1, -1):
            pre = f"dec.txf.layers.{l}."
            d = w["dec"][l]
            x_in = w["dec"][l - 1]["x3" if self.vae else "x2"] if l > 0 else w["X0"]
            if self.vae:
                ds, tmp = others(dy)
                dy = self._ff_bwd(d, dy, d["x2"], pre, "norm3", "s3", "st3", R, D, drop, 0, l, ds, tmp, w["df"])
                ds, tmp = others(dy)
                # (the scratch for d(ctx) is the feed-forward gradient buffer's first R*D floats: free at this point)
                dy = self._cross_attn_bwd(d, dy, w["mem"], pre, R, D, B, Lq, drop, l, ds, tmp, w["df"].view(-1)[:R * D].view(R, D),
                                          w["dmem"], w["dv"])
            else:
                ds, tmp = others(dy)
                dy = self._ff_bwd(d, dy, d["x1"], pre, "norm2", "s2", "st2", R, D, drop, 0, l, ds, tmp, w["df"])
            ds, tmp = others(dy)
            dy = self._self_attn_bwd(d, dy, x_in, pre, R, D, B, Lq, True, None, drop, 0, l, ds, tmp, w["df"].view(-1)[:R * D].view(R, D),
                                     w["dqkv"], w["dscore"])
        _call("ark_tok_scatter", L.ptr(seq), L.i64(ld_seq), L.ptr(dy), L.ptr(g["dec.tok_emb.weight"]), L.i32(B), L.i32(Lq), L.i32(D),
              L.i32(V), st)
        self._colsum(dy, D, g["dec.pos_emb.weight"], B, D, n_batch=Lq, bs_in=B * D, bs_out=D)
        if not (self.vae and self._enc_on):
            return
        yield "decoder"
        st = L.cur_stream()
        # ---- latent + encoder half (t-SAIL)
        Z, T = self.Z, self._T
        W3, Re = 3 * D, T * B
        self._colsum(w["dmem"], D, g["dec.z_proj.bias"], B, D)
        self._gemm(MM, MM, L.EPI_NONE, w["dmem"], D, w["z"], Z, g["dec.z_proj.weight"], Z, D, Z, B, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, w["dmem"], D, p["dec.z_proj.weight"], Z, w["dz"], Z, B, Z, D)
        _call("ark_latent_bwd_ex", L.ptr(w["dz"]), L.ptr(w["head"]), L.ptr(self._eps), L.ptr(self.hyper), L.ptr(w["dhead"]), L.i32(B), L.i32(Z),
              L.i32(0), st)
        if ext_dhead is not None:
            _call("ark_axpy", L.ptr(w["dhead"]), L.ptr(ext_dhead.contiguous()), L.i64(B * 2 * Z), L.f32(1.0), st)
        self._colsum(w["dhead"], 2 * Z, g["enc.mu.bias"], B, 2 * Z)
        self._gemm(MM, MM, L.EPI_NONE, w["dhead"], 2 * Z, w["g"], W3, g["enc.mu.weight"], W3, 2 * Z, W3, B, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, w["dhead"], 2 * Z, p["enc.mu.weight"], W3, w["dg"], W3, B, W3, 2 * Z)
        km = w["kmask"] if self.pad_rid is not None else None
        ebufs = [w["eA"], w["eB"], w["eC"]]
        dy = ebufs[0]
        _call("ark_seq_pool_bwd", L.ptr(w["dg"]), L.ptr(km), L.ptr(w["inv_cnt"]), L.ptr(dy), L.i32(B), L.i32(T), L.i32(W3), st)

        def eothers(cur):
            o = [b for b in ebufs if b is not cur]
            return o[0], o[1]

        for l in range(self.n_enc - 1, -1, -1):
            pre = f"enc.txf.layers.{l}."
            d = w["enc"][l]
            x_in = w["enc"][l - 1]["x2"] if l > 0 else w["E0"]
            ds, tmp = eothers(dy)
            dy = self._ff_bwd(d, dy, d["x1"], pre, "norm2", "s2", "st2", Re, W3, drop, 1, l, ds, tmp, w["e_df"])
            ds, tmp = eothers(dy)
            dy = self._self_attn_bwd(d, dy, x_in, pre, Re, W3, B, T, False, km, drop, 1, l, ds, tmp, self._e_scratch(w, Re, W3),
                                     w["e_dqkv"], w["e_dscore"])
        _call("ark_triple_scatter", L.ptr(self._triples), L.ptr(dy), L.ptr(g["enc.e_emb.weight"]), L.ptr(g["enc.r_emb.weight"]), L.i32(B),
              L.i32(T), L.i32(D), L.i64(-1 if self.pad_eid is None else self.pad_eid), L.i64(-1 if self.pad_rid is None else self.pad_rid), st)

    def _e_scratch(self, w, Re, W3):
        """[Re, 3D] scratch of the encoder's attention backward (FF = 2048 columns may be narrower than 3D: own buffer)"""
        if "e_scr" not in w:
            w["e_scr"] = torch.empty(Re, W3, device=self.device, dtype=torch.float32)
        return w["e_scr"]

    # ------------------------------------------------------------------ whole step
    def _dp_segments(self, triples, seq, eps, ce_count):
        """the data-parallel step as a generator: every next() launches the work that completes one gradient bucket [lo, hi)
        and yields it (packed into the bf16 transport buffer when `ark_dp_bf16`)"""
        tot, split = self.layout.total, self.dp_split()
        self.forward(triples if self.vae else None, seq, eps if self.vae else None, ce_count=ce_count)
        two = False
        for _ in self._backward_steps():
            two = True
            self._dp_pack(split, tot)
            yield (split, tot)       # its all-reduce flies underneath the latent / encoder half
        lo_hi = (0, split) if two else (0, tot)
        self._dp_pack(*lo_hi)
        yield lo_hi

    def train_step(self, triples, seq, eps=None, grad_sync=None, ce_count=None, dp=False):
        """forward + loss + backward (+ bucketed gradient all-reduce) + Adam; returns out4 on the device"""
        self._default_norms(seq.shape[0])
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        if dp:
            import torch.distributed as dist
            handles = []
            for item in self._dp_segments(triples, seq, eps, ce_count):
                handles.append((item, dist.all_reduce(self.dp_bucket(*item), op=dist.ReduceOp.SUM, async_op=True)))
            self._dp_nseg = len(handles)
            for item, h in handles:
                h.wait()
                self._dp_unpack(*item)
        else:
            self.forward(triples if self.vae else None, seq, eps if self.vae else None, ce_count=ce_count)
            self.backward()
        if grad_sync is not None:
            grad_sync(self.G)
        self.adam()
        return self.ws["out4"]

    def capture_train_step(self, triples, seq, eps=None, ce_count=None, grad_sync=None, dp=False):
        """the train step (forward + loss + backward + Adam, ~200 short launches) as hipGraphs on fixed-address inputs;
        returns replay().  Single process: ONE graph.  Data parallel (`dp=True`): one graph per gradient bucket (t-SAIL: decoder
        half | latent + encoder half; t-ARK: one), that bucket's asynchronous all-reduce launched right after its replay, then
        one graph for [widen the reduced buckets, Adam].  Everything that changes per step lives on the device: dropout and
        noise draw counters, Adam's step scalars, the CE count."""
        if grad_sync is not None:
            raise L.ArkError(f"{self.mt}: capture with a caller-supplied grad_sync is not supported; use dp=True")
        self._default_norms(seq.shape[0])
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):   # warm-up outside capture (allocates the workspace); this IS a step
            self.train_step(triples, seq, eps, ce_count=ce_count, dp=dp)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        steps0 = self.adam_steps
        graphs, segs = [], []
        if dp:
            gen = self._dp_segments(triples, seq, eps, ce_count)
            for _ in range(self._dp_nseg):   # (counted by the eager warm-up step above)
                box = []
                gseg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gseg, capture_error_mode=CAPTURE_MODE):
                    box.append(next(gen))
                graphs.append(gseg)
                segs.append((gseg, box[0]))
            assert next(gen, None) is None, "data-parallel schedule changed between warm-up and capture"

            def tail():
                for _, item in segs:
                    self._dp_unpack(*item)
                self.adam()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                tail()
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                self.train_step(triples, seq, eps, ce_count=ce_count)
        graphs.append(g)
        self.adam_steps = steps0   # capture does not execute
        out4 = self.ws["out4"]
        gstream = torch.cuda.Stream(device=self.device)

        def replay():
            # (as Engine.capture_train_step: replays run on an explicit stream fenced against the caller's)
            cur = torch.cuda.current_stream()
            use = cur if cur.cuda_stream != 0 else gstream
            if use is not cur:
                use.wait_stream(cur)
            with torch.cuda.stream(use):
                if dp:
                    import torch.distributed as dist
                    handles = []
                    for gseg, item in segs:
                        gseg.replay()
                        handles.append(dist.all_reduce(self.dp_bucket(*item), op=dist.ReduceOp.SUM, async_op=True))
                    for h in handles:
                        h.wait()
                g.replay()
            if use is not cur:
                cur.wait_stream(use)
            self.adam_steps += 1
            return out4

        self._graphs[("train", self.ws_key, bool(dp))] = graphs
        self._pinned.add(self.ws_key)
        return replay

    def graphed_train_step(self, triples, seq, ce_count=None, dp=False):
        """train_step() through cached hipGraphs (data parallel: one graph per gradient bucket, see capture_train_step)"""
        assert self.training, "graphed_train_step is a training-mode step"
        key = (None if triples is None else tuple(triples.shape), tuple(seq.shape), ce_count is None, bool(dp))
        ent = self._graph_steps.get(key)
        if ent is None:
            tri_s = None if triples is None else triples.clone()
            seq_s = seq.clone()
            replay = self.capture_train_step(tri_s, seq_s, None, ce_count=ce_count, dp=dp)
            self._graph_steps[key] = (tri_s, seq_s, replay)
            return self.ws["out4"]
        tri_s, seq_s, replay = ent
        if tri_s is not None:
            tri_s.copy_(triples, non_blocking=True)
        seq_s.copy_(seq, non_blocking=True)
        self._default_norms(seq.shape[0])
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        self._workspace(seq.shape[0], seq.shape[1] - 1, 0 if tri_s is None else tri_s.shape[1])
        return replay()

    def eval_loss(self, triples, seq, eps=None):
        self._default_norms(seq.shape[0])
        was = self.training
        self.training = False
        try:
            w = self.forward(triples if self.vae else None, seq, eps if self.vae else None, with_dlogits=False)
        finally:
            self.training = was
        return w["out4"]

    @torch.no_grad()
    def prefix_logits(self, prefix, z=None):
        """logits [B, V] of the position after `prefix` [B, t] (the reference re-runs the whole prefix per generated
        token, models.py:291,430; so does this: the Transformer has no recurrent state to carry).  t-SAIL: z [B, Z]."""
        was = self.training
        self.training = False
        try:
            t, B = prefix.shape[1], prefix.shape[0]
            w = self.forward(None, prefix.contiguous(), None, with_loss=False, L_run=t, z_given=z, decode_ws=True)
        finally:
            self.training = was
        return self._logits(w)[(t - 1) * B:t * B, :self.V]

    def decode_begin(self, *a, **k):
        raise L.ArkError(f"{self.mt} has no incremental decoder state: use prefix_logits()")

    @torch.no_grad()
    def greedy_decode(self, z, max_len=None, bos=1, eos=2):
        """token sequences of SAIL.decode_latent(z, beam=1) for t-SAIL (reference models.py:282-300): prefix re-run,
        argmax of the last position, stop once every row ends in EOS"""
        assert self.vae
        B = z.shape[0]
        Lmax = (self.seq_len - 1) if max_len is None else max_len
        z = z.to(self.device, dtype=torch.float32)
        s = torch.full((B, 1), bos, dtype=torch.int64, device=self.device)
        nxt = torch.empty(B, dtype=torch.int64, device=self.device)
        for _ in range(Lmax):
            logits = self.prefix_logits(s, z)
            _call("ark_argmax_rows", L.ptr(logits), L.i64(self.ldl), L.ptr(nxt), L.i32(B), L.i32(self.V), L.cur_stream())
            s = torch.cat([s, nxt[:, None]], 1)
            if bool((s[:, -1] == eos).all()):
                break
        return s

    @torch.no_grad()
    def beam_decode(self, z, beam, max_len=None, bos=1, eos=2):
        """the reference's batch-shared beam (models.py:282-300) on prefix re-runs: candidates ranked by the batch-MEAN
        accumulated log-probability (stable descending sort), stop when every kept beam ends in EOS everywhere"""
        assert self.vae and beam >= 1
        B = z.shape[0]
        Lmax = (self.seq_len - 1) if max_len is None else max_len
        z = z.to(self.device, dtype=torch.float32)
        beams = [(torch.full((B, 1), bos, dtype=torch.int64, device=self.device), torch.zeros(B, device=self.device))]
        for _ in range(Lmax):
            cand = []
            for s, lp in beams:
                logp = torch.log_softmax(self.prefix_logits(s, z).float(), dim=-1)
                top_lp, ids = logp.topk(beam, dim=-1)
                for k in range(beam):
                    cand.append((torch.cat([s, ids[:, k:k + 1]], 1), lp + top_lp[:, k]))
            means = torch.stack([c[1].mean() for c in cand])
            order = torch.sort(means, descending=True, stable=True).indices[:beam].tolist()
            beams = [cand[i] for i in order]
            if all(bool((s[:, -1] == eos).all()) for s, _ in beams):
                break
        return beams[0][0]
