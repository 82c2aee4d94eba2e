"""Host-side engine for the SAIL / ARK training step on MI355X.

Owns the flat fp32 parameter / gradient / Adam-moment buffers and the per-batch activation
workspace in HBM, and drives the hand-written HIP kernels (include/ark_amd.h) in the order of
the reference's train step:

    SAIL.forward          kgvae/model/models.py:317-320   (encoder :46-64, decoder :136-142)
    ARK.forward           kgvae/model/models.py:395-405   (decoder :340-345)
    ce + b*kl, backward   kgvae/experiments/ablation_study.py:59-76
    optim.Adam.step       kgvae/experiments/ablation_study.py:571

PyTorch is used for device memory, streams and (optionally) hipGraph capture only; there is no
torch arithmetic and no CPU fallback on this path.

Layout notes
  * decoder activations are time-major: row (t, b) = t*B + b, so each GRU timestep is one
    contiguous [B, D] block; layer outputs are stored with an extra leading slot holding h0.
  * the two latent heads share one [2Z, 3D] weight block (mu rows first) so they run as one GEMM.
"""
import math
from collections import OrderedDict

import torch

from . import _lib as L

HP = dict(LR=0, BETA=1, KL_NORM=2, CE_INV_COUNT=3, CE_COUNT=4, ADAM_STEP=5, ADAM_BC1=6, ADAM_BC2=7, ADAM_B1=8,
          ADAM_B2=9, ADAM_EPS=10, GRAD_SCALE=11, DROP_STEP=12, NOISE_STEP=13, COUNT=16)

# precision policy -> (forward products, backward products)
#   "mixed": fp16 operands forward (3 more mantissa bits than bf16 at the same MFMA rate keeps the
#            ELBO within 1e-4 of the fp32 reference), bf16 operands backward (gradient range-safe)
PREC = {"f32": (L.PREC_F32, L.PREC_F32), "fp32": (L.PREC_F32, L.PREC_F32), "float32": (L.PREC_F32, L.PREC_F32),
        "bf16": (L.PREC_BF16, L.PREC_BF16), "bfloat16": (L.PREC_BF16, L.PREC_BF16),
        "f16": (L.PREC_F16, L.PREC_F16), "mixed": (L.PREC_F16, L.PREC_BF16)}


def _rup(x, m):
    return (x + m - 1) // m * m


# Graph captures run in THREAD-LOCAL error mode: with torch.distributed initialised, ProcessGroupNCCL's watchdog thread polls
# the completion events of earlier collectives; under the default (global) mode such a query from another thread while THIS
# thread captures is "operation not permitted when stream is capturing" and the watchdog takes the process down (seen in
# round 4: a second capture in a process that had already run RCCL collectives).
CAPTURE_MODE = "thread_local"

_calls = [0]   # library calls made so far (txf_engine: "nothing has run since the previous product")


def _ranks_on_this_device(world_size):
    """how many ranks of this job share one GPU: 1 on a real node (one process per GPU); torchrun's LOCAL_WORLD_SIZE against
    the visible devices otherwise (the two-process tests on a one-GPU box)"""
    if world_size <= 1:
        return 1
    import os
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world_size))
    ndev = max(1, torch.cuda.device_count())
    return max(1, -(-local // ndev))


_stamps = None   # diagnostics (tools/step_stamps.py): {"buf": uint64 device tensor, "log": [(library call, stream)]}


def _call(name, *args):
    _calls[0] += 1
    L.check(getattr(L.lib(), name)(*args), name)
    if _stamps is not None and len(_stamps["log"]) < _stamps["buf"].numel():
        # every library call takes its stream last: a one-thread stamp launch right behind it on the same queue
        st = args[-1]
        _stamps["log"].append((name, getattr(st, "value", None)))
        L.check(L.lib().ark_stamp(L.ptr(_stamps["buf"]), L.i32(len(_stamps["log"]) - 1), st), "ark_stamp")


class ParamLayout:
    """name -> (offset, shape) inside the flat buffers.  Blocks start on 16-byte boundaries."""

    def __init__(self, cfg):
        mt = cfg["model_type"]
        D, n, V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        self.tied = bool(cfg.get("tie_weights", True))
        blocks = []  # list of lists of (name, shape): tensors inside a block are packed back to back
        if mt == "SAIL":
            Z, H = cfg["d_latent"], 3 * D
            blocks += [[("enc.e_emb.weight", (cfg["n_entities"], D))], [("enc.r_emb.weight", (cfg["n_relations"], D))]]
            for i in range(n):
                blocks += [[(f"enc.mlp.{2 * i}.weight", (H, H))], [(f"enc.mlp.{2 * i}.bias", (H,))]]
            blocks += [[("enc.mu.weight", (Z, H)), ("enc.logv.weight", (Z, H))],
                       [("enc.mu.bias", (Z,)), ("enc.logv.bias", (Z,))]]
            # z_proj's gradient is produced by the latent/encoder half of the backward pass, so it sits with
            # the encoder block: everything from dec.tok_emb on is complete once the decoder half is done
            blocks += [[("dec.z_proj.weight", (D, Z))], [("dec.z_proj.bias", (D,))], [("dec.tok_emb.weight", (V, D))]]
        elif mt == "ARK":
            blocks += [[("dec.tok_emb.weight", (V, D))], [("dec.pos_emb.weight", (cfg["seq_len"], D))]]
        else:
            raise NotImplementedError(f"Unknown model_type: {mt}")
        for l in range(n):
            blocks += [[(f"dec.gru.weight_ih_l{l}", (3 * D, D))], [(f"dec.gru.weight_hh_l{l}", (3 * D, D))],
                       [(f"dec.gru.bias_ih_l{l}", (3 * D,))], [(f"dec.gru.bias_hh_l{l}", (3 * D,))]]
        if not self.tied:
            blocks += [[("dec.out.weight", (V, D))]]
        blocks += [[("dec.out.bias", (V,))]]
        self.entries = OrderedDict()
        off = 0
        for blk in blocks:
            off = _rup(off, 4)
            for name, shape in blk:
                numel = 1
                for s in shape:
                    numel *= s
                self.entries[name] = (off, tuple(shape), numel)
                off += numel
        self.total = _rup(off, 4)
        # gradients in [dec_grad_offset, total) are final after Engine.backward_decoder(): data-parallel
        # runs all-reduce that bucket while the encoder half is still computing
        self.dec_grad_offset = self.entries["dec.tok_emb.weight"][0]


class Engine:
    def __init__(self, cfg, device, precision="f32", world_size=1, rank=0):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.ArkError("ark_amd.Engine needs a GPU device (no CPU fallback exists)")
        L.lib()  # fail loudly if the extension is missing
        self.prec_fwd, self.prec_bwd = PREC[precision]
        self.prec = self.prec_fwd  # products issued by the next kernel wrapper call
        self.precision = precision
        self.mt = cfg["model_type"]
        self.D, self.n, self.V = cfg["d_model"], cfg["n_layers"], cfg["vocab_size"]
        self.Z = cfg.get("d_latent", 0) if self.mt == "SAIL" else 0
        self.seq_len = cfg["seq_len"]
        self.L = self.seq_len - 1
        self.pad_eid = cfg.get("pad_eid")
        self.pad_rid = cfg.get("pad_rid")
        self.p_drop = float(cfg.get("dec_dropout", 0.1)) if self.n > 1 else 0.0
        self.world_size = world_size
        self.rank = rank     # data-parallel rank: mixed into the dropout seed so row b of every rank draws its own mask
        if self.D % 32 != 0:
            raise L.ArkError("d_model must be a multiple of 32 for the gfx950 GRU tiles")
        self.layout = ParamLayout(cfg)
        n = self.layout.total
        dev = self.device
        self.P = torch.zeros(n, device=dev, dtype=torch.float32)
        self.G = torch.zeros(n, device=dev, dtype=torch.float32)
        self.M = torch.zeros(n, device=dev, dtype=torch.float32)
        self.Vv = torch.zeros(n, device=dev, dtype=torch.float32)
        self.p = OrderedDict()
        self.g = OrderedDict()
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
        self.hyper = init.to(dev)   # device-resident step scalars (include/ark_amd.h ARK_HP_*)
        self._hp = {"LR": float(cfg.get("learning_rate", 1e-3)), "BETA": 1.0, "GRAD_SCALE": 1.0}   # host mirror of host-owned slots
        self.adam_steps = 0
        self.ws = None
        self.ws_key = None
        self.training = True
        self.drop_seed = int(cfg.get("dropout_seed", 0x5A11))
        self.noise_seed = int(cfg.get("noise_seed", cfg.get("seed", 0)))
        self._graphs = {}
        self._ws_cache = {}
        self._pinned_B = set()
        self._graph_steps = {}
        self.fwd_gen = 0   # forward-call generation (the autograd wrappers check that backward follows ITS forward)
        # fast path ("v2"): 16-bit operands streamed by LDS-DMA, layer-diagonal GRU kernels; needs 16-bit shadows of
        # the GEMM weights, refreshed with every optimiser step: W_ih, W_hh, W_tok, encoder MLP / heads in the forward
        # type; W_ih^T, W_hh^T, W_tok^T, W_mlp^T in the backward type
        self.use_dma = self.prec_fwd != L.PREC_F32 and self.prec_bwd != L.PREC_F32 and self.D % 64 == 0
        self.tune = L.diag_tuning(**dict(cfg.get("ark_diag_tuning") or {})) if self.use_dma else None
        self.wg_tune = L.wgrad_tuning(**dict(cfg.get("ark_wgrad_tuning") or {})) if self.use_dma else None
        if self.use_dma:
            D, V = self.D, self.V
            i16 = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.int16)
            self.wih16 = [i16(3 * D, D) for _ in range(self.n)]
            self.whh16 = [i16(3 * D, D) for _ in range(self.n)]
            self.wihT16 = [i16(D, 3 * D) for _ in range(self.n)]
            self.whhT16 = [i16(D, 3 * D) for _ in range(self.n)]
            self.wtok16 = i16(V, D)
            self.Vp = _rup(V, 64)                                    # K-padded vocabulary for the dY product
            self.wtokT16 = torch.zeros(D, self.Vp, device=dev, dtype=torch.int16)   # W_tok^T, backward type, zero pad
            if self.mt == "SAIL":
                H = 3 * D
                self.wm16 = [i16(H, H) for _ in range(self.n)]      # encoder MLP weights, forward type
                self.wmT16 = [i16(H, H) for _ in range(self.n)]     # their transposes, backward type
                self.wh16 = i16(2 * self.Z, H)                      # [mu; logv] head, forward type
            # small vocabularies: layer 0's input projection per TOKEN, x_tab[V, 3D] = W_tok16 W_ih0_16^T (fp32 sums of the
            # same 16-bit products the cells would form), refreshed with the shadows; the forward diagonals' layer-0 roles
            # add its rows instead of streaming x_t W_ih0^T (the backward has used the same identity since round 1: token sums)
            self.xtab = (torch.zeros(V, 3 * D, device=dev) if (self.mt == "SAIL" and self.Vp <= 256 and cfg.get("ark_fwd_tab", True))
                         else None)
            self._shadow_jobs = self._build_shadow_jobs()
            off, tot = self.layout.dec_grad_offset, self.layout.total
            self._adam_jobs = {"all": self._build_adam_jobs(0, tot), "enc": self._build_adam_jobs(0, off),
                               "dec": self._build_adam_jobs(off, tot)}
            if self.mt == "SAIL":   # the encoder bucket = embeddings | MLP (99 % of it) | heads + z-projection
                m0, m1 = self.layout.entries["enc.mlp.0.weight"][0], self.layout.entries["enc.mu.weight"][0]
                self._adam_jobs["mlp"] = self._build_adam_jobs(m0, m1)
                self._adam_jobs["enc_rest"] = self._build_adam_jobs((0, m0), (m1, off))   # (one launch)
                self._adam_jobs["emb"] = self._build_adam_jobs(0, m0)        # entity / relation tables (wd-articles: 31 M parameters)
                self._adam_jobs["heads"] = self._build_adam_jobs(m1, off)
        self._shadow_ok = False
        self._side = None
        self._chain_streams = []
        # (measured on MI355X, syn-paths B = 1024: see DESIGN.md; 1 = the single dependent chain of rounds 1-2)
        self.diag_chains = max(1, int(cfg.get("ark_diag_chains", 2)))
        # small batches of long sequences: the whole forward recurrence as ONE persistent launch (csrc/gru_sweep.hip);
        # "auto" = where the diagonal launches are all fixed cost (see _use_sweep)
        self.sweep = cfg.get("ark_sweep", "auto")
        # processes that share THIS device (data-parallel tests put two ranks on one card; a real rank owns its GPU): every one
        # of them may hold a persistent sweep grid at the same time, and a sweep only makes progress once ALL its workgroups
        # are resident -- the host therefore admits a sweep only if the grids of all co-located ranks fit the chip together
        self.ranks_per_device = max(1, int(cfg.get("ark_ranks_per_device", 0)) or _ranks_on_this_device(world_size))
        self.sweep_bwd = bool(cfg.get("ark_sweep_bwd", True))   # (0: persistent forward, diagonal backward)
        # unit slices per physical sweep workgroup: 1 = one per CU (default), 2 = two slices share a 512-thread workgroup (half
        # the CUs held, the kernels beside the sweep get the rest).  Measured on MI355X, same box: wd-articles 7.13 -> 8.01
        # ms/step, wd-movies 1.93 -> 2.00 with 2 -- a sweep step is three memory round trips, and two slices on one CU lengthen
        # every one of them by more than the vocabulary CE gains from 208 instead of 160 CUs
        self.sweep_ws = 2 if int(cfg.get("ark_sweep_wg_slices", 1)) == 2 else 1
        self.ce_dw_after_latent = True   # (round 5: the A/B switches nothing sets became constants -- INTEGRATION.md lists what is left)
        self.ce_chunks = int(cfg.get("ark_ce_chunks", 4))   # time chunks of the sweep + CE pipeline (1: off)
        self._ce_dw_pending = None
        self._wgrads_forked = False   # this step's GRU weight gradients are on the side queue (backward_decoder)
        self._pad_bufs = {}
        self._n_valid = 0
        self._side_used = False
        self._dp_pending = None      # data parallel: event behind the decoder bucket's reduction + Adam on the update queue
        self._dp_flush_graph = None  # (captured mode: that Adam as a graph)
        self._dp_upd = None          # the update queue
        self._dp_mode = False        # inside a pipelined data-parallel step (see _forward_steps)
        self.dp_pipeline = bool(cfg.get("ark_dp_pipeline", True))
        # all-reduce bf16 copies of the gradient buckets (half the bytes over xGMI; fp32 master gradients, weights and
        # moments; measured drift after 3 steps: mean < 1e-4, tests/test_dp_gpu.py).  `ark_dp_bf16: false` = fp32 buckets
        self.dp_bf16 = bool(cfg.get("ark_dp_bf16", True))
        # order of the data-parallel step (SAIL, fast path).  "beside": the single-process order -- GRU weight gradients on the
        # side queue beside the latent / encoder backward; both buckets complete at the end of the step, the encoder bucket's
        # all-reduce is exposed, the decoder bucket's is pipelined into the next step.  "chain-first" (rounds 1-3): the
        # dependent chain first, the GRU weight gradients after it so that the encoder bucket's all-reduce has something to
        # hide under -- costs ~0.2 ms of compute on every rank before a byte moves
        self.dp_order = str(cfg.get("ark_dp_order", "beside"))
        if self.dp_order not in ("beside", "chain-first"):
            raise L.ArkError(f"ark_dp_order: {self.dp_order!r} (beside | chain-first)")
        self._defer_wgrads = False
        self._fork_pending = None
        self._dlog16_only = False
        self.dlog16_only = True
        self.h0_ride = True
        self._finalize = None
        self._defer_finalize = False   # set by train_step / _dp_steps around forward(): backward follows at once
        self.fork_after = int(cfg.get("ark_fork_after", 1))   # measured: 0 -> 1.281, 1 -> 1.266, 2 -> 1.296 ms/step
        self.overlap_wgrad = bool(cfg.get("ark_overlap_wgrad", True))
        self.fused_latent = True
        self.fused_prologue = True   # encoder pool + token gather as one launch
        self._x0_ready = False
        # engine of the encoder's 16-bit products (csrc/gemm16.hip): 0 = the library's choice, 1 = shared ring, 2 = wave-private
        # K-slices; forward and input-gradient products separately (the latter run beside the weight-gradient launch)
        self.g16_fwd = int(cfg.get("ark_g16_fwd", 0))
        # measured (same box, alternating): forward on the wave-private engine 1.103 -> 1.093 ms per step; the input
        # gradients on it 1.093 -> 1.110 -- its workgroups want a CU's whole LDS and queue behind the weight-gradient
        # launch's two 64-KB workgroups per CU, where the ring's 24-KB workgroups slip in beside them
        self.g16_bwd = int(cfg.get("ark_g16_bwd", 1))
        # round-5 schedule choices, measured and fixed (no switches): the embedding scatter in front of the MLP
        # weight gradients (wd-articles 6.96 -> 6.70 ms, syn-paths +-0); a deferred fused-CE weight gradient behind the whole
        # encoder backward (wd-movies 1.87 -> 1.84); sweep workloads whose side queue is the long one: GRU weight gradients on a
        # third queue, the latent reductions and the MLP block's Adam on the chain's own queue (wd-articles -0.2 ms)
        self.pool_bwd_early = True
        self.ce_dw_last = True
        self.wgrad_third_queue = True
        self.early_dec_adam = True
        self.early_mlp_adam = True
        self._adam_dec_early = False
        self._adam_dec_done = False
        self._adam_mlp_done = self._adam_emb_done = False
        self._ticked = None
        self.emb_gemm = True
        # large vocabularies: tied projection fused with the cross-entropy -- [B*L, V] logits / dlogits never exist
        # (csrc/vocab_ce.hip); small ones keep the three short launches (W_tok is a few KB there)
        self.fused_ce = bool(cfg.get("ark_fused_ce", self.use_dma and self.V >= 2048 and self.D in (64, 128, 256, 512)))
        self._fused_ce_step = False

    def _noise_seed(self):
        """seed of this rank's latent-noise stream (the draw counter lives in the device scalar array)"""
        return (self.noise_seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * (self.rank + 1)) & 0xFFFFFFFFFFFFFFFF

    def noise_draws(self):
        """latent-noise draws so far (device uint32; synchronises)"""
        return int(self.hyper.view(torch.int32)[HP["NOISE_STEP"]].item()) & 0xFFFFFFFF

    def set_noise_draws(self, n):
        self.hyper.view(torch.int32)[HP["NOISE_STEP"]:HP["NOISE_STEP"] + 1].fill_(int(n) - (1 << 32) if int(n) >= (1 << 31) else int(n))

    def _zero(self, t):
        """zero a contiguous device tensor on the current queue (library kernel)"""
        _call("ark_zero", L.ptr(t), L.i64(t.numel() * t.element_size()), L.cur_stream())

    def _layer_seed(self, l):
        """dropout stream of layer l's output on this rank (hashed with the draw counter and the element index)"""
        return (self.drop_seed + 7919 * l + 104729 * self.rank) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ parameters
    def load_params(self, named):
        self.dp_flush()
        with torch.no_grad():
            for k, v in named.items():
                if k == "dec.out.weight" and self.layout.tied:
                    continue
                self.p[k].copy_(v.to(self.device, dtype=torch.float32))
        self._shadow_ok = False

    def mark_params_dirty(self):
        """call after anything other than Engine.adam() changed the parameters"""
        self.dp_flush()
        self._shadow_ok = False

    def _build_shadow_jobs(self):
        """job table of ark_weight_shadows: every GEMM weight -> plain (forward type) / transposed (backward type) copy"""
        import ctypes
        jobs = []
        for l in range(self.n):
            jobs.append((self.p[f"dec.gru.weight_ih_l{l}"], self.wih16[l], self.wihT16[l], 3 * self.D, self.D))
            jobs.append((self.p[f"dec.gru.weight_hh_l{l}"], self.whh16[l], self.whhT16[l], 3 * self.D, self.D))
        jobs.append((self.p["dec.out.weight"], self.wtok16, self.wtokT16, self.V, self.D, self.Vp))
        if self.mt == "SAIL":
            H = 3 * self.D
            for i in range(self.n):
                jobs.append((self.p[f"enc.mlp.{2 * i}.weight"], self.wm16[i], self.wmT16[i], H, H))
            jobs.append((self.p["enc.mu.weight"], self.wh16, None, 2 * self.Z, H))
        chunks = []
        for c0 in range(0, len(jobs), 12):
            ch = jobs[c0:c0 + 12]
            n = len(ch)
            src = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in ch])
            dst = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in ch])
            dstT = (ctypes.c_void_p * n)(*[(j[2].data_ptr() if j[2] is not None else 0) for j in ch])
            R = (ctypes.c_int * n)(*[j[3] for j in ch])
            C = (ctypes.c_int * n)(*[j[4] for j in ch])
            pf = (ctypes.c_int * n)(*[self.prec_fwd] * n)
            pb = (ctypes.c_int * n)(*[self.prec_bwd] * n)
            ldT = (ctypes.c_int * n)(*[(j[5] if len(j) > 5 else 0) for j in ch])
            chunks.append((n, src, dst, dstT, R, C, pf, pb, ldT))
        return chunks

    def refresh_shadows(self):
        """16-bit shadows of every GEMM weight from the fp32 parameters (after load_params / an external optimiser;
        Engine.adam() writes them itself, from the registers that hold the updated weights)"""
        if self.use_dma:
            for (n, src, dst, dstT, R, C, pf, pb, ldT) in self._shadow_jobs:
                _call("ark_weight_shadows", L.i32(n), src, dst, dstT, R, C, pf, pb, ldT, L.cur_stream())
            self._refresh_xtab()
        self._shadow_ok = True

    def _refresh_xtab(self):
        """x_tab = W_tok16 W_ih0_16^T from the current shadows (one small product; queued right behind whatever wrote them)"""
        if self.use_dma and self.xtab is not None:
            D = self.D
            _call("ark_gemm16", L.i32(self.prec_fwd), L.i32(L.EPI_NONE), L.ptr(self.wtok16), L.i64(D), L.ptr(self.wih16[0]), L.i64(D),
                  L.ptr(self.xtab), L.i64(3 * D), L.ptr(None), L.ptr(None), L.i32(self.V), L.i32(3 * D), L.i32(D), L.i32(0),
                  L.cur_stream())

    def _build_adam_jobs(self, *ranges):
        """job tables of ark_adam_step_shadows for the flat ranges [lo, hi), ... : one MATRIX job per shadowed weight (its
        16-bit shadows are written from the updated values), LINEAR jobs for everything between them; all ranges share
        launches (ARK_ADAM_MAX_JOBS jobs each)"""
        import ctypes
        if len(ranges) == 2 and not isinstance(ranges[0], (tuple, list)):
            ranges = (ranges,)
        jobs = []
        for lo, hi in ranges:
            jobs += self._adam_job_list(lo, hi)
        chunks = []
        for c0 in range(0, len(jobs), L.ADAM_MAX_JOBS):
            ch = jobs[c0:c0 + L.ADAM_MAX_JOBS]
            n = len(ch)
            chunks.append((n, (ctypes.c_int64 * n)(*[j[0] for j in ch]), (ctypes.c_int * n)(*[j[1] for j in ch]),
                           (ctypes.c_int * n)(*[j[2] for j in ch]),
                           (ctypes.c_void_p * n)(*[(j[3].data_ptr() if j[3] is not None else 0) for j in ch]),
                           (ctypes.c_void_p * n)(*[(j[4].data_ptr() if j[4] is not None else 0) for j in ch]),
                           (ctypes.c_int * n)(*[self.prec_fwd] * n), (ctypes.c_int * n)(*[self.prec_bwd] * n),
                           (ctypes.c_int * n)(*[j[5] for j in ch])))
        return chunks

    def _adam_job_list(self, lo, hi):
        shadows = {}   # parameter name -> (dst, dstT, ldT)
        for l in range(self.n):
            shadows[f"dec.gru.weight_ih_l{l}"] = (self.wih16[l], self.wihT16[l], 0)
            shadows[f"dec.gru.weight_hh_l{l}"] = (self.whh16[l], self.whhT16[l], 0)
        tokname = "dec.tok_emb.weight" if self.layout.tied else "dec.out.weight"
        shadows[tokname] = (self.wtok16, self.wtokT16, self.Vp)
        if self.mt == "SAIL":
            for i in range(self.n):
                shadows[f"enc.mlp.{2 * i}.weight"] = (self.wm16[i], self.wmT16[i], 0)
        jobs, cur = [], lo
        def linear(a, b):
            if b > a:
                jobs.append((a, 0, b - a, None, None, 0))
        ents = list(self.layout.entries.items())
        k = 0
        while k < len(ents):
            name, (off, shape, numel) = ents[k]
            mat = None
            if name in shadows and len(shape) == 2 and shape[1] % 4 == 0:
                mat = (off, shape[0], shape[1]) + shadows[name]
            elif name == "enc.mu.weight" and self.mt == "SAIL" and shape[1] % 4 == 0:
                mat = (off, 2 * self.Z, shape[1], self.wh16, None, 0)   # [mu; logv] are adjacent: one [2Z, 3D] matrix
                k += 1                                                    # (skips enc.logv.weight)
            if mat is not None and lo <= mat[0] and mat[0] + mat[1] * mat[2] <= hi:
                linear(cur, mat[0])
                jobs.append(mat)
                cur = mat[0] + mat[1] * mat[2]
            k += 1
        linear(cur, hi)
        return jobs

    def _adam_launch(self, which, g16=False):
        """g16: the gradient comes from the bf16 transport buffer of the data-parallel buckets (no widening pass)"""
        st = L.cur_stream()
        if self.use_dma:
            for (n, off, R, C, dst, dstT, pf, pb, ldT) in self._adam_jobs[which]:
                if g16:
                    _call("ark_adam_step_shadows_g16", L.ptr(self.P), L.ptr(self._Gh), L.ptr(self.M), L.ptr(self.Vv), L.i32(n), off,
                          R, C, dst, dstT, pf, pb, ldT, L.ptr(self.hyper), st)
                    continue
                _call("ark_adam_step_shadows", L.ptr(self.P), L.ptr(self.G), L.ptr(self.M), L.ptr(self.Vv), L.i32(n), off, R, C,
                      dst, dstT, pf, pb, ldT, L.ptr(self.hyper), st)
            if which in ("all", "dec"):   # (the bucket that holds W_tok and W_ih0)
                self._refresh_xtab()
        else:
            lo, hi = {"all": (0, self.layout.total), "enc": (0, self.layout.dec_grad_offset),
                      "dec": (self.layout.dec_grad_offset, self.layout.total)}[which]
            _call("ark_adam_step", L.ptr(self.P[lo:]), L.ptr(self.G[lo:]), L.ptr(self.M[lo:]), L.ptr(self.Vv[lo:]), L.i64(hi - lo),
                  L.ptr(self.hyper), st)

    def set_hyper(self, lr=None, beta=None, kl_norm=None, ce_count=None, grad_scale=None):
        """update device-resident step scalars (tiny async fills, only when a value changes)"""
        for key, val in (("LR", lr), ("BETA", beta), ("KL_NORM", kl_norm), ("GRAD_SCALE", grad_scale)):
            if val is not None and self._hp.get(key) != float(val):
                self.dp_flush()   # (a decoder-bucket update still on the update queue reads these scalars)
                self._hp[key] = float(val)
                self.hyper[HP[key]:HP[key] + 1].fill_(float(val))
        if ce_count is not None and self._hp.get("CE_COUNT") != float(ce_count):
            self._hp["CE_COUNT"] = float(ce_count)
            self.hyper[HP["CE_COUNT"]:HP["CE_COUNT"] + 1].fill_(float(ce_count))
            self.hyper[HP["CE_INV_COUNT"]:HP["CE_INV_COUNT"] + 1].fill_(1.0 / float(ce_count) if ce_count > 0 else 0.0)

    def reset_optimizer(self):
        self.M.zero_()
        self.Vv.zero_()
        self.adam_steps = 0
        self.hyper[HP["ADAM_STEP"]:HP["ADAM_BC2"] + 1].zero_()

    def set_optimizer_step(self, t):
        """resume: optimiser step count and its bias corrections (what ark_adam_tick leaves behind after t steps)"""
        self.adam_steps = int(t)
        self.hyper[HP["ADAM_STEP"]:HP["ADAM_BC2"] + 1].copy_(torch.tensor(
            [float(t), 1.0 - 0.9 ** t, 1.0 - 0.999 ** t], dtype=torch.float32))

    def dropout_draws(self):
        """number of training forwards so far = the dropout draw counter (device uint32; synchronises)"""
        return int(self.hyper.view(torch.int32)[HP["DROP_STEP"]].item()) & 0xFFFFFFFF

    def set_dropout_draws(self, n):
        self.hyper.view(torch.int32)[HP["DROP_STEP"]:HP["DROP_STEP"] + 1].fill_(int(n) - (1 << 32) if int(n) >= (1 << 31) else int(n))

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, T):
        key = B
        if self.ws_key == key:
            return self.ws
        if key in self._ws_cache:   # captured graphs hold raw workspace addresses: never free a live one
            self.ws, self.ws_key = self._ws_cache[key], key
            return self.ws
        dev, D, n, Lq, V, Z = self.device, self.D, self.n, self.L, self.V, self.Z
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        w = {}
        R = Lq * B
        if self.mt == "SAIL":
            H = 3 * D
            w["g"] = f(B, H)
            w["inv_cnt"] = f(B)
            w["pre"] = [f(B, H) for _ in range(n)]
            w["act"] = [f(B, H) for _ in range(n)]
            w["head"] = f(B, 2 * Z)
            z0 = lambda *sh: torch.zeros(*sh, device=dev)   # (rows of a padded batch are never written: they must read 0)
            w["mu"], w["logv"], w["z"] = z0(B, Z), z0(B, Z), z0(B, Z)
            w["kl"] = torch.zeros(1, device=dev)
            w["kl_rows"] = torch.zeros(B, device=dev)
            w["eps0"] = torch.zeros(B, Z, device=dev)
            w["dz"], w["dhead"] = z0(B, Z), z0(B, 2 * Z)
            w["dH0"] = f(B, D)
            w["dA"], w["dB"] = z0(B, H), z0(B, H)
        # fast path: 16-bit operands + layer-diagonal GRU kernels.  (Batches that are not a multiple of 16 rows run
        # the register-staged kernels: same results, several times slower -- pad the batch instead.)
        v2 = self.use_dma and B % 16 == 0
        w["v2"] = v2
        i16 = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.int16)
        two = self.prec_fwd != self.prec_bwd   # separate backward-type copies of the activations
        if v2 and self.mt == "SAIL":
            H = 3 * D
            # rows padded to a multiple of 64 and zero-initialised: the grouped weight-gradient launch reduces over whole
            # 64-row stages, and rows >= B are never written (they add exact zeros)
            Bk = _rup(B, 64)
            z16 = lambda: torch.zeros(Bk, H, device=dev, dtype=torch.int16)
            w["g16a"], w["g16b"] = z16(), (z16() if two else None)
            w["act16a"] = [z16() for _ in range(n)]
            w["act16b"] = [z16() if two else None for _ in range(n)]
            w["dpre16"] = [z16() for _ in range(n)]
        if v2:
            # "16": row-major 16-bit copies (a: forward type, b: backward type); "_t": tile-native
            # the weight-gradient launch reduces over whole 64-row stages: its operands carry zeroed rows up to the next
            # multiple of 64 (rows >= R of the gate-gradient panels are never written, so whatever the other operand holds
            # there is multiplied by exact zeros -- and it is finite: zero-initialised, written by real cells only)
            Rp = _rup(R, 64)
            z16r = lambda rows, cols: torch.zeros(rows, cols, device=dev, dtype=torch.int16)
            w["X0a"] = z16r(Rp, D)
            w["X0b"] = z16r(Rp, D) if two else None
            w["Y"] = [torch.zeros((Lq + 1) * B, D, device=dev) for _ in range(n)]   # tile-native fp32 state
            w["Y16a"] = [z16r((Lq + 1) * B + 64, D) for _ in range(n)]
            w["Y16b"] = [z16r((Lq + 1) * B + 64, D) if two else None for _ in range(n)]
            for nm in ("SR", "SZ", "SN", "SHN"):
                w[nm] = [i16(R, D) for _ in range(n)]                        # tile-native fp16 saves
            w["dG16"] = [z16r(Rp, 4 * D) for _ in range(n)]                  # gate-gradient panels [dr | dz | dn | dn*r]
            if self.fused_ce:
                w["lse"] = f(R)
            else:
                w["dlog16"] = torch.zeros(R, self.Vp, device=dev, dtype=torch.int16)   # K-padded 16-bit dlogits
            # small vocabularies: token-embedding gradient and layer 0's input weight gradient from the per-token sums
            # of layer 0's gate-gradient panel instead of input-gradient GEMM + scatter + a [3D,D]x[B*L] weight-gradient
            # product (SAIL; ARK also needs dX0 for the position embedding)
            w["emb_gemm"] = bool(self.mt == "SAIL" and self.Vp <= 256 and self.emb_gemm)
            if w["emb_gemm"]:
                w["S_tok"] = f(self.Vp, 3 * D)
                w["tok_scratch"] = torch.empty(4 * R + 32 + 4 * (R // 64 + 1), device=dev, dtype=torch.uint8)
            w["xtab"] = bool(self.xtab is not None)
            if w["xtab"]:
                w["tok_tm"] = torch.zeros(Rp, device=dev, dtype=torch.int32)  # time-major token ids (rows of x_tab)
            w["h0"] = f(B, D)                                                # row-major h0 (z-projection)
            w["dX0"] = f(R, D)
            if self.p_drop > 0:   # the dropout mask itself is regenerated in-kernel from a counter hash
                w["Yd16a"] = [z16r(Rp, D) for _ in range(n - 1)]
                w["Yd16b"] = [z16r(Rp, D) if two else None for _ in range(n - 1)]
            w["carry_l"] = [f(B, D) for _ in range(n)]   # the diagonal BPTT keeps one carry per layer
        else:
            w["X0"] = f(R, D)
            w["GI"] = [f(R, 3 * D) for _ in range(n)]       # gi forward, overwritten by dgi in backward
            w["dGH"] = [f(R, 3 * D) for _ in range(n)]
            w["Y"] = [torch.zeros((Lq + 1) * B, D, device=dev) for _ in range(n)]  # slot 0 = h0
            for nm in ("SR", "SZ", "SN", "SHN"):
                w[nm] = [f(R, D) for _ in range(n)]
            if self.p_drop > 0:
                w["mask"] = [f(R, D) for _ in range(n - 1)]
                w["Ydrop"] = [f(R, D) for _ in range(n - 1)]
        self.ldl = _rup(V, 4)
        if not (v2 and self.fused_ce):   # (the fused path allocates them on first use: reference-style loop, beam search)
            w["logits"] = torch.zeros(R, self.ldl, device=dev)
        w["_R"] = R
        w["row_loss"] = f(R)
        w["out4"] = torch.zeros(4, device=dev)
        w["dYa"], w["dYb"] = f(R, D), f(R, D)
        w["carry"] = f(B, D)
        w["tok_next"] = torch.zeros(B, dtype=torch.int64, device=dev)
        if len(self._ws_cache) >= 6:   # bound the cache; batch sizes with captured graphs stay pinned
            for k in [k for k in self._ws_cache if k not in self._pinned_B][:1]:
                del self._ws_cache[k]
        self._ws_cache[key] = w
        self.ws, self.ws_key = w, key
        return w

    def _logits(self, w):
        """the materialised [B*L, V] logits buffer (and its 16-bit gradient copy on the fast path), created on demand"""
        if "logits" not in w:
            w["logits"] = torch.zeros(w["_R"], self.ldl, device=self.device)
        if w["v2"] and "dlog16" not in w:
            w["dlog16"] = torch.zeros(w["_R"], self.Vp, device=self.device, dtype=torch.int16)
        return w["logits"]

    # ------------------------------------------------------------------ kernel wrappers
    def _gemm(self, a_lay, b_lay, epi, A, lda, Bm, ldb, C, ldc, M, N, K, C2=None, bias=None, aux=None, acc=0):
        _call("ark_gemm", L.i32(self.prec), L.i32(a_lay), L.i32(b_lay), L.i32(epi), L.ptr(A), L.i64(lda), L.ptr(Bm),
              L.i64(ldb), L.ptr(C), L.i64(ldc), L.ptr(C2), L.ptr(bias), L.ptr(aux), L.i32(M), L.i32(N), L.i32(K),
              L.i32(acc), L.cur_stream())

    def _colsum(self, X, ld, out, M, N, n_batch=1, bs_in=0, bs_out=0):
        # gradients accumulate into G, which backward() zeroes once up front
        _call("ark_colsum", L.ptr(X), L.i64(ld), L.i64(bs_in), L.ptr(out), L.i64(bs_out if n_batch > 1 else N), L.i32(M),
              L.i32(N), L.i32(n_batch), L.i32(1), L.cur_stream())

    # ------------------------------------------------------------------ forward
    def forward(self, triples, seq, eps=None, with_loss=True, with_dlogits=True, L_run=None, ce_count=None):
        """Launch the forward pass.  triples [B,T,3] int64 (SAIL), seq [B,seq_len] int64 (device).
        Results stay on the device in the workspace (out4 = loss, ce, kl, token-loss sum)."""
        for _ in self._forward_steps(triples, seq, eps, with_loss, with_dlogits, L_run, ce_count):
            self.dp_flush()   # a decoder-bucket update still owed by a data-parallel step lands before the decoder runs
        return self.ws

    def _forward_steps(self, triples, seq, eps=None, with_loss=True, with_dlogits=True, L_run=None, ce_count=None):
        """forward() as a generator with ONE seam: it yields after the encoder half (encoder MLP, heads, latent,
        z-projection: parameters of the gradient bucket [0, dec_grad_offset)) and before the decoder half (GRU,
        vocabulary projection: the other bucket).  The pipelined data-parallel step finishes the previous
        step's decoder-bucket all-reduce + Adam in that seam."""
        self.prec = self.prec_fwd
        self._dlog16_valid = False
        self.fwd_gen += 1
        if not self._shadow_ok:
            self.refresh_shadows()
        B = seq.shape[0]
        nv = B
        if with_loss and self.use_dma and B % 16 != 0:
            # the layer-diagonal kernels work on 16-row tiles: a ragged batch (the last one of an epoch) is padded with
            # all-PAD rows -- no target, no KL term (the latent kernels run on the real rows only), so every gradient they
            # contribute is exactly zero -- instead of dropping to the register-staged path
            triples, seq, eps = self._pad_batch(triples, seq, eps)
            B = seq.shape[0]
        self._n_valid = nv
        T = triples.shape[1] if (self.mt == "SAIL" and triples is not None) else 0
        w = self._workspace(B, T)
        D, n, V, Z = self.D, self.n, self.V, self.Z
        Lq = self.L if L_run is None else L_run
        R = Lq * B
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p = self.p
        assert seq.dtype == torch.int64 and seq.is_contiguous() and seq.device == self.device
        ld_seq = seq.shape[1]
        self._seq, self._triples, self._B, self._T, self._Lrun = seq, triples, B, T, Lq
        use_drop = self.training and self.p_drop > 0

        if self.mt == "SAIL":
            assert triples.dtype == torch.int64 and triples.is_contiguous() and triples.device == self.device
            H = 3 * D
            if w["v2"]:
                pf, pb = self.prec_fwd, self.prec_bwd
                skip = self._skip_x0(w, B, Lq)   # (small vocabulary, diagonal path: nobody reads the embedding rows)
                if self.fused_prologue and (skip or (self._dp_pending is None and not self._dp_mode)):
                    # the decoder's token gather shares the encoder pool's launch (both only need the batch indices); not
                    # in a pipelined data-parallel step, whose update of the token table lands at the seam (unless only the
                    # token ids are written: `skip`) -- a captured first segment must not read W_tok either
                    _call("ark_pool_gather_fwd16", L.ptr(triples), L.ptr(p["enc.e_emb.weight"]), L.ptr(p["enc.r_emb.weight"]),
                          L.ptr(w["g"]), L.ptr(w["inv_cnt"]), L.ptr(w["g16a"]), L.i32(pf), L.ptr(w["g16b"]), L.i32(pb), L.i32(B),
                          L.i32(T), L.i32(D), L.i64(-1 if self.pad_rid is None else self.pad_rid), L.ptr(seq), L.i64(ld_seq),
                          L.ptr(p["dec.tok_emb.weight"]), L.ptr(None if skip else w["X0a"]), L.ptr(None if skip else w["X0b"]),
                          L.i32(Lq), L.i32(D), L.ptr(w["tok_tm"] if w["xtab"] else None), L.ptr(self.hyper if use_drop else None), st)
                    self._x0_ready = True
                elif T >= 32:
                    # (pipelined data parallel, large vocabulary: the gather of W_tok rows waits for the seam.)  Long graphs in a
                    # small batch still want the pool of the fused launch -- a graph spread over column blocks and its triples
                    # over waves: 20 us where one workgroup per graph takes 230 (wd-articles, T = 212) -- so that launch is
                    # issued with its gather half reduced to writing the time-major token ids (it does not read W_tok then)
                    if "tok_ids_scratch" not in w:
                        w["tok_ids_scratch"] = torch.empty(B * Lq, device=self.device, dtype=torch.int32)
                    _call("ark_pool_gather_fwd16", L.ptr(triples), L.ptr(p["enc.e_emb.weight"]), L.ptr(p["enc.r_emb.weight"]),
                          L.ptr(w["g"]), L.ptr(w["inv_cnt"]), L.ptr(w["g16a"]), L.i32(pf), L.ptr(w["g16b"]), L.i32(pb), L.i32(B),
                          L.i32(T), L.i32(D), L.i64(-1 if self.pad_rid is None else self.pad_rid), L.ptr(seq), L.i64(ld_seq),
                          L.ptr(p["dec.tok_emb.weight"]), L.ptr(None), L.ptr(None), L.i32(Lq), L.i32(D),
                          L.ptr(w["tok_tm"] if w["xtab"] else w["tok_ids_scratch"]), L.ptr(None), st)
                else:
                    _call("ark_enc_pool_fwd16", L.ptr(triples), L.ptr(p["enc.e_emb.weight"]), L.ptr(p["enc.r_emb.weight"]),
                          L.ptr(w["g"]), L.ptr(w["inv_cnt"]), L.ptr(w["g16a"]), L.i32(pf), L.ptr(w["g16b"]), L.i32(pb), L.i32(B),
                          L.i32(T), L.i32(D), L.i64(-1 if self.pad_rid is None else self.pad_rid), st)
            else:
                _call("ark_enc_pool_fwd", L.ptr(triples), L.ptr(p["enc.e_emb.weight"]), L.ptr(p["enc.r_emb.weight"]),
                      L.ptr(w["g"]), L.ptr(w["inv_cnt"]), L.i32(B), L.i32(T), L.i32(D),
                      L.i64(-1 if self.pad_rid is None else self.pad_rid), st)
            if w["v2"]:
                a16 = w["g16a"]
                for i in range(n):
                    _call("ark_gemm16_engine", L.i32(self.g16_fwd), L.i32(pf), L.i32(L.EPI_BIAS_GELU), L.ptr(a16), L.i64(H), L.ptr(self.wm16[i]), L.i64(H),
                          L.ptr(w["pre"][i]), L.i64(H), L.ptr(p[f"enc.mlp.{2 * i}.bias"]), L.ptr(None), L.ptr(w["act16a"][i]),
                          L.ptr(w["act16b"][i]), L.i32(pb), L.ptr(None), L.i32(B), L.i32(H), L.i32(H), st)
                    a16 = w["act16a"][i]
                # (the heads: a handful of tiles over K = 3D -- the K-split flavour of the wave-private engine where it applies)
                _call("ark_gemm16_engine", L.i32(3 if self.g16_fwd == 0 else self.g16_fwd), L.i32(pf), L.i32(L.EPI_BIAS), L.ptr(a16),
                      L.i64(H), L.ptr(self.wh16), L.i64(H), L.ptr(w["head"]), L.i64(2 * Z), L.ptr(p["enc.mu.bias"]), L.ptr(None),
                      L.ptr(None), L.ptr(None), L.i32(pf), L.ptr(None), L.i32(B), L.i32(2 * Z), L.i32(H), st)
            else:
                a = w["g"]
                for i in range(n):
                    self._gemm(KM, KM, L.EPI_BIAS_GELU, a, H, p[f"enc.mlp.{2 * i}.weight"], H, w["pre"][i], H, B, H, H,
                               C2=w["act"][i], bias=p[f"enc.mlp.{2 * i}.bias"])
                    a = w["act"][i]
                self._gemm(KM, KM, L.EPI_BIAS, a, H, p["enc.mu.weight"], H, w["head"], 2 * Z, B, 2 * Z, H,
                           bias=p["enc.mu.bias"])
            if eps is None:   # z = mu + N(0,1) * std, drawn on the device (reference models.py:63) by the library's own
                eps = w["eps0"]   # counter-based generator: fresh every launch / replay, a stream of its own per rank
                _call("ark_normal_fill", L.ptr(eps), L.i64(eps.numel()), L.u64(self._noise_seed()), L.ptr(self.hyper), st)
            self._eps = eps
            self._kl_rows = bool(w["v2"] and Z <= 128 and self.fused_latent)
            if self._kl_rows:
                # reparameterisation + z-projection as ONE launch; per-row KL terms, summed by the loss finalisation
                import ctypes
                yt = (ctypes.c_void_p * n)(*[w["Y"][l].data_ptr() for l in range(n)])
                ya = (ctypes.c_void_p * n)(*[w["Y16a"][l].data_ptr() for l in range(n)])
                yb = (ctypes.c_void_p * n)(*[(w["Y16b"][l].data_ptr() if w["Y16b"][l] is not None else 0) for l in range(n)])
                _call("ark_latent_zproj_fwd", L.i32(self.prec_fwd), L.i32(self.prec_bwd), L.ptr(w["head"]), L.ptr(eps), L.ptr(w["mu"]),
                      L.ptr(w["logv"]), L.ptr(w["z"]), L.ptr(w["kl_rows"]), L.ptr(p["dec.z_proj.weight"]), L.ptr(p["dec.z_proj.bias"]),
                      L.ptr(w["h0"]), L.i32(n), yt, ya, yb, L.i32(B), L.i32(nv), L.i32(Z), L.i32(D), st)
            else:
                _call("ark_latent_fwd", L.ptr(w["head"]), L.ptr(eps), L.ptr(w["mu"]), L.ptr(w["logv"]), L.ptr(w["z"]),
                      L.ptr(w["kl"]), L.i32(nv), L.i32(Z), st)   # (padding rows: z stays 0, no KL term)
                self._decode_h0(w, w["z"], B)
        else:
            for l in range(n):
                self._zero(w["Y"][l][:B])
                if w["v2"]:
                    self._zero(w["Y16a"][l][:B])
                    if w["Y16b"][l] is not None:
                        self._zero(w["Y16b"][l][:B])
        yield
        st = L.cur_stream()
        fused = bool(w["v2"] and self.fused_ce and with_loss)
        self._fused_ce_step = fused and with_dlogits
        chunks = self._ce_chunks(B, Lq) if (fused and with_dlogits) else None
        if chunks:
            # long sequences: the forward sweep in a few time chunks, the fused CE of a chunk on the side queue beside the
            # sweep of the next one (the sweep leaves 160 of the 256 CUs idle; the CE of all rows only started after it)
            self._decoder_prologue(w, seq, ld_seq, B, Lq, use_drop)
            if ce_count is None:
                _call("ark_count_targets", L.ptr(seq), L.i64(ld_seq), L.i32(B), L.i32(Lq), L.ptr(self.hyper), st)
                self._hp.pop("CE_COUNT", None)
            main = torch.cuda.current_stream()
            side = self._side_stream()
            # Queue order matters (round-3 trace: CE(k) was queued first, its one-per-CU workgroups took the whole chip and the
            # next chunk's sweep sat behind it -- 3 % overlap): sweep(k + 1) goes out BEFORE CE(k), so its co-resident
            # workgroups are on their CUs when the CE grid arrives, and the CE grid is capped at the CUs the sweep leaves
            # (`cu_budget`: a 512-thread CE workgroup owns a CU's whole register file), so neither order can starve the sweep.
            # (co-located ranks -- the two-process tests -- each hold a sweep and a capped CE grid: split what is left)
            free_cus = max(32, self._cu_count() // self.ranks_per_device - self._sweep_wgs(B))
            self._check_beside_sweep(side_wgs=free_cus)
            done = []
            for k, (t0, t1) in enumerate(chunks):
                self._sweep_fwd(w, B, Lq, use_drop, True, t0, t1)
                ev = torch.cuda.Event()
                ev.record(main)
                done.append(ev)
                if k > 0:
                    side.wait_event(done[k - 1])
                    with torch.cuda.stream(side):
                        self._fused_ce_fwd(w, seq, ld_seq, B, chunks[k - 1][0], chunks[k - 1][1], True, cu_budget=free_cus)
            side.wait_event(done[-1])
            with torch.cuda.stream(side):   # (the last chunk's CE has the chip to itself)
                self._fused_ce_fwd(w, seq, ld_seq, B, chunks[-1][0], chunks[-1][1], True)
            main.wait_stream(side)
        else:
            self._decoder_forward(w, seq, ld_seq, B, Lq, use_drop, project=not fused)
            if with_loss:
                if ce_count is None:
                    _call("ark_count_targets", L.ptr(seq), L.i64(ld_seq), L.i32(B), L.i32(Lq), L.ptr(self.hyper), st)
                    self._hp.pop("CE_COUNT", None)   # the device slot no longer holds what set_hyper last wrote
        if with_loss and fused:
            # projection + cross-entropy (+ dY of the top layer, consumed by the backward diagonals) in one sweep over V
            if not chunks:
                self._fused_ce_fwd(w, seq, ld_seq, B, 0, Lq, with_dlogits)
            self._finalize = self._make_finalize(w, R, B)
            if not (self._defer_finalize and with_dlogits):
                self._finalize()
                self._finalize = None
        elif with_loss:
            self._logits(w)
            d16 = w["dlog16"] if (with_dlogits and w["v2"]) else None
            # the diagonal backward takes every consumer of dlogits (dY, dW_tok, db_out) from the 16-bit copy:
            # the fp32 gradient is then never written
            only16 = d16 is not None and R % 64 == 0 and self.dlog16_only
            _call("ark_ce_fwd_bwd", L.ptr(w["logits"]), L.i64(self.ldl), L.ptr(seq), L.i64(ld_seq), L.ptr(self.hyper),
                  L.ptr(w["row_loss"]), L.ptr(w["logits"] if (with_dlogits and not only16) else None), L.ptr(d16),
                  L.i32(self.prec_bwd if d16 is not None else 0), L.i64(self.Vp if d16 is not None else 0), L.i32(B), L.i32(Lq),
                  L.i32(V), st)
            self._dlog16_valid = d16 is not None
            self._dlog16_only = bool(only16)
            self._finalize = self._make_finalize(w, R, B)
            if not (self._defer_finalize and with_dlogits and w["v2"]):
                self._finalize()
                self._finalize = None
            # (else: the two-queue backward launches it on its side queue -- the loss scalars are not on the
            #  dependent chain)

    def _ce_chunks(self, B, Lq):
        """[(t0, t1), ...] time chunks of a chunked forward sweep + CE pipeline, or None (one launch each)"""
        nc = self.ce_chunks
        if nc <= 1 or not self.overlap_wgrad or not self._use_sweep(B, Lq) or Lq < 64 * nc:
            return None
        if self._sweep_wgs(B) > 128:   # the sweep has to leave CUs for the CE that runs beside it
            return None
        step = -(-Lq // nc)
        return [(t, min(Lq, t + step)) for t in range(0, Lq, step)]

    def _fused_ce_fwd(self, w, seq, ld_seq, B, t0, t1, with_dlogits, cu_budget=0):
        """tied projection + cross-entropy (+ dY of the top layer) of the rows of steps [t0, t1) in one sweep over V;
        cu_budget: the CUs the launch may fill (beside a persistent sweep: the ones it leaves free; 0 = all)"""
        D, n, V, p = self.D, self.n, self.V, self.p
        st = L.cur_stream()
        Lc, o = t1 - t0, t0 * B
        Rc = Lc * B
        y = w["Y16a"][n - 1][B + o:]
        sq = seq[:, t0:]   # (the kernels index seq[b * ld + t + 1] with t counted from the chunk's first step)
        dY = w["dYa"][o:] if with_dlogits else None
        nv = L.lib().ark_vocab_ce_fwd_splits(L.i32(Rc), L.i32(V), L.i32(D), L.i32(cu_budget)) if with_dlogits else 1
        if nv > 1:   # few rows, wide model: the vocabulary sweep is split over workgroups (+ one merging launch)
            # (one scratch per chunk position: the CE launches of consecutive chunks may overlap on the side queue's tail)
            key = ("ce_ws", t0)
            if key not in w or w[key].numel() < nv * (Rc * D + 4 * Rc):
                w[key] = torch.empty(nv * (Rc * D + 4 * Rc), device=self.device)
            _call("ark_vocab_ce_fwd_ws", L.i32(self.prec_fwd), L.ptr(y), L.ptr(self.wtok16), L.ptr(p["dec.out.bias"]), L.ptr(sq),
                  L.i64(ld_seq), L.ptr(self.hyper), L.ptr(w["row_loss"][o:]), L.ptr(w["lse"][o:]), L.ptr(dY), L.ptr(w[key]),
                  L.i64(w[key].numel()), L.i32(B), L.i32(Lc), L.i32(V), L.i32(D), L.i32(cu_budget), st)
        else:
            _call("ark_vocab_ce_fwd", L.i32(self.prec_fwd), L.ptr(y), L.ptr(self.wtok16), L.ptr(p["dec.out.bias"]), L.ptr(sq),
                  L.i64(ld_seq), L.ptr(self.hyper), L.ptr(w["row_loss"][o:]), L.ptr(w["lse"][o:]), L.ptr(dY), L.i32(B), L.i32(Lc),
                  L.i32(V), L.i32(D), st)

    def _make_finalize(self, w, R, B):
        """the launch that turns per-row losses (and the KL terms) into out4 = [loss, ce, kl, token-loss sum]"""
        if self.mt == "SAIL" and getattr(self, "_kl_rows", False):
            nv = self._n_valid
            scale = -0.5 / (nv * self.Z)
            return lambda: _call("ark_loss_finalize_rows", L.ptr(w["row_loss"]), L.i32(R), L.ptr(w["kl_rows"]), L.i32(nv),
                                 L.f32(scale), L.ptr(self.hyper), L.ptr(w["out4"]), L.cur_stream())
        return lambda: _call("ark_loss_finalize", L.ptr(w["row_loss"]), L.i32(R), L.ptr(w["kl"] if self.mt == "SAIL" else None),
                             L.ptr(self.hyper), L.ptr(w["out4"]), L.cur_stream())

    def _decode_h0(self, w, z, B):
        p = self.p
        st = L.cur_stream()
        zp = (L.ptr(z), L.ptr(p["dec.z_proj.weight"]), L.ptr(p["dec.z_proj.bias"]))
        if w["v2"]:
            import ctypes
            n = self.n
            yt = (ctypes.c_void_p * n)(*[w["Y"][l].data_ptr() for l in range(n)])
            ya = (ctypes.c_void_p * n)(*[w["Y16a"][l].data_ptr() for l in range(n)])
            yb = (ctypes.c_void_p * n)(*[(w["Y16b"][l].data_ptr() if w["Y16b"][l] is not None else 0) for l in range(n)])
            _call("ark_zproj_fwd_v2", L.i32(self.prec_fwd), L.i32(self.prec_bwd), *zp, L.ptr(w["h0"]), L.i32(n), yt, ya, yb,
                  L.i32(B), L.i32(self.Z), L.i32(self.D), st)
            return
        # layer buffers are separate allocations: write h0 into slot 0 of every layer
        for l in range(self.n):
            _call("ark_zproj_fwd", *zp, L.ptr(w["Y"][l]), L.i64(0), L.i32(1), L.i32(B), L.i32(self.Z), L.i32(self.D), st)

    def _decoder_prologue(self, w, seq, ld_seq, B, Lq, use_drop):
        """layer 0's inputs: 16-bit token (+ position) embeddings of every step, unless the step's first launch made them"""
        p = self.p
        if self._x0_ready:   # (gathered by the step's first launch, beside the encoder pool)
            self._x0_ready = False
            return
        tick = self.hyper if use_drop else None
        if w["xtab"]:
            _call("ark_tok_time_major", L.ptr(seq), L.i64(ld_seq), L.ptr(w["tok_tm"]), L.i32(B), L.i32(Lq), L.ptr(tick), L.cur_stream())
            tick = None   # (one dropout draw per training forward)
        if not self._skip_x0(w, B, Lq):
            _call("ark_tok_gather16", L.i32(self.prec_fwd), L.i32(self.prec_bwd), L.ptr(seq), L.i64(ld_seq),
                  L.ptr(p["dec.tok_emb.weight"]), L.ptr(p["dec.pos_emb.weight"] if self.mt == "ARK" else None), L.ptr(w["X0a"]),
                  L.ptr(w["X0b"]), L.i32(B), L.i32(Lq), L.i32(self.D), L.ptr(tick), L.cur_stream())

    def _skip_x0(self, w, B, Lq):
        """True where no kernel reads the 16-bit embedding rows X0: the forward diagonals take layer 0's input projection from
        x_tab, the backward takes dW_ih0 / dW_tok from the token sums; the persistent sweep still streams X0"""
        return bool(w.get("xtab") and w.get("emb_gemm") and not self._use_sweep(B, Lq))

    def _decoder_forward_v2(self, w, seq, ld_seq, B, Lq, use_drop, save=True, project=True):
        """fast decoder forward: token gather -> layer-diagonal GRU sweep -> tied vocabulary projection
        (project=False: the caller fuses the projection into the cross-entropy)"""
        D, n, V = self.D, self.n, self.V
        R = Lq * B
        st = L.cur_stream()
        p = self.p
        pf, pb = self.prec_fwd, self.prec_bwd
        self._decoder_prologue(w, seq, ld_seq, B, Lq, use_drop)
        self._diag_sweep(w, B, Lq, use_drop, save)
        if project:
            _call("ark_gemm16", L.i32(pf), L.i32(L.EPI_BIAS), L.ptr(w["Y16a"][n - 1][B:]), L.i64(D), L.ptr(self.wtok16), L.i64(D),
                  L.ptr(self._logits(w)), L.i64(self.ldl), L.ptr(p["dec.out.bias"]), L.ptr(None), L.i32(R), L.i32(V), L.i32(D),
                  L.i32(0), st)

    def _chains(self, B):
        """Row-block chains of the diagonal sweeps.  A cell's rows depend only on the SAME rows of its two predecessor
        cells (the recurrence couples units, never graphs), so the batch splits into independent chains of dependent
        launches.  Run on parallel queues they fill each other's launch boundaries, ramps and tails: one chain's launch
        drains while the other's is in its main loop (`ark_diag_chains`; every chain keeps >= 256 rows of 64-row tiles)."""
        n = self.diag_chains
        while n > 1 and (B % (64 * n) != 0 or B // n < 256):
            n -= 1
        return [(c * (B // n), B // n) for c in range(n)]

    def _chain_stream(self, c):
        while len(self._chain_streams) < c:
            self._chain_streams.append(torch.cuda.Stream(device=self.device))
        return self._chain_streams[c - 1]

    def _run_chains(self, B, body):
        """body(b0, Bc) once per chain: chain 0 on the current queue, the others on their own queues, forked and joined
        with events (capturable)"""
        chains = self._chains(B)
        if len(chains) == 1:
            body(0, B)
            return
        main = torch.cuda.current_stream()
        used = []
        for c, (b0, Bc) in enumerate(chains):
            if c == 0:
                continue
            st = self._chain_stream(c)
            st.wait_stream(main)
            used.append(st)
        for c, (b0, Bc) in enumerate(chains):
            if c == 0:
                body(b0, Bc)
            else:
                with torch.cuda.stream(used[c - 1]):
                    body(b0, Bc)
        for st in used:
            main.wait_stream(st)

    def _use_sweep(self, B, Lq):
        """the persistent sweep needs all n * (B/16) * (D/16) workgroups resident at once (one per CU) and pays where a
        diagonal launch is all fixed cost: few rows, many steps"""
        if self.sweep in (0, False, "0", "off") or not self.use_dma:
            return False
        if self.D not in (128, 256, 512) or self.n > L.SWEEP_MAX_LAYERS or B % 16 != 0:
            return False
        fits = L.lib().ark_gru_sweep_row_tiles(L.i32(self.n), L.i32(B), L.i32(self.D)) > 0
        shared_ok = True
        if fits and self.ranks_per_device > 1:
            # every co-located rank runs the same sweeps on its shard: all their grids must be resident together (one
            # workgroup per CU each), or two partly resident grids starve each other until the bounded waits give up
            grid = max(self._sweep_wgs(B), self._sweep_wgs(B, backward=True) if self.sweep_bwd else 0)
            shared_ok = grid * self.ranks_per_device <= self._cu_count()
        if self.sweep in (1, True, "1", "on"):
            if not fits:
                raise L.ArkError(f"ark_sweep=1: {self.n} x {B // 16} x {self.D // 16} workgroups cannot be co-resident")
            if not shared_ok:
                raise L.ArkError(f"ark_sweep=1: {self.ranks_per_device} ranks share this device and their sweep grids "
                                 f"({self._sweep_wgs(B)} workgroups each, one per CU) do not fit its {self._cu_count()} CUs together")
            return True
        return fits and shared_ok and Lq >= 32

    def _cu_count(self):
        return int(torch.cuda.get_device_properties(self.device).multi_processor_count)

    def _check_beside_sweep(self, side_wgs=0):
        """refuse (ArkError) a step that could put something chip-filling beside a persistent sweep: the sweep's workgroups
        spin on each other and must ALL stay resident.  (1) a gradient all-reduce that may still be in flight -- an RCCL
        kernel holds its channels' CUs while it spins on its peers; the pipelined schedule joins it at the forward's seam,
        before the decoder, so finding one pending here is a scheduling bug; (2) a side launch capped at `side_wgs`
        one-per-CU workgroups (the chunked CE) that does not fit the CUs the sweeps of all co-located ranks leave."""
        if self._dp_pending is not None:
            raise L.ArkError("a gradient all-reduce may still be running: a persistent sweep must not share the chip with an "
                             "RCCL kernel (join it first: Engine.dp_flush())")
        if side_wgs and (self._sweep_wgs(self._B) + side_wgs) * self.ranks_per_device > self._cu_count():
            raise L.ArkError(f"{side_wgs} one-per-CU workgroups beside a {self._sweep_wgs(self._B)}-workgroup sweep on "
                             f"{self.ranks_per_device} rank(s) of this device exceed its {self._cu_count()} CUs")

    def _sweep_wgs(self, B, backward=False):
        """CUs (= physical workgroups) the persistent forward / backward sweep of this batch holds"""
        return int(L.lib().ark_gru_sweep_cus(L.i32(self.n), L.i32(B), L.i32(self.D), L.i32(1 if backward else 0), L.i32(self.sweep_ws)))

    def _sweep_fwd(self, w, B, Lq, use_drop, save=True, t0=0, t1=None):
        """the forward recurrence of all layers over the steps [t0, t1) (default: all) as ONE launch (ark_gru_sweep_fwd);
        same inputs and outputs as the diagonal launches.  A later chunk starts from the state the previous one left in
        slot t0 of the state arrays."""
        D, n, p = self.D, self.n, self.p
        self._check_beside_sweep()
        t1 = Lq if t1 is None else t1
        Lc = t1 - t0
        key = ("sweep", Lq, t0, t1)
        if key not in w:   # (zeroed once: the launches keep monotone counters; the error word is sticky)
            w[key] = (torch.empty(2 * n * Lc * B * D, device=self.device, dtype=torch.int16), self._new_sweep_sync(B, Lc))
        exch, sync = w[key]
        o = t0 * B   # rows of the time-major arrays in front of this chunk
        at = lambda t: None if t is None else t[o:]
        a = L.GruSweep()
        for l in range(n):
            drop = use_drop and l < n - 1
            y = a.layer[l]
            y.w_ih16, y.w_hh16 = L.dptr(self.wih16[l]), L.dptr(self.whh16[l])
            y.b_ih, y.b_hh = L.dptr(p[f"dec.gru.bias_ih_l{l}"]), L.dptr(p[f"dec.gru.bias_hh_l{l}"])
            y.y_t, y.y16a = L.dptr(at(w["Y"][l])), L.dptr(at(w["Y16a"][l]))
            y.y16b = L.dptr(at(w["Y16b"][l]))
            y.yd16a = L.dptr(at(w["Yd16a"][l]) if drop else None)
            y.yd16b = L.dptr(at(w["Yd16b"][l]) if drop else None)
            if save:
                y.save_r, y.save_z = L.dptr(at(w["SR"][l])), L.dptr(at(w["SZ"][l]))
                y.save_n, y.save_hn = L.dptr(at(w["SN"][l])), L.dptr(at(w["SHN"][l]))
            y.drop_seed = self._layer_seed(l)
            y.drop_p = self.p_drop if drop else 0.0
        a.x0_16, a.exch, a.sync, a.hyper = L.dptr(at(w["X0a"])), L.dptr(exch), L.dptr(sync), L.dptr(self.hyper)
        a.n_layers, a.B, a.D, a.L, a.t0, a.wg_slices = n, B, D, Lc, t0, self.sweep_ws
        import ctypes
        _call("ark_gru_sweep_fwd", L.i32(self.prec_fwd), L.i32(self.prec_bwd), ctypes.byref(a), L.cur_stream())

    def _sweep_bwd(self, w, B, Lq, use_drop):
        """the backward recurrence of all layers and steps (+ the initial-state gradient, SAIL) as ONE launch
        (ark_gru_sweep_bwd); same inputs and outputs as the backward diagonal launches"""
        D, n, g = self.D, self.n, self.g
        self._check_beside_sweep()
        key = ("sweep_bwd", Lq)
        if key not in w:
            w[key] = (torch.empty(n * Lq * B * 4 * D, device=self.device, dtype=torch.int16), self._new_sweep_sync(B, Lq))
        exch, sync = w[key]
        a = L.GruSweepBwd()
        for l in range(n):
            y = a.layer[l]
            top = l == n - 1
            y.w_hhT16 = L.dptr(self.whhT16[l])
            y.w_ihT_up16 = 0 if top else L.dptr(self.wihT16[l + 1])
            y.save_r, y.save_z = L.dptr(w["SR"][l]), L.dptr(w["SZ"][l])
            y.save_n, y.save_hn = L.dptr(w["SN"][l]), L.dptr(w["SHN"][l])
            y.y_t, y.dg16 = L.dptr(w["Y"][l]), L.dptr(w["dG16"][l])
            y.db_ih, y.db_hh = L.dptr(g[f"dec.gru.bias_ih_l{l}"]), L.dptr(g[f"dec.gru.bias_hh_l{l}"])
            y.drop_seed = self._layer_seed(l)
            y.drop_p = self.p_drop if (use_drop and not top) else 0.0
        a.dy_t = L.dptr(w["dYa"])
        a.dh0 = L.dptr(w["dH0"]) if self.mt == "SAIL" else 0
        a.exch, a.sync, a.hyper = L.dptr(exch), L.dptr(sync), L.dptr(self.hyper)
        a.n_layers, a.B, a.D, a.L, a.wg_slices = n, B, D, Lq, self.sweep_ws
        import ctypes
        _call("ark_gru_sweep_bwd", L.i32(self.prec_bwd), ctypes.byref(a), L.cur_stream())

    def _new_sweep_sync(self, B, Lc):
        """a zeroed sync workspace of a persistent sweep (header: sticky error word, detail, epoch, leave count, the record of
        the wait that gave up; then one padded monotone counter per (layer, step, row block)); registered by WEAK reference so
        that sweep_error() sees every live one and a workspace evicted from the cache takes its sync words with it"""
        import weakref
        sy = torch.zeros(L.sweep_sync_words(self.n, B, Lc), device=self.device, dtype=torch.int32)
        self.__dict__.setdefault("_sweep_sync_all", []).append(weakref.ref(sy))
        return sy

    def _live_sweep_syncs(self):
        refs = self.__dict__.get("_sweep_sync_all", [])
        live = [(r, r()) for r in refs]
        self._sweep_sync_all = [r for r, t in live if t is not None]
        return [t for _, t in live if t is not None]

    def sweep_error(self):
        """(error word, detail) of the persistent sweeps: non-zero = a workgroup of SOME sweep since the workspaces were
        last reset gave up waiting -- that launch's outputs and everything computed from them are invalid.  The word is
        sticky: no launch clears it, later sweeps on the same workspace leave at once.  Synchronises."""
        rec = self.sweep_error_record()
        return (0, 0) if rec is None else (1, rec["detail"])

    def sweep_error_record(self):
        """None, or the record the workgroup that gave up left behind: who waited where for what, and what it last saw"""
        syncs = self._live_sweep_syncs()
        if not syncs:
            return None
        v = torch.stack([sy[:9] for sy in syncs]).cpu().tolist()
        for h in v:
            if h[0] != 0:
                d, need, seen0, seen1, ticks, ns = h[1] & 0xFFFFFFFF, h[4] & 0xFFFFFFFF, h[5] & 0xFFFFFFFF, h[6] & 0xFFFFFFFF, h[7] & 0xFFFFFFFF, h[8]
                behind = [(need - s) & 0xFFFFFFFF for s in (seen0, seen1) if s != 0xFFFFFFFF]
                worst = max(behind) if behind else 0
                # a counter idles at (launches completed) x ns; `need` is one launch further
                verdict = ("its producers never signalled in this launch: they were not resident (another grid held their CUs) "
                           "or were themselves stuck" if ns and worst >= ns else
                           f"{ns - worst} of its {ns} producers had signalled: late, not absent" if ns else "no record")
                return {"detail": d, "workgroup": d >> 12 & 0x7FFFF, "step": d & 4095, "second_dependency": bool(d >> 31),
                        "epoch": h[2], "need": need, "seen": [seen0, seen1], "signals_per_counter": ns,
                        "waited_s": ticks * 1e-8, "verdict": verdict}
        return None

    def clear_sweep_error(self):
        """reset the sweep workspaces (error words, epochs, counters).  NOT a recovery: Adam has been applying updates
        computed from invalid activations since the failure -- reload a checkpoint (or the initial weights) first."""
        for sy in self._live_sweep_syncs():
            _call("ark_gru_sweep_sync_reset", L.ptr(sy), L.i64(sy.numel()), L.cur_stream())

    def sweep_error_flag(self):
        """device tensor [1], 1.0 if any sweep workspace is poisoned: add it to what the ranks all-reduce anyway (MAX), so
        that every rank of a data-parallel run raises together instead of one rank leaving the others in the next collective"""
        syncs = self._live_sweep_syncs()
        if not syncs:
            return torch.zeros(1, device=self.device)
        return torch.stack([sy[0] for sy in syncs]).ne(0).any().float().reshape(1)

    def raise_on_sweep_error(self, flag=None):
        """fail loudly if a persistent sweep of ANY step since the last check gave up waiting (sticky error word: a
        failure in step k of an epoch is still there after step n).  Call it where the step's scalars are read anyway;
        `flag`: the all-reduced (MAX) sweep_error_flag() of a data-parallel run -- every rank raises, whichever rank failed.
        FATAL for the run: every optimiser step since the failure consumed invalid activations (the sticky word makes later
        sweeps leave at once), so the parameters are corrupt -- reload a checkpoint, then clear_sweep_error()."""
        rec = self.sweep_error_record()
        if rec is None and not (flag is not None and float(flag) != 0.0):
            return
        if rec is None:
            raise L.ArkError("a persistent GRU sweep gave up waiting on ANOTHER rank: the parameters of every rank are invalid "
                             "since then (reload a checkpoint; ark_sweep: 0 selects the layer-diagonal launches)")
        raise L.ArkError(f"persistent GRU sweep gave up waiting after {rec['waited_s']:.2f} s (workgroup {rec['workgroup']}, step "
                         f"{rec['step']}, {'second' if rec['second_dependency'] else 'first'} dependency; counter needed {rec['need']}, "
                         f"last seen {rec['seen']}, launch {rec['epoch']} on this workspace): {rec['verdict']}.  Every result since "
                         "is invalid and the parameters are corrupt: reload a checkpoint, then clear_sweep_error(); "
                         "ark_sweep: 0 selects the layer-diagonal launches")

    def _diag_sweep(self, w, B, Lq, use_drop, save=True):
        """Layer-diagonal forward recurrence: cells (l, d-l) of one anti-diagonal are independent -> ONE
        launch per diagonal (one per ARK_DIAG_MAX_ROLES cells for deeper stacks), each role doing its own input
        projection (no gi buffers, no per-layer input GEMM): L+n-1 dependent launches instead of n*L + n.
        The batch is split into independent row-block chains on parallel queues (_chains)."""
        if self._use_sweep(B, Lq):
            self._sweep_fwd(w, B, Lq, use_drop, save)
            return
            return
        self._run_chains(B, lambda b0, Bc: self._diag_chain(w, B, b0, Bc, Lq, use_drop, save))

    def _diag_chain(self, w, B, b0, Bc, Lq, use_drop, save=True, diagonals=None):
        """the forward diagonals of rows [b0, b0 + Bc) of every timestep (row (t, b) = t*B + b); `diagonals`: a subset
        (profiling tools re-issue single launches)"""
        import ctypes
        D, n, p = self.D, self.n, self.p
        pf, pb = self.prec_fwd, self.prec_bwd
        for d in (range(Lq + n - 1) if diagonals is None else diagonals):
            cells = [l for l in range(n) if 0 <= d - l < Lq]
            for c0 in range(0, len(cells), L.DIAG_MAX_ROLES):
                roles = (L.GruDiagRole * L.DIAG_MAX_ROLES)()
                chunk = cells[c0:c0 + L.DIAG_MAX_ROLES]
                for k, l in enumerate(chunk):
                    t = d - l
                    drop = use_drop and l < n - 1
                    Y, Ya, Yb = w["Y"][l], w["Y16a"][l], w["Y16b"][l]
                    sl = slice(t * B + b0, t * B + b0 + Bc)
                    nx = slice((t + 1) * B + b0, (t + 1) * B + b0 + Bc)
                    r = roles[k]
                    if l == 0 and w["xtab"]:   # layer 0, small vocabulary: rows of x_tab by token id, no input K-segment
                        x = None
                        r.x_tab, r.x_tok = L.dptr(self.xtab), L.dptr(w["tok_tm"][t * B + b0:])
                    elif l == 0:
                        x = w["X0a"][sl]
                    else:
                        x = w["Yd16a"][l - 1][sl] if use_drop else w["Y16a"][l - 1][nx]
                    r.x16, r.h_prev16 = L.dptr(x), L.dptr(Ya[sl])
                    r.w_ih16, r.w_hh16 = L.dptr(None if x is None else self.wih16[l]), L.dptr(self.whh16[l])
                    r.b_ih, r.b_hh = L.dptr(p[f"dec.gru.bias_ih_l{l}"]), L.dptr(p[f"dec.gru.bias_hh_l{l}"])
                    r.y_prev_t, r.y_out_t = L.dptr(Y[sl]), L.dptr(Y[nx])
                    r.y16a, r.y16b = L.dptr(Ya[nx]), L.dptr(Yb[nx] if Yb is not None else None)
                    r.yd16a = L.dptr(w["Yd16a"][l][sl] if drop else None)
                    r.yd16b = L.dptr(w["Yd16b"][l][sl] if (drop and w["Yd16b"][l] is not None) else None)
                    if save:
                        r.save_r, r.save_z = L.dptr(w["SR"][l][sl]), L.dptr(w["SZ"][l][sl])
                        r.save_n, r.save_hn = L.dptr(w["SN"][l][sl]), L.dptr(w["SHN"][l][sl])
                    r.drop_seed = self._layer_seed(l)
                    r.drop_base = (t * B + b0) * D
                    r.drop_p = self.p_drop if drop else 0.0
                _call("ark_gru_diag_fwd", L.i32(pf), L.i32(pb), L.i32(len(chunk)), roles, L.ptr(self.hyper), L.i32(Bc),
                      L.i32(D), ctypes.byref(self.tune), L.cur_stream())

    def _decoder_forward(self, w, seq, ld_seq, B, Lq, use_drop, save=True, project=True):
        if w["v2"]:
            return self._decoder_forward_v2(w, seq, ld_seq, B, Lq, use_drop, save, project)
        D, n, V = self.D, self.n, self.V
        R = Lq * B
        st = L.cur_stream()
        p = self.p
        KM = L.LAY_KMAJ
        _call("ark_tok_gather", L.ptr(seq), L.i64(ld_seq), L.ptr(p["dec.tok_emb.weight"]),
              L.ptr(p["dec.pos_emb.weight"] if self.mt == "ARK" else None), L.ptr(w["X0"]), L.i32(B), L.i32(Lq), L.i32(D),
              L.ptr(self.hyper if use_drop else None), st)
        xin = w["X0"]
        for l in range(n):
            self._gemm(KM, KM, L.EPI_BIAS, xin, D, p[f"dec.gru.weight_ih_l{l}"], D, w["GI"][l], 3 * D, R, 3 * D, D,
                       bias=p[f"dec.gru.bias_ih_l{l}"])
            Y = w["Y"][l]
            drop = use_drop and l < n - 1
            if drop:
                _call("ark_dropout_mask", L.ptr(w["mask"][l]), L.i64(R * D), L.f32(self.p_drop),
                      L.u64(self._layer_seed(l)), L.ptr(self.hyper), st)
            for t in range(Lq):
                sl = slice(t * B, (t + 1) * B)
                common = (L.ptr(p[f"dec.gru.bias_hh_l{l}"]), L.ptr(w["GI"][l][sl]), L.ptr(Y[(t + 1) * B:]))
                tail = (L.ptr(w["Ydrop"][l][sl] if drop else None), L.ptr(w["mask"][l][sl] if drop else None),
                        L.ptr(w["SR"][l][sl] if save else None), L.ptr(w["SZ"][l][sl] if save else None),
                        L.ptr(w["SN"][l][sl] if save else None), L.ptr(w["SHN"][l][sl] if save else None),
                        L.i32(B), L.i32(D), st)
                _call("ark_gru_cell_fwd", L.i32(self.prec), L.ptr(Y[t * B:]), L.ptr(p[f"dec.gru.weight_hh_l{l}"]),
                      *common, *tail)
            xin = w["Ydrop"][l] if drop else Y[B:]
        self._xin_top = xin
        self._gemm(KM, KM, L.EPI_BIAS, w["Y"][n - 1][B:], D, p["dec.out.weight"], D, self._logits(w), self.ldl, R, V, D,
                   bias=p["dec.out.bias"])

    # ------------------------------------------------------------------ backward
    def backward(self, ext_dhead=None):
        """Launch the backward pass of the last forward (dlogits already sit in ws['logits'])."""
        if self.backward_decoder():
            self.backward_encoder(ext_dhead)

    def backward_decoder(self):
        """decoder half of the backward pass: fills every decoder gradient (the flat buffer from
        `dec_grad_offset` on); returns True when an encoder half remains (SAIL)."""
        self.prec = self.prec_bwd
        w, B, T, Lq = self.ws, self._B, self._T, self._Lrun
        D, n, V, Z = self.D, self.n, self.V, self.Z
        R = Lq * B
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p, g = self.p, self.g
        seq = self._seq
        ld_seq = seq.shape[1]
        use_drop = self.training and self.p_drop > 0
        # ONE fill of the flat gradient buffer; every reduction below (split-K weight gradients, bias
        # column sums, embedding scatters) then accumulates into it without its own memset launch
        if w["v2"]:
            self._backward_decoder_diag(w, B, Lq, seq, use_drop)   # fills, scatters and forks streams itself
            if self.mt == "ARK":
                self._join_side()
                return False
            return True
        if self._finalize is not None:   # (deferred by forward for the two-queue backward only)
            self._finalize()
            self._finalize = None
        self._zero(self.G)
        dX0 = self._backward_decoder_v1(w, B, Lq, seq, use_drop)
        _call("ark_tok_scatter", L.ptr(seq), L.i64(ld_seq), L.ptr(dX0), L.ptr(g["dec.tok_emb.weight"]), L.i32(B), L.i32(Lq),
              L.i32(D), L.i32(V), st)
        if self.mt == "ARK":
            gp = g["dec.pos_emb.weight"]
            self._colsum(dX0, D, gp, B, D, n_batch=Lq, bs_in=B * D, bs_out=D)
            self._join_side()
            return False
        return True

    def backward_encoder(self, ext_dhead=None):
        """latent + encoder half of the backward pass (SAIL): fills flat-buffer gradients [0, dec_grad_offset)
        (which includes dec.z_proj)."""
        self.prec = self.prec_bwd
        w, B, T, Lq = self.ws, self._B, self._T, self._Lrun
        D, n, V, Z = self.D, self.n, self.V, self.Z
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p, g = self.p, self.g
        nv = self._n_valid   # real rows (< B when a ragged batch was padded: those rows of dhead / dA stay zero)
        # latent path
        H = 3 * D
        h0rm = w["h0"] if w["v2"] else w["Y"][0]   # row-major h0 (the v2 state buffers are tile-native)
        fused = w["v2"] and Z <= 128
        if fused:
            # ONE launch for the per-row chain dh0 -> dz -> dhead -> dA (it sits on the dependent chain);
            # the reductions over the batch it feeds go to the side queue
            pb = self.prec_bwd
            a16b = lambda i: (w["act16b"][i] if w["act16b"][i] is not None else w["act16a"][i])
            g16b = w["g16b"] if w["g16b"] is not None else w["g16a"]
            ext = ext_dhead.contiguous() if ext_dhead is not None else None
            _call("ark_latent_chain_bwd", L.ptr(w["dH0"]), L.ptr(h0rm), L.ptr(p["dec.z_proj.weight"]), L.ptr(w["head"]),
                  L.ptr(self._eps), L.ptr(self.hyper), L.ptr(ext), L.ptr(p["enc.mu.weight"]), L.ptr(w["pre"][n - 1]),
                  L.ptr(w["dhead"]), L.ptr(w["dA"]), L.ptr(w["dpre16"][n - 1]), L.i32(pb),
                  L.ptr(g[f"enc.mlp.{2 * (n - 1)}.bias"]), L.i32(nv), L.i32(Z), L.i32(D), L.i32(H), st)
            if self._ce_dw_pending is not None and not self.ce_dw_last:
                self._ce_dw_pending()
                self._ce_dw_pending = None
            if self._fork_pending is not None:
                self._fork_pending()
            main = torch.cuda.current_stream()
            side = self._side_stream() if self.overlap_wgrad else main
            if self._side_heavy():   # (the side queue is the long one: this reduction stays on the chain's queue)
                side = main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ss = L.cur_stream()
                _call("ark_latent_reduce_bwd", L.ptr(w["dH0"]), L.ptr(w["z"]), L.ptr(g["dec.z_proj.weight"]),
                      L.ptr(g["dec.z_proj.bias"]), L.ptr(w["dhead"]), L.ptr(a16b(n - 1)), L.i32(pb), L.ptr(g["enc.mu.weight"]),
                      L.ptr(g["enc.mu.bias"]), L.i32(nv), L.i32(Z), L.i32(D), L.i32(H), ss)
            self._side_used = self._side_used or (side is not main)
        else:
            if self._ce_dw_pending is not None:
                self._ce_dw_pending()
                self._ce_dw_pending = None
            if self._fork_pending is not None:
                self._fork_pending()
            _call("ark_zproj_bwd", L.ptr(w["dH0"]), L.ptr(h0rm), L.ptr(w["z"]), L.ptr(p["dec.z_proj.weight"]),
                  L.ptr(w["dz"]), L.ptr(g["dec.z_proj.weight"]), L.ptr(g["dec.z_proj.bias"]), L.i32(B), L.i32(Z), L.i32(D), L.i32(1), st)
            _call("ark_latent_bwd", L.ptr(w["dz"]), L.ptr(w["head"]), L.ptr(self._eps), L.ptr(self.hyper), L.ptr(w["dhead"]),
                  L.i32(B), L.i32(Z), st)
            if ext_dhead is not None:
                w["dhead"].add_(ext_dhead)
            self._colsum(w["dhead"], 2 * Z, g["enc.mu.bias"], B, 2 * Z)
        if w["v2"]:
            pb = self.prec_bwd
            a16b = lambda i: (w["act16b"][i] if w["act16b"][i] is not None else w["act16a"][i])
            g16b = w["g16b"] if w["g16b"] is not None else w["g16a"]
            if not fused:
                _call("ark_gemm_wgrad", L.i32(pb), L.ptr(w["dhead"]), L.i32(0), L.i64(2 * Z), L.ptr(a16b(n - 1)), L.i32(1), L.i64(H),
                      L.ptr(g["enc.mu.weight"]), L.i64(H), L.i32(2 * Z), L.i32(H), L.i32(B), L.i32(1), st)
                # K = 2Z is not a multiple of 64 -> register-staged engine for this one product, then a 16-bit copy
                self._gemm(KM, MM, L.EPI_MUL_DGELU, w["dhead"], 2 * Z, p["enc.mu.weight"], H, w["dA"], H, B, H, 2 * Z,
                           aux=w["pre"][n - 1])
                _call("ark_cast16", L.i32(pb), L.ptr(w["dA"]), L.ptr(w["dpre16"][n - 1]), L.i64(B * H), st)
            dpre, other = w["dA"], w["dB"]
            group = []
            for i in range(n - 1, -1, -1):
                inp16 = a16b(i - 1) if i > 0 else g16b
                # bias gradient = column sums of dpre: formed by the epilogue that produced dpre (the latent kernel for
                # the last layer, the product below for the others); only the unfused latent path still needs a pass
                if i == n - 1 and not fused:
                    self._colsum(dpre, H, g[f"enc.mlp.{2 * i}.bias"], B, H)
                group.append((w["dpre16"][i], H, inp16, H, g[f"enc.mlp.{2 * i}.weight"], H, H, H, _rup(B, 64)))
                if i > 0:
                    # (no fp32 output: this gradient is only read as the next product's 16-bit operand)
                    _call("ark_gemm16_engine", L.i32(self.g16_bwd), L.i32(pb), L.i32(L.EPI_MUL_DGELU), L.ptr(w["dpre16"][i]), L.i64(H), L.ptr(self.wmT16[i]),
                          L.i64(H), L.ptr(None), L.i64(H), L.ptr(None), L.ptr(w["pre"][i - 1]), L.ptr(w["dpre16"][i - 1]), L.ptr(None),
                          L.i32(pb), L.ptr(g[f"enc.mlp.{2 * (i - 1)}.bias"]), L.i32(B), L.i32(H), L.i32(H), st)
                else:
                    _call("ark_gemm16_engine", L.i32(self.g16_bwd), L.i32(pb), L.i32(L.EPI_NONE), L.ptr(w["dpre16"][0]), L.i64(H),
                          L.ptr(self.wmT16[0]), L.i64(H), L.ptr(other), L.i64(H), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None),
                          L.i32(pb), L.ptr(None), L.i32(B), L.i32(H), L.i32(H), st)
                dpre, other = other, dpre
            pooled = False
            if self.pool_bwd_early:
                # the embedding scatter only needs the last input gradient: it goes out in front of the MLP weight gradients
                # (with the embedding tables' Adam behind it) instead of behind them and the side queue's fork
                self._enc_pool_bwd(w, dpre, B, T, st)
                pooled = True
            for i0 in range(0, len(group), 8):
                self._wgrad_group(group[i0:i0 + 8])
            if self._adam_dec_early and self._adam_dec_done and self._side is not None and self.early_mlp_adam:
                # the MLP block of the encoder bucket (99 % of it) is complete: its Adam runs on the side queue (which
                # already ticked the step for the decoder bucket) underneath the embedding scatter
                main = torch.cuda.current_stream()
                if self._side_heavy() and self._ticked is not None:
                    main.wait_event(self._ticked)
                    self._adam_launch("mlp")
                else:
                    self._side.wait_stream(main)
                    with torch.cuda.stream(self._side):
                        self._adam_launch("mlp")
                    self._side_used = True
                self._adam_mlp_done = True
        else:
            self._gemm(MM, MM, L.EPI_NONE, w["dhead"], 2 * Z, w["act"][n - 1], H, g["enc.mu.weight"], H, 2 * Z, H, B, acc=1)
            self._gemm(KM, MM, L.EPI_MUL_DGELU, w["dhead"], 2 * Z, p["enc.mu.weight"], H, w["dA"], H, B, H, 2 * Z,
                       aux=w["pre"][n - 1])
            dpre, other = w["dA"], w["dB"]
            for i in range(n - 1, -1, -1):
                inp = w["act"][i - 1] if i > 0 else w["g"]
                self._colsum(dpre, H, g[f"enc.mlp.{2 * i}.bias"], B, H)
                self._gemm(MM, MM, L.EPI_NONE, dpre, H, inp, H, g[f"enc.mlp.{2 * i}.weight"], H, H, H, B, acc=1)
                if i > 0:
                    self._gemm(KM, MM, L.EPI_MUL_DGELU, dpre, H, p[f"enc.mlp.{2 * i}.weight"], H, other, H, B, H, H,
                               aux=w["pre"][i - 1])
                else:
                    self._gemm(KM, MM, L.EPI_NONE, dpre, H, p[f"enc.mlp.{2 * i}.weight"], H, other, H, B, H, H)
                dpre, other = other, dpre
        if not (w["v2"] and pooled):
            self._enc_pool_bwd(w, dpre, B, T, st)
        if self._ce_dw_pending is not None:
            # the deferred fused-CE weight gradient (a sweep that fills most of the chip: wd-movies) goes out LAST: its
            # one-per-CU, whole-register-file workgroups (1.5 rounds of ~240 us) kept the encoder backward's ten-microsecond
            # launches off the chip for ~570 us when it was queued right behind the latent kernel (stamps, round 5)
            self._ce_dw_pending()
            self._ce_dw_pending = None
        self._join_side()

    def _side_heavy(self):
        """True where the side queue of the backward is the LONG one: a sweep workload whose fused-CE weight gradient runs
        beside the backward sweep and outlasts it (wd-articles); small work then stays on the chain's own queue"""
        return bool(self.wgrad_third_queue and getattr(self, "_fused_ce_step", False) and self.overlap_wgrad
                    and self._use_sweep(self._B, self._Lrun) and self.sweep_bwd
                    and self._sweep_wgs(self._B, backward=True) <= 128)

    def _enc_pool_bwd(self, w, dg, B, T, st):
        """embedding gradients of the encoder (scatter of the pooled input's gradient) and, in a single-process step whose
        optimiser tick is already out, the Adam of the two embedding tables right behind it on the same queue"""
        g, D = self.g, self.D
        _call("ark_enc_pool_bwd", L.ptr(self._triples), L.ptr(dg), L.ptr(w["inv_cnt"]), L.ptr(g["enc.e_emb.weight"]),
              L.ptr(g["enc.r_emb.weight"]), L.i32(B), L.i32(T), L.i32(D), L.i32(g["enc.e_emb.weight"].shape[0]),
              L.i32(g["enc.r_emb.weight"].shape[0]), L.i64(-1 if self.pad_eid is None else self.pad_eid),
              L.i64(-1 if self.pad_rid is None else self.pad_rid), st)
        if self._adam_dec_early and self._adam_dec_done and self._ticked is not None and w["v2"] and "emb" in self._adam_jobs:
            # the embedding tables' gradients are complete: their Adam goes out HERE, on the chain's own queue, beside whatever
            # the side queue still holds (wd-articles: 31 M parameters, 156 us that used to wait behind the side queue's fused-CE
            # weight gradient, GRU weight gradients and decoder-bucket Adam)
            torch.cuda.current_stream().wait_event(self._ticked)
            self._adam_launch("emb")
            self._adam_emb_done = True

    def _join_side(self):
        if self._side_used:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_used = False

    def _backward_decoder_v1(self, w, B, Lq, seq, use_drop):
        D, n, V = self.D, self.n, self.V
        R = Lq * B
        st = L.cur_stream()
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        p, g = self.p, self.g
        dlog = self._logits(w)
        ytop = w["Y"][n - 1][B:]
        # tied vocabulary projection
        self._colsum(dlog, self.ldl, g["dec.out.bias"], R, V)
        self._gemm(MM, MM, L.EPI_NONE, dlog, self.ldl, ytop, D, g["dec.out.weight"], D, V, D, R, acc=1)
        self._gemm(KM, MM, L.EPI_NONE, dlog, self.ldl, p["dec.out.weight"], D, w["dYa"], D, R, D, V)
        dy, dy_other = w["dYa"], w["dYb"]
        for l in range(n - 1, -1, -1):
            Y = w["Y"][l]
            dGI, dGH = w["GI"][l], w["dGH"][l]
            whh = p[f"dec.gru.weight_hh_l{l}"]
            for t in range(Lq - 1, -1, -1):
                sl = slice(t * B, (t + 1) * B)
                first = (t == Lq - 1)
                mid = (L.ptr(dy[sl]), L.ptr(w["carry"]), L.ptr(w["SR"][l][sl]), L.ptr(w["SZ"][l][sl]), L.ptr(w["SN"][l][sl]),
                       L.ptr(w["SHN"][l][sl]), L.ptr(Y[sl]), L.ptr(dGI[sl]), L.ptr(dGH[sl]))
                _call("ark_gru_cell_bwd", L.i32(self.prec), L.ptr(None if first else dGH[(t + 1) * B:]), L.ptr(whh),
                      *mid, L.i32(B), L.i32(D), L.i32(1 if first else 0), st)
            if self.mt == "SAIL":
                _call("ark_gru_h0_bwd", L.i32(self.prec), L.ptr(dGH), L.ptr(whh), L.ptr(w["carry"]), L.ptr(w["dH0"]),
                      L.i32(0 if l == n - 1 else 1), L.i32(B), L.i32(D), st)
            drop_below = use_drop and l > 0
            xin = w["X0"] if l == 0 else (w["Ydrop"][l - 1] if drop_below else w["Y"][l - 1][B:])
            self._gemm(MM, MM, L.EPI_NONE, dGH, 3 * D, Y, D, g[f"dec.gru.weight_hh_l{l}"], D, 3 * D, D, R, acc=1)
            self._gemm(MM, MM, L.EPI_NONE, dGI, 3 * D, xin, D, g[f"dec.gru.weight_ih_l{l}"], D, 3 * D, D, R, acc=1)
            self._colsum(dGI, 3 * D, g[f"dec.gru.bias_ih_l{l}"], R, 3 * D)
            self._colsum(dGH, 3 * D, g[f"dec.gru.bias_hh_l{l}"], R, 3 * D)
            self._gemm(KM, MM, L.EPI_MUL_AUX if drop_below else L.EPI_NONE, dGI, 3 * D, p[f"dec.gru.weight_ih_l{l}"], D,
                       dy_other, D, R, D, 3 * D, aux=w["mask"][l - 1] if drop_below else None)
            dy, dy_other = dy_other, dy
        return dy

    def _backward_decoder_diag(self, w, B, Lq, seq, use_drop):
        """Decoder backward in layer-diagonal order.  Main stream = the dependent chain only
        (dY of the top layer -> one launch per backward anti-diagonal -> h0 gradients); everything that
        merely accumulates into the flat gradient buffer runs on the side stream underneath it:
        the buffer fill, db_out / dW_tok, then -- once the gate-gradient panels are complete -- all GRU
        weight gradients as one grouped launch, layer 0's input gradient and the embedding scatter.
        The gradient arriving from the layer above is formed inside the cell (dgi_above x W_ih_above,
        through this layer's dropout mask), so no per-layer input-gradient GEMM remains."""
        D, n, V = self.D, self.n, self.V
        R = Lq * B
        p, g = self.p, self.g
        pb = self.prec_bwd
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        ld_seq = seq.shape[1]
        yb = lambda l: (w["Y16b"][l] if w["Y16b"][l] is not None else w["Y16a"][l])
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else main   # ark_overlap_wgrad=0: everything in one queue
        side.wait_stream(main)
        filled = torch.cuda.Event()
        with torch.cuda.stream(side):
            self._zero(self.G)   # ONE fill; every reduction of the step accumulates into it
            if self.mt == "SAIL":
                self._zero(w["dH0"])   # the initial-state roles add into it
            filled.record(side)   # (the first backward diagonal waits for this: nothing else goes in front of it)
            if self._adam_dec_early and side is not main:
                # the optimiser tick of THIS step, early: every Adam launch of the step -- decoder bucket, MLP block and the
                # embedding tables, on whichever queue -- only has to follow this event (round 4 ticked on the side queue behind
                # the weight gradients, so nothing could update before them)
                _call("ark_adam_tick", L.ptr(self.hyper), L.cur_stream())
                self._ticked = torch.cuda.Event()
                self._ticked.record(side)
            if self._finalize is not None:   # loss scalars of this step (deferred by forward)
                self._finalize()
                self._finalize = None
            ce_dw_later = ce_dw_beside = None
            self._wgrads_forked = False
            if self._fused_ce_step:
                # dW_tok and db_out from the recomputed softmax (no dlogits buffer): csrc/vocab_ce.hip
                def ce_dw():
                    _call("ark_vocab_ce_dw", L.i32(self.prec_fwd), L.ptr(w["Y16a"][n - 1][B:]), L.ptr(self.wtok16), L.ptr(p["dec.out.bias"]),
                          L.ptr(seq), L.i64(ld_seq), L.ptr(self.hyper), L.ptr(w["lse"]), L.ptr(g["dec.out.weight"]),
                          L.ptr(g["dec.out.bias"]), L.i32(B), L.i32(Lq), L.i32(V), L.i32(D), L.cur_stream())
                # a persistent sweep that fills most of the chip cannot share it with this launch's one-workgroup-per-CU
                # grid (measured, wd-movies: the sweep took 880 us beside it instead of 420): queue it behind the sweep,
                # where it runs beside the short kernels of the latent / encoder backward instead
                if side is not main and self._use_sweep(B, Lq) and self.sweep_bwd and self._sweep_wgs(B, backward=True) > 128:
                    ce_dw_later = ce_dw
                elif side is not main and self._use_sweep(B, Lq) and self.sweep_bwd:
                    # a sweep that leaves most of the chip free (wd-articles: 96 CUs): the two run side by side, but the
                    # sweep's co-resident workgroups must be dispatched FIRST -- this launch's ~1 000 one-per-CU workgroups
                    # would otherwise take every CU and the sweep's would trickle in as they drain, spinning on each other
                    ce_dw_beside = ce_dw
                else:
                    ce_dw()
            elif getattr(self, "_dlog16_valid", False) and self._dlog16_only:
                _call("ark_colsum16", L.i32(pb), L.ptr(w["dlog16"]), L.i64(self.Vp), L.ptr(g["dec.out.bias"]), L.i32(R), L.i32(V),
                      L.i32(1), L.cur_stream())
                _call("ark_wgrad16_rows", L.i32(pb), L.ptr(w["dlog16"]), L.i64(self.Vp), L.ptr(yb(n - 1)[B:]), L.i64(D),
                      L.ptr(g["dec.out.weight"]), L.i64(D), L.i32(self.Vp), L.i32(V), L.i32(D), L.i32(R), self.wg_tune, L.cur_stream())
            else:
                dlog = self._logits(w)
                self._colsum(dlog, self.ldl, g["dec.out.bias"], R, V)
                _call("ark_gemm_wgrad", L.i32(pb), L.ptr(dlog), L.i32(0), L.i64(self.ldl), L.ptr(yb(n - 1)[B:]), L.i32(1), L.i64(D),
                      L.ptr(g["dec.out.weight"]), L.i64(D), L.i32(V), L.i32(D), L.i32(R), L.i32(1), L.cur_stream())
        self._side_used = side is not main
        st = L.cur_stream()
        if self._fused_ce_step:
            pass   # dY of the top layer was written by the fused forward kernel (w["dYa"], tile-native)
        elif getattr(self, "_dlog16_valid", False):
            _call("ark_gemm16", L.i32(pb), L.i32(L.EPI_NONE), L.ptr(w["dlog16"]), L.i64(self.Vp), L.ptr(self.wtokT16),
                  L.i64(self.Vp), L.ptr(w["dYa"]), L.i64(D), L.ptr(None), L.ptr(None), L.i32(R), L.i32(D), L.i32(self.Vp),
                  L.i32(1), st)
        else:   # external dlogits (autograd path): register-staged engine on the fp32 buffer, then re-tile
            self._gemm(KM, MM, L.EPI_NONE, self._logits(w), self.ldl, p["dec.out.weight"], D, w["dX0"], D, R, D, V)
            _call("ark_to_tiled", L.ptr(w["dX0"]), L.ptr(w["dYa"]), L.i32(R), L.i32(D), st)
        main.wait_event(filled)   # the cells accumulate bias gradients into the flat buffer

        def fork_wgrads():
            # the big weight-gradient launch starves small dependent kernels that run beside it (the 13-us dh0
            # launch took 83 us: its 48 KB of LDS do not fit next to two 64-KB weight-gradient workgroups), so it
            # is forked off the main queue only after that launch (fork_after: 0 right after the diagonals,
            # 1 after the dh0 roles, 2 after the latent kernel as well)
            side.wait_stream(torch.cuda.current_stream())
            self._wgrads_forked = True
            if ce_dw_beside is not None and self.wgrad_third_queue and side is not main:
                # the side queue still holds the fused-CE weight gradient that ran beside the sweep (wd-articles: ~400 us
                # longer than the sweep): the GRU weight gradients and the token scatter only need the sweep's panels, so they
                # go to a queue of their own and start now; the decoder bucket's Adam (side queue) waits for both
                wq = self._wgrad_stream()
                wq.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(wq):
                    self._gru_wgrads(w, B, Lq, seq, use_drop, range(n), emb=True)
                side.wait_stream(wq)
            else:
                with torch.cuda.stream(side):
                    self._gru_wgrads(w, B, Lq, seq, use_drop, range(n), emb=True)
            with torch.cuda.stream(side):
                if self._adam_dec_early and self._ce_dw_pending is None:
                    # every gradient of the decoder bucket [dec_grad_offset, total) is now queued: its Adam (+ shadows,
                    # 40 % of the parameters, HBM-bound) runs here, underneath the latent / encoder backward chain
                    if self._ticked is None:
                        _call("ark_adam_tick", L.ptr(self.hyper), L.cur_stream())
                    self._adam_launch("dec")
                    self._adam_dec_done = True
            self._side_used = side is not main
            self._fork_pending = None

        def after_cells():
            self._side_used = side is not main
            self._fork_pending = None
            if ce_dw_beside is not None:   # (queued behind the sweep's launch, not behind its completion)
                with torch.cuda.stream(side):
                    ce_dw_beside()
            if ce_dw_later is not None:
                def go():
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        ce_dw_later()
                        if self._adam_dec_early and not self._adam_dec_done and self._wgrads_forked:   # (the fork left the update to us)
                            if self._ticked is None:
                                _call("ark_adam_tick", L.ptr(self.hyper), L.cur_stream())
                            self._adam_launch("dec")
                            self._adam_dec_done = True
                    self._side_used = True
                # SAIL: behind the latent kernel, the head of the encoder chain (its one-per-CU grid leaves that kernel 4 CUs)
                if self.mt == "SAIL" and self.ce_dw_after_latent and not self._defer_wgrads:
                    # dW_tok / db_out of the tied projection are then written AFTER the weight-gradient fork: the decoder
                    # bucket's early Adam must follow THEM, not the fork (rounds 3-4 queued it at the fork: a single-process
                    # step at the wd-movies shape never trained dec.out.bias nor the output side of the tied embedding -- found
                    # in round 5 by the two-rank test at the real vocabulary, whose data-parallel step has no early Adam)
                    self._ce_dw_pending = go
                else:
                    go()
            if not self._defer_wgrads:   # (data parallel schedules them itself, bucket by bucket)
                if self.mt == "SAIL" and self.fork_after > 0 and side is not main:
                    self._fork_pending = fork_wgrads
                else:
                    fork_wgrads()

        self._diag_bwd_sweep(w, B, Lq, use_drop, after_cells)
        if self.mt == "SAIL" and self._fork_pending is not None and self.fork_after == 1:
            self._fork_pending()

    def _diag_bwd_sweep(self, w, B, Lq, use_drop, after_cells=None):
        """the dependent chain of the decoder backward: one launch per backward anti-diagonal, then (SAIL) the
        initial-state roles.  Needs dY of the top layer in w["dYa"] and a zeroed w["dH0"].  Independent row-block
        chains run on parallel queues (_chains); `after_cells` runs once every chain's cells are queued and joined."""
        if self._use_sweep(B, Lq) and self.sweep_bwd:
            self._sweep_bwd(w, B, Lq, use_drop)   # every cell and the initial-state gradient in one persistent launch
            if after_cells is not None:
                after_cells()
            return
        h0_done = {}
        self._run_chains(B, lambda b0, Bc: h0_done.setdefault(b0, self._diag_bwd_chain(w, B, b0, Bc, Lq, use_drop)))
        if after_cells is not None:
            after_cells()
        if self.mt == "SAIL":   # continues on the main stream into the encoder half
            # dH0 = sum over layers of (carry + dgh_0 W_hh): all remaining layers as roles of one more launch
            import ctypes
            done = set.intersection(*h0_done.values()) if h0_done else set()
            rest = [l for l in range(self.n) if l not in done]
            for c0 in range(0, len(rest), L.DIAG_MAX_ROLES):
                roles = (L.GruDiagBwdRole * L.DIAG_MAX_ROLES)()
                chunk = rest[c0:c0 + L.DIAG_MAX_ROLES]
                for k, l in enumerate(chunk):
                    self._h0_role(roles[k], w, l, 0)
                _call("ark_gru_diag_bwd", L.i32(self.prec_bwd), L.i32(len(chunk)), roles, L.ptr(self.hyper), L.i32(B),
                      L.i32(self.D), ctypes.byref(self.tune), L.cur_stream())

    def _diag_bwd_chain(self, w, B, b0, Bc, Lq, use_drop, diagonals=None):
        """the backward diagonals of rows [b0, b0 + Bc); returns the layers whose initial-state role rode a diagonal
        (`diagonals`: a subset -- timing tools re-issue single launches)"""
        import ctypes
        D, n, g = self.D, self.n, self.g
        pb = self.prec_bwd
        tune = ctypes.byref(self.tune)

        def launch(roles, k):
            _call("ark_gru_diag_bwd", L.i32(pb), L.i32(k), roles, L.ptr(self.hyper), L.i32(Bc), L.i32(D), tune, L.cur_stream())

        h0_done = set()
        for e in (range(Lq + n - 1) if diagonals is None else diagonals):
            cells = [l for l in range(n - 1, -1, -1) if 0 <= Lq - 1 - (e - (n - 1 - l)) < Lq]
            for c0 in range(0, len(cells), L.DIAG_MAX_ROLES):
                roles = (L.GruDiagBwdRole * L.DIAG_MAX_ROLES)()
                chunk = cells[c0:c0 + L.DIAG_MAX_ROLES]
                k = 0
                for l in chunk:
                    t = Lq - 1 - (e - (n - 1 - l))
                    sl = slice(t * B + b0, t * B + b0 + Bc)
                    r = roles[k]
                    top = l == n - 1
                    if top:
                        r.dy_t = L.dptr(w["dYa"][sl])
                    else:
                        r.dgi_up16, r.w_ihT_up16 = L.dptr(w["dG16"][l + 1][sl]), L.dptr(self.wihT16[l + 1])
                    r.first = 1 if t == Lq - 1 else 0
                    r.dg_next16 = 0 if r.first else L.dptr(w["dG16"][l][(t + 1) * B + b0:])
                    r.w_hhT16 = L.dptr(self.whhT16[l])
                    r.carry_t = L.dptr(w["carry_l"][l][b0:])
                    r.save_r, r.save_z = L.dptr(w["SR"][l][sl]), L.dptr(w["SZ"][l][sl])
                    r.save_n, r.save_hn = L.dptr(w["SN"][l][sl]), L.dptr(w["SHN"][l][sl])
                    r.y_prev_t = L.dptr(w["Y"][l][sl])
                    r.dg16 = L.dptr(w["dG16"][l][sl])
                    r.db_ih, r.db_hh = L.dptr(g[f"dec.gru.bias_ih_l{l}"]), L.dptr(g[f"dec.gru.bias_hh_l{l}"])
                    drop = use_drop and not top
                    r.drop_seed = self._layer_seed(l)
                    r.drop_base = (t * B + b0) * D
                    r.drop_p = self.p_drop if drop else 0.0
                    k += 1
                if self.mt == "SAIL" and self.h0_ride and c0 + L.DIAG_MAX_ROLES >= len(cells):
                    # a layer's initial-state role (dH0 += carry + dgh_0 W_hh) may ride any diagonal after the one
                    # that finished its step 0; layers >= 1 fit into the under-filled last diagonals
                    for l in range(n - 1, 0, -1):
                        if e == (n - 1 - l) + Lq and k < L.DIAG_MAX_ROLES:
                            self._h0_role(roles[k], w, l, b0)
                            h0_done.add(l)
                            k += 1
                launch(roles, k)
        return h0_done

    def _h0_role(self, r, w, l, b0):
        r.dg_next16, r.w_hhT16 = L.dptr(w["dG16"][l][b0:]), L.dptr(self.whhT16[l])
        r.carry_t, r.dh0 = L.dptr(w["carry_l"][l][b0:]), L.dptr(w["dH0"][b0:])

    def _gru_wgrads(self, w, B, Lq, seq, use_drop, layers, emb):
        """weight gradients of the given GRU layers as ONE grouped launch on the current stream, plus
        (emb=True) the input-embedding gradient, which needs layer 0's gate-gradient panel.
        Panel columns (ld 4D): [dr | dz | dn | dn*r] -> dW_ih = panel[:, 0:3D]^T x, dW_hh[0:2D] = panel[:, 0:2D]^T h,
        dW_hh[2D:3D] = panel[:, 3D:4D]^T h."""
        D, n, V = self.D, self.n, self.V
        R = Lq * B
        p, g = self.p, self.g
        pb = self.prec_bwd
        KM, MM = L.LAY_KMAJ, L.LAY_MMAJ
        ld_seq = seq.shape[1]
        ss = L.cur_stream()
        yb = lambda l: (w["Y16b"][l] if w["Y16b"][l] is not None else w["Y16a"][l])
        items = []
        for l in layers:
            if l == 0:
                xin = w["X0b"] if w["X0b"] is not None else w["X0a"]
            elif use_drop:
                xin = w["Yd16b"][l - 1] if w["Yd16b"][l - 1] is not None else w["Yd16a"][l - 1]
            else:
                xin = yb(l - 1)[B:]
            G16 = w["dG16"][l]
            ghh = g[f"dec.gru.weight_hh_l{l}"]
            Rp = _rup(R, 64)   # (whole 64-row stages: the operands are allocated with zeroed rows up to there)
            items += [(G16, 4 * D, yb(l), D, ghh, D, 2 * D, D, Rp),
                      (G16[:, 3 * D:], 4 * D, yb(l), D, ghh[2 * D:], D, D, D, Rp)]
            if not (l == 0 and emb and w["emb_gemm"]):   # (layer 0's dW_ih comes from the token sums below)
                items.append((G16, 4 * D, xin, D, g[f"dec.gru.weight_ih_l{l}"], D, 3 * D, D, Rp))
        if R % 64 != 0 and R < w["_R"]:   # a run shorter than the allocation: the tail rows may hold an earlier run's panel
            for l in layers:
                w["dG16"][l][R:_rup(R, 64)].zero_()
        for i0 in range(0, len(items), L.WGRAD_MAX_GROUP):
            self._wgrad_group(items[i0:i0 + L.WGRAD_MAX_GROUP])
        if not emb:
            return
        G0 = w["dG16"][0]
        if w["emb_gemm"]:
            # S = per-token sums of layer 0's gate-gradient panel [Vp, 3D]; then two small exact-fp32 products:
            #   dW_tok += onehot^T dX0 = S W_ih_0   (no [R,D] input gradient, no scatter)
            #   dW_ih_0 += dgi_0^T X0 = S^T W_tok   (X0 rows are rows of W_tok: that weight-gradient product is not run)
            self._zero(w["S_tok"])
            _call("ark_token_sums16", L.i32(pb), L.ptr(seq), L.i64(ld_seq), L.ptr(G0), L.i64(4 * D), L.ptr(w["S_tok"]),
                  L.i64(3 * D), L.ptr(w["tok_scratch"]), L.i64(w["tok_scratch"].numel()), L.i32(B), L.i32(Lq), L.i32(self.Vp),
                  L.i32(3 * D), ss)
            _call("ark_gemm", L.i32(L.PREC_F32), L.i32(KM), L.i32(MM), L.i32(L.EPI_NONE), L.ptr(w["S_tok"]), L.i64(3 * D),
                  L.ptr(p["dec.gru.weight_ih_l0"]), L.i64(D), L.ptr(g["dec.tok_emb.weight"]), L.i64(D), L.ptr(None),
                  L.ptr(None), L.ptr(None), L.i32(V), L.i32(D), L.i32(3 * D), L.i32(1), ss)
            _call("ark_gemm", L.i32(L.PREC_F32), L.i32(MM), L.i32(MM), L.i32(L.EPI_NONE), L.ptr(w["S_tok"]), L.i64(3 * D),
                  L.ptr(p["dec.tok_emb.weight"]), L.i64(D), L.ptr(g["dec.gru.weight_ih_l0"]), L.i64(D), L.ptr(None),
                  L.ptr(None), L.ptr(None), L.i32(3 * D), L.i32(D), L.i32(V), L.i32(1), ss)
        else:
            _call("ark_gemm16", L.i32(pb), L.i32(L.EPI_NONE), L.ptr(G0), L.i64(4 * D), L.ptr(self.wihT16[0]),
                  L.i64(3 * D), L.ptr(w["dX0"]), L.i64(D), L.ptr(None), L.ptr(None), L.i32(R), L.i32(D), L.i32(3 * D),
                  L.i32(0), ss)
            _call("ark_tok_scatter", L.ptr(seq), L.i64(ld_seq), L.ptr(w["dX0"]), L.ptr(g["dec.tok_emb.weight"]), L.i32(B),
                  L.i32(Lq), L.i32(D), L.i32(V), ss)
            if self.mt == "ARK":
                self._colsum(w["dX0"], D, g["dec.pos_emb.weight"], B, D, n_batch=Lq, bs_in=B * D, bs_out=D)

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _wgrad_stream(self):
        if getattr(self, "_wgq", None) is None:
            self._wgq = torch.cuda.Stream(device=self.device)
        return self._wgq

    def _wgrad_group(self, items):
        import ctypes
        n = len(items)
        vp = lambda k: (ctypes.c_void_p * n)(*[it[k].data_ptr() for it in items])
        i64 = lambda k: (ctypes.c_int64 * n)(*[it[k] for it in items])
        i32 = lambda k: (ctypes.c_int * n)(*[it[k] for it in items])
        _call("ark_wgrad16_group", L.i32(self.prec_bwd), L.i32(n), vp(0), i64(1), vp(2), i64(3), vp(4), i64(5), i32(6), i32(7),
              i32(8), self.wg_tune, L.cur_stream())

    # ------------------------------------------------------------------ optimiser
    def adam(self):
        """fused Adam over the flat buffer; on the fast path the same launch writes the 16-bit weight shadows"""
        self.dp_flush()
        _call("ark_adam_tick", L.ptr(self.hyper), L.cur_stream())
        self._adam_launch("all")
        self.adam_steps += 1
        self._shadow_ok = True

    def _adam_part(self, which):
        """Adam (+ shadows) of ONE gradient bucket (pipelined data parallel), with the step scalars _dp_tick() left:
        `enc` = [0, dec_grad_offset), `dec` = [dec_grad_offset, total) -- together exactly adam()."""
        off, tot = self.layout.dec_grad_offset, self.layout.total
        fused = self.dp_bf16 and self.use_dma   # Adam reads the reduced bf16 bucket itself (ark_adam_step_shadows_g16)
        if not fused:
            self._dp_unpack(*((0, off) if which == "enc" else (off, tot)))
        self._adam_launch(which, g16=fused)

    def _dp_tick(self):
        """advance the optimiser step scalars once for both bucket updates of a pipelined data-parallel step"""
        _call("ark_adam_tick", L.ptr(self.hyper), L.cur_stream())
        self.adam_steps += 1

    def _dp_finish(self, h_enc, h_dec, adam_enc, adam_dec):
        """the tail of a pipelined data-parallel step.  Current queue: wait for the encoder bucket's all-reduce, tick, Adam
        of that bucket (the next step's encoder forward needs it).  UPDATE queue, concurrently: wait for the decoder
        bucket's all-reduce, Adam of that bucket -- it runs as soon as its reduction lands, underneath Adam(enc) and the
        next step's encoder forward, and is joined at that forward's seam (dp_flush).  Round 3 ran it AT the seam, on the
        critical path (45 us of the one-rank cost of the schedule)."""
        main = torch.cuda.current_stream()
        if self._dp_upd is None:
            self._dp_upd = torch.cuda.Stream(device=self.device)
        h_enc.wait()
        self._dp_tick()
        ticked = torch.cuda.Event()
        ticked.record(main)
        self._dp_upd.wait_event(ticked)   # (both updates read the ticked scalars; everything of this step is queued before)
        with torch.cuda.stream(self._dp_upd):
            h_dec.wait()
            adam_dec()
            done = torch.cuda.Event()
            done.record(self._dp_upd)
        adam_enc()
        self._dp_pending = done

    def dp_flush(self):
        """Join a pipelined data-parallel step: the current queue waits for the decoder bucket's update (its all-reduce +
        Adam on the update queue).  Called automatically before anything reads or writes the decoder parameters or the step
        scalars (the seam of the next forward, adam(), load_params, mark_params_dirty, set_hyper); call it yourself before
        reading parameters from outside the engine (checkpoints, state_dict)."""
        if self._dp_pending is None:
            return
        ev, self._dp_pending = self._dp_pending, None
        torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------ whole step
    def _pad_batch(self, triples, seq, eps):
        """copies of the inputs with the batch padded to a multiple of 16 rows (token 0 = PAD everywhere, eps 0);
        persistent buffers, so the copies can be captured into a hipGraph"""
        B = seq.shape[0]
        Bp = (B + 15) // 16 * 16
        key = (B, tuple(seq.shape[1:]), None if triples is None else tuple(triples.shape[1:]), eps is not None)
        bufs = self._pad_bufs.get(key)
        if bufs is None:
            dev = self.device
            bufs = (None if triples is None else torch.zeros((Bp,) + tuple(triples.shape[1:]), dtype=torch.int64, device=dev),
                    torch.zeros((Bp,) + tuple(seq.shape[1:]), dtype=torch.int64, device=dev),
                    None if eps is None else torch.zeros(Bp, self.Z, device=dev))
            self._pad_bufs[key] = bufs
        tp, sp, ep = bufs
        if tp is not None:
            tp[:B].copy_(triples)
        sp[:B].copy_(seq)
        if ep is not None:
            ep[:B].copy_(eps)
        return tp, sp, ep

    def _default_norms(self, B):
        # KL_NORM scales the KL GRADIENT only (loss.hip: latent_bwd / latent_chain_bwd); the reported KL value is always
        # the mean over THIS process's rows (kl_scale = -0.5 / (rows * Z) in the loss finalisation), so evaluation of
        # whole batches on one rank of a data-parallel job reports the same KL as a single process
        if self.mt == "SAIL":
            self.set_hyper(kl_norm=1.0 / (B * self.world_size * self.Z))

    def _dp_steps(self, triples, seq, eps, ce_count):
        """Data-parallel schedule of one step as a generator: every next() launches the work that COMPLETES one
        contiguous bucket [lo, hi) of the flat gradient buffer and yields it, so the caller can start that
        bucket's all-reduce while the next bucket is still being computed.  The buckets tile [0, total).

        SAIL on the diagonal path: the dependent chain goes first -- forward, decoder chain (biases, tied
        projection, dh0), latent and encoder backward -> bucket [0, dec_grad_offset) -- and the GRU weight
        gradients (one grouped launch that depends on no collective) + the token-embedding gradient last
        -> bucket [dec_grad_offset, total): the larger (encoder) reduction rides underneath them and the
        exposed one is the smaller.  Measured alternative, not kept: one bucket per MLP / GRU layer -- a
        weight-gradient launch takes ~190 us whether it carries 2 or 6 products (every workgroup walks the
        whole K = B*L alone), so per-layer launches cost more than the finer overlap returns
        (1-rank step 1.76 ms vs 1.5 ms).  Other configurations: decoder bucket, then encoder bucket."""
        tot, off = self.layout.total, self.layout.dec_grad_offset
        self._dp_mode = bool(self.dp_pipeline)
        try:
            fwd = self._forward_steps(triples, seq, eps, ce_count=ce_count)
            next(fwd)
        finally:
            self._dp_mode = False
        yield "seam"   # (encoder half done; the caller joins the previous step's decoder-bucket update here)
        for _ in fwd:
            pass
        w = self.ws
        if not (self.mt == "SAIL" and w["v2"]):
            more = self.backward_decoder()
            self._join_side()
            self._dp_pack(off, tot)
            yield (off, tot)
            if more:
                self.backward_encoder()
                self._dp_pack(0, off)
                yield (0, off)
            return
        if self.dp_order == "beside":
            self.backward_decoder()   # (forks the GRU weight gradients + token-embedding gradient onto the side queue)
            self.backward_encoder()   # (joins the side queue: every gradient is queued)
            self._dp_pack(0, off)
            yield (0, off)            # the encoder's parameters are needed first: its bucket goes out first
            self._dp_pack(off, tot)
            yield (off, tot)
            return
        self._defer_wgrads = True
        try:
            self.backward_decoder()
        finally:
            self._defer_wgrads = False
        self.backward_encoder()   # (joins the side queue)
        self._dp_pack(0, off)
        yield (0, off)
        self._gru_wgrads(w, self._B, self._Lrun, self._seq, self.training and self.p_drop > 0, range(self.n), emb=True)
        self._dp_pack(off, tot)
        yield (off, tot)

    # optional 16-bit transport of the gradient buckets (`ark_dp_bf16`): the all-reduce moves bf16 copies -- half the
    # bytes over xGMI -- and the reduced values return to the fp32 gradient buffer before Adam (fp32 master weights,
    # fp32 moments).  Off by default: the sums are formed in bf16 by the collective.
    def _dp_pack(self, lo, hi):
        if self.dp_bf16:
            if getattr(self, "_Gh", None) is None:
                self._Gh = torch.empty(self.layout.total, device=self.device, dtype=torch.bfloat16)
            _call("ark_cast16", L.i32(L.PREC_BF16), L.ptr(self.G[lo:]), L.ptr(self._Gh[lo:]), L.i64(hi - lo), L.cur_stream())

    def _dp_unpack(self, lo, hi):
        if self.dp_bf16:
            _call("ark_uncast16", L.i32(L.PREC_BF16), L.ptr(self._Gh[lo:]), L.ptr(self.G[lo:]), L.i64(hi - lo), L.cur_stream())

    def dp_bucket(self, lo, hi):
        """the tensor to all-reduce for gradient bucket [lo, hi)"""
        return self._Gh[lo:hi] if self.dp_bf16 else self.G[lo:hi]

    def train_step(self, triples, seq, eps=None, grad_sync=None, ce_count=None, dp=False):
        """forward + ELBO + backward (+ gradient all-reduce) + Adam.  Returns the device tensor
        out4 = [loss, ce, kl, token-loss sum] (no host sync).  `dp=True`: bucketed all-reduce (sum) over
        the default process group (RCCL), every bucket launched as soon as its gradients are complete
        (`_dp_steps`); `grad_sync`: a caller-supplied callable(flat_grad) run between backward and Adam."""
        self._default_norms(seq.shape[0])
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        if dp:
            import torch.distributed as dist
            handles = {}
            nseg = 0
            for item in self._dp_steps(triples, seq, eps, ce_count):
                nseg += 1
                if item == "seam":
                    self.dp_flush()
                    continue
                handles[item] = dist.all_reduce(self.dp_bucket(*item), op=dist.ReduceOp.SUM, async_op=True)
            self._dp_nseg = nseg
            off, tot = self.layout.dec_grad_offset, self.layout.total
            if self.dp_pipeline and grad_sync is None and set(handles) == {(0, off), (off, tot)}:
                # Adam of the encoder bucket now; the decoder bucket's reduction keeps flying underneath it and
                # the next step's encoder forward, and is finished in that forward's seam (dp_flush)
                self._dp_finish(handles[(0, off)], handles[(off, tot)], lambda: self._adam_part("enc"), lambda: self._adam_part("dec"))
                return self.ws["out4"]
            for h in handles.values():
                h.wait()
            for item in handles:
                self._dp_unpack(*item)
            # (unpipelined: ARK's single bucket, `ark_dp_pipeline: false`, or a caller-supplied grad_sync on top)
            if grad_sync is not None:
                grad_sync(self.G)
            self.adam()
        else:
            self._local_step(triples, seq, eps, ce_count, grad_sync)
        return self.ws["out4"]

    def _local_step(self, triples, seq, eps, ce_count, grad_sync=None):
        """forward + backward + Adam of ONE process (also the body of the captured single-process graph).  The decoder
        bucket's Adam is launched on the side queue as soon as its gradients are queued (see _backward_decoder_diag)."""
        self._defer_finalize = True
        # (fork_after = 0 would launch that Adam -- which rewrites the W_hh^T shadows -- beside the h0 roles that read them)
        self._adam_dec_early = bool(self.early_dec_adam and grad_sync is None and self.use_dma and self.overlap_wgrad
                                    and self.mt == "SAIL" and self.fork_after >= 1)
        self._adam_dec_done = False
        self._adam_mlp_done = self._adam_emb_done = False
        self._ticked = None
        try:
            self.forward(triples, seq, eps, ce_count=ce_count)
            if self.backward_decoder():
                self.backward_encoder()
        finally:
            self._defer_finalize = False
            self._adam_dec_early = False
        if grad_sync is not None:
            grad_sync(self.G)
        ticked, self._ticked = self._ticked is not None, None
        if self._adam_dec_done:   # the decoder bucket was updated on the side queue (the step is ticked): the rest
            self._adam_dec_done = False
            if self._adam_mlp_done and self._adam_emb_done:
                self._adam_launch("heads")
            elif self._adam_mlp_done:
                self._adam_launch("enc_rest")
            else:
                assert not self._adam_emb_done
                self._adam_launch("enc")
            self._adam_mlp_done = self._adam_emb_done = False
            self.adam_steps += 1
            self._shadow_ok = True
        elif ticked:   # (ticked early, but no bucket went out early: everything now, without a second tick)
            assert not self._adam_emb_done and not self._adam_mlp_done
            self.dp_flush()
            self._adam_launch("all")
            self.adam_steps += 1
            self._shadow_ok = True
        else:
            self.adam()

    def eval_loss(self, triples, seq, eps=None):
        self._default_norms(seq.shape[0])
        was = self.training
        self.training = False
        try:
            w = self.forward(triples, seq, eps, with_dlogits=False)
        finally:
            self.training = was
        return w["out4"]

    # ------------------------------------------------------------------ graph capture
    def capture_train_step(self, triples, seq, eps=None, ce_count=None, grad_sync=None, dp=False):
        """Capture the train step for fixed-address inputs into hipGraphs; returns replay().

        Single process: ONE graph (fwd + bwd + Adam).  Data parallel (`dp=True`): one graph per gradient
        bucket of `_dp_steps` with that bucket's asynchronous all-reduce launched right after its replay,
        then [Adam] once all reductions have landed -- every collective except the last (smallest) bucket's
        rides underneath the segments that follow it.  With `ark_dp_pipeline` (default) the decoder bucket's
        reduction and Adam are finished in the NEXT step's forward seam (see dp_flush; call Engine.dp_flush()
        after the last step).  `grad_sync` (callable) is the unbucketed alternative: [fwd+bwd] | call | [Adam].
        The caller refreshes the CONTENTS of `triples`, `seq`, `eps` in place between replays; step
        scalars live in the device `hyper` array."""
        self._default_norms(seq.shape[0])
        if ce_count is not None:
            self.set_hyper(ce_count=ce_count)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up outside capture (allocates the workspace)
            self.train_step(triples, seq, eps, ce_count=ce_count, grad_sync=grad_sync, dp=dp)
            self.dp_flush()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graphs = []

        def cap(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                fn()
            graphs.append(g)
            return g

        steps0 = self.adam_steps
        if dp:
            # one graph per bucket of the data-parallel schedule; the collectives run between the replays
            segs = []
            gen = self._dp_steps(triples, seq, eps, ce_count)
            for _ in range(self._dp_nseg):   # (counted by the eager warm-up step above)
                box = []
                gseg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gseg, capture_error_mode=CAPTURE_MODE):
                    box.append(next(gen))
                graphs.append(gseg)
                segs.append((gseg, box[0]))
            assert next(gen, None) is None, "data-parallel schedule changed between warm-up and capture"
            off, tot = self.layout.dec_grad_offset, self.layout.total
            pipelined = self.dp_pipeline and {it for _, it in segs if it != "seam"} == {(0, off), (off, tot)}
            ga, gb = None, None
            if pipelined:
                gc = cap(lambda: self._adam_part("enc"))
                self._dp_flush_graph = cap(lambda: self._adam_part("dec"))
            else:
                def unpack_adam():
                    for _, it in segs:
                        if it != "seam":
                            self._dp_unpack(*it)
                    self.adam()
                gc = cap(unpack_adam)
        elif grad_sync is not None:
            def a():
                self.forward(triples, seq, eps, ce_count=ce_count)
                self.backward()
            ga = cap(a)
            gb = None
            gc = cap(self.adam)
        else:
            ga, gb, gc = cap(lambda: self._local_step(triples, seq, eps, ce_count)), None, None
        self.adam_steps = steps0  # capture does not execute
        out4 = self.ws["out4"]
        gstream = torch.cuda.Stream(device=self.device)

        def replay():
            # hipGraphLaunch is NOT ordered against plain work queued on the legacy null stream
            # (measured on ROCm 7.2: input-refresh copies raced with the replay), so replays always
            # run on an explicit stream that is fenced against the caller's current stream.
            cur = torch.cuda.current_stream()
            use = cur if cur.cuda_stream != 0 else gstream
            if use is not cur:
                use.wait_stream(cur)
            with torch.cuda.stream(use):
                if dp:
                    import torch.distributed as dist
                    handles = {}
                    for gseg, item in segs:
                        gseg.replay()
                        if item == "seam":
                            self.dp_flush()   # previous step's decoder bucket: wait for its reduction, replay its Adam
                        else:
                            handles[item] = dist.all_reduce(self.dp_bucket(*item), op=dist.ReduceOp.SUM, async_op=True)
                    if pipelined:
                        self._dp_finish(handles[(0, off)], handles[(off, tot)], gc.replay, self._dp_flush_graph.replay)
                        self.adam_steps -= 1   # (counted below, as for the other modes)
                    else:
                        for h in handles.values():
                            h.wait()
                        gc.replay()
                else:
                    ga.replay()
                    if grad_sync is not None:
                        grad_sync(self.G)
                    if gc is not None:
                        gc.replay()
            if use is not cur:
                cur.wait_stream(use)
            self.adam_steps += 1
            return out4

        self._graphs[("train", seq.shape[0], dp)] = graphs
        self._pinned_B.add(seq.shape[0])
        return replay

    def dp_prefers_eager(self, B, Lq):
        """True where the data-parallel step should be launched eagerly instead of replayed from its segment graphs: the sweep
        workloads, whose step is a few dozen LONG launches whose OVERLAP is what matters (sweep beside fused CE, both directions).
        ROCm's graph executor re-deals the nodes of a captured segment to its own queues and then runs a launch with no open
        dependency behind unrelated ones (seen in kernel traces: the CE weight gradient queued behind the whole encoder
        backward instead of beside the sweep) -- measured on one rank at wd-articles: captured 8.4 ms, eager 7.0 (single-process
        captured step: 6.4; there the same graph keeps the overlap).  Eager costs nothing here: ~100 launches per 7 ms."""
        Bp = (B + 15) // 16 * 16
        # only where the sweep is the narrow one (<= 128 workgroups: wd-articles).  wd-movies (1.9 ms, ~150 launches): eager 1.88 ms
        # against 1.92 captured on an idle host -- not worth leaving the host a third of a step for the rest of the loop
        return bool(self.mt == "SAIL" and self.use_dma and self._use_sweep(Bp, Lq) and self._sweep_wgs(Bp) <= 128)

    def graphed_train_step(self, triples, seq, ce_count=None, dp=False):
        """train_step() through a cached hipGraph: the first call for a batch shape runs the step eagerly
        (that IS the step for this batch) and captures it on private input buffers; later calls copy the
        batch into those buffers and replay.  The latent noise is drawn inside the graph."""
        assert self.training, "graphed_train_step is a training-mode step"
        if dp and self.dp_prefers_eager(seq.shape[0], seq.shape[1] - 1):
            return self.train_step(triples, seq, None, ce_count=ce_count, dp=True)
        key = (None if triples is None else tuple(triples.shape), tuple(seq.shape), bool(dp), ce_count is None)
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
        self._workspace(seq.shape[0], 0)
        if not self._shadow_ok:   # parameters were written from outside since the last step
            self.refresh_shadows()
        return replay()

    # ------------------------------------------------------------------ incremental decode
    def _decode_ws(self, B):
        """persistent buffers of the one-token-at-a-time decoder (per batch size): nothing is allocated per call"""
        ws = self.__dict__.setdefault("_dec_cache", {})
        d = ws.get(B)
        if d is None:
            dev, D, n = self.device, self.D, self.n
            f = lambda *sh: torch.zeros(*sh, device=dev, dtype=torch.float32)
            d = {"x0": f(B, D), "gi": f(B, 3 * D), "logits": f(B, _rup(self.V, 4)), "Y": [f(2 * B, D) for _ in range(n)],
                 "nxt": torch.zeros(B, dtype=torch.int64, device=dev), "z": f(B, max(self.Z, 1)),
                 "toks": torch.zeros(B, self.seq_len, dtype=torch.int64, device=dev)}
            if len(ws) >= 4:
                ws.pop(next(iter(ws)))
            ws[B] = d
        return d

    @torch.no_grad()
    def decode_begin(self, B, z=None):
        """start an incremental decode of B sequences: h0 = tanh(z_proj(z)) for every layer (SAIL, reference
        models.py:139-140) or zeros (ARK, models.py:340-345).  Exact-fp32 register-staged kernels."""
        self.dp_flush()
        d = self._decode_ws(B)
        st = L.cur_stream()
        if self.mt == "SAIL":
            d["z"].copy_(z.to(self.device, dtype=torch.float32))
            for l in range(self.n):
                _call("ark_zproj_fwd", L.ptr(d["z"]), L.ptr(self.p["dec.z_proj.weight"]), L.ptr(self.p["dec.z_proj.bias"]),
                      L.ptr(d["Y"][l]), L.i64(0), L.i32(1), L.i32(B), L.i32(self.Z), L.i32(self.D), st)
        else:
            for l in range(self.n):
                d["Y"][l][:B].zero_()
        return d

    @torch.no_grad()
    def decode_step(self, d, cur, t):
        """advance the causal GRU by ONE token: `cur` [B] int64 are the tokens at position t; returns the logits
        [B, V] of position t + 1 (a view of the persistent buffer).  Replaces the reference's re-run of the whole
        prefix per generated token (models.py:291, :427)."""
        B, D, n, V = cur.shape[0], self.D, self.n, self.V
        p = self.p
        st = L.cur_stream()
        KM = L.LAY_KMAJ
        prec = L.PREC_F32
        pos = p["dec.pos_emb.weight"][t:] if self.mt == "ARK" else None
        _call("ark_tok_gather", L.ptr(cur), L.i64(1), L.ptr(p["dec.tok_emb.weight"]), L.ptr(pos), L.ptr(d["x0"]),
              L.i32(B), L.i32(1), L.i32(D), L.ptr(None), st)
        xin = d["x0"]
        for l in range(n):
            _call("ark_gemm", L.i32(prec), L.i32(KM), L.i32(KM), L.i32(L.EPI_BIAS), L.ptr(xin), L.i64(D),
                  L.ptr(p[f"dec.gru.weight_ih_l{l}"]), L.i64(D), L.ptr(d["gi"]), L.i64(3 * D), L.ptr(None),
                  L.ptr(p[f"dec.gru.bias_ih_l{l}"]), L.ptr(None), L.i32(B), L.i32(3 * D), L.i32(D), L.i32(0), st)
            a, b = d["Y"][l][(t % 2) * B:], d["Y"][l][((t + 1) % 2) * B:]
            _call("ark_gru_cell_fwd", L.i32(prec), L.ptr(a), L.ptr(p[f"dec.gru.weight_hh_l{l}"]),
                  L.ptr(p[f"dec.gru.bias_hh_l{l}"]), L.ptr(d["gi"]), L.ptr(b), L.ptr(None),
                  L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.i32(B), L.i32(D), st)
            xin = b
        ldl = d["logits"].shape[1]
        _call("ark_gemm", L.i32(prec), L.i32(KM), L.i32(KM), L.i32(L.EPI_BIAS), L.ptr(xin), L.i64(D), L.ptr(p["dec.out.weight"]),
              L.i64(D), L.ptr(d["logits"]), L.i64(ldl), L.ptr(None), L.ptr(p["dec.out.bias"]), L.ptr(None), L.i32(B), L.i32(V),
              L.i32(D), L.i32(0), st)
        return d["logits"][:, :V]

    @torch.no_grad()
    def greedy_decode(self, z, max_len=None, bos=1, eos=2):
        """token sequences of SAIL.decode_latent(z, beam=1) (reference models.py:282-300): the decoder is a causal
        GRU, so instead of re-running the whole prefix we advance one step per token.  Exact-fp32 kernels on
        persistent buffers; every step is queued without a host round trip and the reference's stopping rule
        (first position at which EVERY row's token is EOS) is applied once at the end -- positions up to there do
        not depend on later ones, so the result is the reference's, token for token."""
        assert self.mt == "SAIL"
        B = z.shape[0]
        Lmax = (self.seq_len - 1) if max_len is None else max_len
        d = self.decode_begin(B, z)
        toks = d["toks"]
        toks.fill_(eos)
        toks[:, 0] = bos
        st = L.cur_stream()
        ldl = d["logits"].shape[1]
        for t in range(Lmax):
            self.decode_step(d, toks[:, t].contiguous() if t == 0 else d["nxt"], t)
            _call("ark_argmax_rows", L.ptr(d["logits"]), L.i64(ldl), L.ptr(d["nxt"]), L.i32(B), L.i32(self.V), st)
            toks[:, t + 1] = d["nxt"]
        done = (toks[:, 1:Lmax + 1] == eos).all(dim=0)          # ONE synchronisation, after the last step
        hit = torch.nonzero(done)
        stop = int(hit[0]) + 1 if hit.numel() else Lmax
        return toks[:, :stop + 1].clone()

    @torch.no_grad()
    def beam_decode(self, z, beam, max_len=None, bos=1, eos=2):
        """token sequences of SAIL.decode_latent(z, beam > 1) (reference models.py:282-300: a batch-shared beam,
        candidates ranked by the batch-MEAN log-probability, stable descending sort).  The reference re-runs the decoder
        on every beam's whole prefix per generated token; here every beam keeps its GRU state (beam x B rows of one
        incremental decode), advances ONE token per step, and the surviving beams' states are gathered block-wise.
        Exact-fp32 kernels.  One host synchronisation per step for the reference's stopping rule."""
        assert self.mt == "SAIL" and beam >= 1
        B = z.shape[0]
        Lmax = (self.seq_len - 1) if max_len is None else max_len
        Bt = beam * B
        zr = z.to(self.device, dtype=torch.float32).repeat(beam, 1)
        d = self.decode_begin(Bt, zr)
        dev = self.device
        toks = torch.full((beam, B, Lmax + 1), eos, dtype=torch.int64, device=dev)
        toks[:, :, 0] = bos
        scores = torch.zeros(beam, B, device=dev)
        active = 1          # distinct beams so far (all blocks start as copies of the single BOS beam)
        length = 1
        for t in range(Lmax):
            logits = self.decode_step(d, toks[:, :, t].reshape(-1).contiguous(), t)
            logp = torch.log_softmax(logits.float(), dim=-1).view(beam, B, -1)
            top_lp, ids = logp.topk(beam, dim=-1)                         # [beam, B, beam]
            cand = (scores[:, :, None] + top_lp)[:active]                 # candidate (j, k) = beam j extended by its k-th token
            order = torch.sort(cand.mean(dim=1).reshape(-1), descending=True, stable=True).indices[:beam]
            j, k = order // beam, order % beam
            new_tok = ids[j, :, k]                                        # [beam, B]
            scores = scores[j] + top_lp[j, :, k]
            toks = toks[j]
            toks[:, :, t + 1] = new_tok
            half = (t + 1) % 2                                            # decode_step left the new state in this half
            for l in range(self.n):
                y = d["Y"][l].view(2, beam, B, self.D)
                y[half].copy_(y[half].index_select(0, j))
            active = beam
            length = t + 2
            if bool((new_tok == eos).all()):                                # every beam's last token is EOS
                break
        return toks[0, :, :length].clone()

