#!/usr/bin/env python
"""Headline benchmark: SAIL (VAE) training graphs/sec on synthetic syn-paths-shaped batches.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one minibatch: index tensors (resident in HBM when the timed region starts: one
device-to-device copy into the step's input buffers) -> encoder -> reparameterise -> GRU decoder -> tied logits ->
CE + beta*KL -> full backward -> (RCCL gradient all-reduce when N>1) -> fused Adam.  `ms_per_step_with_upload` is the same
step fed from pinned host memory (one host -> device copy per step on the dependent chain: the figure of rounds 1-4).  Weak scaling: 1024 graphs per GPU per step
(BASELINE.json configs[1]: autoreg_syn-paths, model_type SAIL, batch 1024, D=512 Z=10 n=3).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required for RCCL on this pool

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# workload presets: the reference YAML of each dataset with model_type SAIL (BASELINE.json configs);
# dataset shapes (entities, relations, edges) are the synthetic generator's parameters.
# The headline metric is quoted on syn-paths (configs[1]); the others are here for development runs.
WORKLOADS = {
    "syn-paths": dict(d_model=512, d_latent=10, nE=49, nR=3, max_triples=3, min_triples=3, padded=False, learning_rate=1e-4, batch=1024),
    "syn-types": dict(d_model=1024, d_latent=24, nE=30, nR=3, max_triples=3, min_triples=3, padded=False, learning_rate=1e-3, batch=256),
    "wd-movies": dict(d_model=128, d_latent=64, nE=24093, nR=3, max_triples=23, min_triples=2, padded=True, learning_rate=1e-3, batch=256),
    "wd-articles": dict(d_model=512, d_latent=128, nE=60932, nR=6, max_triples=212, min_triples=4, padded=True, learning_rate=1e-4, batch=16),
}


def build_cfg(dec_dropout, workload="syn-paths"):
    c = dict(WORKLOADS[workload], model_type="SAIL", n_layers=3, n_heads=4, beta=0.1)
    nE, nR, T = c["nE"], c["nR"], c["max_triples"]
    nE2, nR2 = (nE + 1, nR + 1) if c["padded"] else (nE, nR)
    c.update(n_entities=nE2, n_relations=nR2, pad_eid=nE if c["padded"] else None, pad_rid=nR if c["padded"] else None,
             ENT_BASE=3, REL_BASE=3 + nE2, vocab_size=3 + nE2 + nR2, seq_len=2 + 3 * T, dec_dropout=dec_dropout,
             special_tokens={"PAD": 0, "BOS": 1, "EOS": 2})
    return c


def synth_global_batch(cfg, B, seed):
    from ark_amd.datasets import synthetic_batch
    return synthetic_batch(cfg["nE"], cfg["nR"], cfg["max_triples"], B, seed, padded=cfg["padded"],
                           min_triples=cfg["min_triples"])


def host_threads():
    """threads for the CPU baseline: this process's CPU share (the GPU box gives 16 per GPU)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("ARK_CPU_THREADS", "16"))))


PMC_FILE = "profiles/r05_pmc_summary.json"   # committed PMC passes of this command (tools/profile_round.sh)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def flops_per_graph(cfg):
    """F_fwd = 2*(27 D^2 + 7 D Z + 18 L D^2 + L D V) MAC->FLOP, train = 3x"""
    D, Z, V = cfg["d_model"], cfg["d_latent"], cfg["vocab_size"]
    Lq = cfg["seq_len"] - 1
    return 3 * 2 * (27 * D * D + 7 * D * Z + 18 * Lq * D * D + Lq * D * V)


def cpu_baseline(cfg, B, steps=6, warmup=2):
    """the CPU oracle (PyTorch-CPU restatement of the reference step, pinned to reference goldens)
    timed on this box's host cores: fwd + loss + autograd bwd + Adam, fp32, with the same inter-layer GRU
    dropout rate as the GPU leg (masks drawn inside the step, as nn.GRU(dropout=p) does)."""
    import torch
    from oracle import sail_oracle as O
    torch.set_num_threads(host_threads())
    P = O.init_params(cfg, 0)
    state = O.adam_init(O.leaf_params(P))
    triples, seq = synth_global_batch(cfg, B, 1)
    p = float(cfg.get("dec_dropout", 0.0))
    n, Lq, D = cfg["n_layers"], cfg["seq_len"] - 1, cfg["d_model"]
    ts = []
    for s in range(warmup + steps):
        torch.manual_seed(1000 + s)
        eps = torch.randn(B, cfg["d_latent"])
        t0 = time.perf_counter()
        masks = None
        if p > 0:
            masks = [torch.empty(B, Lq, D).bernoulli_(1 - p).div_(1 - p) for _ in range(n - 1)]
        O.train_step(P, state, (triples, seq), cfg, cfg["learning_rate"], beta=cfg["beta"], eps=eps, drop_masks=masks)
        ts.append(time.perf_counter() - t0)
    dt = sum(ts[warmup:]) / steps
    return {"value": B / dt, "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} SAIL train steps of batch {B} (syn-paths, fp32, dec_dropout {p}) after {warmup} warm-ups, "
                      f"{dt * 1e3:.1f} ms/step"}


def elbo_parity(dev, cfg, B):
    """|ELBO(HIP forward) - ELBO(CPU oracle)| / |ELBO(oracle)| on the same seed-initialised weights, batch and noise
    (dropout off: an evaluation forward), for the shipped `mixed` mode (fp16 forward operands) and for LITERAL bf16 forward
    operands (BASELINE configs[1] says "bf16"): the number beside north_star's 1e-4 bar.  Part of the cpu_baseline leg:
    the oracle is the checker here, never the thing measured."""
    import torch
    from oracle import sail_oracle as O
    from ark_amd.engine import Engine
    c = dict(cfg, dec_dropout=0.0)
    P = O.init_params(c, 0)
    triples, seq = synth_global_batch(c, B, 1)
    torch.manual_seed(1000)
    eps = torch.randn(B, c["d_latent"])
    with torch.no_grad():
        ref = float(O.sail_elbo(P, triples, seq, eps, c["beta"], c)[0])
    out = {"batch": B, "oracle_elbo": ref, "bar": 1e-4}
    for prec in ("mixed", "bf16"):
        eng = Engine(c, dev, precision=prec)
        eng.load_params(P)
        eng.set_hyper(beta=c["beta"])
        got = float(eng.eval_loss(triples.to(dev), seq.to(dev), eps.to(dev))[0])
        out[prec] = {"elbo": got, "elbo_rel_err": abs(got - ref) / abs(ref)}
        del eng
    return out


def time_diag_kernels(eng, B, reps=8):
    """average launch duration of the two layer-diagonal GRU kernels (forward, BPTT), measured live with HIP
    events on the launch stream.  Each sweep replays the real step's launch sequence (every anti-diagonal,
    its own state / save / panel buffers, so cache residency matches the train step) from one hipGraph, so the
    figures are GPU-bound and contain nothing but that kernel."""
    import torch
    w = eng.ws
    n, Lq = eng.n, eng.L
    use_drop = eng.training and eng.p_drop > 0
    st = torch.cuda.Stream()
    out = {}

    def bwd_sweep():
        # the dependent chain of the decoder backward only (bias gradients pile up in the gradient buffer: timing only)
        eng._diag_bwd_sweep(w, B, Lq, use_drop)

    if eng._use_sweep(B, Lq):
        # small batch x long sequence: each direction of the recurrence is ONE persistent launch (csrc/gru_sweep.hip)
        sweeps = {"gru_sweep_fwd_kernel": (lambda: eng._diag_sweep(w, B, Lq, use_drop, True), 1),
                  "gru_sweep_bwd_kernel": (bwd_sweep, 1)}
    else:
        # full batches: one launch per anti-diagonal of the (layer, time) grid, both directions
        sweeps = {"gru_diag_fwd_kernel": (lambda: eng._diag_sweep(w, B, Lq, use_drop, True), Lq + n - 1),
                  "gru_diag_bwd_kernel": (bwd_sweep, Lq + n - 1 + (1 if eng.mt == "SAIL" else 0))}
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for name, (fn, launches) in sweeps.items():
            fn()
            torch.cuda.synchronize()
            from ark_amd.engine import CAPTURE_MODE
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                fn()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            e1.synchronize()
            out[name] = (e0.elapsed_time(e1) / (reps * launches) * 1e-3, launches)
    return out


def diag_byte_models(eng, B):
    """algorithmic bytes per launch of the two diagonal kernels, averaged over the launches of one step.
    `min`: SURVEY 8d's per-cell minimum (x, h and the five saved D-vectors of a cell in 16 bits + the 16-bit
    weights; backward: the saved vectors, the two incoming and the outgoing gate-gradient panels);
    `impl`: what this implementation's layout moves (fp32 state, both 16-bit types, dropped copies)."""
    D, n, Lq = eng.D, eng.n, eng.L
    two = eng.prec_fwd != eng.prec_bwd
    drop = eng.training and eng.p_drop > 0
    cells = n * Lq
    wbytes = 2 * 3 * D * D * 2            # W_ih + W_hh (or their transposes), 16-bit
    f_min = cells * (B * D * 2 * 7 + wbytes)
    f_impl = cells * (B * D * (2 + 2 + 4 + 4 + 2 + (2 if two else 0) + 8) + wbytes + 2 * 3 * D * 4)
    if drop:
        f_impl += (n - 1) * Lq * B * D * 2 * (2 if two else 1)
    b_min = cells * (B * D * 2 * (5 + 3 + 3 + 4) + wbytes)
    # per cell: panel of the layer above (3D, non-top) + own panel at t+1 (3D) + dy/carry/h_prev fp32 + 4 fp16 saves in;
    # panel (4D) + carry out
    b_impl = 0
    for l in range(n):
        up = 0 if l == n - 1 else 3 * D * 2
        dy = 4 * D if l == n - 1 else 0
        b_impl += Lq * (B * (up + 3 * D * 2 + dy + 4 * D + 4 * D + 8 * D + 4 * D * 2 + 4 * D) + wbytes)
    fl = cells * 2.0 * B * D * 6 * D
    # small vocabularies: layer 0's input projection is a table row per token (W_tok W_ih0^T, Engine.xtab), not a product
    tab = getattr(eng, "xtab", None) is not None and not eng._use_sweep(B, Lq)
    fl_f = fl - (Lq * 2.0 * B * D * 3 * D if tab else 0.0)
    if tab:   # layer 0 reads neither x_t nor W_ih0: one pass over the fp32 table [V, 3D] and the token ids per launch instead
        cut = Lq * (B * D * 2 + 3 * D * D * 2) - Lq * (eng.V * 3 * D * 4 + B * 4)
        f_min -= cut
        f_impl -= cut
    # operand bytes the workgroups pull from L2 into LDS by LDS-DMA (default tiles: forward 64 rows x 32 units = 96 weight
    # rows over K = 2D -- K = D for layer 0 with the token table; backward 32 rows x 64 columns over K = 3D from the layer
    # above + 3D recurrent), summed over the step
    f_dma = (cells * 2 * D - (Lq * D if tab else 0)) * ((B + 63) // 64) * (D // 32) * (64 + 96) * 2
    b_dma = 0
    for l in range(n):
        for t in range(Lq):
            K = (3 * D if l < n - 1 else 0) + (3 * D if t < Lq - 1 else 0)
            b_dma += ((B + 31) // 32) * (D // 64) * (32 + 64) * K * 2
    return {"gru_diag_fwd_kernel": dict(min=f_min, impl=f_impl, flops=fl_f, dma=f_dma),
            "gru_diag_bwd_kernel": dict(min=b_min, impl=b_impl, flops=fl, dma=b_dma)}


def time_workload(dev, workload, precision, dropout, batch, steps, warmup, settle, extra_cfg=None, world=1, rank=0, dist=None,
                  use_dp=False, no_graph=False):
    """K timed train steps of one workload preset on this rank's GPU (the contract's timed region: barrier + synchronize on
    both sides, max over ranks); returns the engine and the timing."""
    import torch
    from ark_amd.engine import Engine
    from ark_amd import initlib
    cfg = build_cfg(dropout, workload)
    cfg.update(extra_cfg or {})
    B = batch or cfg["batch"]
    Bg = B * world
    eng = Engine(cfg, dev, precision=precision, world_size=world, rank=rank)
    eng.load_params(initlib.init_state(cfg, seed=0))
    eng.set_hyper(lr=cfg["learning_rate"], beta=cfg["beta"])

    # synthetic data ring in PINNED HOST memory.  Each batch is ONE packed byte buffer [triples int64 | seq int64 |
    # eps f32]; the step's fixed-address device inputs are views of one staging buffer, so the per-step host->device
    # transfer (part of the metric) is a single asynchronous copy inside the timed region.
    NB = 8
    T = cfg["max_triples"]
    n_tri, n_seq, n_eps = B * T * 3 * 8, B * cfg["seq_len"] * 8, B * cfg["d_latent"] * 4

    def views(buf):
        tri = buf[:n_tri].view(torch.int64).view(B, T, 3)
        sq = buf[n_tri:n_tri + n_seq].view(torch.int64).view(B, cfg["seq_len"])
        ep = buf[n_tri + n_seq:n_tri + n_seq + n_eps].view(torch.float32).view(B, cfg["d_latent"])
        return tri, sq, ep

    ring = []
    ce_counts = []
    for i in range(NB):
        tr, sq = synth_global_batch(cfg, Bg, seed=1 + i)
        torch.manual_seed(1000 + i)
        eps = torch.randn(Bg, cfg["d_latent"])
        sl = slice(rank * B, (rank + 1) * B)
        buf = torch.empty(n_tri + n_seq + n_eps, dtype=torch.uint8).pin_memory()
        a_, b_, c_ = views(buf)
        a_.copy_(tr[sl]); b_.copy_(sq[sl]); c_.copy_(eps[sl])
        ring.append(buf)
        ce_counts.append(float((sq[:, 1:] != 0).sum()))   # non-PAD targets of the GLOBAL batch
    ce_count = ce_counts[0]
    stage = ring[0].to(dev)
    tri_in, seq_in, eps_in = views(stage)
    # The timed region starts with the inputs RESIDENT IN HBM (the tier's measurement contract): the NB batches sit in a device
    # ring and every step copies its batch device-to-device into the captured step's fixed input buffers, on the run queue.
    # The PCIe-inclusive figure -- the same steps fed from the pinned host ring, one host -> device copy per step on the
    # dependent chain: what rounds 1-4 reported as `value` -- is timed right behind it and reported as `ms_per_step_with_upload`.
    # (Measured on one box, 800 timed steps each: resident 1.058-1.066 ms, in-queue upload 1.091-1.093; batch i + 1 uploaded
    #  on a copy queue into one of two landing buffers while step i computes, then a device-to-device copy in front of the
    #  step: 1.10-1.12, SLOWER than the in-queue upload -- the cross-queue events cost more than the 27 us they hide.)
    ring_dev = [b.to(dev) for b in ring]

    def feed(i, upload=False):
        eng.set_hyper(ce_count=ce_counts[i % NB])   # device-side scalar; a no-op while the count is unchanged
        stage.copy_(ring[i % NB] if upload else ring_dev[i % NB], non_blocking=True)   # on the run stream

    # everything (input H2D copies, graph replays, collectives) runs on ONE explicit stream:
    # ordering between plain copies on the legacy null stream and hipGraphLaunch is not relied on
    run_stream = torch.cuda.Stream(device=dev)
    run_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(run_stream):
        eager = bool(no_graph or (use_dp and eng.dp_prefers_eager(B, cfg["seq_len"] - 1)))
        if eager:
            # (data parallel at the sweep workloads: eager launches keep the sweep / CE overlap the segment graphs lose --
            #  Engine.dp_prefers_eager; the training loop takes the same decision in graphed_train_step)
            def step():
                return eng.train_step(tri_in, seq_in, eps_in, ce_count=ce_count, dp=use_dp)
        else:
            step = eng.capture_train_step(tri_in, seq_in, eps_in, ce_count=ce_count, dp=use_dp)

        # steady state: the first ~300 replays after capture run 3-4 % slower than the rest (same box: 1.21 ms/step timed
        # after 20 untimed steps, 1.16 after 300 or more, independent of the number of timed steps), so a fixed number of
        # untimed settle steps runs in front of the W warm-up steps the caller asked for
        log(f'{workload}: captured/ready; settle + warmup')
        for i in range(settle):
            feed(i)
            step()
        for i in range(warmup):
            feed(i)
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            feed(warmup + i)
            out4 = step()
        eng.dp_flush()   # pipelined data parallel: the last step's decoder-bucket update is part of the K steps
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.raise_on_sweep_error()   # a persistent sweep that gave up waiting voids the run
        loss_k = [float(x) for x in out4.cpu()]   # (of the K timed steps: the secondary leg below trains on)
        dt_up = None
        if dist is None and not use_dp and steps >= 100:
            # the same steps fed over PCIe (pinned host -> device per step, on the dependent chain): a secondary figure, never `value`
            nup = min(steps, 300)
            for i in range(max(50, settle)):   # (the steps after the change of feed run slower for a while, as after a capture: not timed)
                feed(i, upload=True)
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(nup):
                feed(i, upload=True)
                step()
            torch.cuda.synchronize()
            dt_up = (time.perf_counter() - t1) / nup
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return {"eng": eng, "cfg": cfg, "B": B, "Bg": Bg, "dt": dt, "loss": loss_k,
            "h2d_bytes": n_tri + n_seq + n_eps, "steps": steps, "dt_upload": dt_up, "eager": eager}


def other_workloads(dev, precision, dropout, mfma_peak):
    """the other BASELINE.json configurations at their YAML batch sizes, a few dozen captured steps each (development
    numbers in front of the driver: same kernels, same step definition as the headline)"""
    import torch
    out = {}
    for name, steps in (("syn-types", 100), ("wd-movies", 100), ("wd-articles", 30)):
        try:
            r = time_workload(dev, name, precision, dropout, 0, steps, 10, 40)   # (40 settle + 10 warm-up steps, then `steps` timed)
            eng, cfg, B, dt = r["eng"], r["cfg"], r["B"], r["dt"]
            gps = B * steps / dt
            fl = flops_per_graph(cfg)
            ent = {"batch": B, "d_model": cfg["d_model"], "seq_len": cfg["seq_len"], "vocab": cfg["vocab_size"], "steps": steps,
                   "warmup": 10, "settle_steps": 40,
                   "ms_per_step": dt / steps * 1e3, "graphs_per_s": gps, "model_tflops": gps * fl / 1e12,
                   "model_mfma_frac": gps * fl / 1e12 / mfma_peak, "final_loss": r["loss"][0]}
            if eng.ws["v2"]:
                times = time_diag_kernels(eng, B, reps=3)
                ent["diag_kernels"] = {k: {"kernel_avg_us": kt * 1e6, "launches_per_step": n, "us_per_step": kt * 1e6 * n}
                                       for k, (kt, n) in times.items()}
                if eng._use_sweep(B, eng.L):
                    steps_cp = eng.L + eng.n - 1
                    ent["persistent_sweep"] = {"workgroups": eng._sweep_wgs(B), "recurrence_steps": steps_cp,
                                               "us_per_recurrence_step": {k: kt * 1e6 / steps_cp for k, (kt, n) in times.items()}}
                    # these kernels are bound by the serial dependency chain, not by HBM or MFMA: the critical path
                    # is (L + n - 1) recurrence steps of t_step each; t_step is three dependent trips through the memory system
                    # (drain of the write-through stores, the counter, the handed-off fragments) around ~0.3 us of MFMA work
                    ent["roofline_sweeps"] = {
                        k: {"bound": "critical path (latency)", "recurrence_steps": steps_cp, "t_step_us": kt * 1e6 / steps_cp,
                            "achieved_us": kt * 1e6, "mfma_work_us_per_step": 0.3,
                            "floor_us": steps_cp * 0.3, "frac_of_floor": steps_cp * 0.3 / (kt * 1e6),
                            "phase_split_us_per_step": {"source": "profiles/r05_sweep_stamps.txt (stamped build, wd-articles, the layer that paces the sweep; round-4 hand-off protocol: monotone epochs)",
                                                        "wait_counters": 1.58, "barrier": 0.07, "fragments_land": 0.64,
                                                        "mfma_partials_barrier": 0.29, "reduce_gate_math_tiles": 0.57,
                                                        "store_drain": 0.31, "atomic_bulk_stores_loop": 0.90}}
                        for k, (kt, n) in times.items()}
                ent["diag_share_of_step"] = sum(v["us_per_step"] for v in ent["diag_kernels"].values()) / (dt / steps * 1e6)
            ent["kernel_profile"] = f"profiles/r05_{name}_kernel_stats.csv"
            out[name] = ent
            log(f'{name}: {ent["ms_per_step"]:.3f} ms/step, {gps:.0f} graphs/s')
            del eng, r
            torch.cuda.empty_cache()
        except Exception as e:   # a development leg must never take the headline line down
            out[name] = {"error": repr(e)}
    return out


def dp_overhead_1rank(dev, args, extra, plain_ms):
    """what the data-parallel SCHEDULE costs before a byte moves: the same workload through the N-rank code path (bucketed
    gradient all-reduce over a ONE-rank RCCL group, bf16 transport, split / pipelined Adam, one hipGraph per bucket) against
    the single-process step measured above, on this GPU, for both bucket orders (Engine.dp_order)."""
    import torch
    import torch.distributed as dist
    res = {"plain_ms_per_step": plain_ms, "ranks": 1, "backend": "nccl (RCCL), one rank: no transfer, the schedule's own cost",
           "steps": 200, "warmup": 20, "settle_steps": 100}
    try:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            for order in ("beside", "chain-first"):
                r = time_workload(dev, args.workload, args.precision, args.dropout, args.batch, 200, 20, 100,
                                  dict(extra, ark_dp_order=order), world=1, rank=0, dist=dist, use_dp=True)
                ms = r["dt"] / 200 * 1e3
                res[order.replace("-", "_") + "_ms_per_step"] = ms
                res[order.replace("-", "_") + "_ratio"] = ms / plain_ms
                log(f"dp schedule on one rank, order {order}: {ms:.3f} ms/step = {ms / plain_ms:.3f} x plain")
                del r
                torch.cuda.empty_cache()
            res["default_order"] = "beside"
            res["ratio"] = res["beside_ratio"]
        finally:
            dist.destroy_process_group()
    except Exception as e:   # a diagnostic leg must never take the headline line down
        res["error"] = repr(e)
    return res


def dp_overhead_sweep_workloads(args, plain):
    """The same one-rank leg at the shapes whose recurrences run as PERSISTENT SWEEPS (BASELINE configs 4 and 5 are data-parallel
    configurations): a sweep's co-resident workgroups and a live RCCL communicator in one schedule (Engine._check_beside_sweep
    refuses a step that could run them side by side).  Each in a CHILD process (`bench.py --force-dist --workload ...`: the
    N-rank entry with one rank) -- its own RCCL life cycle and the launch mode Engine.dp_prefers_eager chooses for the shape
    (eager at wd-articles, segment graphs at wd-movies) -- against the single-process step of `other_workloads`."""
    import subprocess
    out = {}
    # (settle steps: the first ~100 steps of a fresh process run 10 % slower at wd-movies -- 2.13 ms after 20, 1.92 after 100 or 300)
    for wl, nt, ns in (("wd-movies", 100, 100), ("wd-articles", 30, 30)):
        try:
            p_ms = plain[wl]["ms_per_step"]
            cmd = [sys.executable, os.path.abspath(__file__), "--force-dist", "--workload", wl, "--no-other", "--no-cpu-baseline",
                   "--steps", str(nt), "--warmup", "10", "--settle", str(ns), "--precision", args.precision, "--dropout", str(args.dropout)]
            env = dict(os.environ, MASTER_PORT=str(29600 + (os.getpid() % 300)))
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            ms = d["ms_per_step"]
            out[wl] = {"plain_ms_per_step": p_ms, "dp_ms_per_step": ms, "ratio": ms / p_ms, "steps": nt,
                       "launch": d.get("config", {}).get("launch")}
            log(f"dp schedule on one rank, {wl}: {ms:.3f} ms/step = {ms / p_ms:.3f} x plain")
        except Exception as e:   # a diagnostic leg must never take the headline line down
            out[wl] = {"error": repr(e)}
    return out


def transformer_variants(dev, precision, dropout):
    """development numbers of the Transformer variants at the syn-paths shape, B = 1024: captured train steps
    (H2D of the batch excluded: fixed device inputs), 40 timed after 10 untimed"""
    import torch
    from ark_amd import initlib
    from ark_amd.txf_engine import TxfEngine
    out = {}
    for mt, wl, nt, nw in (("t-ARK", "syn-paths", 40, 10), ("t-SAIL", "syn-paths", 40, 10),
                           # the wd-articles shape (16 graphs x 637 tokens, V = 60 943): matrix-core flash attention, fused CE
                           ("t-ARK", "wd-articles", 12, 4), ("t-SAIL", "wd-articles", 12, 4)):
        name = mt if wl == "syn-paths" else f"{mt}@{wl}"
        try:
            cfg = dict(build_cfg(dropout, wl), model_type=mt)
            B = cfg["batch"]
            eng = TxfEngine(cfg, dev, precision=precision)
            eng.load_params(initlib.init_state(cfg, seed=0))
            eng.set_hyper(lr=cfg["learning_rate"], beta=cfg["beta"])
            tri, seq = synth_global_batch(cfg, B, 1)
            tri, seq = tri.to(dev), seq.to(dev)
            st = torch.cuda.Stream(device=dev)
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                step = eng.capture_train_step(tri, seq)
                for _ in range(nw):
                    o4 = step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nt):
                    o4 = step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / nt
            out[name] = {"batch": B, "seq_len": cfg["seq_len"], "ms_per_step": dt * 1e3, "graphs_per_s": B / dt, "steps": nt, "warmup": nw,
                         "final_loss": float(o4[0]),
                         "note": "16-bit MFMA products; attention: one-wave-per-head fp32 kernels for L <= 16, matrix-core flash kernels "
                                 "beyond; fused vocabulary CE for V >= 2048; fp32 LayerNorm"}
            log(f'{name}: {dt * 1e3:.3f} ms/step')
            del eng
            torch.cuda.empty_cache()
        except Exception as e:
            out[name] = {"error": repr(e)}
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) as children of this process --
    which has not touched the GPU -- and relay their exit code; rank 0's JSON line goes to our stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("launching", " ".join(cmd))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle", type=int, default=300, help="untimed steps in front of the warm-up (clock / cache settle)")
    ap.add_argument("--batch", type=int, default=0, help="graphs per GPU per step (default: the workload's)")
    ap.add_argument("--workload", default="syn-paths", choices=list(WORKLOADS), help="headline = syn-paths")
    ap.add_argument("--precision", default="mixed", choices=["mixed", "bf16", "f16", "f32"])
    ap.add_argument("--dropout", type=float, default=0.1, help="dec_dropout (reference default 0.1)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--no-other", action="store_true", help="skip the other BASELINE workloads (development legs)")
    ap.add_argument("--cfg", default="", help="extra engine config ints, e.g. ark_overlap_wgrad=0,ark_fork_after=0")
    ap.add_argument("--diag", default="", help="diagonal-kernel tiles, e.g. fwd_rows=64,fwd_units=32,bwd_rows=32,bwd_ki=2")
    ap.add_argument("--force-dist", action="store_true", help="run the data-parallel code path even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL, one GPU per rank (the measured configuration); gloo = functional rehearsal, ranks may share a GPU")
    ap.add_argument("--wgrad", default="", help="weight-gradient kernel choices, e.g. tile=128,nbuf=2,target_wgs=300,balance=1")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when its first communicator is
    # created, so everything written to fd 1 from here on -- native libraries included -- goes to stderr, and the result line
    # is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:   # rehearsal of the multi-rank script on fewer GPUs than ranks (collectives staged through the host)
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    extra = {}
    for kv in filter(None, args.cfg.split(",")):   # engine options, e.g. ark_overlap_wgrad=0
        k, v = kv.split("=")
        extra[k] = int(v)
    if args.diag:
        extra["ark_diag_tuning"] = {k: int(v) for k, v in (kv.split("=") for kv in args.diag.split(","))}
    if args.wgrad:
        extra["ark_wgrad_tuning"] = {k: int(v) for k, v in (kv.split("=") for kv in args.wgrad.split(","))}
    use_dp = world > 1 or args.force_dist
    run = time_workload(dev, args.workload, args.precision, args.dropout, args.batch, args.steps, args.warmup, args.settle, extra,
                        world=world, rank=rank, dist=dist, use_dp=use_dp, no_graph=args.no_graph)
    eng, cfg, B, Bg, dt, loss, h2d_bytes = (run[k] for k in ("eng", "cfg", "B", "Bg", "dt", "loss", "h2d_bytes"))
    log(f'timed {args.steps} steps in {dt:.3f}s loss={loss[0]:.4f}')

    if rank == 0:
        gps = Bg * args.steps / dt
        fl = flops_per_graph(cfg)
        mfma_peak = 2500.0 if args.precision != "f32" else 157.3
        roof = None
        if eng.ws["v2"]:
            times = time_diag_kernels(eng, B)
            models = diag_byte_models(eng, B)
            traffic = {}
            try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (FETCH_SIZE x2 + WRITE_SIZE)
                pm = json.load(open(os.path.join(ROOT, PMC_FILE)))
                for name in times:   # several instantiations of one kernel template: the one the step launches most
                    rows = [v for k, v in pm.items() if name in k]
                    if rows:
                        traffic[name] = max(rows, key=lambda v: v["launches"])["hbm_bytes_corrected"]
            except Exception:
                pass
            # unit of the roofline object = ONE anti-diagonal of the full batch.  With row-block chains (Engine._chains) that is
            # `chains` concurrent launches of B / chains rows each: the event-timed sweep gives the time per anti-diagonal, the
            # committed PMC passes report bytes per LAUNCH -> x chains
            chains = len(eng._chains(B))
            traffic = {k: v * chains for k, v in traffic.items()}
            kern = {}
            alias = {"gru_sweep_fwd_kernel": "gru_diag_fwd_kernel", "gru_sweep_bwd_kernel": "gru_diag_bwd_kernel"}
            for name, (kt, launches) in times.items():
                m = models[alias.get(name, name)]   # (the sweeps move the same algorithmic bytes as the launches they replace)
                kern[name] = {"kernel_avg_us": kt * 1e6, "launches_per_step": launches, "us_per_step": kt * 1e6 * launches,
                              "bytes_per_launch": m["impl"] / launches, "bytes_per_launch_min_8d": m["min"] / launches,
                              "achieved": m["impl"] / launches / kt / 1e9, "frac": m["impl"] / launches / kt / 1e9 / 8000.0,
                              "frac_min_8d": m["min"] / launches / kt / 1e9 / 8000.0, "traffic": traffic.get(name),
                              "concurrent_launches_per_unit": chains, "flops_per_launch": m["flops"] / launches,
                              "mfma_frac": m["flops"] / launches / kt / 1e12 / mfma_peak,
                              # what actually bounds these kernels: the per-CU L2 -> LDS operand stream.
                              # ceiling = 110 GB/s per CU, measured with a consumer-less LDS-DMA ring (tools/l2_stream_bench.hip)
                              "l2_to_lds": {"bytes_per_cu_per_launch": m["dma"] / launches / 256,
                                            "achieved_gbs_per_cu": m["dma"] / launches / 256 / kt / 1e9,
                                            "measured_ceiling_gbs_per_cu": 110.0,
                                            "frac": m["dma"] / launches / 256 / kt / 1e9 / 110.0}}
                if name in alias:   # persistent: one launch walks every (layer, step); no LDS-DMA operand stream
                    del kern[name]["l2_to_lds"]
                    kern[name]["recurrence_steps_per_launch"] = eng.L + eng.n - 1
                    kern[name]["us_per_recurrence_step"] = kt * 1e6 / (eng.L + eng.n - 1)
                log(f'{name}: {kt * 1e6:.2f} us/launch x {launches}')
            dom = max(kern, key=lambda k: kern[k]["us_per_step"])   # dominant = most time per step
            d = kern[dom]
            # both kernels sit below the chip's ~310 FLOP/B balance point, so the HBM roofline bounds them
            # headline fraction = SURVEY 8d's ALGORITHMIC bytes (the per-cell minimum); what the implementation's layout moves
            # (fp32 state, both 16-bit types, dropped copies) is the secondary figure
            roof = {"bound": "hbm", "kernel": dom, "achieved": d["bytes_per_launch_min_8d"] / (d["kernel_avg_us"] * 1e-6) / 1e9,
                    "peak": 8000.0, "unit": "GB/s", "frac": d["frac_min_8d"],
                    "traffic": d["traffic"], "traffic_source": PMC_FILE + " (rocprofv3 --pmc passes of this command)",
                    "kernel_avg_us": d["kernel_avg_us"], "launches_per_step": d["launches_per_step"],
                    "bytes_per_launch": d["bytes_per_launch_min_8d"], "bytes_per_launch_impl": d["bytes_per_launch"],
                    "achieved_impl": d["achieved"], "frac_impl": d["frac"],
                    "mfma": {"achieved": d["flops_per_launch"] / (d["kernel_avg_us"] * 1e-6) / 1e12, "peak": mfma_peak,
                             "unit": "TFLOP/s", "frac": d["mfma_frac"]},
                    "kernels": kern}
        res = {
            "metric": "training graphs/sec", "value": gps, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"mixed": "f16 fwd / bf16 bwd MFMA operands (BASELINE configs[1] says bf16: bf16 forward operands miss the 1e-4 "
                               "ELBO bar, fp16 has the same width and MFMA rate), f32 accumulate+state",
                      "bf16": "bf16", "f16": "f16", "f32": "f32"}[args.precision],
            "data": f"synthetic (IntelliGraphs {args.workload}-shaped, uniform ids; random-init weights)",
            "config": {"workload": f"autoreg_{args.workload} SAIL train step (fwd+ELBO+bwd+Adam; batches resident in HBM)", "batch_per_gpu": B,
                       "global_batch": Bg, "d_model": cfg["d_model"], "d_latent": cfg["d_latent"], "n_layers": 3, "seq_len": cfg["seq_len"],
                       "vocab": cfg["vocab_size"], "dec_dropout": args.dropout, "hipgraph": not run["eager"],
                       "launch": "eager" if run["eager"] else "hipGraph replay", "settle_steps": args.settle,
                       "input_bytes_per_step": h2d_bytes,
                       "inputs": "resident in HBM when the timed region starts: 8 device-resident batches, one device-to-device copy per "
                                 "step into the captured step's input buffers (run queue); ms_per_step_with_upload = the same steps fed "
                                 "from pinned host memory, one host -> device copy per step on the dependent chain (the figure rounds "
                                 "1-4 reported as value)",
                       "parallelism": f"dp{world}"},
            "final_loss": loss[0],
            # secondary, never `value`: the PCIe-inclusive step (see config.inputs)
            "ms_per_step_with_upload": (run["dt_upload"] * 1e3 if run.get("dt_upload") else None),
            "model_tflops": gps * fl / 1e12,
            "model_mfma_frac": gps * fl / 1e12 / mfma_peak,
            "roofline": roof,
        }
        if not args.no_other and world == 1 and args.workload == "syn-paths" and not args.force_dist:
            del eng
            torch.cuda.empty_cache()
            res["dp_overhead_1rank"] = dp_overhead_1rank(dev, args, extra, dt / args.steps * 1e3)
            res["other_workloads"] = other_workloads(dev, args.precision, args.dropout, mfma_peak)
            res["dp_overhead_1rank"].update(dp_overhead_sweep_workloads(args, res["other_workloads"]))
            if args.precision != "f32":
                res["transformer_variants"] = transformer_variants(dev, args.precision, args.dropout)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(cfg, B, steps=args.cpu_steps, warmup=5)
            if args.workload == "syn-paths":
                try:
                    res["elbo_parity"] = elbo_parity(dev, cfg, B)
                except Exception as e:   # a diagnostic leg must never take the headline line down
                    res["elbo_parity"] = {"error": repr(e)}
            if B == 1024:   # the plumbing batch and the YAML batch beside it
                for b2 in (32, 256):
                    c2 = cpu_baseline(cfg, b2, steps=20, warmup=5)
                    res["cpu_baseline"][f"batch_{b2}"] = {"value": c2["value"], "sample": c2["sample"]}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()   # leave together: rank 0 is still timing its kernels while the others are done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
