"""Worker of tests/test_dp_gpu.py: one rank of a 2-process data-parallel run that shares ONE GPU (gloo
collectives on device tensors), so the real multi-process schedule -- bucket order, pipelined decoder
bucket, captured graphs -- is exercised where only one GPU exists.  Not a test module itself."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def case(variant):
    """(model config, global batch, padded graphs?) of a data-parallel test variant.
      sail / nopipe / ark: the syn-paths golden shape (L = 10, V = 55), B = 128
      wd-movies[-drop]:   BASELINE config 4's kind of shape (reference configs/autoreg_wd-movies.yaml:7-12): D = 128,
                          Z = 64, padded graphs of 1..12 triples (per-rank target counts differ), L = 37 -> the persistent
                          sweeps, V = 3 010 -> the fused / vocabulary-split CE; 32 graphs per rank
      wd-articles:        config 5's (configs/autoreg_wd-articles.yaml:5-11): D = 512, Z = 128, GLOBAL batch 16 -> 8 graphs
                          per rank, padded to the 16-row tiles; L = 259 -> sweep + CE in time chunks
      wd-movies-full / wd-articles-full: the two YAML shapes at their REAL vocabularies and lengths (V = 24 101, L = 70,
                          B = 256; V = 60 943, L = 637, B = 16): the vocabulary split of the fused CE at the sizes it was
                          built for, on two ranks"""
    from tests.parity_util import load_golden
    from tests.test_configs_gpu import _cfg, SHAPES
    if variant == "wd-movies-full":     # the YAML shape itself: V = 24 101, L = 70, GLOBAL batch 256 (128 graphs per rank)
        return SHAPES["wd-movies"][0], SHAPES["wd-movies"][1], True
    if variant == "wd-articles-full":   # V = 60 943, T = 212 -> L = 637, GLOBAL batch 16 (8-graph shards padded to 16 rows)
        return SHAPES["wd-articles-full"][0], SHAPES["wd-articles-full"][1], True
    if variant.startswith("wd-movies"):
        return dict(_cfg(128, 64, 3000, 3, 12, True), dec_dropout=0.1 if variant.endswith("-drop") else 0.0), 64, True
    if variant == "wd-articles":
        return _cfg(512, 128, 2500, 6, 86, True), 16, True
    if variant in ("tsail-long", "tark-long"):
        # 40 decoder positions (t-SAIL: 13 triples at width 3 D), V = 3 008: the matrix-core flash attention and the fused
        # vocabulary CE under the bucketed step; padded graphs, unequal per-rank target counts
        mt = "t-SAIL" if variant == "tsail-long" else "t-ARK"
        return dict(_cfg(128, 16, 3000, 3, 13, True), model_type=mt, dec_dropout=0.0, ark_txf_dropout=0.0), 32, True
    if variant in ("tsail", "tark"):   # the Transformer variants (bucketed, captured steps of ark_amd.txf_engine)
        _, cfg = load_golden("tsail_small" if variant == "tsail" else "tark_small")
        return dict(cfg, dec_dropout=0.0, ark_txf_dropout=0.0), 64, False
    _, cfg = load_golden("ark_synpaths_b32_s0" if variant == "ark" else "sail_synpaths_b32_s0")
    return dict(cfg, dec_dropout=0.0), 128, False


def main(rank, world, port, out_path, graph, steps, bf16=False, variant="sail"):
    from oracle import sail_oracle as O
    from tests.parity_util import synth_batch
    from ark_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # variants: "sail" (pipelined two-bucket schedule, weight gradients beside the chain), "chainfirst" (the same with
    # `ark_dp_order: chain-first`), "nopipe" (ark_dp_pipeline: false), "ark" (decoder-only: one bucket), and the wd-shaped
    # ones of case()
    cfg, B, padded = case(variant)
    cfg = dict(cfg, learning_rate=1e-3, ark_dp_bf16=bf16, ark_dp_pipeline=(variant != "nopipe"),
               ark_dp_order="chain-first" if variant == "chainfirst" else "beside")
    sail = cfg["model_type"] == "SAIL"
    P = O.init_params(cfg, 0)
    txf = cfg["model_type"].startswith("t-")
    if txf:
        from ark_amd.txf_engine import TxfEngine
        eng = TxfEngine(cfg, dev, precision="mixed", world_size=world, rank=rank)
        sail = cfg["model_type"] == "t-SAIL"
    else:
        eng = Engine(cfg, dev, precision="mixed", world_size=world, rank=rank)
    eng.load_params(P)
    eng.set_hyper(lr=1e-3, beta=0.5)
    Bl = B // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    batches = []
    for s in range(steps):
        tri, seq = synth_batch(cfg, B, seed=20 + s, padded=padded)
        torch.manual_seed(40 + s)
        eps = torch.randn(B, cfg["d_latent"])
        batches.append((tri, seq, eps, int((seq[:, 1:] != 0).sum())))
    tri_in = batches[0][0][sl].contiguous().to(dev) if sail else None
    seq_in = batches[0][1][sl].contiguous().to(dev)
    eps_in = batches[0][2][sl].contiguous().to(dev) if sail else None
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng._default_norms(Bl)
        if graph:
            eng.set_hyper(ce_count=batches[0][3])
            step = eng.capture_train_step(tri_in, seq_in, eps_in, ce_count=batches[0][3], dp=True)
            eng.load_params(P)          # the capture's warm-up step moved the weights: start over
            eng.reset_optimizer()
            eng.refresh_shadows()       # (a bare replay() does not do this; graphed_train_step does)
        else:
            def step():
                return eng.train_step(tri_in, seq_in, eps_in, ce_count=eng._hp["CE_COUNT"], dp=True)
        losses = []
        for (tri, seq, eps, cnt) in batches:
            seq_in.copy_(seq[sl].to(dev))
            if sail:
                tri_in.copy_(tri[sl].to(dev)); eps_in.copy_(eps[sl].to(dev))
            eng.set_hyper(ce_count=cnt)
            losses.append(step()[:3].clone())
        eng.dp_flush()
        torch.cuda.synchronize()
    mine = eng.P.detach().cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    Lq = cfg["seq_len"] - 1
    Bp = (Bl + 15) // 16 * 16
    # which paths this rank took, its sweep error words, its own non-PAD target count of the first batch, its
    # (token-loss / global count, kl) per step
    if not txf and eng.sweep_error()[0]:
        print(f"[dp_worker rank {rank}] sweep error {eng.sweep_error()} (workgroup {eng.sweep_error()[1] >> 12 & 0x7FFFF})",
              file=sys.stderr, flush=True)
    paths = [0.0, 0.0, 0.0, 0.0] if txf else [float(eng._use_sweep(Bp, Lq)), float(eng.fused_ce),
                                              float(eng._ce_chunks(Bp, Lq) is not None), float(eng.sweep_error()[0])]
    info = torch.tensor(paths + [float((batches[0][1][sl][:, 1:] != 0).sum()), float(Bp)] +
                        [float(x) for l3 in losses for x in l3.cpu()])
    infos = [torch.empty_like(info) for _ in range(world)]
    dist.all_gather(infos, info)
    if rank == 0:
        torch.save({"P": gathered, "adam_steps": eng.adam_steps, "info": infos}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1", int(sys.argv[6]),
         len(sys.argv) > 7 and sys.argv[7] == "1", sys.argv[8] if len(sys.argv) > 8 else "sail")
