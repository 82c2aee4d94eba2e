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


def main(rank, world, port, out_path, graph, steps, bf16=False, variant="sail"):
    from oracle import sail_oracle as O
    from tests.parity_util import load_golden, synth_batch
    from ark_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # variants: "sail" (pipelined two-bucket schedule), "nopipe" (ark_dp_pipeline: false), "ark" (decoder-only: one bucket)
    _, cfg = load_golden("ark_synpaths_b32_s0" if variant == "ark" else "sail_synpaths_b32_s0")
    cfg = dict(cfg, dec_dropout=0.0, learning_rate=1e-3, ark_dp_bf16=bf16, ark_dp_pipeline=(variant != "nopipe"))
    sail = cfg["model_type"] == "SAIL"
    P = O.init_params(cfg, 0)
    B = 128
    eng = Engine(cfg, dev, precision="mixed", world_size=world, rank=rank)
    eng.load_params(P)
    eng.set_hyper(lr=1e-3, beta=0.5)
    Bl = B // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    batches = []
    for s in range(steps):
        tri, seq = synth_batch(cfg, B, seed=20 + s)
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
        for (tri, seq, eps, cnt) in batches:
            seq_in.copy_(seq[sl].to(dev))
            if sail:
                tri_in.copy_(tri[sl].to(dev)); eps_in.copy_(eps[sl].to(dev))
            eng.set_hyper(ce_count=cnt)
            step()
        eng.dp_flush()
        torch.cuda.synchronize()
    mine = eng.P.detach().cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        torch.save({"P": gathered, "adam_steps": eng.adam_steps}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1", int(sys.argv[6]),
         len(sys.argv) > 7 and sys.argv[7] == "1", sys.argv[8] if len(sys.argv) > 8 else "sail")
