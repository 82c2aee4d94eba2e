"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
or gloo on CPU for tests.  The hot path shards naturally over graphs: rank k of N
takes rows [k*B/N, (k+1)*B/N) of every global batch and of that step's eps; the only exchange is the
all-reduce(sum) of the flat gradient buffer (two buckets, overlapped / pipelined by the engine:
Engine._dp_steps, Engine.dp_flush; `make_grad_sync` below is the plain single-call alternative).
Exactness needs global normalisers:

  * CE is a mean over the non-PAD targets of the GLOBAL batch -> every rank divides by the global
    count (computed on the host from the index tensor, no collective needed);
  * kl_mean is a mean over B_global * Z elements -> KL_NORM = 1 / (B_global * Z).

With those, the summed shard gradients equal the single-process gradient (tests/test_dp_cpu.py,
tests/test_engine_gpu.py::test_shard_gradients_sum_to_full_batch, tests/test_dp_gpu.py).
"""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)"""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                # ARK_DP_BACKEND=gloo: functional rehearsal with several ranks on ONE GPU (tests); nccl = RCCL
                backend = os.environ.get("ARK_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard(t, rank, world):
    """contiguous rows [rank*B/world, (rank+1)*B/world) of a global-batch tensor"""
    B = t.shape[0]
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return t[rank * per:(rank + 1) * per].contiguous()


def count_targets(seq, pad=0):
    """number of non-PAD target tokens of a (global) batch: the CE denominator"""
    return int((seq[:, 1:] != pad).sum())


def make_grad_sync(world):
    """all-reduce(sum) of the flat gradient buffer (no-op for a single rank)"""
    if world <= 1:
        return None
    import torch.distributed as dist

    def sync(flat_grad):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)

    return sync
