"""Data-parallel correctness on CPU: 2 ranks over gloo.  Each rank computes the gradient of its
shard with the GLOBAL normalisers (CE token count, KL element count) -- here with the CPU oracle as
the compute engine -- all-reduces with ark_amd.dp's helpers and must obtain the single-process
full-batch gradient.  Uses a padded batch so per-rank token counts differ."""
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ark_amd import dp
    from oracle import sail_oracle as O
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, _, w = dp.init(backend="gloo")
    assert (r, w) == (rank, world)
    z = np.load(os.path.join(GOLD, "sail_small_pad.npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg_json"]))
    P = O.init_params(cfg, int(z["seed"]))
    leaves = O.leaf_params(P)
    triples, seq = torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"])
    eps = torch.from_numpy(z["eps0"])
    B, Z = seq.shape[0], cfg["d_latent"]
    beta = 0.7
    count = dp.count_targets(seq)
    tri_k, seq_k, eps_k = dp.shard(triples, rank, world), dp.shard(seq, rank, world), dp.shard(eps, rank, world)
    for _, p in leaves:
        p.requires_grad_(True)
    _, mu, logv = O.encoder_forward(P, tri_k, eps_k, cfg)[0:3]
    zlat = mu + eps_k * torch.exp(0.5 * logv)
    logits = O.decoder_forward(P, zlat, seq_k[:, :-1], cfg)
    tok = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), seq_k[:, 1:].reshape(-1), ignore_index=0,
                                            reduction="sum")
    kl_sum = -0.5 * torch.sum(1 + logv - mu.pow(2) - logv.exp())
    (tok / count + beta * kl_sum / (B * Z)).backward()          # global normalisers
    flat = torch.cat([p.grad.reshape(-1) for _, p in leaves])
    dp.make_grad_sync(world)(flat)
    if rank == 0:
        np.save(out_path, flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_full_batch(tmp_path):
    from oracle import sail_oracle as O
    out = str(tmp_path / "dp.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    z = np.load(os.path.join(GOLD, "sail_small_pad.npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg_json"]))
    P = O.init_params(cfg, int(z["seed"]))
    leaves = O.leaf_params(P)
    for _, p in leaves:
        p.requires_grad_(True)
    loss, *_ = O.sail_elbo(P, torch.from_numpy(z["triples"]), torch.from_numpy(z["seq"]), torch.from_numpy(z["eps0"]), 0.7, cfg)
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for _, p in leaves]).numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-8


def test_shard_and_count_helpers():
    from ark_amd import dp
    t = torch.arange(24).reshape(8, 3)
    assert torch.equal(dp.shard(t, 1, 4), t[2:4])
    seq = torch.tensor([[1, 5, 6, 7, 2, 0, 0], [1, 5, 2, 0, 0, 0, 0]])
    assert dp.count_targets(seq) == 4 + 2
    assert dp.make_grad_sync(1) is None
    try:
        dp.shard(t, 0, 3)
        assert False
    except ValueError:
        pass
