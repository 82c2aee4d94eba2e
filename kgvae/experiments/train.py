"""Training entry point:  python -m kgvae.experiments.train --config configs/<file>.yaml

Keeps the reference's CLI, flat-YAML schema, vocabulary layout, epoch loop, logging keys and
checkpoint format (reference kgvae/experiments/train.py:241-624) and -- like the reference's
ablation_study.py:549-591 -- dispatches on `model_type`, so `model_type: SAIL` trains the VAE with
`loss = ce + b*kl` and the beta0->beta1 schedule while `model_type: ARK` trains the decoder-only model.
The step itself (forward, ELBO, backward, Adam) runs on the MI355X engine (ark_amd.engine).

Differences that are deliberate:
  * `wandb` and `intelligraphs` are optional: without them metrics go to stdout / metrics.jsonl,
    data falls back to synthetic IntelliGraphs-shaped graphs, semantic verification is skipped;
  * whole splits are tokenised once per epoch (GraphSeqDataset.tensorize) instead of per item;
  * launched under torchrun it trains data-parallel (one process per GPU, RCCL all-reduce);
  * `resume_from_checkpoint: true` + `checkpoint_path` (keys the reference's YAMLs carry but never read,
    configs/autoreg_syn-paths.yaml:35-36) continue a run from a checkpoint this module wrote;
  * extra optional keys: precision (mixed|bf16|f16|f32), synthetic_sizes, max_steps_per_epoch,
    permute_rng (python|numpy), seed (data-parallel runs are always seeded: every rank must build the same model
    and draw the same epoch order).
"""
import argparse
import json
import math
import os
import random
import time
import uuid
import warnings

import torch
import yaml

from ark_amd import dp
from ark_amd.datasets import load_data_as_list
from kgvae.model.models import ARK, SAIL
from kgvae.model.utils import GraphSeqDataset, ints_to_labels, seq_to_triples
from kgvae.model.verification import get_verifier, run_semantic_evaluation


class _Tracker:
    """wandb when installed, else a local stand-in with the same init/log/finish surface"""

    def __init__(self, project, entity, config, name, run_root, enabled=True):
        self.wandb = None
        self.config = {}
        if enabled:
            try:
                import wandb
                kw = dict(project=project, config=config, name=name, anonymous="allow")
                if entity:
                    kw["entity"] = entity
                wandb.init(**kw)
                self.wandb = wandb
                self.config = dict(wandb.config)
                self.run_id = wandb.run.id
            except Exception:
                self.wandb = None
        if self.wandb is None:
            self.run_id = uuid.uuid4().hex[:8]
        self._file = None
        self._root = run_root

    def open_local(self, run_dir):
        if self.wandb is None:
            self._file = open(os.path.join(run_dir, "metrics.jsonl"), "a")

    def log(self, d):
        if self.wandb is not None:
            self.wandb.log(d)
        elif self._file is not None:
            self._file.write(json.dumps(d) + "\n")
            self._file.flush()

    def finish(self):
        if self.wandb is not None:
            self.wandb.finish()
        elif self._file is not None:
            self._file.close()


def cosine_lr(base_lr, epoch, t_max, eta_min):
    """closed form of CosineAnnealingLR stepped once per epoch (reference train.py:452-457, 561-563)"""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def iterate_batches(tri, seq, batch_size, shuffle, drop_last, generator=None):
    n = seq.shape[0]
    order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
    stop = (n // batch_size) * batch_size if drop_last else n
    for i in range(0, stop, batch_size):
        idx = order[i:i + batch_size]
        yield (tri[idx] if tri is not None else None), seq[idx]


def epoch_batches(tri, seq, batch_size, shuffle, pad, rank=0, nranks=1, generator=None):
    """Host side of one epoch: (optionally shuffled) full batches only (drop_last), stacked as
    [n_batches, B_local, ...] with this rank's contiguous row shard of every global batch, plus the
    GLOBAL target-token count of every batch (CE is a mean over the global batch's non-PAD targets;
    counting here keeps the ranks free of a per-step collective)."""
    B = batch_size
    nb = seq.shape[0] // B
    if shuffle:
        order = torch.randperm(seq.shape[0], generator=generator)   # data parallel: the SAME generator state on every rank
        seq = seq[order]
        tri = tri[order] if tri is not None else None
    seq = seq[:nb * B].view(nb, B, -1)
    tri = tri[:nb * B].view(nb, B, *tri.shape[1:]) if tri is not None else None
    counts = (seq[:, :, 1:] != pad).sum(dim=(1, 2)).tolist()
    if nranks > 1:
        assert B % nranks == 0, "batch_size must divide evenly over the ranks"
        Bl = B // nranks
        seq = seq[:, rank * Bl:(rank + 1) * Bl]
        tri = tri[:, rank * Bl:(rank + 1) * Bl] if tri is not None else None
    return tri, seq, counts


def train_epoch(model, dataset, config, device, b=1.0, lr=None, world=(0, 1), max_steps=None, epoch_seed=None):
    """one pass over the training split; returns epoch means of (loss, recon, kl, 0) over batches,
    as the reference's train_epoch does (ablation_study.py:31-88).

    The epoch's index tensors go to the device in one copy and every step is a hipGraph replay
    (`use_hip_graph: false` in the config falls back to eager launches), so the host only slices."""
    model.train()
    rank, nranks = world
    gen = None
    if epoch_seed is not None:
        # every rank draws the same per-graph permutations and the same epoch order, then takes its row shard
        random.seed(epoch_seed)
        gen = torch.Generator().manual_seed(epoch_seed)
        if getattr(dataset, "fast_rng", None) is not None:
            import numpy as np
            dataset.fast_rng = np.random.default_rng(epoch_seed)
    tri, seq = dataset.tensorize()   # redraws the per-graph permutations, like a fresh DataLoader pass
    tri, seq, counts = epoch_batches(tri, seq, config["batch_size"], config["shuffle_train"],
                                     config["special_tokens"]["PAD"], rank, nranks, generator=gen)
    nb_all = seq.shape[0]
    seq = seq.contiguous().to(device, non_blocking=True)
    tri = tri.contiguous().to(device, non_blocking=True) if tri is not None else None
    graph = bool(config.get("use_hip_graph", True))
    sums = torch.zeros(4, device=device)
    nb = 0
    for i in range(nb_all):
        out4 = model.train_step(tri[i] if tri is not None else None, seq[i], beta=b, lr=lr, ce_count=counts[i],
                                dp=nranks > 1, graph=graph)
        sums += out4
        nb += 1
        if max_steps and nb >= max_steps:
            break
    model.engine().dp_flush()    # pipelined data parallel: land the last step's decoder-bucket update
    # a persistent GRU sweep that gave up waiting poisons its rank's parameters; its flag rides the epoch's all-reduce
    # (a sum over ranks: > 0 on EVERY rank if any failed), so all ranks raise together -- a rank raising alone would
    # leave the others waiting in their next collective until the RCCL timeout
    red = torch.cat([sums, model.engine().sweep_error_flag()])
    if nranks > 1:
        import torch.distributed as dist
        dist.all_reduce(red)     # token-loss sums add up; ce/loss are re-derived below
    s = red.tolist()
    model.engine().raise_on_sweep_error(flag=s[4])
    if nb == 0:
        return 0.0, 0.0, 0.0, 0.0
    if nranks > 1:
        # every rank divides its token-loss sum by the GLOBAL count, so the per-rank ce values add up
        # to the global ce; kl_mean is a per-rank mean over equal shards, so the global kl is their mean
        ce, kl = s[1] / nb, s[2] / nb / nranks
        return ce + b * kl, ce, kl, 0.0
    return s[0] / nb, s[1] / nb, s[2] / nb, 0.0


@torch.no_grad()
def validate(model, dataset, config, device, compute_compression=False, b=1.0, special_tokens=None, world=(0, 1)):
    """mean loss / recon / kl over validation batches (+ compression bits when asked),
    reference train.py:74-129 / ablation_study.py:92-187.  Data parallel: rank k evaluates batches k, k+N, ...
    and the sums are all-reduced, so every rank sees the same numbers and nobody idles; the compression bits
    (a `sample_frac` pass) stay on rank 0."""
    model.eval()
    rank, nranks = world
    tri, seq = dataset.tensorize()
    tot = torch.zeros(5, device=device)
    for i, (tb, sb) in enumerate(iterate_batches(tri, seq, config["batch_size"], False, False)):
        if i % nranks != rank:
            continue
        tot[:4] += model.eval_loss(tb.to(device), sb.to(device), beta=b)
        tot[4] += 1
    tot = torch.cat([tot, model.engine().sweep_error_flag()])   # (evaluation forwards of long sequences run the sweep too)
    if nranks > 1:
        import torch.distributed as dist
        dist.all_reduce(tot)
    t = (tot[:4] / tot[4].clamp(min=1)).tolist()
    model.engine().raise_on_sweep_error(flag=float(tot[5]))
    res = [t[0], t[1], t[2], 0.0]
    if compute_compression and rank == 0:
        bits = model.posterior_bits(dataset, device, pad_id=config["special_tokens"]["PAD"],
                                    sample_frac=config.get("sample_frac", 0.1))
        res += [bits["avg_total_bits"], bits["avg_kl_bits"], bits["avg_ar_bits"], 0.0]
    return tuple(res)


def scheduler_state(base_lr, t_max, eta_min, last_epoch):
    """state_dict() of the reference's CosineAnnealingLR(T_max=num_epochs, eta_min) after `last_epoch` epoch-end steps
    (reference train.py:452-457, 561-563)"""
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=base_lr)
    sd = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t_max, eta_min=eta_min).state_dict()
    sd.update(last_epoch=last_epoch, _step_count=last_epoch + 1, _last_lr=[cosine_lr(base_lr, last_epoch, t_max, eta_min)])
    return sd


def optimizer_state(model, lr, initial_lr=None):
    """torch.optim.Adam(model.parameters(), lr).state_dict() with the engine's moments (loadable by load_state_dict)"""
    eng = model.engine()
    eng.dp_flush()
    names = [k for k, _ in model.named_parameters()]
    group = dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=lr).state_dict()["param_groups"][0])
    group.update(lr=lr, params=list(range(len(names))))
    if initial_lr is not None:
        group["initial_lr"] = initial_lr   # the key CosineAnnealingLR adds to the groups of its optimizer
    state = {}
    if eng.adam_steps > 0:
        for i, k in enumerate(names):
            o, sh, n = eng.layout.entries[k]
            state[i] = {"step": torch.tensor(float(eng.adam_steps)), "exp_avg": eng.M[o:o + n].view(sh).detach().cpu().clone(),
                        "exp_avg_sq": eng.Vv[o:o + n].view(sh).detach().cpu().clone()}
    return {"state": state, "param_groups": [group]}


def save_checkpoint(path, epoch, model, config, val_loss, vocabs, dataset_meta, base_lr, use_sched, best_val_loss=None):
    """Same dict layout and legacy serialisation as the reference (train.py:566-618), written at the same point of the
    epoch: AFTER scheduler.step(), so `epoch` = finished epochs, the scheduler's last_epoch = epoch and the optimizer's
    lr is the NEXT epoch's.  `ark_amd` is an extra key (resume state the reference has no place for)."""
    eng = model.engine()
    t_max, eta_min = config["num_epochs"], config.get("eta_min", 1e-6)
    lr_next = cosine_lr(base_lr, epoch, t_max, eta_min) if use_sched else base_lr
    ckpt = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "optimizer_state_dict": optimizer_state(model, lr_next, base_lr if use_sched else None),
            "scheduler_state_dict": scheduler_state(base_lr, t_max, eta_min, epoch) if use_sched else None,
            "val_loss": val_loss, "config": {k: v for k, v in config.items()}, "vocabs": vocabs, "dataset_meta": dataset_meta,
            "ark_amd": {"adam_steps": eng.adam_steps, "dropout_draws": eng.dropout_draws(), "noise_draws": eng.noise_draws(),
                        "best_val_loss": best_val_loss,
                        "cuda_rng_state": torch.cuda.get_rng_state(eng.device), "cpu_rng_state": torch.get_rng_state(),
                        "python_rng_state": random.getstate()}}
    torch.save(ckpt, path, _use_new_zipfile_serialization=False)


def load_checkpoint(path, model, device):
    """resume_from_checkpoint: weights, Adam moments and step, dropout draw counter, RNG streams.  Returns
    (finished epochs, best validation loss so far).  The file is one this module wrote; it is read with the
    restricted unpickler (tensors and plain containers only)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    eng = model.engine()
    names = [k for k, _ in model.named_parameters()]
    st = ckpt["optimizer_state_dict"]["state"]
    eng.reset_optimizer()
    step = 0
    for i, k in enumerate(names):
        if i in st:
            o, sh, n = eng.layout.entries[k]
            eng.M[o:o + n].copy_(st[i]["exp_avg"].reshape(-1))
            eng.Vv[o:o + n].copy_(st[i]["exp_avg_sq"].reshape(-1))
            step = int(st[i]["step"])
    extra = ckpt.get("ark_amd") or {}
    eng.set_optimizer_step(int(extra.get("adam_steps", step)))
    eng.set_dropout_draws(int(extra.get("dropout_draws", 0)))
    eng.set_noise_draws(int(extra.get("noise_draws", 0)))
    if extra.get("cuda_rng_state") is not None:
        torch.cuda.set_rng_state(extra["cuda_rng_state"], device)
        torch.set_rng_state(extra["cpu_rng_state"])
        ps = extra["python_rng_state"]
        random.setstate((ps[0], tuple(ps[1]), ps[2]))
    best = extra.get("best_val_loss")
    return int(ckpt["epoch"]), (float("inf") if best is None else float(best))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True, help="Path to config file")
    ap.add_argument("--wandb-project", type=str, default="submission", help="Weights & Biases project name")
    ap.add_argument("--wandb-entity", type=str, default=None, help="Weights & Biases entity")
    ap.add_argument("--checkpoint-dir", type=str, default="checkpoints", help="Directory to save checkpoints")
    args = ap.parse_args(argv)

    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    rank, local_rank, nranks = dp.init()
    main_rank = rank == 0
    model_type = config.get("model_type", "ARK")
    seeded = nranks > 1 or "seed" in config
    seed = int(config.get("seed", 0))
    if seeded:   # data parallel: identical initial weights and data order on every rank
        random.seed(seed)
        # (`seed_per_rank`: test switch -- ranks build DIFFERENT models, which the broadcast below must repair)
        torch.manual_seed(seed + (rank if config.get("seed_per_rank", False) else 0))

    tracker = _Tracker(args.wandb_project, args.wandb_entity or os.getenv("WANDB_ENTITY"), config,
                       config.get("experiment_name", "ARK_experiment"), args.checkpoint_dir, enabled=main_rank)
    for k, v in tracker.config.items():   # sweep overrides (reference train.py:252-273)
        config[k] = v
    config["learning_rate"] = float(config.get("learning_rate", 1e-3))
    run_dir = os.path.join(args.checkpoint_dir, tracker.run_id)
    if main_rank:
        os.makedirs(run_dir, exist_ok=True)
        tracker.open_local(run_dir)
        with open(os.path.join(run_dir, "effective_config.yaml"), "w") as f:
            yaml.safe_dump(config, f)
    best_comp_bits = 1e12
    tracker.log({"objective": best_comp_bits})

    if not torch.cuda.is_available():
        raise SystemExit("kgvae.experiments.train needs an AMD GPU: the model runs on hand-written gfx950 kernels only")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if config.get("use_test_for_final_eval", False):
        warnings.warn("Test set evaluation ENABLED! Only use for final evaluation, NOT for hyperparameter tuning!",
                      UserWarning, stacklevel=2)

    dataset_name = config["dataset"]
    sizes = config.get("synthetic_sizes") or {}
    (train_g, val_g, test_g, (e2i, i2e), (r2i, i2r), (min_edges, max_edges), _) = load_data_as_list(dataset_name, **sizes)
    num_entities, num_relations = len(e2i), len(r2i)
    use_padding = config.get("use_padding", dataset_name.startswith("wd-"))
    if use_padding:
        PAD_EID, PAD_RID = num_entities, num_relations
        num_entities += 1
        num_relations += 1
    else:
        PAD_EID = PAD_RID = None
    special_tokens = {"PAD": 0, "BOS": 1, "EOS": 2}
    ENT_BASE = 3
    REL_BASE = ENT_BASE + num_entities
    VOCAB_SIZE = REL_BASE + num_relations
    seq_len = 1 + max_edges * 3 + 1

    def make_ds(graphs):
        return GraphSeqDataset(graphs=graphs, i2e=i2e, i2r=i2r, triple_order=config["triple_order"],
                               permute=config.get("permute_triples", False), use_padding=use_padding, pad_eid=PAD_EID,
                               pad_rid=PAD_RID, max_triples=max_edges, special_tokens=special_tokens, ent_base=ENT_BASE,
                               rel_base=REL_BASE, seq_len=seq_len)

    train_ds, val_ds, test_ds = make_ds(train_g), make_ds(val_g), make_ds(test_g)
    if config.get("permute_rng", "python") == "numpy":   # extension: vectorised per-epoch permutations
        import numpy as np
        train_ds.fast_rng = np.random.default_rng(int(config.get("seed", 0)))
    config.update({"n_entities": num_entities, "n_relations": num_relations, "pad_eid": PAD_EID, "pad_rid": PAD_RID,
                   "seq_len": seq_len, "vocab_size": VOCAB_SIZE, "special_tokens": special_tokens, "ENT_BASE": ENT_BASE,
                   "REL_BASE": REL_BASE})
    if main_rank:
        print(f"Dataset: {dataset_name}")
        print(f"Entities: {len(e2i)}, Relations: {len(r2i)}")
        print(f"Train batches: {len(train_ds) // config['batch_size']}, Val batches: {math.ceil(len(val_ds) / config['batch_size'])}")
    verifier = get_verifier(dataset_name)
    if verifier is None and main_rank:
        print(f"Warning: No verifier available for dataset {dataset_name}")

    if model_type in ("ARK", "t-ARK"):
        model = ARK(config).to(device)
    elif model_type in ("SAIL", "t-SAIL"):
        model = SAIL(config).to(device)
    else:
        raise NotImplementedError(f"Model type '{model_type}' is not implemented. Use one of: 'ARK','t-ARK','SAIL','t-SAIL'.")
    if main_rank:
        print(f"Using model: {model_type}")
    eng = model.engine()
    eng.world_size, eng.rank = nranks, rank
    if nranks > 1:
        # the replicas must START equal (as DistributedDataParallel does): rank 0's weights win
        import torch.distributed as dist
        dist.broadcast(eng.P, src=0)
        eng.mark_params_dirty()

    base_lr = config["learning_rate"]
    use_sched = bool(config.get("lr_scheduler", False))
    eta_min = config.get("eta_min", 1e-6)
    num_epochs = config["num_epochs"]
    vocabs = {"e2i": e2i, "i2e": i2e, "r2i": r2i, "i2r": i2r}
    dataset_meta = {"dataset": dataset_name, "n_entities": len(i2e), "n_relations": len(i2r)}
    best_val_loss = float("inf")
    start_epoch = 0
    if config.get("resume_from_checkpoint", False):
        start_epoch, best_val_loss = load_checkpoint(config["checkpoint_path"], model, device)
        if main_rank:
            print(f"Resumed from {config['checkpoint_path']} at epoch {start_epoch}")

    for epoch in range(start_epoch, num_epochs):
        if main_rank:
            print(f"\nEpoch {epoch + 1}/{num_epochs}")
        b = 1
        if model_type in ("SAIL", "t-SAIL"):
            b = config["beta0"] + (config["beta1"] - config["beta0"]) * epoch / num_epochs
        lr = cosine_lr(base_lr, epoch, num_epochs, eta_min) if use_sched else base_lr
        t0 = time.time()
        train_loss, train_recon, train_kl, _ = train_epoch(model, train_ds, config, device, b, lr, (rank, nranks),
                                                           config.get("max_steps_per_epoch"),
                                                           epoch_seed=(seed * 1000003 + epoch) if nranks > 1 else None)
        dt = time.time() - t0
        do_comp = ((epoch + 1) % int(config.get("compression_log_every", 5)) == 0)
        val = validate(model, val_ds, config, device, compute_compression=do_comp, b=b, special_tokens=special_tokens,
                       world=(rank, nranks))
        if not main_rank:
            continue
        val_loss, val_recon, val_kl = val[:3]
        if do_comp and len(val) == 8:
            tracker.log({"val/compression_bits": val[4], "val/compression_kl_bits": val[5],
                         "val/compression_edge_bits": val[6], "val/compression_entity_bits": val[7]})
            if math.isfinite(float(val[4])) and float(val[4]) < best_comp_bits:
                best_comp_bits = float(val[4])
        tracker.log({"objective": best_comp_bits})
        log = {"epoch": epoch + 1, "train/loss": train_loss, "train/reconstruction_loss": train_recon,
               "val/loss": val_loss, "val/reconstruction_loss": val_recon, "learning_rate": lr,
               "train/graphs_per_sec": (len(train_ds) // config["batch_size"]) * config["batch_size"] / max(dt, 1e-9)}
        if model_type in ("SAIL", "t-SAIL"):
            log["train/kl_loss"], log["val/kl_loss"] = train_kl, val_kl

        if verifier and (epoch + 1) % config.get("verify_every", 10) == 0:
            target_n = config.get("num_generated_latent_graphs", 1000)
            if model_type in ("SAIL", "t-SAIL"):
                zs = torch.randn(target_n, config["d_latent"], device=device)
                graphs = model.decode_latent(zs, seq_len, special_tokens, seq_to_triples, ENT_BASE, REL_BASE, beam=1)
            else:
                chunks = []
                while sum(c.size(0) for c in chunks) < target_n:
                    chunks.append(model.generate(seq_len, special_tokens, device=device, batch_size=50, beam=1, sample=True,
                                                 temperature=config.get("temperature", 1.0), top_p=config.get("top_p", 0.9),
                                                 top_k=config.get("top_k", 0)).cpu())
                rows = torch.cat(chunks, 0)[:target_n]
                graphs = [seq_to_triples(r, special_tokens, ENT_BASE, REL_BASE) for r in rows]
            labels = ints_to_labels(graphs, i2e, i2r)
            ev = run_semantic_evaluation(labels, train_g, i2e, i2r, verifier, title=f"{model_type} samples")
            res = ev.organized_results["results"]
            tracker.log({"verification/validity_rate": res.get("semantics", 0.0) / 100.0,
                         "verification/novelty_rate": res.get("novel", 0.0) / 100.0,
                         "verification/valid_novelty_rate": res.get("novel_semantics", 0.0) / 100.0})
        tracker.log(log)
        print(f"Train Loss: {train_loss:.4f} (Recon: {train_recon:.4f})  [{log['train/graphs_per_sec']:.0f} graphs/s]")
        print(f"Val   Loss: {val_loss:.4f} (Recon: {val_recon:.4f})")

        if val_loss < best_val_loss:
            best_val_loss = val_loss
            save_checkpoint(os.path.join(run_dir, f"{dataset_name}_{model_type}_best_model.pt"), epoch + 1, model, config,
                            val_loss, vocabs, dataset_meta, base_lr, use_sched, best_val_loss)
            print(f"Saved best model with validation loss: {val_loss:.4f}")
        if (epoch + 1) % config.get("save_every", 10) == 0:
            save_checkpoint(os.path.join(run_dir, f"{dataset_name}_{model_type}_checkpoint_epoch_{epoch + 1}.pt"), epoch + 1,
                            model, config, val_loss, vocabs, dataset_meta, base_lr, use_sched, best_val_loss)

    if main_rank:
        final = {}
        for split, ds in (("val", val_ds),) + ((("test", test_ds),) if config.get("use_test_for_final_eval", False) else ()):
            r = validate(model, ds, config, device, compute_compression=True, b=1.0, special_tokens=special_tokens)
            final.update({f"final_{split}/loss": r[0], f"final_{split}/reconstruction_loss": r[1],
                          f"final_{split}/kl_loss": r[2], f"final_{split}/compression_bits": r[4]})
        tracker.log(final)
        tracker.finish()
        print("\nTraining and evaluation completed!")
    if config.get("dump_final_params"):   # test hook: every rank's flat parameter buffer
        eng.dp_flush()
        torch.save(eng.P.detach().cpu(), f"{config['dump_final_params']}.rank{rank}.pt")
        Bl = config["batch_size"] // nranks   # (workspace of this rank's training batches: the GRU engine keys by batch
        wtrain = eng._ws_cache.get(Bl)        #  size, the Transformer engine by (batch, decoder length, triples))
        if wtrain is None:
            wtrain = next((w for k, w in eng._ws_cache.items() if isinstance(k, tuple) and k[0] == Bl and "eps0" in w), None)
        if model_type in ("SAIL", "t-SAIL") and wtrain is not None:   # the last training step's device-drawn latent noise
            torch.save(wtrain["eps0"].detach().cpu(), f"{config['dump_final_params']}.eps.rank{rank}.pt")
    if nranks > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
