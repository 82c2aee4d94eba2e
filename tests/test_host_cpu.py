"""CPU-only checks of the host side: C-ABI surface, parameter layout / state-dict compatibility,
init-order compatibility, data codec, config schema, and loud failure without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return z, json.loads(str(z["cfg_json"]))


def test_library_loads_and_exports_every_declared_symbol():
    """every function declared in include/ark_amd.h is exported by the built library"""
    hdr = open(os.path.join(ROOT, "include", "ark_amd.h")).read()
    names = sorted(set(re.findall(r"^(?:int|void)\s+(ark_\w+)\s*\(", hdr, flags=re.M)))
    assert len(names) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, "ark_amd", "lib", "libark_amd.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.ark_version() >= 100


def test_argument_errors_are_negative_codes_without_touching_the_gpu():
    lib = ctypes.CDLL(os.path.join(ROOT, "ark_amd", "lib", "libark_amd.so"))
    wt = (ctypes.c_int * 4)(96, 2, 300, 1)                    # ArkWgradTuning with a tile size that does not exist
    one = ctypes.c_void_p(256)                                # never dereferenced: the argument check comes first
    assert lib.ark_wgrad16(2, one, ctypes.c_int64(128), one, ctypes.c_int64(128), one, ctypes.c_int64(128), 128, 128, 256, wt,
                           ctypes.c_void_p(0)) < 0
    bad = (ctypes.c_int * 11)(48, 1, 2, 1, 32, 32, 2, 2, 4, 0, 0)   # ArkDiagTuning with a forward row count that does not exist
    role = (ctypes.c_char * 256)()
    assert lib.ark_gru_diag_fwd(2, 1, 1, role, ctypes.c_void_p(0), 64, 128, bad, ctypes.c_void_p(0)) < 0
    null = ctypes.c_void_p(0)
    assert lib.ark_adam_step(null, null, null, null, ctypes.c_int64(0), null, null) < 0


def test_vocabulary_split_count_follows_the_cu_budget():
    """ark_vocab_ce_fwd_splits is host arithmetic: the split count times the row tiles stays within the CUs the caller
    leaves to the cross-entropy (the chunk pipeline passes the CUs a resident sweep does not hold), and row counts that
    fill the chip on their own are not split"""
    lib = ctypes.CDLL(os.path.join(ROOT, "ark_amd", "lib", "libark_amd.so"))
    f = lib.ark_vocab_ce_fwd_splits
    # wd-articles: 16 sequences x 259 steps, V = 60 943, D = 512; one quarter of the steps as a chunk
    whole = f(16 * 259, 60943, 512, 0)
    assert 2 <= whole <= 16 and (16 * 259 + 63) // 64 * whole <= 256 * 2
    chunk_rows = 16 * 65
    for budget in (256, 160, 96, 32):
        nv = f(chunk_rows, 60943, 512, budget)
        assert 1 <= nv <= 16
        assert nv == 1 or (60943 + 63) // 64 // nv >= 32          # a split keeps at least 32 vocabulary tiles
    assert f(chunk_rows, 60943, 512, 96) <= f(chunk_rows, 60943, 512, 256)   # a smaller budget never splits finer
    assert f(1024 * 10, 55, 512, 0) == 1 and f(256 * 64, 60943, 512, 0) == 1   # small vocabulary / chip-filling rows
    assert f(chunk_rows, 60943, 96, 0) == 1 and f(chunk_rows + 3, 60943, 512, 0) == 1   # shapes the fused kernel lacks


@pytest.mark.parametrize("name", ["sail_tiny", "sail_tiny_pad", "ark_tiny"])
def test_state_dict_keys_order_and_init_match_reference(name):
    """same keys in the same order as the reference's state_dict, same initial values under the seed"""
    from kgvae.model.models import ARK, SAIL
    z, cfg = _golden(name)
    ref_keys = [k[3:] for k in z.files if k.startswith("w0/")]
    torch.manual_seed(int(z["seed"]))
    model = (SAIL if cfg["model_type"] == "SAIL" else ARK)(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert np.array_equal(sd[k].numpy(), z["w0/" + k]), k
    assert sd["dec.out.weight"].data_ptr() == sd["dec.tok_emb.weight"].data_ptr()


def test_param_layout_is_aligned_and_heads_are_adjacent():
    from ark_amd.engine import ParamLayout
    _, cfg = _golden("sail_tiny")
    lay = ParamLayout(cfg)
    e = lay.entries
    for name, (off, shape, numel) in e.items():
        assert numel == int(np.prod(shape))
    assert e["enc.logv.weight"][0] == e["enc.mu.weight"][0] + e["enc.mu.weight"][2]   # one [2Z,3D] block
    assert e["enc.logv.bias"][0] == e["enc.mu.bias"][0] + e["enc.mu.bias"][2]
    assert e["enc.mu.weight"][0] % 4 == 0 and lay.total % 4 == 0
    assert "dec.out.weight" not in e   # tied to dec.tok_emb.weight
    # a constant stride between layers lets one launch refresh every layer's weight shadows
    assert (e["dec.gru.weight_hh_l2"][0] - e["dec.gru.weight_hh_l1"][0]) == (e["dec.gru.weight_hh_l1"][0] - e["dec.gru.weight_hh_l0"][0])


def test_model_refuses_to_run_on_cpu():
    from ark_amd import ArkError
    from kgvae.model.models import SAIL
    _, cfg = _golden("sail_tiny")
    m = SAIL(cfg)
    with pytest.raises(ArkError):
        m(torch.zeros(2, 3, 3, dtype=torch.long), torch.ones(2, 10, dtype=torch.long))
    t = SAIL(dict(cfg, model_type="t-SAIL"))   # the Transformer VAE is built too (same keys as the reference's state dict)
    assert "enc.txf.layers.0.self_attn.in_proj_weight" in t.state_dict() and "dec.txf.layers.0.multihead_attn.in_proj_bias" in t.state_dict()
    with pytest.raises(ArkError):
        t(torch.zeros(2, 3, 3, dtype=torch.long), torch.ones(2, 10, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        SAIL(dict(cfg, model_type="nope"))
    with pytest.raises(NotImplementedError):   # (model-type families as in the reference: ARK builds ARK / t-ARK, models.py:372-393)
        SAIL(dict(cfg, model_type="ARK"))
    with pytest.raises(KeyError):
        SAIL({"model_type": "SAIL"})


def test_codec_and_dataset_match_reference_vectors():
    from kgvae.model import utils as U
    st = {"PAD": 0, "BOS": 1, "EOS": 2}
    for c in json.load(open(os.path.join(GOLD, "codec.json"))):
        if c["triples"] is not None:
            assert U.triples_to_seq([tuple(t) for t in c["triples"]], st, 3, 13, len(c["seq"])).tolist() == c["seq"]
        assert [list(t) for t in U.seq_to_triples(torch.tensor(c["seq"]), st, 3, 13)] == c["decoded"]
    # golden batches were produced by the reference's GraphSeqDataset: rebuild them from the triples
    for name in ["sail_tiny", "sail_tiny_pad"]:
        z, cfg = _golden(name)
        tri = z["triples"]
        graphs = []
        for b in range(tri.shape[0]):
            graphs.append([tuple(int(x) for x in t) for t in tri[b] if cfg["pad_rid"] is None or t[1] != cfg["pad_rid"]])
        ds = U.GraphSeqDataset(graphs, None, None, use_padding=cfg["pad_rid"] is not None, pad_eid=cfg["pad_eid"],
                               pad_rid=cfg["pad_rid"], max_triples=cfg["max_triples"], special_tokens=st,
                               ent_base=cfg["ENT_BASE"], rel_base=cfg["REL_BASE"], seq_len=cfg["seq_len"])
        t, s = ds.tensorize()
        assert np.array_equal(t.numpy(), tri) and np.array_equal(s.numpy(), z["seq"])
        t0, s0 = ds[0]
        assert np.array_equal(t0.numpy(), tri[0]) and np.array_equal(s0.numpy(), z["seq"][0])
    assert U.canonical_graph_string([(2, 0, 1), (1, 0, 2)]) == U.canonical_graph_string([(1, 0, 2), (2, 0, 1)])
    assert U.ints_to_labels([[(0, 0, 1), (9, 0, 1)]], {0: "a", 1: "b"}, {0: "r"}) == [[("a", "r", "b")]]


def test_shipped_configs_carry_the_reference_schema():
    keys = {"model_type", "d_model", "d_latent", "n_heads", "n_layers", "batch_size", "learning_rate", "num_epochs", "beta0",
            "beta1", "dataset", "shuffle_train", "use_padding", "triple_order", "permute_triples", "num_diversity_samples",
            "num_generated_test_graphs", "num_generated_latent_graphs", "sample_frac", "beam_width", "lr_scheduler",
            "save_every", "resume_from_checkpoint", "checkpoint_path", "verify_every", "experiment_name",
            "use_test_for_final_eval", "compression_log_every"}
    cdir = os.path.join(ROOT, "configs")
    files = [f for f in os.listdir(cdir) if f.endswith(".yaml")]
    assert len(files) == 5
    for f in files:
        cfg = yaml.safe_load(open(os.path.join(cdir, f)))
        assert keys <= set(cfg), (f, keys - set(cfg))
        assert cfg["model_type"] in ("SAIL", "ARK")


def test_synthetic_dataset_has_the_intelligraphs_tuple_shape():
    from ark_amd.datasets import synthetic_as_list
    tr, va, te, (e2i, i2e), (r2i, i2r), (lo, hi), _ = synthetic_as_list("wd-movies", n_train=20, n_val=5, n_test=5)
    assert len(tr) == 20 and len(va) == 5 and len(te) == 5
    assert all(lo <= len(g) <= hi for g in tr)
    assert all(0 <= h < len(e2i) and 0 <= r < len(r2i) and 0 <= t < len(e2i) for g in tr for h, r, t in g)
    assert i2e[e2i["e3"]] == "e3" and i2r[r2i["r1"]] == "r1"


def test_schedules_match_torch():
    from kgvae.experiments.train import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-6)
    for ep in range(10):
        assert abs(opt.param_groups[0]["lr"] - cosine_lr(1e-3, ep, 10, 1e-6)) < 1e-10
        opt.step()
        sch.step()


def test_epoch_batches_shards_and_counts():
    """train.py's host-side epoch preparation agrees with dp.shard / dp.count_targets batch by batch"""
    import torch
    from ark_amd import dp
    from kgvae.experiments.train import epoch_batches
    g = torch.Generator().manual_seed(0)
    n, B, L, T = 37, 8, 7, 3
    seq = torch.randint(0, 9, (n, L), generator=g)
    tri = torch.randint(0, 5, (n, T, 3), generator=g)
    full_t, full_s, counts = epoch_batches(tri, seq, B, False, 0)
    assert full_s.shape == (n // B, B, L) and full_t.shape == (n // B, B, T, 3) and len(counts) == n // B
    for r in range(4):
        t_r, s_r, c_r = epoch_batches(tri, seq, B, False, 0, rank=r, nranks=4)
        assert c_r == counts
        for i in range(n // B):
            assert torch.equal(s_r[i], dp.shard(seq[i * B:(i + 1) * B], r, 4))
            assert torch.equal(t_r[i], dp.shard(tri[i * B:(i + 1) * B], r, 4))
            assert counts[i] == dp.count_targets(seq[i * B:(i + 1) * B], 0)
    t_n, s_n, c_n = epoch_batches(None, seq, B, True, 0)
    assert t_n is None and s_n.shape == full_s.shape and sum(c_n) <= int((seq[:, 1:] != 0).sum())


def test_sample_perms_is_random_sample_bit_for_bit():
    """kgvae.model.utils.sample_perms == n successive random.sample(range(T), T) calls on Python's global
    generator: same permutations, same generator state afterwards (so everything drawn later matches too)"""
    import random
    import numpy as np
    from kgvae.model.utils import sample_perms
    for T in (2, 3, 4, 6, 23):
        for seed in (0, 7):
            random.seed(seed)
            for _ in range(seed):
                random.random()
            st = random.getstate()
            ref = np.array([random.sample(range(T), T) for _ in range(777)])
            after = random.getstate()
            random.setstate(st)
            got = sample_perms(777, T)
            assert (ref == got).all(), (T, seed)
            assert random.getstate() == after, (T, seed)
    assert sample_perms(0, 3).shape == (0, 3) and sample_perms(4, 1).tolist() == [[0]] * 4


def test_shipped_configs_are_the_reference_yamls_with_model_type_overridden():
    """the benchmark configs are 'the shipped YAML with model_type overridden'.  Every key of the
    reference's five YAMLs (tests/golden/reference_yaml_values.json, parsed by tools/make_golden_r3.py) has the same value
    in configs/sail_<dataset>.yaml, except the model type, the run's own names / paths and the added `precision`"""
    import json
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "reference_yaml_values.json")))
    own = {"model_type", "checkpoint_path", "experiment_name"}
    assert set(ref) == {"syn-paths", "syn-types", "syn-tipr", "wd-movies", "wd-articles"}
    for name, want in ref.items():
        got = yaml.safe_load(open(os.path.join(root, "configs", f"sail_{name}.yaml")))
        assert set(got) - set(want) <= {"precision"}, (name, set(got) - set(want))
        for k, v in want.items():
            if k not in own:
                assert got[k] == v, (name, k, got[k], v)
        assert got["model_type"] == "SAIL"


def test_ranks_sharing_a_device_are_counted_from_the_launcher_environment(monkeypatch):
    """Engine admits a persistent sweep only if the grids of all ranks on ITS device fit the chip together; the count comes from
    torchrun's LOCAL_WORLD_SIZE and the visible devices (one process per GPU on a real node -> 1)"""
    import torch
    from ark_amd import engine as E
    assert E._ranks_on_this_device(1) == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert E._ranks_on_this_device(8) == 1            # a real 8-GPU node
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert E._ranks_on_this_device(8) == 8            # eight ranks rehearsed on one card
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert E._ranks_on_this_device(2) == 2            # the two-process tests
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert E._ranks_on_this_device(2) == 2            # (spawned without torchrun: every rank is local)
