import os, sys, yaml, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from kgvae.experiments import train as T
cfg = yaml.safe_load(open(os.path.join(root, "configs", "sail_wd-articles.yaml")))
cfg.update(num_epochs=3, save_every=100, compression_log_every=100, verify_every=100, num_generated_test_graphs=4,
           num_generated_latent_graphs=4, num_diversity_samples=4,
           synthetic_sizes={"n_train": 640, "n_val": 64, "n_test": 64})
out = os.path.join(root, "gpurun_out", "articles_run")
os.makedirs(out, exist_ok=True)
cp = os.path.join(out, "c.yaml")
yaml.safe_dump(cfg, open(cp, "w"))
t0 = time.time()
T.main(["--config", cp, "--checkpoint-dir", os.path.join(out, "ck")])
print("done in", round(time.time() - t0, 1), "s")
