"""`python -m kgvae.experiments.ablation_study --config <yaml>` -- the reference's VAE training script
(reference kgvae/experiments/ablation_study.py:348-797).  Its CLI, config keys and checkpoint layout
are a superset of train.py's and the SAIL / ARK dispatch it adds is already part of this repository's
kgvae.experiments.train, so this module is the same entry point under the reference's second name."""
from kgvae.experiments.train import cosine_lr, iterate_batches, main, save_checkpoint, train_epoch, validate  # noqa: F401

if __name__ == "__main__":
    main()
