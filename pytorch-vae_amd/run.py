# -*- coding: utf-8 -*-
"""
Two-stage VQ-VAE training entry on MI355X -- same command line as the reference's run.py:97-106.

  python run.py -c configs/stage1_ae.yaml                                   # stage 1: AE pre-training
  python run.py -c configs/stage2_vq.yaml --warm_start_ckpt S1.ckpt --init_codebook centroids.npy
  python run.py -c configs/stage2_vq.yaml --resume_ckpt checkpoints/.../last.ckpt
  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run.py -c configs/stage2_vq.yaml   # 8 x MI355X, RCCL
"""
import argparse
import os
import random
import time
from pathlib import Path

import numpy as np
import torch
import yaml

from experiment import VQVAEExperiment
from trainer import ModelCheckpoint, Trainer


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)


def maybe_warm_start(model, ckpt_path):
    """Load `model.*` weights of a previous run (stage 1 -> stage 2); not a resume (reference run.py:55-71)."""
    if not ckpt_path or not os.path.isfile(ckpt_path):
        print(f"[Warm-start] skipped (no valid ckpt at {ckpt_path})")
        return
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    state = ckpt.get("state_dict", ckpt)
    stripped = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    missing, unexpected = model.load_state_dict(stripped, strict=False)
    print(f"[Warm-start] Loaded weights from {ckpt_path}: missing={len(missing)}, unexpected={len(unexpected)}")


def maybe_init_codebook(model, path):
    if not path or not os.path.isfile(path):
        print(f"[Codebook init] skipped (invalid path: {path})")
        return
    C = torch.from_numpy(np.load(path).astype(np.float32))
    model.init_codebook_from_centroids(C)
    print(f"[Codebook init] Loaded centroids {tuple(C.shape)} from {path}")


def main():
    ap = argparse.ArgumentParser(description="Train VQ-VAE (two-stage compatible) on MI355X.")
    ap.add_argument("--config", "-c", type=str, required=True, help="Path to YAML config file.")
    ap.add_argument("--warm_start_ckpt", type=str, default="", help="Stage-1 checkpoint for warm start (ignored on resume).")
    ap.add_argument("--init_codebook", type=str, default="", help=".npy centroids for the codebook (ignored on resume).")
    ap.add_argument("--resume_ckpt", type=str, default="", help="Resume model + optimizer + schedule + epoch.")
    args = ap.parse_args()

    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    model_params, exp_params, data_params = cfg["model_params"], cfg["exp_params"], cfg["data_params"]
    trainer_params = dict(cfg.get("trainer_params", {}))
    logging_params = cfg.get("logging_params", {})

    seed = exp_params.get("manual_seed", 42)
    seed_everything(seed)
    print(f"[Seed] manual_seed={seed}")
    experiment = VQVAEExperiment(model_params, exp_params, data_params)
    model = experiment.model

    resume = bool(args.resume_ckpt)
    if resume:
        if not os.path.isfile(args.resume_ckpt):
            raise FileNotFoundError(f"[Resume] ckpt not found: {args.resume_ckpt}")
        print(f"[Resume] Will resume full state from: {args.resume_ckpt}")
        experiment.exp_params["warm_start_ckpt"] = ""
        experiment._warm_start_ckpt = None
    else:
        warm = args.warm_start_ckpt or exp_params.get("warm_start_ckpt", "")
        if warm:
            experiment._warm_start_ckpt = warm      # on_fit_start re-applies it with quantizer.* dropped
            try:
                maybe_warm_start(model, warm)
            except Exception as e:
                print(f"[Warm-start] failed: {e}")
        else:
            print("[Warm-start] skipped (no warm_start_ckpt provided).")
        book = args.init_codebook or model_params.get("codebook_init_path", "")
        if model_params.get("use_vq", True) and book:
            experiment._init_codebook_path = book
            try:
                maybe_init_codebook(model, book)
            except Exception as e:
                print(f"[Codebook init] failed: {e}")
        else:
            print("[Codebook init] skipped (use_vq=False or no path provided).")

    ckpt_dir = Path(exp_params.get("checkpoint_dir", "./checkpoints/aeot_sigmoid"))
    ckpt_cb = ModelCheckpoint(dirpath=str(ckpt_dir), filename=exp_params.get("checkpoint_name_pattern", "epochepoch={epoch:03d}"),
                              every_n_epochs=int(exp_params.get("save_every_epochs", 10)), save_last=True, save_top_k=-1)
    clip = trainer_params.pop("gradient_clip_val", 5.0)
    trainer = Trainer(callbacks=[ckpt_cb], gradient_clip_val=clip, **trainer_params)

    print("======= Training {} =======".format(model_params.get("name", "VQVAE")))
    print("use_vq =", model_params.get("use_vq", True))
    t0 = time.time()
    trainer.fit(experiment, ckpt_path=args.resume_ckpt if resume else None)
    print(f"[Done] Training completed in {(time.time() - t0) / 60:.2f} minutes.")
    print(f"[Checkpoint dir] {str(ckpt_dir.resolve())}")


if __name__ == "__main__":
    main()
