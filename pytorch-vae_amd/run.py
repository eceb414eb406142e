# -*- coding: utf-8 -*-
"""
Two-stage VQ-VAE training entry on MI355X -- same command line as the reference's run.py:97-106.

  python run.py -c configs/stage1_ae.yaml                                   # stage 1: AE pre-training
  python run.py -c configs/stage2_vq.yaml --warm_start_ckpt S1.ckpt --init_codebook centroids.npy
  python run.py -c configs/stage2_vq.yaml --resume_ckpt checkpoints/.../last.ckpt

Multi-GPU like the reference (Lightning `devices: N, strategy: ddp`, run.py:191-218, configs/stage2_vq.yaml:209-212): with
trainer_params.devices > 1 this one command starts N ranks itself (child processes, one per MI355X, RCCL over xGMI).
Running it under torchrun works as well (the launcher's RANK / WORLD_SIZE win over `devices`):
  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run.py -c configs/stage2_vq.yaml
"""
import argparse
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch
import yaml

from experiment import VQVAEExperiment
from trainer import ModelCheckpoint, ScalarLogger, Trainer


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


def resolve_devices(trainer_params) -> int:
    """trainer_params.devices as a rank count: int, "auto" / -1 (every visible GPU), a list of device ids, or a digit string.
    Capped to the GPUs present (with a notice) unless the ranks are told to share devices over gloo (VQH_DIST_BACKEND=gloo:
    the two-ranks-on-one-GPU tests)."""
    from vqvae_hip import launch
    dev = trainer_params.get("devices", 1)
    ngpu = launch.visible_gpus()
    if isinstance(dev, (list, tuple)):
        n = len(dev)
    elif isinstance(dev, str):
        n = ngpu if dev.strip().lower() in ("auto", "-1") else int(dev)
    else:
        n = int(dev)
        if n < 0:
            n = ngpu
    n = max(1, n)
    if str(trainer_params.get("accelerator", "gpu")) == "gpu" and os.environ.get("VQH_DIST_BACKEND", "nccl") == "nccl" \
            and ngpu >= 1 and n > ngpu:
        print(f"[Launch] trainer_params.devices={n} but only {ngpu} GPU(s) are visible: using {ngpu}")
        n = ngpu
    return n


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

    # `devices: N` + `strategy: ddp`: one process per GPU.  The parent has not touched the GPU yet; it starts the ranks as
    # children (never a re-exec) and leaves with their exit code.  Inside a rank (RANK set) this block is skipped.
    from vqvae_hip import launch
    if not launch.under_launcher():
        n_ranks = resolve_devices(trainer_params)
        if n_ranks > 1:
            print(f"[Launch] starting {n_ranks} ranks (strategy={trainer_params.get('strategy', 'ddp')}, one process per GPU)", flush=True)
            raise SystemExit(launch.spawn_ranks(n_ranks, os.path.abspath(__file__), sys.argv[1:]))

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
    logger_name = logging_params.get("name", model_params.get("name", "VQVAE")) + ("-resume" if resume else "")
    logger = ScalarLogger(logging_params.get("save_dir", "./logs"), logger_name)
    clip = trainer_params.pop("gradient_clip_val", 5.0)
    trainer = Trainer(logger=logger, callbacks=[ckpt_cb], gradient_clip_val=clip, **trainer_params)

    print("======= Training {} =======".format(model_params.get("name", "VQVAE")))
    print("use_vq =", model_params.get("use_vq", True))
    t0 = time.time()
    trainer.fit(experiment, ckpt_path=args.resume_ckpt if resume else None)
    print(f"[Done] Training completed in {(time.time() - t0) / 60:.2f} minutes.")
    print(f"[Checkpoint dir] {str(ckpt_dir.resolve())}")


if __name__ == "__main__":
    main()
