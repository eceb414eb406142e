# -*- coding: utf-8 -*-
"""A small fit loop standing in for pytorch_lightning.Trainer as the reference drives it
(run.py:191-218): one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torchrun, backend "nccl" =
RCCL), epoch hooks in Lightning's order, gradient clipping value forwarded to the fused step, and
Lightning-layout checkpoints:
    {"epoch", "global_step", "state_dict": {"model.<key>": tensor}, "optimizer_states": [AdamW state_dict],
     "lr_schedulers": [...], "hyper_parameters": {...}}
so a checkpoint written here loads through the reference's warm-start code (it strips the "model." prefix,
run.py:63-69) and vice versa."""
import os
from typing import List, Optional

import torch


class ModelCheckpoint:
    def __init__(self, dirpath, filename="epoch{epoch:03d}", every_n_epochs=1, save_last=True, save_top_k=-1, verbose=False):
        self.dirpath, self.filename = str(dirpath), filename
        self.every_n_epochs, self.save_last = int(every_n_epochs), bool(save_last)

    def on_epoch_end(self, trainer, epoch):
        if trainer.global_rank != 0:
            return
        os.makedirs(self.dirpath, exist_ok=True)
        if self.every_n_epochs > 0 and (epoch + 1) % self.every_n_epochs == 0:
            name = self.filename.format(epoch=epoch) if "{" in self.filename else self.filename
            trainer.save_checkpoint(os.path.join(self.dirpath, name + ".ckpt"))
        if self.save_last:
            trainer.save_checkpoint(os.path.join(self.dirpath, "last.ckpt"))


class ScalarLogger:
    """Stand-in for the reference's TensorBoardLogger (run.py:166-170): the scalars the harness logs (train/*, val/*, epoch/*,
    lr) appended as JSON lines to <save_dir>/<name>/version_<n>/scalars.jsonl by rank 0.  No tensorboard dependency."""

    def __init__(self, save_dir, name):
        self.root = os.path.join(str(save_dir), str(name))
        self.path = None

    def _open(self):
        if self.path is None:
            os.makedirs(self.root, exist_ok=True)
            taken = [int(d.split("_")[1]) for d in os.listdir(self.root) if d.startswith("version_") and d.split("_")[1].isdigit()]
            vdir = os.path.join(self.root, f"version_{max(taken) + 1 if taken else 0}")
            os.makedirs(vdir, exist_ok=True)
            self.path = os.path.join(vdir, "scalars.jsonl")
        return self.path

    def log_metrics(self, metrics, step):
        import json
        rec = {"step": int(step)}
        rec.update({k: float(v) for k, v in metrics.items()})
        with open(self._open(), "a") as f:
            f.write(json.dumps(rec) + "\n")


class Trainer:
    def __init__(self, max_epochs=1, gradient_clip_val=0.0, callbacks: Optional[List] = None, limit_val_batches=1.0,
                 limit_train_batches=1.0, accelerator="gpu", devices=1, strategy="ddp", logger=None, **ignored):
        self.max_epochs = int(max_epochs)
        self.gradient_clip_val = float(gradient_clip_val or 0.0)
        self.callbacks = list(callbacks or [])
        self.limit_val_batches, self.limit_train_batches = limit_val_batches, limit_train_batches
        self.accelerator, self.devices, self.strategy = accelerator, devices, strategy
        self.logger = logger
        self.ckpt_path = None
        self.current_epoch, self.global_step = 0, 0
        self.global_rank = int(os.environ.get("RANK", 0))
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.experiment = None
        self.optimizers = []
        self.callback_metrics = {}

    # ---- distributed bootstrap: one rank per GPU, RCCL via torch.distributed ----------------------------
    def _init_distributed(self):
        if self.world_size > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = "nccl" if (self.accelerator == "gpu" and torch.cuda.is_available()) else "gloo"
            backend = os.environ.get("VQH_DIST_BACKEND", backend)     # tests: several ranks sharing one GPU need gloo
            torch.distributed.init_process_group(backend, rank=self.global_rank, world_size=self.world_size)
        if self.accelerator == "gpu" and torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())

    def _limit(self, n, lim):
        return max(1, int(n * lim)) if isinstance(lim, float) and lim <= 1.0 else min(n, int(lim))

    def fit(self, experiment, ckpt_path: Optional[str] = None):
        self._init_distributed()
        exp = self.experiment = experiment
        exp.trainer, exp.global_rank = self, self.global_rank
        self.ckpt_path = ckpt_path
        if self.accelerator == "gpu":
            if not torch.cuda.is_available():
                raise RuntimeError("Trainer(accelerator='gpu'): no MI355X visible -- the HIP training step has no CPU fallback")
            exp.model = exp.model.to(torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()))
        exp.setup("fit")
        policy = exp.configure_optimizers()
        start_epoch = 0
        if ckpt_path:
            start_epoch = self.load_checkpoint(ckpt_path) + 1
        exp.current_epoch = start_epoch
        exp.on_fit_start()
        train_loader, val_loader = exp.train_dataloader(), exp.val_dataloader()
        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = exp.current_epoch = epoch
            if hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            exp.model.train()
            exp.on_train_epoch_start()
            nb = self._limit(len(train_loader), self.limit_train_batches)
            for i, batch in enumerate(train_loader):
                if i >= nb:
                    break
                exp.training_step(batch, i)
                self.global_step += 1
            if val_loader is not None and len(val_loader) > 0 and self.limit_val_batches:
                exp.model.eval()
                exp.on_validation_epoch_start()
                nv = self._limit(len(val_loader), self.limit_val_batches)
                for i, batch in enumerate(val_loader):
                    if i >= nv:
                        break
                    exp.validation_step(batch, i)
                exp.on_validation_epoch_end()
                exp.model.train()
            exp.on_train_epoch_end()
            if self.logger is not None and self.global_rank == 0 and getattr(exp, "logged", None):
                rec = dict(exp.logged, epoch=epoch)
                if policy is not None:
                    rec["lr"] = policy.current()[0]            # LearningRateMonitor(logging_interval="epoch"), run.py:186
                self.logger.log_metrics(rec, self.global_step)
            if policy is not None:
                policy.on_epoch()
            for cb in self.callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(self, epoch)
        return exp

    # ---- checkpoints ---------------------------------------------------------------------------------
    def save_checkpoint(self, path):
        exp = self.experiment
        sd = {"model." + k: v.detach().cpu().clone() for k, v in exp.model.state_dict().items()}
        ckpt = {"epoch": int(self.current_epoch), "global_step": int(self.global_step),
                "pytorch-lightning_version": "1.9.0", "state_dict": sd, "hyper_parameters": dict(exp.hparams),
                "optimizer_states": [exp.model.optimizer_state()] if hasattr(exp.model, "optimizer_state") else [],
                "lr_schedulers": [exp.lr_policy.state_dict()] if exp.lr_policy is not None else []}
        if hasattr(exp.model, "engine_state"):
            ckpt["vqh_engine"] = exp.model.engine_state()      # dropout counter [seed, step]: masks continue after a resume
        tmp = path + ".tmp"
        torch.save(ckpt, tmp)
        os.replace(tmp, path)

    def load_checkpoint(self, path):
        exp = self.experiment
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        state = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        exp.model.load_state_dict(state, strict=True)
        if ckpt.get("optimizer_states") and hasattr(exp.model, "load_optimizer_state"):
            exp.model.load_optimizer_state(ckpt["optimizer_states"][0])
        if ckpt.get("lr_schedulers") and exp.lr_policy is not None:
            exp.lr_policy.load_state_dict(ckpt["lr_schedulers"][0])
        if ckpt.get("vqh_engine") is not None and hasattr(exp.model, "load_engine_state"):
            exp.model.load_engine_state(ckpt["vqh_engine"])
        self.global_step = exp.global_step = int(ckpt.get("global_step", 0))
        return int(ckpt.get("epoch", -1))
