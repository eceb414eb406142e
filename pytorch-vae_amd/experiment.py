# -*- coding: utf-8 -*-
"""
Training harness around the HIP model: the host-side mirror of the reference's
VQVAEExperiment (experiment.py:49-502) without pytorch_lightning.  Same constructor
(model_params, exp_params, data_params), same hook names and the same epoch-schedule semantics, so
`run.py` and the YAML configs drive it unchanged.  It performs no tensor arithmetic of its own:
training_step hands the batch to VQVAE.train_step (one fused GPU step, hipGraph replayed).

`model_cls` exists for the CPU plumbing tests, which inject a stand-in with the same train_step /
eval_step interface; the product always uses models.VQVAE (GPU only).
"""
import os
from typing import Dict, List, Optional, Tuple

import torch
import yaml
from torch.utils.data import DataLoader

from dataset import CurveDataset, SyntheticCurveDataset, pad_collate

WEIGHT_DEFAULTS = [("ss_weight", 1.0), ("bond_length_weight", 0.0), ("bond_angle_weight", 0.0), ("xyz_tv_lambda", 0.0),
                   ("dir_weight", 0.0), ("dih_weight", 0.0), ("rmsd_weight", 1.0), ("pdm_weight", 0.0),
                   ("win_kabsch_weight", 0.0), ("kappa_weight", 0.0), ("tau_weight", 0.0), ("lr_pdm_weight", 0.0),
                   ("pdm_window", 8), ("win_kabsch_size", 16), ("win_kabsch_stride", 8), ("lr_min_sep", 24),
                   ("lr_stride", 8), ("lr_max_offsets", 8)]
INT_KEYS = ["pdm_window", "win_kabsch_size", "win_kabsch_stride", "lr_min_sep", "lr_stride", "lr_max_offsets"]
LOSS_KWARGS = [k for k, _ in WEIGHT_DEFAULTS]


def interpolate_schedule(schedules: Dict[str, List[List[float]]], epoch: int) -> Dict[str, float]:
    """Piecewise-linear [epoch, value] knots -> value at `epoch` (reference experiment.py:14-34):
    constant before the first knot and after the last one."""
    out: Dict[str, float] = {}
    for key, knots in (schedules or {}).items():
        if not knots:
            continue
        if epoch <= knots[0][0]:
            out[key] = float(knots[0][1])
            continue
        val = float(knots[-1][1])
        for (e0, v0), (e1, v1) in zip(knots[:-1], knots[1:]):
            if e0 <= epoch < e1:
                val = float(v0 + (epoch - e0) / max(1e-8, (e1 - e0)) * (v1 - v0))
                break
        out[key] = val
    return out


def _clean_path(p):
    return None if (p is None or (isinstance(p, str) and not p.strip())) else p


class LRPolicy:
    """Per-step / per-epoch learning-rate (and Adam beta1) policy: torch's OneCycleLR (cos, cycle_momentum) and
    CosineAnnealingLR restated in closed form (reference experiment.py:169-197)."""

    def __init__(self, kind, base_lr, max_epochs=1, steps_per_epoch=1, pct_start=0.15, div_factor=25.0,
                 final_div=1500.0):
        self.kind, self.base_lr = kind, float(base_lr)
        self.max_epochs, self.total = int(max_epochs), max(1, int(max_epochs) * int(steps_per_epoch))
        self.pct_start, self.div, self.final_div = float(pct_start), float(div_factor), float(final_div)
        self.step_num, self.epoch_num = 0, 0
        self.manual_lr = None

    @staticmethod
    def _cos(start, end, pct):
        import math
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def current(self) -> Tuple[float, float]:
        """(lr, beta1) for the optimizer step about to run."""
        if self.manual_lr is not None:
            return self.manual_lr, 0.9
        if self.kind == "onecycle":
            init, mx = self.base_lr / self.div, self.base_lr
            mn = init / self.final_div
            e1 = float(self.pct_start * self.total) - 1.0
            e2 = float(self.total - 1)
            s = min(self.step_num, self.total - 1)
            if s <= e1:
                pct = s / e1 if e1 > 0 else 1.0
                return self._cos(init, mx, pct), self._cos(0.95, 0.85, pct)
            pct = (s - e1) / (e2 - e1) if e2 > e1 else 1.0
            return self._cos(mx, mn, pct), self._cos(0.85, 0.95, pct)
        if self.kind == "cosine":
            import math
            eta_min = self.base_lr * 1e-6
            t = min(self.epoch_num, self.max_epochs)
            return eta_min + (self.base_lr - eta_min) * (1 + math.cos(math.pi * t / max(1, self.max_epochs))) / 2, 0.9
        return self.base_lr, 0.9

    def on_step(self):
        self.step_num += 1

    def on_epoch(self):
        self.epoch_num += 1

    def state_dict(self):
        return {"step_num": self.step_num, "epoch_num": self.epoch_num, "manual_lr": self.manual_lr}

    def load_state_dict(self, sd):
        self.step_num, self.epoch_num = int(sd.get("step_num", 0)), int(sd.get("epoch_num", 0))
        self.manual_lr = sd.get("manual_lr")


class VQVAEExperiment:
    def __init__(self, model_params: dict, exp_params: dict, data_params: dict, model_cls=None):
        self.LR = float(exp_params.get("LR", 1e-3))
        self.weight_decay = float(exp_params.get("weight_decay", 0.0))
        self.manual_seed = int(exp_params.get("manual_seed", 42))
        self.hparams = {"LR": self.LR, "weight_decay": self.weight_decay, "manual_seed": self.manual_seed,
                        "model_name": model_params.get("name", "VQVAE")}
        if model_cls is None:
            from models import vae_models
            model_cls = vae_models["VQVAE"]
        self.model = model_cls(**model_params)
        self.exp_params, self.data_params, self.model_params = exp_params, data_params, model_params
        self._warm_start_ckpt = _clean_path(exp_params.get("warm_start_ckpt"))
        self._init_codebook_path = _clean_path(exp_params.get("init_codebook_path")) or \
            _clean_path(model_params.get("codebook_init_path"))
        self.schedules = exp_params.get("schedules", {}) or {}
        self.current_weights: Dict[str, float] = {k: float(exp_params.get(k, d)) for k, d in WEIGHT_DEFAULTS}
        self.current_weights.update(label_smoothing=float(model_params.get("label_smoothing", 0.0)),
                                    usage_entropy_lambda=float(model_params.get("usage_entropy_lambda", 0.0)),
                                    beta=float(model_params.get("beta", 0.25)))
        self.example_input_array = (torch.zeros(1, 64, 6), torch.ones(1, 64, dtype=torch.bool))
        torch.manual_seed(self.manual_seed)
        self.train_dataset = self.val_dataset = None
        self.trainer = None
        self.current_epoch, self.global_rank, self.global_step = 0, 0, 0
        self.lr_policy: Optional[LRPolicy] = None
        self.logged: Dict[str, float] = {}
        self._ep_n = 0
        self._val_sum, self._val_n = None, 0

    # ---- data (reference :122-153) -------------------------------------------------------------------
    def setup(self, stage: Optional[str] = None):
        dp = self.data_params
        syn = dp.get("synthetic")
        if syn:
            n, Lmax = int(syn.get("n", 1024)), int(syn.get("max_len", 64))
            self.train_dataset = SyntheticCurveDataset(n, Lmax, syn.get("min_len"), seed=int(syn.get("seed", 0)))
            self.val_dataset = SyntheticCurveDataset(int(syn.get("n_val", max(1, n // 8))), Lmax, syn.get("min_len"),
                                                     seed=int(syn.get("seed", 0)) + 1)
        else:
            root = dp["npy_dir"]
            res = lambda p: p if os.path.isabs(p) else os.path.join(root, p)
            self.train_dataset = CurveDataset(root, list_path=res(dp["train_list"]), train=True)
            self.val_dataset = CurveDataset(root, list_path=res(dp["val_list"]), train=False)
        if self.global_rank == 0:
            print(f"[Data] Train files: {len(self.train_dataset)} | Val files: {len(self.val_dataset)}")

    def _loader(self, ds, bs, train):
        nw = int(self.data_params.get("num_workers", 8))
        sampler = None
        d = torch.distributed
        if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=train, drop_last=False)
        return DataLoader(ds, batch_size=bs, shuffle=(train and sampler is None), sampler=sampler, num_workers=nw,
                          pin_memory=bool(self.data_params.get("pin_memory", True)), collate_fn=pad_collate,
                          drop_last=train, persistent_workers=nw > 0)

    def train_dataloader(self):
        return self._loader(self.train_dataset, int(self.data_params.get("train_batch_size", 256)), True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, int(self.data_params.get("val_batch_size", 256)), False)

    # ---- optimiser policy (reference :169-197) ------------------------------------------------------
    def configure_optimizers(self):
        max_epochs = int(getattr(self.trainer, "max_epochs", 1) or 1)
        if self.schedules and "LR" in self.schedules:
            kind = "none"
        else:
            kind = str(self.exp_params.get("lr_scheduler", "cosine")).lower()
        spe = 1
        if kind == "onecycle":
            # like the reference, the step count comes from an un-sharded loader (experiment.py:181)
            spe = max(1, len(self.train_dataset) // int(self.data_params.get("train_batch_size", 256)))
        self.lr_policy = LRPolicy(kind, self.LR, max_epochs, spe,
                                  pct_start=float(self.exp_params.get("onecycle_pct_start", 0.15)),
                                  div_factor=float(self.exp_params.get("onecycle_div_factor", 25.0)),
                                  final_div=float(self.exp_params.get("onecycle_final_div", 1500.0)))
        return self.lr_policy

    # ---- warm start / codebook init (reference :202-307) ----------------------------------------------
    @staticmethod
    def _strip_model_prefix(state, prefix="model."):
        return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state.items()}

    @staticmethod
    def _filter_state_dict_for_warmstart(candidate, model_state, drop_prefixes=("quantizer.",), require_shape_match=True):
        kept, skipped_prefix, skipped_shape = {}, [], []
        for k, v in candidate.items():
            if any(k.startswith(p) for p in drop_prefixes):
                skipped_prefix.append(k)
            elif k not in model_state:
                continue
            elif require_shape_match and tuple(v.shape) != tuple(model_state[k].shape):
                skipped_shape.append(k)
            else:
                kept[k] = v
        return kept, skipped_prefix, skipped_shape

    def _maybe_init_codebook(self):
        if not getattr(self.model, "use_vq", False) or self._init_codebook_path is None:
            return
        if not os.path.isfile(self._init_codebook_path):
            if self.global_rank == 0:
                print(f"[CodebookInit] Path not found: {self._init_codebook_path}")
            return
        try:
            import numpy as np
            C = torch.from_numpy(np.load(self._init_codebook_path).astype("float32"))
            self.model.init_codebook_from_centroids(C)
            if self.global_rank == 0:
                print(f"[CodebookInit] Loaded centroids from: {self._init_codebook_path} shape={tuple(C.shape)}")
        except Exception as e:
            if self.global_rank == 0:
                print(f"[CodebookInit] Failed to init codebook: {e}")

    def on_fit_start(self):
        resume = _clean_path(getattr(self.trainer, "ckpt_path", None))
        if resume is not None:
            if self.global_rank == 0:
                print(f"[Resume] ckpt_path detected, skip warm-start/codebook-init. resume_epoch={int(self.current_epoch)}")
            return
        if self._warm_start_ckpt and os.path.isfile(self._warm_start_ckpt):
            if self.global_rank == 0:
                print(f"[WarmStart] Loading model weights from: {self._warm_start_ckpt}")
            try:
                ckpt = torch.load(self._warm_start_ckpt, map_location="cpu", weights_only=True)
                state = self._strip_model_prefix(ckpt.get("state_dict", ckpt))
                kept, sp, ss = self._filter_state_dict_for_warmstart(state, self.model.state_dict())
                missing, unexpected = self.model.load_state_dict(kept, strict=False)
                if self.global_rank == 0:
                    print(f"[WarmStart] loaded kept={len(kept)} missing={len(missing)} unexpected={len(unexpected)} "
                          f"skipped_prefix={len(sp)} skipped_shape={len(ss)}")
            except Exception as e:
                if self.global_rank == 0:
                    print(f"[WarmStart] Failed to load: {e}")
        self._maybe_init_codebook()          # always after the warm start so it cannot be overwritten

    # ---- epoch schedule (reference :309-343) ----------------------------------------------------------
    def on_train_epoch_start(self):
        epoch = int(self.current_epoch)
        new_vals = interpolate_schedule(self.schedules, epoch) if self.schedules else {}
        for k, v in new_vals.items():
            if k in self.current_weights:      # keys that are not loss weights (rigid_aug_prob, ...) are ignored
                self.current_weights[k] = float(v)
        for k in INT_KEYS:
            self.current_weights[k] = int(round(float(self.current_weights.get(k, 0))))
        self.model.label_smoothing = self.current_weights["label_smoothing"]
        self.model.usage_entropy_lambda = self.current_weights["usage_entropy_lambda"]
        q = getattr(self.model, "quantizer", None)
        if q is not None and hasattr(q, "reset_epoch_stats"):
            q.reset_epoch_stats()
        if self.global_rank == 0:
            brief = {k: round(float(self.current_weights[k]), 6) for k in ("beta", "ss_weight", "rmsd_weight")}
            print(f"[Schedule] Epoch {epoch}: {brief}")
        self._ep_n = 0
        if hasattr(self.model, "reset_metric_sums"):
            self.model.reset_metric_sums()
        self.model.beta = float(self.current_weights["beta"])
        if "LR" in new_vals and self.lr_policy is not None:
            self.lr_policy.manual_lr = float(new_vals["LR"])

    def loss_weights(self):
        return {k: self.current_weights[k] for k in LOSS_KWARGS}

    # ---- steps (reference :351-479) --------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        x, mask = batch
        lr, beta1 = self.lr_policy.current() if self.lr_policy is not None else (self.LR, 0.9)
        clip = float(getattr(self.trainer, "gradient_clip_val", 0.0) or 0.0)
        metrics = self.model.train_step(x, mask, self.loss_weights(), lr, self.weight_decay, clip, betas=(beta1, 0.999))
        self._ep_n += 1
        self.global_step += 1
        if self.lr_policy is not None:
            self.lr_policy.on_step()
        n = int(self.exp_params.get("print_every", 0))
        if n > 0 and batch_idx % n == 0:
            # the reference logs ~20 scalars per step with sync_dist=True (experiment.py:402-437: ~10 scalar all-reduces
            # every step); here the whole metric vector is averaged over the ranks in ONE all-reduce per logging interval
            metrics = self._rank_mean(metrics)
        if n > 0 and batch_idx % n == 0 and self.global_rank == 0:
            names = self.model.metric_names()
            md = dict(zip(names, [float(v) for v in metrics.tolist()]))       # the only host sync, every n steps
            self.logged = {f"train/{k}": v for k, v in md.items()}
            print(f"step={batch_idx:05d} | loss={md['loss']:.3f} | xyz={md['Reconstruction_Loss_XYZ']:.3f} | "
                  f"xyz_raw={md['XYZ_MSE_Raw']:.3f} | xyz_aln={md['XYZ_MSE_Aligned']:.3f} | vq={md['VQ_Loss']:.3f} | "
                  f"ppl={md['VQ_Perplexity']:.3f} | dead={md['VQ_DeadRatio']:.3f} | ss_acc={md['SS_Accuracy']:.3f} | "
                  f"ss_loss={md['Reconstruction_Loss_SS']:.3f} | lr={lr:.6f}", flush=True)
        return metrics

    def _rank_mean(self, vec):
        """Mean of a metric vector over the data-parallel ranks (one batched all-reduce; identity for one process)."""
        d = torch.distributed
        if d.is_available() and d.is_initialized() and d.get_world_size() > 1 and torch.is_tensor(vec):
            vec = vec.clone()
            d.all_reduce(vec)
            vec = vec / d.get_world_size()
        return vec

    def validation_step(self, batch, batch_idx):
        x, mask = batch
        metrics = self.model.eval_step(x, mask, self.loss_weights())
        # running sums on the device (no host sync per batch); averaged, rank-reduced and logged at epoch end
        if torch.is_tensor(metrics):
            self._val_sum = metrics.clone() if self._val_sum is None else self._val_sum + metrics
        self._val_n += 1
        return metrics

    def on_validation_epoch_start(self):
        self._val_sum, self._val_n = None, 0
        q = getattr(self.model, "quantizer", None)
        if q is not None and hasattr(q, "reset_epoch_stats"):
            q.reset_epoch_stats()

    def on_validation_epoch_end(self):
        if self._val_n > 0 and self._val_sum is not None:
            mean = self._rank_mean(self._val_sum / float(self._val_n))          # val/* metrics (experiment.py:478-479, :402-437)
            names = self.model.metric_names()
            md = dict(zip(names, [float(v) for v in mean.tolist()]))
            self.logged.update({f"val/{k}": v for k, v in md.items()})
            if self.trainer is not None:
                self.trainer.callback_metrics.update({f"val/{k}": v for k, v in md.items()})
            if self.global_rank == 0:
                print(f"[Val {int(self.current_epoch)}] loss={md['loss']:.4f} xyz={md['Reconstruction_Loss_XYZ']:.4f} "
                      f"ss_loss={md['Reconstruction_Loss_SS']:.4f} vq={md['VQ_Loss']:.4f}")
        q = getattr(self.model, "quantizer", None)
        if q is not None and hasattr(q, "get_epoch_stats"):
            st = q.get_epoch_stats()
            if self.global_rank == 0:
                print(f"[Val Stats] PPL: {st.get('perplexity', 0):.2f}, Dead Ratio: {st.get('dead_ratio', 0):.3f}")

    def on_train_epoch_end(self):
        sums = self._rank_mean(self.model.metric_sums()) if (self._ep_n > 0 and hasattr(self.model, "metric_sums")) else None
        if sums is not None and self.global_rank == 0:
            names = self.model.metric_names()
            md = dict(zip(names, [float(v) / self._ep_n for v in sums.tolist()]))
            lr = self.lr_policy.current()[0] if self.lr_policy is not None else self.LR
            print(f"[Epoch {int(self.current_epoch)}] loss={md['loss']:.4f} xyz={md['Reconstruction_Loss_XYZ']:.4f} "
                  f"ss_loss={md['Reconstruction_Loss_SS']:.4f} rmsd_aln={md['RMSD_Aligned']:.4f}A "
                  f"rmsd_raw={md['RMSD_Raw']:.4f}A vq={md['VQ_Loss']:.4f} lr={lr:.6f}")
            self.logged.update({f"epoch/{k}": v for k, v in md.items()})


def build_experiment_from_yaml(yaml_path: str):
    with open(yaml_path) as f:
        config = yaml.safe_load(f)

    def expand(o):
        if isinstance(o, str):
            return os.path.expandvars(o)
        if isinstance(o, dict):
            return {k: expand(v) for k, v in o.items()}
        if isinstance(o, list):
            return [expand(v) for v in o]
        return o
    config = expand(config)
    return VQVAEExperiment(config["model_params"], config["exp_params"], config["data_params"]), config
