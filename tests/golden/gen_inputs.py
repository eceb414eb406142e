# -*- coding: utf-8 -*-
"""Seeded synthetic inputs shared by make_golden.py (fixture generation, build container only)
and the tests (CPU and GPU box).  Fixtures store seeds + expected outputs + an input checksum;
the weights/inputs themselves are regenerated from the seed by these functions (torch CPU
generator: bit-stable for one torch build, which the container and the GPU box share)."""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import vqvae_oracle as O  # noqa: E402

# ------------------------------------------------------------------------------------------
# model configurations used by the fixtures
# ------------------------------------------------------------------------------------------
SMALL_VQ = dict(hidden_dim=64, num_layers=1, num_heads=4, max_seq_len=40, codebook_size=32,
                code_dim=16, latent_tokens=8, tokenizer_heads=4, tokenizer_layers=2,
                use_vq=True, num_quantizers=1, beta=0.25, label_smoothing=0.01, ss_tv_lambda=0.002,
                xyz_align_alpha=0.95, reinit_dead_codes=False, print_init=False)
SMALL_RVQ = dict(SMALL_VQ, num_quantizers=3, codebook_size=16, residual_vq=True)
SMALL_AE = dict(hidden_dim=64, num_layers=1, num_heads=4, max_seq_len=40, codebook_size=32,
                code_dim=16, latent_tokens=8, tokenizer_heads=4, tokenizer_layers=2,
                use_vq=False, beta=0.0, xyz_align_alpha=1.0, latent_sigmoid=True,
                latent_sigmoid_ae_only=True, reinit_dead_codes=False, print_init=False)
# true-width C2 model (stage2_vq.yaml with Q=1, K=512, D=64, N=64)
C2_MODEL = dict(hidden_dim=512, num_layers=4, num_heads=8, max_seq_len=350, codebook_size=512,
                code_dim=64, latent_tokens=64, tokenizer_heads=8, tokenizer_layers=2,
                use_vq=True, num_quantizers=1, residual_vq=True, beta=0.0, label_smoothing=0.01,
                ss_tv_lambda=0.002, xyz_align_alpha=0.95, reinit_dead_codes=False, print_init=False)

ALL_LOSS_WEIGHTS = dict(ss_weight=0.8, bond_length_weight=0.015, bond_angle_weight=0.006,
                        xyz_tv_lambda=0.001, dir_weight=0.02, dih_weight=0.03, rmsd_weight=1.8,
                        pdm_weight=0.001, win_kabsch_weight=0.0006, kappa_weight=0.004,
                        tau_weight=0.005, lr_pdm_weight=0.003, pdm_window=8, win_kabsch_size=16,
                        win_kabsch_stride=8, lr_min_sep=24, lr_stride=8, lr_max_offsets=8)
BASE_LOSS_WEIGHTS = dict(ss_weight=0.8, rmsd_weight=1.8)


def curve_batch(B, L, seed, ragged=False, min_len=None):
    """SURVEY.md 8d synthetic input: xyz ~ 5*N(0,1) centred per sample, random SS one-hot."""
    g = torch.Generator().manual_seed(seed)
    xyz = 5.0 * torch.randn(B, L, 3, generator=g)
    ss = torch.nn.functional.one_hot(torch.randint(0, 3, (B, L), generator=g), 3).float()
    mask = torch.ones(B, L, dtype=torch.bool)
    if ragged:
        lo = min_len if min_len is not None else max(3, L // 2)
        lens = torch.randint(lo, L + 1, (B,), generator=g)
        lens[0] = L                                  # keep L_max == L like pad_collate would
        mask = torch.arange(L)[None, :] < lens[:, None]
    m = mask.float()[..., None]
    xyz = xyz - (xyz * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)
    x = torch.cat([xyz, ss], -1) * m                 # zero padding like pad_sequence
    return x.contiguous(), mask


def smooth_curve_batch(B, L, seed, ragged=False):
    """A chain-like curve (random walk with ~3.8 A steps) so the geometric loss terms are
    evaluated in their realistic regime (non-degenerate bond vectors)."""
    g = torch.Generator().manual_seed(seed)
    step = torch.randn(B, L, 3, generator=g)
    step = 3.8 * step / step.norm(dim=-1, keepdim=True)
    xyz = torch.cumsum(step, 1)
    ss = torch.nn.functional.one_hot(torch.randint(0, 3, (B, L), generator=g), 3).float()
    mask = torch.ones(B, L, dtype=torch.bool)
    if ragged:
        lens = torch.randint(max(5, L // 2), L + 1, (B,), generator=g)
        lens[0] = L
        mask = torch.arange(L)[None, :] < lens[:, None]
    m = mask.float()[..., None]
    xyz = xyz - (xyz * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)
    return (torch.cat([xyz, ss], -1) * m).contiguous(), mask


def model_state(cfg_kw, seed):
    cfg = O.make_cfg(**cfg_kw)
    return O.random_state(cfg, seed)


def vq_inputs(R, K, D, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    z = scale * torch.randn(R, D, generator=g)
    emb = torch.randn(K, D, generator=g) / math.sqrt(D)
    return z, emb


def checksum(t):
    """Order-sensitive fp64 checksum of a tensor (detects RNG drift between torch builds)."""
    t = t.detach().double().reshape(-1)
    w = torch.arange(1, t.numel() + 1, dtype=torch.float64)
    return float((t * torch.cos(w)).sum())


def vq_row_masks(B, M, seed, steps):
    """Valid-position masks [B, M] for VectorQuantizerEMA(mask=...) fixtures: ragged prefixes; step 1 has no valid
    position at all (the reference then skips the EMA update entirely, models/vq_vae.py:196)."""
    g = torch.Generator().manual_seed(seed + 7777)
    out = []
    for s in range(steps):
        lens = torch.randint(M // 4, M + 1, (B,), generator=g)
        m = torch.arange(M)[None, :] < lens[:, None]
        if s == 1:
            m = torch.zeros(B, M, dtype=torch.bool)
        out.append(m)
    return out
