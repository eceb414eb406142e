# -*- coding: utf-8 -*-
"""Data side of the hot path (reference: dataset.py:30-139): variable-length curve tensors
[L_i, 6] = centred xyz + secondary-structure one-hot, zero-padded to [B, L_max, 6] with a bool mask.
SyntheticCurveDataset (ours) produces seeded curves of the same layout for the benchmark configs,
since the reference's .npy corpus is not available offline."""
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


def pad_collate(batch: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """list of [L_i, 6] -> (x [B, L_max, 6] zero padded, mask [B, L_max] True = real point)."""
    if len(batch) == 0:
        raise RuntimeError("Empty batch given to pad_collate.")
    lens = torch.tensor([int(t.shape[0]) for t in batch])
    Lmax = int(lens.max())
    x = torch.zeros(len(batch), Lmax, batch[0].shape[-1], dtype=batch[0].dtype)
    for i, t in enumerate(batch):
        x[i, :t.shape[0]] = t
    mask = torch.arange(Lmax)[None, :] < lens[:, None]
    return x, mask


class CurveDataset(Dataset):
    """One curve per .npy/.npz file holding {'curve_coords': [L,3], 'ss_one_hot': [L,3]}; xyz is centred per curve."""

    def __init__(self, npy_dir: str, list_path: Optional[str] = None, list_file: Optional[str] = None, train: bool = True):
        super().__init__()
        listing = list_path or list_file
        if listing is None:
            raise ValueError("CurveDataset requires a valid list_path or list_file.")
        with open(listing) as f:
            names = [ln.strip() for ln in f if ln.strip()]
        if not names:
            raise FileNotFoundError(f"No files found using list file: {listing}")
        self.file_paths = [os.path.join(str(npy_dir), n) for n in names]
        self.train = train
        print(f"[Dataset] {'Train' if train else 'Val'} set: {len(self.file_paths)} curves from {npy_dir}")

    def __len__(self):
        return len(self.file_paths)

    def __getitem__(self, i):
        path = self.file_paths[i]
        raw = np.load(path, allow_pickle=True)   # the user's own data files (dict saved with np.save)
        rec = {k: raw[k] for k in raw.files} if isinstance(raw, np.lib.npyio.NpzFile) else raw.item()
        xyz = np.asarray(rec["curve_coords"], dtype=np.float32)
        ss = np.asarray(rec["ss_one_hot"], dtype=np.float32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError(f"Bad coords shape at {path}: {xyz.shape}")
        if ss.shape != xyz.shape:
            raise ValueError(f"Bad ss_one_hot shape at {path}: {ss.shape}")
        xyz = xyz - xyz.mean(axis=0, keepdims=True)
        full = np.concatenate([xyz, ss], axis=-1).astype(np.float32)
        if not np.isfinite(full).all():
            full = np.nan_to_num(full, nan=0.0, posinf=0.0, neginf=0.0)
        return torch.from_numpy(full)


class SyntheticCurveDataset(Dataset):
    """Seeded chain-like curves (3.8 A steps) with random SS labels; lengths uniform in [min_len, max_len]."""

    def __init__(self, n: int, max_len: int = 64, min_len: Optional[int] = None, seed: int = 0):
        self.n, self.max_len, self.min_len, self.seed = int(n), int(max_len), int(min_len or max_len), int(seed)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        Lc = int(torch.randint(self.min_len, self.max_len + 1, (1,), generator=g))
        step = torch.randn(Lc, 3, generator=g)
        xyz = torch.cumsum(3.8 * step / step.norm(dim=-1, keepdim=True), 0)
        xyz = xyz - xyz.mean(0, keepdim=True)
        ss = torch.nn.functional.one_hot(torch.randint(0, 3, (Lc,), generator=g), 3).float()
        return torch.cat([xyz, ss], -1)


def synthetic_curve_batch(B: int, L: int, seed: int, ragged: bool = False, min_len: Optional[int] = None):
    """One seeded synthetic batch in the layout pad_collate produces: xyz ~ 5*N(0,1) centred per sample
    (dataset.py:121-122 of the reference centres every curve), random secondary-structure one-hot, zero padding; the first
    sample always has the full length so that L_max == L.  Used by bench.py and the full-size tests (the reference's .npy
    corpus is not available offline)."""
    g = torch.Generator().manual_seed(seed)
    xyz = 5.0 * torch.randn(B, L, 3, generator=g)
    ss = torch.nn.functional.one_hot(torch.randint(0, 3, (B, L), generator=g), 3).float()
    mask = torch.ones(B, L, dtype=torch.bool)
    if ragged:
        lo = min_len if min_len is not None else max(3, L // 2)
        lens = torch.randint(lo, L + 1, (B,), generator=g)
        lens[0] = L
        mask = torch.arange(L)[None, :] < lens[:, None]
    m = mask.float()[..., None]
    xyz = xyz - (xyz * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)
    return (torch.cat([xyz, ss], -1) * m).contiguous(), mask
