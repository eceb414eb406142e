# -*- coding: utf-8 -*-
"""vqh_vq_nearest: plane-tensor form (default when the workspace allows) vs the register-resident split form (vq flags bit 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
ws = torch.empty(160 << 20, device=dev)
for R, K, D in [(262144, 8192, 256), (65536, 8192, 256), (16384, 8192, 256), (65536, 1024, 128), (262144, 1024, 128)]:
    z = torch.randn(R, D, device=dev)
    emb = torch.randn(K, D, device=dev) / D ** 0.5
    idx = torch.empty(R, device=dev, dtype=torch.int64)
    res = []
    for flags in (0, 32):
        old = L.lib().vqh_vq_set_flags(flags)
        try:
            form = L.lib().vqh_vq_nearest_form(R, K, D, ws.numel())
            for _ in range(2):
                L.call("vqh_vq_nearest", z, D, emb, D, idx, 0, R, K, D, 3e-5, ws, ws.numel())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                L.call("vqh_vq_nearest", z, D, emb, D, idx, 0, R, K, D, 3e-5, ws, ws.numel())
            torch.cuda.synchronize()
            res.append((form, (time.perf_counter() - t0) / 5 * 1e3, idx.clone()))
        finally:
            L.lib().vqh_vq_set_flags(old)
    fl = 2.0 * R * K * D
    print(f"R={R} K={K} D={D}: form {res[0][0]} {res[0][1]:.3f} ms ({fl / res[0][1] / 1e9:.0f} TF)  | form {res[1][0]} {res[1][1]:.3f} ms "
          f"({fl / res[1][1] / 1e9:.0f} TF)  same={bool(torch.equal(res[0][2], res[1][2]))}", flush=True)
