# -*- coding: utf-8 -*-
"""vqh_vq_segment_sum: the (row chunk, code range) kernel against the one-workgroup-per-code kernel (flag bit 2) over shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
ws = torch.empty(48 << 20, device=dev)


def t(z, idx, K, D, flags=0, n=10):
    R = z.shape[0]
    cnt, ssum = torch.empty(K, device=dev), torch.empty(K, D, device=dev)
    old = L.lib().vqh_vq_set_flags(flags)
    try:
        for _ in range(3):
            L.call("vqh_vq_segment_sum", z, D, idx, R, D, 0, K, cnt, ssum, ws, ws.numel())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            L.call("vqh_vq_segment_sum", z, D, idx, R, D, 0, K, cnt, ssum, ws, ws.numel())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    finally:
        L.lib().vqh_vq_set_flags(old)


for R, K, D in [(262144, 8192, 256), (65536, 8192, 256), (16384, 8192, 256), (4096, 8192, 256), (8192, 1024, 512), (16384, 1024, 512),
                (65536, 1024, 512), (16384, 4096, 512)]:
    z = torch.randn(R, D, device=dev)
    idx = torch.randint(0, K, (R,), device=dev)
    sk = (torch.randint(0, 5, (R,), device=dev) * 3).clamp(max=K - 1)            # collapsed usage: 5 codes take every row
    print(f"R={R} K={K} D={D}: sorted {t(z, idx, K, D):7.1f} us (skewed {t(z, sk, K, D):7.1f}) | range kernel {t(z, idx, K, D, flags=16):7.1f} "
          f"(skewed {t(z, sk, K, D, flags=16):7.1f}) | per-code {t(z, idx, K, D, flags=4):7.1f} (skewed {t(z, sk, K, D, flags=4):7.1f})", flush=True)
