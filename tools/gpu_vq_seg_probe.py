# -*- coding: utf-8 -*-
"""vqh_vq_segment_sum at R = 262144, K = 8192, D = 256: time with real indices, with no matching row (scan only) and with the
one-workgroup-per-code kernel (flag bit 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
R, K, D = 262144, 8192, 256
z = torch.randn(R, D, device=dev)
ws = torch.empty(48 << 20, device=dev)
cnt, ssum = torch.empty(K, device=dev), torch.empty(K, D, device=dev)


def t(idx, flags=0, n=10):
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


idx = torch.randint(0, K, (R,), device=dev)
print("random codes      : %.1f us (incl. 2 reduce launches)" % t(idx))
print("no matching rows  : %.1f us" % t(torch.full((R,), -1, device=dev, dtype=torch.int64)))
print("sorted codes      : %.1f us" % t(torch.sort(idx)[0]))
print("per-code kernel   : %.1f us" % t(idx, flags=4))
