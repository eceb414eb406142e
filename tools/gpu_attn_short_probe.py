# -*- coding: utf-8 -*-
"""Short-sequence attention (C2: B = 256, 8 heads x 64, T = S = 64): the fused short kernels (flags 0) against the general
kernels on the bf16 pipes (flags 1) and on the fp32 MFMA (flags 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
B, nh, dh = 256, 8, 64
E = nh * dh
rng = torch.tensor([5, 1], device=dev, dtype=torch.int64)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for T, S in [(64, 64), (64, 32), (48, 64)]:
    q, do = torch.randn(B, T, E, device=dev), torch.randn(B, T, E, device=dev)
    k, v = torch.randn(B, S, E, device=dev), torch.randn(B, S, E, device=dev)
    valid = torch.ones(B, S, dtype=torch.bool, device=dev)
    o, lse = torch.empty(B, T, E, device=dev), torch.empty(B * nh * T, device=dev)
    dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=dev)
    for flags in (0, 1, 5):
        old = L.lib().vqh_attn_set_flags(flags)
        try:
            tf = timeit(lambda: L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, valid, B, nh, T, S, dh, 0, rng, 5, 0.1))
            tb = timeit(lambda: L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, valid, B, nh, T, S, dh, 0, rng, 5, 0.1))
        finally:
            L.lib().vqh_attn_set_flags(old)
        byf = 4.0 * B * E * (2 * T + 2 * S)
        byb = 4.0 * B * E * (4 * T + 4 * S)
        print(f"T={T} S={S} flags={flags}: fwd {tf:6.1f} us ({byf / tf / 1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({byb / tb / 1e6:5.2f} TB/s)", flush=True)
