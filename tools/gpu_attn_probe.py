# -*- coding: utf-8 -*-
"""General attention kernels at the real-data shapes (stage2_vq.yaml, B = 128, L = 350, N = 64 latent tokens, 8 heads x 64):
bf16x3 (default) vs native fp32 MFMA (vqh_attn_set_flags bit 2), forward and backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
B, nh, dh = 128, 8, 64
E = nh * dh
rng = torch.tensor([5, 1], device=dev, dtype=torch.int64)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for T, S in [(350, 350), (64, 350), (350, 64), (256, 256)]:
    q, do = torch.randn(B, T, E, device=dev), torch.randn(B, T, E, device=dev)
    k, v = torch.randn(B, S, E, device=dev), torch.randn(B, S, E, device=dev)
    valid = torch.ones(B, S, dtype=torch.bool, device=dev)
    o, lse = torch.empty(B, T, E, device=dev), torch.empty(B * nh * T, device=dev)
    dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=dev)
    out = []
    for flags in (0, 4):
        old = L.lib().vqh_attn_set_flags(flags)
        try:
            tf = timeit(lambda: L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, valid, B, nh, T, S, dh, 0, rng, 5, 0.1))
            tb = timeit(lambda: L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, valid, B, nh, T, S, dh, 0, rng, 5, 0.1))
        finally:
            L.lib().vqh_attn_set_flags(old)
        out.append((tf, tb))
    fl = 4.0 * B * nh * T * S * dh
    print(f"T={T} S={S}: fwd x3 {out[0][0]:7.1f} us ({fl / out[0][0] / 1e6:6.1f} TF) | fp32 {out[1][0]:7.1f} us ({fl / out[1][0] / 1e6:6.1f} TF)   "
          f"bwd x3 {out[0][1]:7.1f} us | fp32 {out[1][1]:7.1f} us ({2.5 * fl / out[1][1] / 1e6:6.1f} TF)", flush=True)
