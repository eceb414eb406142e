# -*- coding: utf-8 -*-
"""p3 GEMM on stage images (pitch 0) vs row-pitched plane tensors vs the x3 kernel on fp32 operands, C2 shapes (rows = 16384)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
ws = torch.empty(48 << 20, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


R = 16384
shapes = [("fwd QKV   ", 1, 1, R, 1536, 512), ("fwd out   ", 1, 1, R, 512, 512), ("fwd FF1   ", 1, 1, R, 2048, 512), ("fwd FF2   ", 1, 1, R, 512, 2048),
          ("dgrad QKV ", 1, 0, R, 512, 1536), ("dgrad out ", 1, 0, R, 512, 512), ("dgrad FF2 ", 1, 0, R, 2048, 512), ("dgrad FF1 ", 1, 0, R, 512, 2048),
          ("wgrad FF1 ", 0, 0, 2048, 512, R), ("wgrad QKV ", 0, 0, 1536, 512, R)]
for name, akc, bkc, M, N, K in shapes:
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    Ap, Bp, Ai, Bi = L.p3_split(A), L.p3_split(B), L.p3_image(A), L.p3_image(B)
    t_x3 = timeit(lambda: L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), C, N, ws=ws))
    t_p3 = timeit(lambda: L.gemm_p3(akc, bkc, M, N, K, Ap, Bp, C, N, ws=ws))
    t_im = timeit(lambda: L.gemm_p3(akc, bkc, M, N, K, Ai, Bi, C, N, ws=ws, pitch_a=0, pitch_b=0))
    fl = 12.0 * M * N * K
    print(f"{name} M={M:6d} N={N:5d} K={K:6d}: x3 {t_x3:7.1f} us ({fl / t_x3 / 1e6 / 2500:5.3f})  p3 planes {t_p3:7.1f} us  p3 images {t_im:7.1f} us "
          f"({fl / t_im / 1e6 / 2500:5.3f})  images / x3 = {t_im / t_x3:5.3f}", flush=True)
