# -*- coding: utf-8 -*-
"""Isolated launch times of the plane-tensor GEMM (gemm_p3) against the in-loop-split tile (gemm_f32_x3) on the C2 shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

dev = "cuda:0"
ws = torch.empty(48 << 20, device=dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


shapes = [(1, 1, 16384, 512, 512), (1, 1, 16384, 1536, 512), (1, 1, 16384, 2048, 512), (1, 1, 16384, 512, 2048),
          (1, 0, 16384, 512, 512), (1, 0, 16384, 2048, 512), (1, 0, 16384, 512, 2048), (0, 0, 2048, 512, 16384), (0, 0, 512, 512, 16384)]
for rnd in range(2):
    for akc, bkc, M, N, K in shapes:
        A = torch.randn((M, K) if akc else (K, M), device=dev)
        B = torch.randn((N, K) if bkc else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        Ap, Bp = L.p3_split(A), L.p3_split(B)
        t3 = timeit(lambda: L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), C, N, ws=ws))
        tp = timeit(lambda: L.gemm_p3(akc, bkc, M, N, K, Ap, Bp, C, N, ws=ws))
        fl = 2.0 * M * N * K
        print(f"{'NT' if akc and bkc else 'NN' if akc else 'TN'} {M}x{N}x{K}: x3 {t3 * 1e6:7.1f} us {fl / t3 / 1e12:6.1f} TF | "
              f"p3 {tp * 1e6:7.1f} us {fl / tp / 1e12:6.1f} TF fp32-equiv = {6 * fl / tp / 1e15:5.3f} PF bf16 ({6 * fl / tp / 2.5e15:.3f} of peak)", flush=True)
