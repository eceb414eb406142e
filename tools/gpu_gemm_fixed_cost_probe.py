# -*- coding: utf-8 -*-
"""Fixed cost of an x3 GEMM launch: kernel time (rocprofv3 --kernel-trace, or wall time of back-to-back launches) against K for the
C2 forward shape M = 16384, N = 512 (256 tiles = one workgroup per CU) and N = 2048 (4 per CU): t(K) = a + b K."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda:0"
M = 16384
ws = torch.empty(48 << 20, device=dev)
for N in (512, 2048):
    for K in (128, 256, 512, 1024, 2048, 4096):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        C = torch.empty(M, N, device=dev)
        for _ in range(3):
            L.gemm(1, 1, M, N, K, A, K, W, K, C, N, ws=ws)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            L.gemm(1, 1, M, N, K, A, K, W, K, C, N, ws=ws)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n * 1e6
        ksteps = K // 32
        print(f"N={N} K={K}: {t:7.1f} us per launch (back to back)  = {t / ksteps:6.2f} us per K-step; "
              f"{12.0 * M * N * K / t / 1e6 / 2500:5.3f} of the bf16 peak", flush=True)
