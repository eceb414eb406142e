import sys, os
sys.path.insert(0, "/root/repo/pytorch-vae_amd")
import torch
from vqvae_hip import lib as L
dev = "cuda"
ws = torch.empty(64 * 1024 * 1024, device=dev)
def t(M, N, K, iters=20):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    for _ in range(3): L.gemm(0, 0, M, N, K, A, M, B, N, C, N, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm(0, 0, M, N, K, A, M, B, N, C, N, ws=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"TN {M}x{N}x{K}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
for M, N in ((512, 512), (1536, 512), (2048, 512), (4608, 512), (6144, 512), (8192, 512)):
    t(M, N, 16384)
