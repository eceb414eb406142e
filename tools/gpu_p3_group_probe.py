# -*- coding: utf-8 -*-
"""Grouped weight gradients of one C2 encoder layer: gemm_f32_x3_group (fp32 operands, split in the K loop) vs gemm_p3_group (plane operands)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

dev = "cuda:0"
ws = torch.empty(48 << 20, device=dev)
rows = 16384
shapes = [(512, 512), (1536, 512), (2048, 512), (512, 2048)]      # out_proj, in_proj, linear1, linear2
i32, ip3, flops = [], [], 0.0
for n_out, k_in in shapes:
    dY, X = torch.randn(rows, n_out, device=dev), torch.randn(rows, k_in, device=dev)
    gW, gb = torch.empty(n_out, k_in, device=dev), torch.empty(n_out, device=dev)
    i32.append((dY, n_out, X, k_in, rows, gW, gb))
    dYp, Xp = L.p3_split(dY), L.p3_split(X)
    ip3.append((dYp, L.p3_pitch(dYp), Xp, L.p3_pitch(Xp), rows, gW, gb))
    flops += 2.0 * rows * n_out * k_in


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for rnd in range(3):
    t3 = timeit(lambda: L.wgrad_group(i32, ws))
    tp = timeit(lambda: L.wgrad_group_p3(ip3, ws))
    print(f"x3 group {t3 * 1e6:7.1f} us ({flops / t3 / 1e12:5.1f} TF, {6 * flops / t3 / 2.5e15:.3f} of bf16 peak) | "
          f"p3 group {tp * 1e6:7.1f} us ({flops / tp / 1e12:5.1f} TF, {6 * flops / tp / 2.5e15:.3f})  [incl. the split-K reduce launch]", flush=True)
