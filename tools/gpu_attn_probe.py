"""GPU probe (not a pytest): short-sequence attention kernels at the C2 call shape (B=256, 8 heads, T=S=64, dh=64), packed QKV.
   python tools/gpu_attn_probe.py [flags,...]   (vqh_attn_set_flags: 0 = current, 2 = round-1 backward, 1 = general kernels)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

torch.manual_seed(0)
dev = "cuda"
B, nh, T, S, dh = 256, 8, 64, 64, 64
E = nh * dh
qkv = torch.randn(B * T, 3 * E, device=dev)
do = torch.randn(B * T, E, device=dev)
o, lse = torch.empty(B * T, E, device=dev), torch.empty(B * nh * T, device=dev)
dqkv, dsum = torch.empty_like(qkv), torch.empty(B * nh * T, device=dev)
valid = torch.ones(B, S, dtype=torch.bool, device=dev)
rng = torch.tensor([7, 5], device=dev, dtype=torch.int64)


def fwd():
    L.call("vqh_attn_fwd", qkv, 3 * E, qkv[:, E:], 3 * E, qkv[:, 2 * E:], 3 * E, o, E, lse, valid, B, nh, T, S, dh, 0, rng, 5, 0.1)


def bwd():
    L.call("vqh_attn_bwd", qkv, 3 * E, qkv[:, E:], 3 * E, qkv[:, 2 * E:], 3 * E, o, E, lse, do, E, dsum, dqkv, 3 * E, dqkv[:, E:], 3 * E,
           dqkv[:, 2 * E:], 3 * E, valid, B, nh, T, S, dh, 0, rng, 5, 0.1)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for f in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "2"])]:
    L.lib().vqh_attn_set_flags(f)
    fwd()
    print(f"flags={f}: fwd {timeit(fwd):7.1f} us   bwd {timeit(bwd):7.1f} us", flush=True)
