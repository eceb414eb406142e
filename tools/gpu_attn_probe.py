"""GPU probe (not a pytest): attention kernel timings at the C2 shape, dropout off vs on."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda"
B, nh, T, S, dh = 256, 8, 64, 64, 64
E = nh * dh
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * E, device=dev)
q, k, v = qkv, qkv[:, E:], qkv[:, 2 * E:]
o, do = torch.empty(B * T, E, device=dev), torch.randn(B * T, E, device=dev)
lse, dsum = torch.empty(B * nh * T, device=dev), torch.empty(B * nh * T, device=dev)
dqkv = torch.empty_like(qkv)
rng = torch.tensor([1, 2], device=dev, dtype=torch.int64)
mask = torch.ones(B, S, dtype=torch.bool, device=dev)

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for p, m in ((0.0, None), (0.0, mask), (0.1, mask)):
    f = lambda: L.call("vqh_attn_fwd", q, 3 * E, k, 3 * E, v, 3 * E, o, E, lse, m, B, nh, T, S, dh, 0, rng, 1, p)
    b = lambda: L.call("vqh_attn_bwd", q, 3 * E, k, 3 * E, v, 3 * E, o, E, lse, do, E, dsum, dqkv, 3 * E, dqkv[:, E:], 3 * E,
                       dqkv[:, 2 * E:], 3 * E, m, B, nh, T, S, dh, 0, rng, 1, p)
    flops = 4.0 * B * nh * T * S * dh
    tf, tb = t(f), t(b)
    print(f"p={p} mask={'yes' if m is not None else 'no'}: fwd {tf:7.1f} us ({flops/tf/1e6:5.1f} TF)  bwd(dq+dkv) {tb:7.1f} us ({2.5*flops/tb/1e6:5.1f} TF)", flush=True)

# layout experiment: the same 2048 (batch, head) problems with every operand tile contiguous (nh = 1, ld = dh)
B2 = B * nh
q2, k2, v2 = (torch.randn(B2 * T, dh, device=dev) for _ in range(3))
o2, do2 = torch.empty(B2 * T, dh, device=dev), torch.randn(B2 * T, dh, device=dev)
lse2, dsum2 = torch.empty(B2 * T, device=dev), torch.empty(B2 * T, device=dev)
dq2, dk2, dv2 = (torch.empty(B2 * T, dh, device=dev) for _ in range(3))
f = lambda: L.call("vqh_attn_fwd", q2, dh, k2, dh, v2, dh, o2, dh, lse2, None, B2, 1, T, S, dh, 0, rng, 1, 0.1)
b = lambda: L.call("vqh_attn_bwd", q2, dh, k2, dh, v2, dh, o2, dh, lse2, do2, dh, dsum2, dq2, dh, dk2, dh, dv2, dh, None,
                   B2, 1, T, S, dh, 0, rng, 1, 0.1)
print(f"contiguous tiles (nh=1, B={B2}): fwd {t(f):7.1f} us   bwd {t(b):7.1f} us", flush=True)
