"""GPU probe (not a pytest): cost of the fused GEMM epilogues (bias / ReLU+dropout / dropout+residual / masks)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
torch.manual_seed(0)
dev = "cuda"
rng = torch.tensor([1234, 5], dtype=torch.int64, device=dev)

def timeit(akc, bkc, M, N, K, iters=20, **kw):
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev)
    args = dict(kw)
    if args.pop("use_bias", False):
        args["bias"] = bias
    if args.pop("use_aux", False):
        args["aux_in"], args["ldaux"] = aux, N
    if args.pop("use_auxout", False):
        args["aux_out"], args["ldaux"] = aux, N
    def run():
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, rng=rng, site=3, **args)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

CASES = [
    ("NT 16384x2048x512 plain", (1,1,16384,2048,512), {}),
    ("NT 16384x2048x512 bias", (1,1,16384,2048,512), dict(use_bias=True)),
    ("NT 16384x2048x512 relu_drop p=0", (1,1,16384,2048,512), dict(use_bias=True, mode=L.EPI_RELU_DROP, p=0.0)),
    ("NT 16384x2048x512 relu_drop p=.1", (1,1,16384,2048,512), dict(use_bias=True, mode=L.EPI_RELU_DROP, p=0.1)),
    ("NT 16384x512x2048 plain", (1,1,16384,512,2048), {}),
    ("NT 16384x512x2048 drop_resid p=0", (1,1,16384,512,2048), dict(use_bias=True, use_aux=True, mode=L.EPI_DROP_RESID, p=0.0)),
    ("NT 16384x512x2048 drop_resid p=.1", (1,1,16384,512,2048), dict(use_bias=True, use_aux=True, mode=L.EPI_DROP_RESID, p=0.1)),
    ("NT 16384x512x512 plain", (1,1,16384,512,512), {}),
    ("NT 16384x512x512 drop_resid p=0", (1,1,16384,512,512), dict(use_bias=True, use_aux=True, mode=L.EPI_DROP_RESID, p=0.0)),
    ("NT 16384x512x512 drop_resid p=.1", (1,1,16384,512,512), dict(use_bias=True, use_aux=True, mode=L.EPI_DROP_RESID, p=0.1)),
    ("NN 16384x2048x512 plain", (1,0,16384,2048,512), {}),
    ("NN 16384x2048x512 posmask", (1,0,16384,2048,512), dict(use_aux=True, mode=L.EPI_MUL_POSMASK, p=0.1)),
    ("NT 16384x1536x512 bias", (1,1,16384,1536,512), dict(use_bias=True)),
]
for rep in range(2):
    for name, sh, kw in CASES:
        us = timeit(*sh, **kw)
        print(f"{name:40s} {us:8.1f} us  {2.0*sh[2]*sh[3]*sh[4]/us/1e6:6.1f} TF", flush=True)
    print()
