"""GPU probe (not a pytest): does a power-of-two leading dimension (channel camping) cost GEMM throughput?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
torch.manual_seed(0)
dev = "cuda"
ws = torch.empty(64 * 1024 * 1024, device=dev)
L.lib().vqh_gemm_set_flags(1)

def timeit(akc, bkc, M, N, K, pa, pb, pc, iters=20):
    ar, ac = (M, K) if akc else (K, M)
    br, bc = (N, K) if bkc else (K, N)
    A = torch.randn(ar, ac + pa, device=dev)[:, :ac]
    B = torch.randn(br, bc + pb, device=dev)[:, :bc]
    Cc = torch.empty(M, N + pc, device=dev)[:, :N]
    def run():
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, Cc.stride(0), ws=ws)
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

SHAPES = [(1,1,16384,2048,512),(1,1,16384,512,2048),(1,1,16384,512,512),(1,0,16384,2048,512),(1,0,16384,512,2048),(1,0,16384,512,512),
          (0,0,512,2048,16384),(0,0,2048,512,16384),(0,0,512,512,16384)]
PADS = [(0,0,0),(32,0,0),(0,32,0),(0,0,32),(32,32,32),(16,16,16),(64,64,64)]
print("shape".ljust(28), *[str(p).rjust(14) for p in PADS])
for rep in range(2):
    for sh in SHAPES:
        row = []
        for p in PADS:
            us = timeit(*sh, *p)
            row.append(f"{2.0*sh[2]*sh[3]*sh[4]/us/1e6:8.1f}TF    ")
        print(str(sh).ljust(28), *row, flush=True)
