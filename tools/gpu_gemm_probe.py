"""GPU probe (not a pytest): GEMM timing on the C2 shapes under the tuning/diagnostic flags."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

torch.manual_seed(0)
dev = "cuda"
ws = torch.empty(64 * 1024 * 1024, device=dev)
SHAPES = [(1,0,16384,512,512),(1,0,16384,1536,512),(1,0,16384,64,512),(1,1,16384,64,512),(1,1,16384,512,512),(1,1,16384,1536,512),(1,1,16384,2048,512),(1,1,16384,512,2048),
          (1,0,16384,512,2048),(1,0,16384,2048,512),(1,0,16384,512,1536),(0,0,512,512,16384),(0,0,2048,512,16384),(0,0,512,2048,16384),(0,0,1536,512,16384)]

def timeit(akc, bkc, M, N, K, iters=20):
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.empty(M, N, device=dev)
    for _ in range(3):
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

flagsets = [int(f) for f in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
print("shape".ljust(28), *[f"flags={f}".rjust(18) for f in flagsets])
tot = {f: 0.0 for f in flagsets}
for sh in SHAPES:
    row = []
    for f in flagsets:
        L.lib().vqh_gemm_set_flags(f)
        us = timeit(*sh)
        tot[f] += us
        row.append(f"{us:8.1f}us {2.0*sh[2]*sh[3]*sh[4]/us/1e6:6.1f}TF")
    print(str(sh).ljust(28), *row, flush=True)
print("total us".ljust(28), *[f"{tot[f]:18.1f}" for f in flagsets])
