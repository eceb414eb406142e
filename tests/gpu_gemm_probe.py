"""GPU probe (not a pytest): GEMM correctness vs torch.matmul and TFLOP/s on the C2 shapes."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

torch.manual_seed(0)
dev = "cuda"
ws = torch.empty(64 * 1024 * 1024, device=dev)

def run(akc, bkc, M, N, K, check=True, iters=20):
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    Cc = torch.empty(M, N, device=dev)
    L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, ws=ws)
    torch.cuda.synchronize()
    err = None
    if check:
        ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double()
        err = ((Cc.double() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3):
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cc, N, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    # torch (hipBLASLt/rocBLAS) for context
    At = (A if akc else A.t()); Bt = (B.t() if bkc else B)
    for _ in range(3): torch.matmul(At, Bt)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.matmul(At, Bt)
    e1.record(); torch.cuda.synchronize()
    tms = e0.elapsed_time(e1) / iters
    print(f"akc={akc} bkc={bkc} M={M} N={N} K={K}: relerr={err} {ms*1e3:.1f} us {tf:.1f} TF/s | torch {tms*1e3:.1f} us {2.0*M*N*K/tms/1e9:.1f} TF/s", flush=True)

print(torch.cuda.get_device_name(0))
# correctness on awkward shapes
for (akc, bkc, M, N, K) in [(1,1,111,6,512),(1,1,130,70,35),(1,0,77,512,6),(0,0,512,3,999),(0,0,64,512,1000),(0,1,33,65,129),(1,1,1,1,1)]:
    run(akc, bkc, M, N, K, iters=2)
# C2 shapes: fwd, dgrad, wgrad
for (akc, bkc, M, N, K) in [(1,1,16384,512,512),(1,1,16384,1536,512),(1,1,16384,2048,512),(1,1,16384,512,2048),
                            (1,0,16384,512,2048),(1,0,16384,2048,512),(0,0,512,512,16384),(0,0,2048,512,16384),(0,0,512,2048,16384),
                            (1,1,4096,4096,4096)]:
    run(akc, bkc, M, N, K, check=(M*N*K < 2e11))
