"""GPU probe (not a pytest): the vendor BLAS (torch.matmul -> hipBLASLt, fp32) and this repo's GEMM kernel on the step's
shapes, interleaved in one process.  Under rocprofv3 the BLAS kernel names show the macro tiles it picks."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
torch.manual_seed(0)
dev = "cuda"
torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [("NT", 16384, 2048, 512), ("NT", 16384, 512, 2048), ("NT", 16384, 512, 512), ("NT", 16384, 1536, 512),
          ("NN", 16384, 512, 2048), ("NN", 16384, 2048, 512), ("TN", 2048, 512, 16384), ("TN", 512, 512, 16384)]
for lay, M, N, K in SHAPES:
    if lay == "NT":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        f = lambda: A @ B.t()
    elif lay == "NN":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        f = lambda: A @ B
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        f = lambda: A.t() @ B
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(64 << 20, device=dev)
    akc, bkc = {"NT": (1, 1), "NN": (1, 0), "TN": (0, 0)}[lay]
    g = lambda: L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), C, N, ws=ws)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    res = {"blas": [], "ours": []}
    for rnd in range(3):                      # interleaved rounds, best of 3 each
        res["blas"].append(timeit(f))
        res["ours"].append(timeit(g))
    ref = f()
    err = float((C - ref).abs().max() / ref.abs().max())
    ub, uo = min(res["blas"]), min(res["ours"])
    print(f"{lay} {M}x{N}x{K}: vendor BLAS {ub:7.1f} us {2.0*M*N*K/ub/1e6:6.1f} TF | vqh_gemm {uo:7.1f} us {2.0*M*N*K/uo/1e6:6.1f} TF"
          f" | max rel diff {err:.1e}", flush=True)
