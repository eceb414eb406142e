"""Diagnostic (not a test): rms and worst-case error of the x3 and the native fp32 MFMA tiles against fp64, per layout and data kind."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "pytorch-vae_amd"), ROOT):
    sys.path.insert(0, p)
import torch
import test_gpu_gemm_x3 as T
L = T._hip()
for akc,bkc,M,N,K in [(1, 1, 4096, 512, 512), (1, 1, 2048, 512, 2048), (1, 0, 4096, 512, 2048), (0, 0, 2048, 512, 16384), (0, 1, 1024, 256, 4096), (1, 1, 512, 128, 32)]:
    for kind in ["normal", "wide", "positive"]:
        A, B = T._operands(kind, (M, K) if akc else (K, M), (N, K) if bkc else (K, N), seed=M + N + K)
        ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double()
        absdot = (A if akc else A.t()).double().abs() @ (B.t() if bkc else B).double().abs()
        ws = torch.empty(1 << 24, device="cuda:0")
        out = {}
        for name, extra in (("x3", 0), ("native", 512)):
            with T.flags(L, extra):
                C = torch.full((M, N), float("nan"), device="cuda:0")
                L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), C, N, ws=ws)
                out[name] = C
        w = lambda C: float(((C.double() - ref).abs() / absdot).max())
        print(akc,bkc,M,N,K,kind, "rms x3 %.3e nat %.3e | worst/absdot x3 %.3e nat %.3e" % (T._rms_err(out["x3"], ref), T._rms_err(out["native"], ref), w(out["x3"]), w(out["native"])))
