"""GPU probe (not a pytest): cost of the fused GEMM epilogues on the C2 shapes, per kernel family (flags 1 = 256x128 LDS-DMA
kernel, 129 = 128x128 register-staged kernel).   python tools/gpu_epilogue_probe.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

torch.manual_seed(0)
dev = "cuda"
ws = torch.empty(64 * 1024 * 1024, device=dev)
rng = torch.tensor([1234, 7], device=dev, dtype=torch.int64)
CASES = [("linear+bias NT", 1, 1, 16384, 512, 512, L.EPI_LINEAR, True, False, 0.0),
         ("linear NT", 1, 1, 16384, 512, 512, L.EPI_LINEAR, False, False, 0.0),
         ("drop_resid NT", 1, 1, 16384, 512, 512, L.EPI_DROP_RESID, True, True, 0.1),
         ("drop_resid p=0 NT", 1, 1, 16384, 512, 512, L.EPI_DROP_RESID, True, True, 0.0),
         ("drop_resid NT K2048", 1, 1, 16384, 512, 2048, L.EPI_DROP_RESID, True, True, 0.1),
         ("linear+bias NT N1536", 1, 1, 16384, 1536, 512, L.EPI_LINEAR, True, False, 0.0),
         ("relu_drop NT N2048", 1, 1, 16384, 2048, 512, L.EPI_RELU_DROP, True, False, 0.1),
         ("linear NT N2048", 1, 1, 16384, 2048, 512, L.EPI_LINEAR, False, False, 0.0),
         ("posmask NN N2048", 1, 0, 16384, 2048, 512, L.EPI_MUL_POSMASK, False, True, 0.1),
         ("linear NN N2048", 1, 0, 16384, 2048, 512, L.EPI_LINEAR, False, False, 0.0),
         ("linear NN N512 K1536", 1, 0, 16384, 512, 1536, L.EPI_LINEAR, False, False, 0.0)]


def timeit(akc, bkc, M, N, K, mode, use_bias, use_aux, p, iters=20):
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    B = torch.randn((N, K) if bkc else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev) if use_bias else None
    aux = torch.randn(M, N, device=dev) if use_aux else None

    def run():
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), C, N, bias=bias, mode=mode, aux_in=aux, ldaux=N, rng=rng, site=3,
               p=p, ws=ws)
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


flagsets = [int(f) for f in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "129"])]
print("case".ljust(26), *[f"flags={f}".rjust(20) for f in flagsets])
for c in CASES:
    row = []
    for f in flagsets:
        L.lib().vqh_gemm_set_flags(f)
        us = timeit(*c[1:])
        row.append(f"{us:9.1f}us {2.0 * c[3] * c[4] * c[5] / us / 1e6:6.1f}TF")
    print(c[0].ljust(26), *row, flush=True)
