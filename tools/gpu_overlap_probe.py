"""GPU probe: do an independent dgrad and wgrad GEMM overlap usefully on two streams?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda"
torch.manual_seed(0)
M = 16384
for (N, K) in ((2048, 512), (512, 2048), (512, 512), (1536, 512)):
    dY = torch.randn(M, N, device=dev); W = torch.randn(N, K, device=dev) / 8; X = torch.randn(M, K, device=dev)
    dX = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ws1 = torch.empty(32 << 20, device=dev); ws2 = torch.empty(32 << 20, device=dev)
    s2 = torch.cuda.Stream()
    def seq():
        L.gemm(1, 0, M, K, N, dY, N, W, K, dX, K)
        L.call("vqh_gemm_wgrad", M, N, K, dY, N, X, K, dW, K, db, 0.0, ws1, ws1.numel())
    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            L.call("vqh_gemm_wgrad", M, N, K, dY, N, X, K, dW, K, db, 0.0, ws2, ws2.numel())
            e2 = torch.cuda.Event(); e2.record()
        L.gemm(1, 0, M, K, N, dY, N, W, K, dX, K)
        torch.cuda.current_stream().wait_event(e2)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    ts, tp = t(seq), t(par)
    fl = 4.0 * M * N * K
    print(f"N={N} K={K}: sequential {ts:7.1f} us ({fl/ts/1e6:5.1f} TF)   two streams {tp:7.1f} us ({fl/tp/1e6:5.1f} TF)", flush=True)
