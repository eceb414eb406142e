"""GPU probe (not a pytest): timing of the skinny (head-shaped) GEMM paths."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L
dev = "cuda"
torch.manual_seed(0)
ws = torch.empty(1 << 24, device=dev)

def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

rows, H = 16384, 512
x = torch.randn(rows, H, device=dev)
for n_out, ld in ((3, 6), (3, 3), (6, 6), (8, 8)):
    dy = torch.randn(rows, ld, device=dev)
    W, b = torch.randn(n_out, H, device=dev), torch.randn(n_out, device=dev)
    out = torch.empty(rows, ld, device=dev)
    dW, db, dx = torch.empty(n_out, H, device=dev), torch.empty(n_out, device=dev), torch.empty(rows, H, device=dev)
    print(f"n_out={n_out} ld={ld}: fwd {t(lambda: L.gemm(1, 1, rows, n_out, H, x, H, W, H, out, ld, bias=b)):6.1f} us   "
          f"dgrad {t(lambda: L.gemm(1, 0, rows, H, n_out, dy, ld, W, H, dx, H)):6.1f} us   "
          f"wgrad {t(lambda: L.call('vqh_gemm_wgrad', rows, n_out, H, dy, ld, x, H, dW, H, db, 0.0, ws, ws.numel())):6.1f} us")
