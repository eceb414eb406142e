"""GPU probe (not a pytest): does a memory-bound kernel on a second stream overlap with gemm_f32_dma (1 workgroup per CU,
144 KB LDS, 4 waves)?   Serial vs two-stream time of {GEMM 16384x512x512 NN} + {sum of 32 one-MB slabs, split-K reduce size}."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-vae_amd"))
import torch
from vqvae_hip import lib as L

dev = "cuda"
M, N, K = 16384, 512, 512
A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
ws = torch.empty(64 << 20, device=dev)
slabs = torch.randn(32, 512 * 512, device=dev); out = torch.empty(512 * 512, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    L.gemm(1, 0, M, N, K, A, K, B, N, C, N, ws=ws)


def red():
    L.call("vqh_reduce_slabs", slabs, 32, 512 * 512, 512 * 512, out, 0.0)


def timed(fn, iters=50):
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


def serial():
    gemm(); red()


def forked():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        red()
        ev2 = torch.cuda.Event(); ev2.record()
    gemm()
    torch.cuda.current_stream().wait_event(ev2)


print(f"gemm alone {timed(gemm):.1f} us, reduce alone {timed(red):.1f} us, serial {timed(serial):.1f} us, two streams {timed(forked):.1f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        forked()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(10):
        serial()
print(f"in a hipGraph (10 pairs): serial {timed(g2.replay, 20) / 10:.1f} us per pair, two streams {timed(g.replay, 20) / 10:.1f} us per pair")
