# -*- coding: utf-8 -*-
"""Diagnostic (not a test): per-tensor gradient error of the HIP path against the fp32 oracle, next to the oracle's own
fp32-vs-fp64 distance, for one edge-case configuration.   python tools/gpu_graderr_probe.py [L] [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-vae_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import gen_inputs as G  # noqa: E402
from gen_inputs import O  # noqa: E402
import test_gpu_edges as E  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 350
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dict(G.SMALL_VQ, max_seq_len=350)
x, mask = G.smooth_curve_batch(B, L, 71, ragged=True)
(out, ld, eng, acts), (out_o, ld_o, sd, taps), (out_64, ld_64, sd64, taps64) = E._both(cfg, x, mask, G.ALL_LOSS_WEIGHTS)
print('ReLU flips hip vs o32:', E._relu_flips(acts, taps, mask))
print('ReLU flips o64 vs o32:', E._relu_flips({k: v.float() for k, v in taps64.items()}, taps, mask))
print(f"{'tensor':48s} {'max|g|':>10s} {'hip-o32':>10s} {'o32-o64':>10s} {'hip-o64':>10s}")
for k in O.param_shapes(O.make_cfg(**cfg)):
    g32, g64, gh = sd[k].grad.double(), sd64[k].grad, eng.G[k].cpu().double()
    print(f"{k:48s} {float(g32.abs().max()):10.3e} {float((gh - g32).abs().max()):10.3e} {float((g32 - g64).abs().max()):10.3e} "
          f"{float((gh - g64).abs().max()):10.3e}")
for k in ld_o:
    print(f"{k:28s} hip {float(ld[k]):.8g} o32 {float(ld_o[k]):.8g} o64 {float(ld_64[k]):.10g}")
