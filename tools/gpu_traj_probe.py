# -*- coding: utf-8 -*-
"""Diagnostic (not a test): per-step metric drift of the fused HIP train_step against the oracle's fp32 trajectory, next to
the oracle's own fp32-vs-fp64 drift.   python tools/gpu_traj_probe.py [vq|rvq]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-vae_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import gen_inputs as G  # noqa: E402
from gen_inputs import O  # noqa: E402
import test_gpu_train_step as T  # noqa: E402
from vqvae_hip.engine import METRIC_KEYS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rvq"
cfg_kw = dict(G.SMALL_VQ if name == "vq" else G.SMALL_RVQ)
sd0 = G.model_state(cfg_kw, 501)
weights = dict(G.BASE_LOSS_WEIGHTS, xyz_tv_lambda=0.001, bond_length_weight=0.01, dih_weight=0.02)
lr, wd, clip = 1e-3, 0.01, 1.0
seq = [("train", G.smooth_curve_batch(5, 24, 600 + i, ragged=True)) for i in range(6)]
m, eng = T._model(cfg_kw, sd0)
m.training_steps = 1
mets, gn, snaps = [], [], []
for kind, (x, mask) in seq:
    m.train_step(x, mask, weights, lr, wd, clip, use_graph=False)
    mets.append(dict(zip(METRIC_KEYS, eng.metrics.tolist())))
    gn.append(float(eng.norm[0]))
    sn = {k: eng.P[k].detach().cpu().double().clone() for k in eng.P}
    sn.update({"quantizer." + k: b.detach().cpu().double().clone() for k, b in m.quantizer.named_buffers()})
    sn.update({"grad:" + k: eng.G[k].detach().cpu().double().clone() for k in eng.P})
    snaps.append(sn)
for n in range(1, len(seq) + 1):      # prefix runs: state after step n-1
    sd32, l32, i32 = T._oracle_traj(cfg_kw, sd0, seq[:n], weights, lr, wd, clip, torch.float32)
    sd64, l64, i64 = T._oracle_traj(cfg_kw, sd0, seq[:n], weights, lr, wd, clip, torch.float64)
    rows = []
    for k in snaps[n - 1]:
        if k not in sd32 or not sd32[k].is_floating_point():
            continue
        a, b, c = snaps[n - 1][k], sd32[k].detach().double(), sd64[k].detach().double()
        rows.append((float((a - b).norm()) / max(float((b - c).norm()), 1e-30), k, float((a - b).norm()), float((b - c).norm())))
    grows = []
    for k in sd32:
        if getattr(sd32[k], "grad", None) is None:
            continue
        a, b, c = snaps[n - 1]["grad:" + k], sd32[k].grad.double(), sd64[k].grad.double()
        grows.append((float((a - b).norm()) / max(float((b - c).norm()), 1e-30), k, float((a - b).norm()), float((b - c).norm()), float(b.norm())))
    print(f"=== clipped gradient of step {n - 1}: worst ||hip-o32|| / ||o32-o64||")
    for r in sorted(grows, reverse=True)[:8]:
        print(f"  {r[1]:48s} ratio {r[0]:8.2f}  hip-o32 {r[2]:.3e}  o32-o64 {r[3]:.3e}  ||g|| {r[4]:.3e}")
    print(f"=== state after step {n - 1}: worst ||hip-o32|| / ||o32-o64||")
    for r in sorted(rows, reverse=True)[:6]:
        print(f"  {r[1]:48s} ratio {r[0]:8.2f}  hip-o32 {r[2]:.3e}  o32-o64 {r[3]:.3e}")
    for k in ("quantizer.embedding", "quantizer.ema_cluster_size", "quantizer.ema_embedding"):
        r = [x for x in rows if x[1] == k][0]
        print(f"  {r[1]:48s} ratio {r[0]:8.2f}  hip-o32 {r[2]:.3e}  o32-o64 {r[3]:.3e}")
for i in range(len(seq)):
    print(f"--- step {i}")
    for k in l32[i]:
        a, b, c = mets[i][k], l32[i][k], l64[i][k]
        den = max(abs(b), 1e-30)
        print(f"  {k:26s} hip-o32 {abs(a - b) / den:9.2e}   o32-o64 {abs(b - c) / den:9.2e}   hip-o64 {abs(a - c) / den:9.2e}")
print("weights after the run: ||hip-o32|| / ||o32-o64||, worst tensors")
rows = []
for k in O.param_shapes(O.make_cfg(**cfg_kw)):
    a, b, c = eng.P[k].detach().cpu().double(), sd32[k].detach().double(), sd64[k].detach()
    rows.append((float((a - b).norm()) / max(float((b - c).norm()), 1e-30), k, float((a - b).norm()), float((b - c).norm())))
for r in sorted(rows, reverse=True)[:12]:
    print(f"  {r[1]:48s} ratio {r[0]:8.2f}  hip-o32 {r[2]:.3e}  o32-o64 {r[3]:.3e}")
