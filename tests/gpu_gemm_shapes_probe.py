"""GPU probe (not a pytest): per-shape breakdown of the GEMM launches of one C2 training step (HIP-event timed)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
import bench
from vqvae_hip import lib as L
from models import vae_models
mp, weights, hp = bench.c2_setup()
torch.manual_seed(hp["seed"])
m = vae_models["VQVAE"](**mp).to("cuda").train()
eng = m._engine()
x, mask = bench.synthetic_batch(256, 64, 1000, "cuda")
for _ in range(3):
    eng.train_step(x, mask, weights, hp["lr"], hp["wd"], hp["clip"], use_graph=False)
L.PROFILE = []
R = 3
for _ in range(R):
    eng.train_step(x, mask, weights, hp["lr"], hp["wd"], hp["clip"], use_graph=False)
torch.cuda.synchronize()
recs, L.PROFILE = L.PROFILE, None
by = {}
for (v, M, N, K, e0, e1) in recs:
    d = by.setdefault((v, M, N, K), [0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1) * 1e-3
tot = sum(d[1] for d in by.values()) / R
print(f"total GEMM time/step {tot*1e3:.2f} ms")
print("variant(ta,tb)  M      N      K      n/step  us/launch  TF/s   ms/step  lost_vs_135TF_ms")
for (v, M, N, K), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * M * N * K
    print(f"{v:8s} {M:6d} {N:6d} {K:6d} {n//R:6d} {t/n*1e6:10.1f} {fl*n/t/1e12:6.1f} {t/R*1e3:8.3f} {(t - fl*n/135e12)/R*1e3:8.3f}")
