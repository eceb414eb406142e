"""GPU probe (not a pytest): per-kernel breakdown of the GEMM launches of one C2 training step (HIP-event timed)."""
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
R = 3
prof = L.gemm_profile(lambda: [eng.train_step(x, mask, weights, hp["lr"], hp["wd"], hp["clip"], use_graph=False) for _ in range(R)])
tot = sum(v[1] for v in prof.values()) / R
print(f"total GEMM main-kernel time/step {tot*1e3:.2f} ms")
print("kernel <a_kc,b_kc,MODE>   n/step  us/launch  TF/s   ms/step")
for k, (n, t, f) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"{L.gemm_kernel_name(k):44s} {n//R:6d} {t/n*1e6:10.1f} {f/t/1e12:6.1f} {t/R*1e3:8.3f}")
