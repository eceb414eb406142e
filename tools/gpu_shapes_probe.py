"""GPU probe (not a pytest): a few timed train steps at the other BASELINE shapes (C4 stage-2 shape, verbatim stage2 RVQ, C5 rank)."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (os.path.join(ROOT, "pytorch-vae_amd"), os.path.join(ROOT, "tests", "golden"), ROOT):
    sys.path.insert(0, p)
import torch
import gen_inputs as G
from models import vae_models
W = dict(ss_weight=0.8, rmsd_weight=1.8, xyz_tv_lambda=0.0008, bond_length_weight=0.015, bond_angle_weight=0.006,
         pdm_weight=0.001, lr_pdm_weight=0.003, win_kabsch_weight=0.0006)
CASES = {"C2": (dict(G.C2_MODEL), 256, 64, 16.846),
         "C4_stage2": (dict(G.C2_MODEL), 1024, 256, 61.849),
         "stage2_verbatim_L256": (dict(G.C2_MODEL, num_quantizers=4, codebook_size=1024, code_dim=512), 256, 256, 63.359),
         "C5_rank": (dict(G.C2_MODEL, codebook_size=8192, code_dim=256), 64, 256, 62.453)}
for name in (sys.argv[1:] or list(CASES)):
    cfg, B, L, gf = CASES[name]
    torch.manual_seed(1265)
    m = vae_models["VQVAE"](**cfg).to("cuda").train()
    eng = m._engine()
    x, mask = G.curve_batch(B, L, 5)
    x, mask = x.cuda(), mask.cuda()
    for _ in range(3):
        m.train_step(x, mask, W, 2e-4, 0.008, 3.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        m.train_step(x, mask, W, 2e-4, 0.008, 3.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: B={B} L={L}: {dt*1e3:.1f} ms/step  {B/dt:.0f} samples/s  {B/dt*gf/1e3:.1f} TF/s step-level  loss={float(eng.metrics[0]):.4f} "
          f"mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del m, eng; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
