"""GPU probe (not a pytest): 300 fused train steps of the C2 model (B = 64, L = 64, dropout on, fresh batch every step) with the
split-operand GEMM tiles and with the native fp32 MFMA tiles, same seeds: the two loss curves must stay together (they are
not bitwise equal: round-off differs, and training amplifies it) and neither may produce a non-finite value."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "pytorch-vae_amd"), ROOT):
    sys.path.insert(0, p)
import torch
import gen_inputs as G
from models import vae_models
from vqvae_hip import lib as L

dev = "cuda:0"
w = dict(G.BASE_LOSS_WEIGHTS)
curves = {}
for name, flag in (("x3", 0), ("native", L.GEMM_FLAG_NATIVE_F32)):
    old = L.lib().vqh_gemm_set_flags(1 | flag)
    torch.manual_seed(11)
    m = vae_models["VQVAE"](**dict(G.C2_MODEL)).to(dev).train()
    eng = m._engine()
    losses = []
    for step in range(300):
        x, mask = G.curve_batch(64, 64, 1000 + step, ragged=True)
        eng.train_step(x.to(dev), mask.to(dev), w, 2e-4, 0.01, 1.0, use_graph=True)
        if step % 10 == 9:
            losses.append(float(eng.metrics_dict(w)["loss"]))
    torch.cuda.synchronize()
    L.lib().vqh_gemm_set_flags(old)
    curves[name] = losses
    assert all(v == v and abs(v) < 1e9 for v in losses), name
print("step   x3        native")
for i, (a, b) in enumerate(zip(curves["x3"], curves["native"])):
    print(f"{10 * i + 9:4d}  {a:9.4f}  {b:9.4f}")
a, b = curves["x3"][-5:], curves["native"][-5:]
print("mean of the last 5 samples: x3 %.4f native %.4f" % (sum(a) / 5, sum(b) / 5))
