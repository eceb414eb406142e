"""GPU probe (not a pytest): 300 fused train steps of the C2 model (dropout on, fresh ragged batch every step) with every kernel
family on the bf16 matrix pipes (split-operand GEMM tiles, attention, VQ scores: the defaults) and with all of them on the native
fp32 MFMA (gemm flag 512, attention flag 4, vq flag 2), same seeds: the two loss curves must stay together (they are not bitwise
equal: round-off differs, and training amplifies it) and neither may produce a non-finite value.  Two shapes: B = 64, L = 64
(short-sequence attention kernels) and B = 32, L = 160 (general attention kernels, two length buckets)."""
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
for B, Lmax in ((64, 64), (32, 160)):
    curves = {}
    for name, gf, af, vf in (("x3", 0, 0, 0), ("native", L.GEMM_FLAG_NATIVE_F32, 4, 2)):
        old = L.lib().vqh_gemm_set_flags(1 | gf)
        old_a = L.lib().vqh_attn_set_flags(af)
        old_v = L.lib().vqh_vq_set_flags(vf)
        torch.manual_seed(11)
        m = vae_models["VQVAE"](**dict(G.C2_MODEL)).to(dev).train()
        eng = m._engine()
        losses = []
        for step in range(300):
            x, mask = G.curve_batch(B, Lmax - (step % 3) * 9, 1000 + step, ragged=True)
            eng.train_step(x.to(dev), mask.to(dev), w, 2e-4, 0.01, 1.0, use_graph=True)
            if step % 10 == 9:
                losses.append(float(eng.metrics_dict(w)["loss"]))
        torch.cuda.synchronize()
        L.lib().vqh_gemm_set_flags(old)
        L.lib().vqh_attn_set_flags(old_a)
        L.lib().vqh_vq_set_flags(old_v)
        curves[name] = losses
        assert all(v == v and abs(v) < 1e9 for v in losses), name
    print(f"B = {B}, L <= {Lmax}:  step   bf16 pipes   native fp32")
    for i, (a, b) in enumerate(zip(curves["x3"], curves["native"])):
        if i % 3 == 2:
            print(f"{10 * i + 9:4d}  {a:9.4f}  {b:9.4f}")
    a, b = curves["x3"][-5:], curves["native"][-5:]
    print("mean of the last 5 samples: bf16 pipes %.4f native %.4f" % (sum(a) / 5, sum(b) / 5), flush=True)
