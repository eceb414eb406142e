# -*- coding: utf-8 -*-
"""-m gpu: the drop-in entry (pytorch-vae_amd/run.py, same flags as the reference run.py:97-106) end to end on the HIP
model with synthetic curves: stage 1 (AE) -> checkpoint -> stage 2 (VQ) with --warm_start_ckpt (quantizer.* dropped)
and --init_codebook -> --resume_ckpt.  Two-stage config C4's flow at a small width."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from conftest import PKG

pytestmark = pytest.mark.gpu
SMALL = dict(hidden_dim=64, num_layers=1, num_heads=4, tokenizer_heads=4, tokenizer_layers=1, max_seq_len=48,
             code_dim=16, latent_tokens=8)


def _cfg(base, tmp, name, **over):
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", base)))
    cfg["model_params"].update(SMALL)
    cfg["model_params"].update(over.pop("model", {}))
    cfg["data_params"].update(train_batch_size=16, val_batch_size=16, num_workers=0, pin_memory=False,
                              synthetic={"n": 64, "n_val": 16, "max_len": 48, "min_len": 30, "seed": 3})
    cfg["exp_params"].update(checkpoint_dir=os.path.join(tmp, name), save_every_epochs=1, print_every=2)
    cfg["exp_params"].update(over.pop("exp", {}))
    cfg["trainer_params"].update(max_epochs=over.pop("epochs", 2), devices=over.pop("devices", 1), limit_val_batches=1.0)
    cfg["logging_params"] = {"save_dir": os.path.join(tmp, "logs"), "name": name}
    path = os.path.join(tmp, name + ".yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    return path


def _run(args):
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "run.py")] + args, cwd=PKG, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_two_stage_training_through_run_py(tmp_path):
    tmp = str(tmp_path)
    s1 = _cfg("stage1_ae.yaml", tmp, "s1")
    out1 = _run(["-c", s1])
    assert "[Epoch 1]" in out1 and "Training VQVAE-AEStage1" in out1
    ck1 = os.path.join(tmp, "s1", "last.ckpt")
    c1 = torch.load(ck1, map_location="cpu", weights_only=True)
    assert c1["epoch"] == 1 and all(k.startswith("model.") for k in c1["state_dict"]) and "optimizer_states" in c1
    assert not any(k.startswith("model.quantizer.") for k in c1["state_dict"])            # stage 1 has no quantizer
    cents = os.path.join(tmp, "centroids.npy")
    np.save(cents, (np.random.RandomState(0).randn(2, 32, 16) * 0.1).astype(np.float32))   # [levels, K_per, D]
    s2 = _cfg("stage2_vq.yaml", tmp, "s2", model=dict(num_quantizers=2, codebook_size=32), epochs=2)
    out2 = _run(["-c", s2, "--warm_start_ckpt", ck1, "--init_codebook", cents])
    assert "[WarmStart] loaded kept=" in out2 and "[Codebook Init] Loaded centroids" in out2 and "[Epoch 1]" in out2
    ck2 = os.path.join(tmp, "s2", "last.ckpt")
    c2 = torch.load(ck2, map_location="cpu", weights_only=True)
    assert c2["state_dict"]["model.quantizer.embedding"].shape == (64, 16)
    # encoder weights came from stage 1 and were then trained: same shape, finite
    assert all(torch.isfinite(v).all() for v in c2["state_dict"].values() if v.dtype.is_floating_point)
    # resume: one more epoch from the saved state (optimizer moments + LR position restored)
    s2b = _cfg("stage2_vq.yaml", tmp, "s2", model=dict(num_quantizers=2, codebook_size=32), epochs=3)
    out3 = _run(["-c", s2b, "--resume_ckpt", ck2])
    assert "[Resume]" in out3 and "[Epoch 2]" in out3 and "[Epoch 1]" not in out3
    c3 = torch.load(ck2, map_location="cpu", weights_only=True)
    assert c3["epoch"] == 2 and c3["global_step"] > c2["global_step"]


def test_two_rank_training_through_run_py(tmp_path):
    """The reference's single command (`python run.py -c cfg` with trainer_params.devices: 2, strategy ddp; run.py:191-218):
    run.py starts the 2 ranks itself as child processes (sharing the one GPU here, hence gloo).  DistributedSampler shards,
    the [grads | EMA stats] all-reduce keeps the ranks' weights identical, rank 0 writes checkpoints and the scalar log."""
    tmp = str(tmp_path)
    s2 = _cfg("stage2_vq.yaml", tmp, "ddp", model=dict(num_quantizers=1, codebook_size=32), epochs=2, devices=2)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PYTHONPATH=PKG, VQH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(PKG, "run.py"), "-c", s2], cwd=PKG, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[Launch] starting 2 ranks" in r.stdout and "[Epoch 1]" in r.stdout
    ck = torch.load(os.path.join(tmp, "ddp", "last.ckpt"), map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and ck["global_step"] == 2 * (64 // 2 // 16)        # 64 samples / 2 ranks / batch 16, 2 epochs
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.dtype.is_floating_point)
    import glob
    import json
    logs = glob.glob(os.path.join(tmp, "logs", "ddp", "version_0", "scalars.jsonl"))
    assert len(logs) == 1
    recs = [json.loads(ln) for ln in open(logs[0])]
    assert len(recs) == 2 and recs[-1]["epoch"] == 1 and "epoch/loss" in recs[-1] and "lr" in recs[-1]


def test_two_rank_training_under_torchrun(tmp_path):
    """The same entry under an external launcher (RANK / WORLD_SIZE exported): run.py must not spawn a second generation."""
    tmp = str(tmp_path)
    s2 = _cfg("stage2_vq.yaml", tmp, "ddp2", model=dict(num_quantizers=1, codebook_size=32), epochs=1, devices=2)
    env = dict(os.environ, PYTHONPATH=PKG, VQH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), os.path.join(PKG, "run.py"), "-c", s2],
                       cwd=PKG, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[Launch]" not in r.stdout and "[Epoch 0]" in r.stdout


def test_generate_sample_and_autograd_bridge_api():
    import gen_inputs as G
    from models import vae_models
    m = vae_models["VQVAE"](**G.SMALL_VQ)
    m.load_state_dict(G.model_state(G.SMALL_VQ, 2), strict=True)
    m = m.to("cuda").eval()
    x, mask = G.curve_batch(3, 20, 4, ragged=True)
    x, mask = x.cuda(), mask.cuda()
    rec = m.generate(x, mask)
    assert rec.shape == (3, 20, 6) and torch.isfinite(rec).all()
    assert torch.allclose(rec, m(x, mask)[0])
    smp = m.sample(5, "cuda", out_len=12)
    assert smp.shape == (5, 12, 6) and torch.isfinite(smp).all()
    assert m.sample(2, "cuda").shape == (2, m.max_seq_len, 6)
    m.train()
    m._engine().drop_scale = 0.0
    out = m(x, mask)
    ld = m.loss_function(*out, ss_weight=0.8, rmsd_weight=1.8)
    assert ld["loss"].requires_grad
    ld["loss"].backward()                                   # Lightning-style call -> HIP backward, grads exposed on params
    g = m.to_code.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    with torch.no_grad():                                   # no tape in no_grad / eval: plain tensors
        ld2 = m.loss_function(*out, ss_weight=0.8, rmsd_weight=1.8)
    assert not ld2["loss"].requires_grad and abs(float(ld2["loss"]) - float(ld["loss"])) < 1e-6


def test_gradient_diagnostics_report_after_hip_backward(capsys):
    """enable_grad_monitor / print_grad_summary (models/vq_vae.py:662-734): host-side prints over the flat gradient buffer."""
    import gen_inputs as G
    from models import vae_models
    m = vae_models["VQVAE"](**G.SMALL_VQ)
    m.load_state_dict(G.model_state(G.SMALL_VQ, 2), strict=True)
    m = m.to("cuda").train()
    m._engine().drop_scale = 0.0
    x, mask = G.curve_batch(3, 20, 4, ragged=True)
    m.enable_grad_monitor(True)
    ld = m.loss_function(*m(x.cuda(), mask.cuda()), ss_weight=0.8, rmsd_weight=1.8)
    m.backward()
    m.print_grad_summary()
    out = capsys.readouterr().out
    assert "[Grad Monitor] Enabled" in out and "[GRAD] to_code.weight: norm=" in out and "[GRAD-ERROR]" not in out
    n_params = len(list(m.parameters()))
    assert f"[Grad Summary] Total params with grad: {n_params}" in out
    for label in ("Geo branch", "SS branch", "Fusion", "Decoder"):
        assert f"  {label}: " in out
    m.eval()
    m.print_grad_summary()
    assert "eval mode" in capsys.readouterr().out
