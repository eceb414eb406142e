# -*- coding: utf-8 -*-
"""CPU suite: oracle/vqvae_oracle.py against the committed golden vectors (recorded from the real
reference by tests/golden/make_golden.py).  Tolerances are fp32 round-off of a different
summation order; indices must be bit-exact."""
import numpy as np
import pytest
import torch

import gen_inputs as G
from gen_inputs import O
from conftest import load_golden

torch.set_num_threads(4)


def _close(a, b, tol, what=""):
    a = (a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).double()
    b = (b.detach() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))).double()
    scale = max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol * scale, f"{what}: err {err:.3e} scale {scale:.3e}"


VQ_CASES = ["vq_k512_d64_fresh", "vq_k512_d64_cinit", "vq_k512_d64_eval", "vq_k512_d64_ties",
            "vq_k8192_d256", "vq_rvq4_k64_d32", "vq_rvq4_k1024_d512", "vq_tiny_store",
            "vq_k512_d64_masked", "vq_rvq4_k64_d32_masked"]


def vq_masks(g):
    """Valid-position masks of the VectorQuantizerEMA(mask=...) fixtures (None per step otherwise)."""
    B, M, steps = int(g["B"]), int(g["M"]), int(g["steps"])
    if "row_mask" in g and int(g["row_mask"]):
        return G.vq_row_masks(B, M, int(g["seed"]), steps)
    return [None] * steps


def vq_setup(g):
    B, M, K_per, D, Q = (int(g[k]) for k in ("B", "M", "K_per", "D", "Q"))
    seed, steps, scale = int(g["seed"]), int(g["steps"]), float(g["scale"])
    R = B * M
    zs = [G.vq_inputs(R, Q * K_per, D, seed + 17 * s, scale)[0] for s in range(steps)]
    emb0 = G.vq_inputs(R, Q * K_per, D, seed, scale)[1]
    if int(g["dup_codes"]):
        emb0[5] = emb0[3]
        emb0[K_per - 1] = emb0[0]
        zs[0][:4] = emb0[[3, 0, 7, 5]]
    assert abs(G.checksum(emb0) - float(g["emb0_sum"])) < 1e-6, "RNG drift: regenerate fixtures"
    return B, M, K_per, D, Q, steps, zs, emb0


@pytest.mark.parametrize("name", VQ_CASES)
def test_oracle_quantizer_matches_reference(name):
    g = load_golden(name)
    B, M, K_per, D, Q, steps, zs, emb0 = vq_setup(g)
    cfg = dict(codebook_size=K_per, code_dim=D, num_quantizers=Q, use_vq=True)
    sd = {k: torch.zeros(s) for k, s in O.buffer_shapes(O.make_cfg(**cfg)).items() if k.startswith("quantizer.")}
    sd["quantizer.embedding"] = emb0.clone()
    if int(g["centroid_init"]):
        sd["quantizer.ema_embedding"] = emb0.clone()
        sd["quantizer.ema_cluster_size"] = torch.ones(Q * K_per)
    orc = O.OracleVQVAE(sd, **cfg)
    orc.training = bool(int(g["train"]))
    masks = vq_masks(g)
    for s in range(steps):
        z = zs[s].view(B, M, D)
        assert abs(G.checksum(z) - float(g[f"z_sum_{s}"])) < 1e-6
        _, zq, idx, st = orc.quantize(z, do_ema_update=True, mask=masks[s])
        assert np.array_equal(idx.reshape(-1).numpy().astype(np.int32), g[f"idx_{s}"]), "indices must be bit-exact"
        _close(st, g[f"stats_{s}"], 1e-5, "stats")
        _close(zq.reshape(-1, D)[:8], g[f"zq_head_{s}"], 1e-6, "zq")
        _close(sd["quantizer.ema_cluster_size"], g[f"ecs_{s}"], 2e-6, "ema_cluster_size")
        if f"emb_{s}" in g:
            _close(sd["quantizer.embedding"], g[f"emb_{s}"], 2e-6, "embedding")
            _close(sd["quantizer.ema_embedding"], g[f"eemb_{s}"], 2e-6, "ema_embedding")
        else:
            _close(sd["quantizer.embedding"][:16], g[f"emb_head_{s}"], 2e-6, "embedding head")
        _close(sd["quantizer._ep_usage"], g[f"ep_usage_{s}"], 0, "ep_usage")
        _close(sd["quantizer._ep_cnt"], g[f"ep_cnt_{s}"], 0, "ep_cnt")


MODEL_CASES = [("model_small_vq_full", G.SMALL_VQ, False), ("model_small_vq_ragged", G.SMALL_VQ, True),
               ("model_small_rvq_ragged", G.SMALL_RVQ, True), ("model_small_ae", G.SMALL_AE, False)]
SOFTVQ_CFG = dict(G.SMALL_VQ, soft_vq_use=True, soft_vq_tau_start=2.0, soft_vq_tau_end=0.5, soft_vq_tau_warm_steps=10,
                  soft_vq_alpha_warm_steps=20)
UENT_CFG = dict(G.SMALL_VQ, usage_entropy_lambda=0.05)
EXTRA_CASES = [("model_small_softvq", SOFTVQ_CFG, True), ("model_small_uent", UENT_CFG, True)]


def model_inputs(g, cfg_kw):
    B, L, seed, steps = int(g["B"]), int(g["L"]), int(g["seed"]), int(g["steps"])
    fn = G.smooth_curve_batch if int(g["smooth"]) else G.curve_batch
    batches = [fn(B, L, seed + 100 + s, bool(int(g["ragged"]))) for s in range(steps)]
    sd0 = G.model_state(cfg_kw, seed)
    assert abs(G.checksum(torch.cat([v.reshape(-1) for v in sd0.values()])) - float(g["state_sum"])) < 1e-5, \
        "RNG drift: regenerate fixtures"
    weights = {k: float(v) for k, v in zip(g["weights_keys"], g["weights_vals"])}
    return batches, sd0, weights


@pytest.mark.parametrize("name,cfg_kw,_r", MODEL_CASES + EXTRA_CASES)
def test_oracle_train_step_matches_reference(name, cfg_kw, _r):
    g = load_golden(name)
    batches, sd0, weights = model_inputs(g, cfg_kw)
    cfg = O.make_cfg(**cfg_kw)
    sd = O.attach_grads({k: v.clone() for k, v in sd0.items()}, cfg)
    orc = O.OracleVQVAE(sd, drop_scale=0.0, **cfg_kw)
    orc.training_steps = int(g["start_steps"]) if "start_steps" in g else 1
    opt = torch.optim.AdamW(orc.params(), lr=float(g["lr"]), weight_decay=float(g["wd"]))
    pnames = list(g["param_names"])
    for s, (x, mask) in enumerate(batches):
        assert abs(G.checksum(x) - float(g[f"x_sum_{s}"])) < 1e-6
        ld, out, gn = orc.train_step(x, mask, opt, float(g["clip"]), weights)
        _close(out[0], g[f"recons_{s}"], 2e-5, "recons")
        _close(out[2][1], g[f"z_e_{s}"], 2e-5, "z_e")
        if cfg["use_vq"]:
            assert np.array_equal(out[2][2].reshape(-1).numpy().astype(np.int32), g[f"idx_{s}"])
        for k, v in zip(g[f"loss_keys_{s}"], g[f"loss_vals_{s}"]):
            _close(ld[str(k)], v, 3e-5, f"loss[{k}]")
        _close(gn, g[f"grad_norm_{s}"], 1e-4, "grad norm")
        gne = np.array([float(sd[k].grad.norm()) for k in pnames])
        assert np.allclose(gne, g[f"gradnorm_each_{s}"], rtol=2e-3, atol=1e-6 * float(g[f"grad_norm_{s}"]) + 1e-9)
        for key in g:
            if key.startswith(f"grad_{s}::"):
                _close(sd[key.split("::")[1]].grad, g[key], 2e-4, key)
        if cfg["use_vq"]:
            _close(sd["quantizer.embedding"], g[f"q_emb_{s}"], 1e-5, "codebook")
            _close(sd["quantizer.ema_cluster_size"], g[f"q_ecs_{s}"], 1e-6, "ema_cluster_size")


@pytest.mark.parametrize("name,cfg_kw,_r", MODEL_CASES)
def test_oracle_eval_forward_and_decode(name, cfg_kw, _r):
    g = load_golden(name)
    batches, sd0, weights = model_inputs(g, cfg_kw)
    orc = O.OracleVQVAE({k: v.clone() for k, v in sd0.items()}, drop_scale=0.0, **cfg_kw)
    orc.training = False
    x, mask = batches[0]
    with torch.no_grad():
        out = orc.forward(x, mask)
        ld = orc.loss_function(*out, **weights)
        _close(out[0], g["eval_recons"], 2e-5, "eval recons")
        if orc.use_vq:
            assert np.array_equal(out[2][2].reshape(-1).numpy().astype(np.int32), g["eval_idx"])
        for k, v in zip(g["eval_loss_keys"], g["eval_loss_vals"]):
            _close(ld[str(k)], v, 3e-5, f"eval loss[{k}]")
        _close(orc.decode(out[2][0], mask), g["eval_decode"], 2e-5, "decode")


@pytest.mark.parametrize("name,cfg_kw", [("loss_all_ragged", dict(G.SMALL_VQ, usage_entropy_lambda=0.01)),
                                         ("loss_all_full", G.SMALL_VQ), ("loss_short", G.SMALL_VQ),
                                         ("loss_datastats", G.SMALL_VQ)])
def test_oracle_loss_function_and_input_grads(name, cfg_kw):
    g = load_golden(name)
    sd0 = G.model_state(cfg_kw, int(g["seed"]))
    orc = O.OracleVQVAE(sd0, drop_scale=0.0, **cfg_kw)
    if "stats_std" in g:
        orc.data_mean = torch.from_numpy(g["stats_mean"]).view(1, 1, 3)
        orc.data_std = torch.from_numpy(g["stats_std"]).view(1, 1, 3)
    weights = {k: float(v) for k, v in zip(g["weights_keys"], g["weights_vals"])}
    x, mask = torch.from_numpy(g["x"]), torch.from_numpy(g["mask"])
    B, Nt = g["ze"].shape[:2]
    for tag, m in (("m", mask), ("nomask", None)):
        rec = torch.from_numpy(g["recons"]).clone().requires_grad_(True)
        ze = torch.from_numpy(g["ze"]).clone().requires_grad_(True)
        pack = (torch.from_numpy(g["zq"]), ze, torch.zeros(B, Nt, dtype=torch.long), torch.tensor(3.0), torch.tensor(0.5))
        ld = orc.loss_function(rec, x, pack, m, **weights)
        ld["loss"].backward()
        for k, v in zip(g[f"{tag}_loss_keys"], g[f"{tag}_loss_vals"]):
            _close(ld[str(k)], v, 2e-5, f"{tag} loss[{k}]")
        _close(rec.grad, g[f"{tag}_d_recons"], 1e-4, "d_recons")
        _close(ze.grad, g[f"{tag}_d_ze"], 1e-5, "d_ze")


def test_oracle_state_dict_layout_matches_reference_keys():
    """Key names/shapes of the reference state_dict (recorded at fixture time) == oracle layout."""
    for name, cfg_kw in (("init_small_vq_seed1265", G.SMALL_VQ), ("init_c2_seed1265", G.C2_MODEL),
                         ("init_small_ae_seed7", G.SMALL_AE)):
        g = load_golden(name)
        cfg = O.make_cfg(**cfg_kw)
        mine = dict(O.param_shapes(cfg))
        mine.update(O.buffer_shapes(cfg))
        ref = {str(k): tuple(int(v) for v in str(s).split(",") if v) for k, s in zip(g["keys"], g["shapes"])}
        assert mine == ref
        assert sum(int(np.prod(s)) for s in O.param_shapes(cfg).values()) == int(g["n_params"])


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="reference tree absent (GPU box)")
def test_oracle_matches_live_reference_with_dropout_off():
    """Build container only: a fresh comparison against the imported reference on a new seed."""
    import importlib.util
    import warnings
    warnings.filterwarnings("ignore")
    # load the reference module under its own name: `models` may already be OUR package in this process
    spec = importlib.util.spec_from_file_location("reference_vq_vae", "/root/reference/models/vq_vae.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Ref = mod.VQVAE
    cfg_kw = dict(G.SMALL_VQ)
    sd0 = G.model_state(cfg_kw, 777)
    ref = Ref(**cfg_kw)
    ref.load_state_dict(sd0, strict=True)
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    ref.train()
    ref.training_steps = 1
    orc = O.OracleVQVAE({k: v.clone() for k, v in sd0.items()}, drop_scale=0.0, **cfg_kw)
    orc.training_steps = 1
    x, mask = G.curve_batch(3, 19, 778, ragged=True)
    with torch.no_grad():
        r, o = ref(x, mask), orc.forward(x, mask)
    _close(o[0], r[0], 2e-5, "recons")
    assert torch.equal(o[2][2], r[2][2])
