# -*- coding: utf-8 -*-
"""
Generate tests/golden/*.npz by running the REAL reference (/root/reference, read-only) on seeded
synthetic inputs, in the build container only.  The reference has no tests or golden vectors of
its own, so these recorded outputs are the parity pin for oracle/ and for
the HIP path.  Fixtures hold seeds, expected outputs and input checksums -- never reference code.

    python tests/golden/make_golden.py            # regenerate everything

While generating, every case is also evaluated with oracle/vqvae_oracle.py and the script aborts
if the oracle disagrees with the reference beyond fp32 round-off, so a committed fixture set
implies "oracle == reference" on these inputs at generation time.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs as G  # noqa: E402
from gen_inputs import O  # noqa: E402

REF = "/root/reference"
if not os.path.isdir(REF):
    print("reference tree not present: nothing to do (fixtures are committed)")
    sys.exit(0)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
from models.vq_vae import VQVAE as RefVQVAE, VectorQuantizerEMA as RefVQ  # noqa: E402

torch.set_num_threads(8)


def np_(t):
    return t.detach().cpu().numpy().copy()


def zero_dropout(m):
    """parity runs force every dropout site to p=0."""
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = max(b.abs().max().item() if b.numel() else 0.0, 1e-30)
    assert err <= tol * max(1.0, ref), f"oracle != reference for {what}: abs {err:.3e} (scale {ref:.3e})"
    return err


def oracle64_step(cfg_kw, state32, x, mask, weights, clip, start_steps, train=True):
    """fp64 re-evaluation of ONE step from the reference's fp32 pre-step state (the arbiter of how much of
    a difference is fp32 round-off).  The reference itself is not dtype-generic (one-hot / eye / zeros are built as
    float32: models/vq_vae.py:81-83, :959), so the re-evaluation runs the oracle -- asserted equal to the reference in
    fp32 on this very step -- on .double() copies of the reference's weights and buffers."""
    cfg = O.make_cfg(**cfg_kw)
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.detach().clone()) for k, v in state32.items()}
    sd = O.attach_grads(sd, cfg)
    orc = O.OracleVQVAE(sd, drop_scale=0.0, **cfg_kw)
    orc.training = train
    orc.training_steps = start_steps
    out = orc.forward(x.double(), mask)
    ld = orc.loss_function(*out, **weights)
    grads, gn = {}, None
    if train:
        ld["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(orc.params(), clip) if clip and clip > 0 else None
        grads = {k: sd[k].grad.detach().clone() for k in O.param_shapes(cfg)}
    return out, ld, grads, gn


def close_step(a, b, lr, what, nsteps=1):
    """Post-AdamW weights: the first Adam update is ~lr*sign(g), so an element whose gradient is
    at round-off level (e.g. the key bias of an attention in_proj, whose true gradient is zero)
    may legitimately move by up to 2*lr; elements with a resolved gradient must agree tightly."""
    d = (a.detach().double() - b.detach().double()).abs()
    assert d.max().item() <= 2.1 * lr * nsteps, \
        f"oracle != reference for {what}: max {d.max().item():.3e} (lr {lr})"


# ------------------------------------------------------------------------------------------
# 1. quantizer alone (models/vq_vae.py:170-283)
# ------------------------------------------------------------------------------------------
def vq_case(name, B, M, K_per, D, Q, seed, steps=1, centroid_init=False, train=True, scale=1.0,
            dup_codes=False, store_inputs=False, row_mask=False):
    R = B * M
    masks = G.vq_row_masks(B, M, seed, steps) if row_mask else [None] * steps
    z_all = [G.vq_inputs(R, Q * K_per, D, seed + 17 * s, scale)[0] for s in range(steps)]
    emb0 = G.vq_inputs(R, Q * K_per, D, seed, scale)[1]
    if dup_codes:            # exact ties (first index must win) and exact hits (distance 0)
        emb0[5] = emb0[3]
        emb0[K_per - 1] = emb0[0]
        z_all[0][:4] = emb0[[3, 0, 7, 5]]
    ref = RefVQ(K_per, D, beta=0.25, decay=0.98, reinit_dead_codes=False, print_init=False,
                num_quantizers=Q)
    ref.embedding.copy_(emb0)
    cfg = dict(codebook_size=K_per, code_dim=D, num_quantizers=Q, use_vq=True, print_init=False)
    sd = {k: torch.zeros(s) for k, s in O.buffer_shapes(O.make_cfg(**cfg)).items() if k.startswith("quantizer.")}
    sd["quantizer.embedding"] = emb0.clone()
    if centroid_init:        # models/vq_vae.py:604-612
        ref.ema_embedding.copy_(emb0)
        ref.ema_cluster_size.fill_(1.0)
        sd["quantizer.ema_embedding"] = emb0.clone()
        sd["quantizer.ema_cluster_size"] = torch.ones(Q * K_per)
    orc = O.OracleVQVAE(sd, **cfg)
    ref.train(train)
    orc.training = train
    out = {"B": B, "M": M, "K_per": K_per, "D": D, "Q": Q, "seed": seed, "steps": steps,
           "centroid_init": int(centroid_init), "train": int(train), "scale": scale,
           "dup_codes": int(dup_codes), "emb0_sum": G.checksum(emb0), "row_mask": int(row_mask)}
    for s in range(steps):
        z = z_all[s].view(B, M, D)
        zst, zq, idx, st = ref(z, do_ema_update=True, allow_reinit=False, mask=masks[s])
        o_zst, o_zq, o_idx, o_st = orc.quantize(z, do_ema_update=True, mask=masks[s])
        assert torch.equal(idx.reshape(-1), o_idx.reshape(-1)), f"{name}: oracle indices differ"
        close(o_zq, zq, 1e-6, f"{name} z_q")
        close(o_st, st, 1e-5, f"{name} stats")
        for k in ("embedding", "ema_cluster_size", "ema_embedding", "_ep_usage", "_ep_cnt"):
            close(sd["quantizer." + k], getattr(ref, k), 2e-6, f"{name} {k}")
        # top-2 gap of the fp32 distances as the reference computes them (level 0 only for RVQ)
        with torch.no_grad():
            out[f"z_sum_{s}"] = G.checksum(z)
            out[f"idx_{s}"] = np_(idx.reshape(-1)).astype(np.int32)
            out[f"stats_{s}"] = np_(st)
            out[f"zq_sum_{s}"] = G.checksum(zq)
            out[f"zq_head_{s}"] = np_(zq.reshape(-1, D)[:8])
            out[f"ecs_{s}"] = np_(ref.ema_cluster_size)
            if Q * K_per * D <= 65536:
                out[f"emb_{s}"] = np_(ref.embedding)
                out[f"eemb_{s}"] = np_(ref.ema_embedding)
            else:
                out[f"emb_sum_{s}"] = G.checksum(ref.embedding)
                out[f"eemb_sum_{s}"] = G.checksum(ref.ema_embedding)
                out[f"emb_head_{s}"] = np_(ref.embedding[:16])
            out[f"ep_usage_{s}"] = np_(ref._ep_usage)
            out[f"ep_cnt_{s}"] = np_(ref._ep_cnt)
            # get_epoch_stats() of the reference itself (models/vq_vae.py:118-164): the dict the harness prints per epoch
            es = ref.get_epoch_stats()
            out[f"epstats_{s}"] = np.array([float(es["perplexity"]), float(es["dead_ratio"]), float(es["n_positions"]),
                                            float(es["margin_mean"]), float(es["qe_mean"]), float(es["qe_p90"])], dtype=np.float64)
            assert np.array_equal(np_(es["usage_hist"]), np_(ref._ep_usage))
    if store_inputs:
        out["z_0_full"] = np_(z_all[0])
        out["emb0_full"] = np_(emb0)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: R={R} K={Q}x{K_per} D={D} steps={steps}")


# ------------------------------------------------------------------------------------------
# 2. whole model: forward + loss + backward + clip + AdamW (experiment.py:351-476)
# ------------------------------------------------------------------------------------------
GRAD_KEYS_FULL = ["input_proj.weight", "input_proj.bias", "ss_input_proj.weight", "to_code.weight",
                  "to_code.bias", "from_code.weight", "head_xyz.weight", "head_ss.weight",
                  "head_xyz.bias", "tokenizer.queries", "query_embed.weight", "enc_ln.weight",
                  "fuse_mlp.3.bias", "mem_ln.weight", "encoder.layers.0.norm1.weight",
                  "encoder.layers.0.self_attn.in_proj_bias", "decoder.layers.0.norm3.bias",
                  "tokenizer.layers.0.ln_kv.weight", "tokenizer.layers.1.ffn.2.bias",
                  "decoder.layers.0.multihead_attn.out_proj.bias"]


def model_case(name, cfg_kw, B, L, seed, ragged, weights, lr=1e-3, wd=0.01, clip=1.0,
               smooth=False, steps=1, full_grads=True, eval_too=True, start_steps=1):
    cfg = O.make_cfg(**cfg_kw)
    sd0 = G.model_state(cfg_kw, seed)
    batches = [(G.smooth_curve_batch if smooth else G.curve_batch)(B, L, seed + 100 + s, ragged)
               for s in range(steps)]
    ref = RefVQVAE(**cfg_kw)
    missing = ref.load_state_dict(sd0, strict=True)          # key/shape compatibility check
    zero_dropout(ref)
    ref.train()
    ref.training_steps = start_steps                         # >0: skip the step-0 grad-summary print
    opt_r = torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=wd)

    sd = O.attach_grads({k: v.clone() for k, v in sd0.items()}, cfg)
    orc = O.OracleVQVAE(sd, drop_scale=0.0, **cfg_kw)
    orc.training_steps = start_steps
    opt_o = torch.optim.AdamW(orc.params(), lr=lr, weight_decay=wd)

    out = {"B": B, "L": L, "seed": seed, "ragged": int(ragged), "lr": lr, "wd": wd, "clip": clip,
           "smooth": int(smooth), "steps": steps, "start_steps": start_steps, "state_sum": G.checksum(torch.cat([v.reshape(-1) for v in sd0.values()]))}
    out["weights_keys"] = np.array(sorted(weights.keys()))
    out["weights_vals"] = np.array([float(weights[k]) for k in sorted(weights.keys())])
    pnames = list(O.param_shapes(cfg).keys())
    for s in range(steps):
        x, mask = batches[s]
        out[f"x_sum_{s}"] = G.checksum(x)
        o64, ld64, g64, gn64 = oracle64_step(cfg_kw, ref.state_dict(), x, mask, weights, clip, ref.training_steps)
        opt_r.zero_grad(set_to_none=True)
        r = ref(x, mask)
        ld = ref.loss_function(*r, **weights)
        ld["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
        grads_r = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}   # post-clip
        opt_r.step()

        ldo, ro, gno = orc.train_step(x, mask, opt_o, clip, weights)
        # ---- oracle vs reference -------------------------------------------------------
        close(ro[0], r[0], 2e-5, f"{name} recons")
        close(ro[2][1], r[2][1], 2e-5, f"{name} z_e")
        if cfg["use_vq"]:
            nmis = int((ro[2][2].reshape(-1) != r[2][2].reshape(-1)).sum())
            assert nmis == 0, f"{name}: {nmis} oracle index mismatches"
        for k in ld:
            close(ldo[k], ld[k], 3e-5, f"{name} loss[{k}]")
        close(gno, gn, 1e-4, f"{name} grad norm")
        for k in pnames:
            close(sd[k].grad, grads_r[k], 2e-4, f"{name} grad[{k}]")
            close_step(sd[k], ref.state_dict()[k], lr, f"{name} post-step {k}", s + 1)
        # ---- record reference outputs --------------------------------------------------
        out[f"recons_{s}"] = np_(r[0])
        out[f"z_e_{s}"] = np_(r[2][1])
        out[f"z_q_{s}"] = np_(r[2][0])
        out[f"idx_{s}"] = np_(r[2][2].reshape(-1)).astype(np.int32)
        out[f"loss_keys_{s}"] = np.array(list(ld.keys()))
        out[f"loss_vals_{s}"] = np.array([float(ld[k]) for k in ld], dtype=np.float64)
        out[f"grad_norm_{s}"] = float(gn)
        out[f"gradnorm_each_{s}"] = np.array([float(grads_r[k].norm()) for k in pnames])
        out[f"post_sum_each_{s}"] = np.array([G.checksum(ref.state_dict()[k]) for k in pnames])
        # fp64 re-evaluation of the same step (arbiter): its distance to the fp32 reference bounds the fp32 round-off
        if cfg["use_vq"]:
            assert torch.equal(o64[2][2].reshape(-1), r[2][2].reshape(-1)), f"{name}: fp64 argmin differs from fp32"
        out[f"loss_vals64_{s}"] = np.array([float(ld64[k]) for k in ld], dtype=np.float64)
        out[f"recons_err64_{s}"] = float((o64[0].detach() - r[0].detach().double()).abs().max())
        out[f"z_e_err64_{s}"] = float((o64[2][1].detach() - r[2][1].detach().double()).abs().max())
        out[f"grad_norm64_{s}"] = float(gn64)
        out[f"gradnorm_each64_{s}"] = np.array([float(g64[k].norm()) for k in pnames])
        out[f"grad_maxerr64_each_{s}"] = np.array([float((g64[k] - grads_r[k].double()).abs().max()) for k in pnames])
        out[f"grad_maxabs_each_{s}"] = np.array([float(grads_r[k].abs().max()) for k in pnames])
        if full_grads:
            for k in GRAD_KEYS_FULL:
                if k in grads_r:
                    out[f"grad_{s}::{k}"] = np_(grads_r[k])
                    out[f"post_{s}::{k}"] = np_(ref.state_dict()[k])
        if cfg["use_vq"]:
            q = ref.quantizer
            out[f"q_emb_{s}"] = np_(q.embedding)
            out[f"q_ecs_{s}"] = np_(q.ema_cluster_size)
            out[f"q_eemb_{s}"] = np_(q.ema_embedding)
            out[f"q_ep_usage_{s}"] = np_(q._ep_usage)
    out["param_names"] = np.array(pnames)
    if eval_too:                                   # eval-mode forward (no dropout, no EMA), initial weights
        ref = RefVQVAE(**cfg_kw)
        ref.load_state_dict(sd0, strict=True)
        ref.eval()
        orc = O.OracleVQVAE({k: v.clone() for k, v in sd0.items()}, drop_scale=0.0, **cfg_kw)
        orc.training = False
        x, mask = batches[0]
        with torch.no_grad():
            r = ref(x, mask)
            ro = orc.forward(x, mask)
            ld = ref.loss_function(*r, **weights)
            ldo = orc.loss_function(*ro, **weights)
            o64, ld64, _, _ = oracle64_step(cfg_kw, sd0, x, mask, weights, clip, 0, train=False)
        out["eval_loss_vals64"] = np.array([float(ld64[k]) for k in ld], dtype=np.float64)
        out["eval_recons_err64"] = float((o64[0] - r[0].double()).abs().max())
        close(ro[0], r[0], 2e-5, f"{name} eval recons")
        for k in ld:
            close(ldo[k], ld[k], 3e-5, f"{name} eval loss[{k}]")
        out["eval_recons"] = np_(r[0])
        out["eval_idx"] = np_(r[2][2].reshape(-1)).astype(np.int32)
        out["eval_loss_keys"] = np.array(list(ld.keys()))
        out["eval_loss_vals"] = np.array([float(ld[k]) for k in ld], dtype=np.float64)
        # decode-only and sample-free API checks
        z = r[2][0]
        out["eval_decode"] = np_(ref.decode(z, mask=mask))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: B={B} L={L} ragged={ragged} steps={steps} keys={len(out)}")


# ------------------------------------------------------------------------------------------
# 3. loss function alone, every term switched on, with input gradients
# ------------------------------------------------------------------------------------------
def loss_case(name, B, L, seed, ragged, weights, cfg_kw, noise=0.7, data_stats=None):
    x, mask = G.smooth_curve_batch(B, L, seed, ragged)
    g = torch.Generator().manual_seed(seed + 1)
    rec = x.clone()
    rec[..., :3] += noise * torch.randn(B, L, 3, generator=g)
    rec[..., 3:] = 2.0 * torch.randn(B, L, 3, generator=g)
    Nt, D = 8, cfg_kw["code_dim"]
    ze = torch.randn(B, Nt, D, generator=g)
    zq = ze + 0.3 * torch.randn(B, Nt, D, generator=g)
    sd0 = G.model_state(cfg_kw, seed)
    ref = RefVQVAE(**cfg_kw)
    ref.load_state_dict(sd0, strict=True)
    ref.train()
    orc = O.OracleVQVAE({k: v.clone() for k, v in sd0.items()}, drop_scale=0.0, **cfg_kw)
    out = {"B": B, "L": L, "seed": seed, "ragged": int(ragged), "noise": noise}
    if data_stats is not None:
        mean, std = torch.tensor(data_stats[0]), torch.tensor(data_stats[1])
        ref.set_data_stats(mean, std)
        orc.data_mean, orc.data_std = mean.view(1, 1, 3), std.view(1, 1, 3)
        out["stats_mean"], out["stats_std"] = np.array(data_stats[0], dtype=np.float32), np.array(data_stats[1], dtype=np.float32)
    for tag, m in (("m", mask), ("nomask", None)):
        if tag == "nomask" and ragged is False:
            pass
        r1, z1 = rec.clone().requires_grad_(True), ze.clone().requires_grad_(True)
        pack = (zq, z1, torch.zeros(B, Nt, dtype=torch.long), torch.tensor(3.0), torch.tensor(0.5))
        ld = ref.loss_function(r1, x, pack, m, **weights)
        ld["loss"].backward()
        r2, z2 = rec.clone().requires_grad_(True), ze.clone().requires_grad_(True)
        pack2 = (zq, z2, pack[2], pack[3], pack[4])
        ldo = orc.loss_function(r2, x, pack2, m, **weights)
        ldo["loss"].backward()
        # fp64 re-evaluation (arbiter of fp32 round-off), same oracle on .double() inputs
        sd64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        orc64 = O.OracleVQVAE(sd64, drop_scale=0.0, **cfg_kw)
        if data_stats is not None:
            orc64.data_mean, orc64.data_std = mean.double().view(1, 1, 3), std.double().view(1, 1, 3)
        r3, z3 = rec.double().requires_grad_(True), ze.double().requires_grad_(True)
        ld64 = orc64.loss_function(r3, x.double(), (zq.double(), z3, pack[2], pack[3].double(), pack[4].double()), m, **weights)
        ld64["loss"].backward()
        out[f"{tag}_loss_vals64"] = np.array([float(ld64[k]) for k in ld], dtype=np.float64)
        out[f"{tag}_d_recons_err64"] = float((r3.grad - r1.grad.double()).abs().max())
        out[f"{tag}_d_ze_err64"] = float((z3.grad - z1.grad.double()).abs().max())
        for k in ld:
            close(ldo[k], ld[k], 2e-5, f"{name}/{tag} loss[{k}]")
        close(r2.grad, r1.grad, 1e-4, f"{name}/{tag} d_recons")
        close(z2.grad, z1.grad, 1e-5, f"{name}/{tag} d_ze")
        out[f"{tag}_loss_keys"] = np.array(list(ld.keys()))
        out[f"{tag}_loss_vals"] = np.array([float(ld[k]) for k in ld], dtype=np.float64)
        out[f"{tag}_d_recons"] = np_(r1.grad)
        out[f"{tag}_d_ze"] = np_(z1.grad)
    out["x"] = np_(x)
    out["mask"] = np_(mask)
    out["recons"] = np_(rec)
    out["ze"] = np_(ze)
    out["zq"] = np_(zq)
    out["weights_keys"] = np.array(sorted(weights.keys()))
    out["weights_vals"] = np.array([float(weights[k]) for k in sorted(weights.keys())])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: B={B} L={L} ragged={ragged}")


# ------------------------------------------------------------------------------------------
# 4. the reference's initial weights at a seed (same torch init calls in the same order)
# ------------------------------------------------------------------------------------------
def init_case(name, cfg_kw, seed):
    torch.manual_seed(seed)
    ref = RefVQVAE(**cfg_kw)
    sdr = ref.state_dict()
    keys = list(sdr.keys())
    out = {"seed": seed, "keys": np.array(keys),
           "shapes": np.array([",".join(map(str, sdr[k].shape)) for k in keys]),
           "sums": np.array([G.checksum(sdr[k]) for k in keys]),
           "n_params": sum(p.numel() for p in ref.parameters())}
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: {len(keys)} keys, {out['n_params']} params")


# ------------------------------------------------------------------------------------------
# 5. harness pieces that live behind `import pytorch_lightning` (not installed): the two PURE functions are taken out
#    of the reference source with ast and executed on their own -- no Lightning import, no stub.
# ------------------------------------------------------------------------------------------
def _extract_function(path, fn_name, namespace):
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == fn_name)
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, f"<{os.path.basename(path)}:{fn_name}>", "exec"), namespace)
    return namespace[fn_name]


def harness_case(name="harness"):
    import yaml
    from typing import Dict, List, Tuple
    from torch.nn.utils.rnn import pad_sequence
    interp = _extract_function(os.path.join(REF, "experiment.py"), "interpolate_schedule", {"Dict": Dict, "List": List})
    collate = _extract_function(os.path.join(REF, "dataset.py"), "pad_collate",
                                {"List": List, "Tuple": Tuple, "torch": torch, "pad_sequence": pad_sequence})
    out = {}
    for stem in ("stage1_ae", "stage2_vq"):
        with open(os.path.join(REF, "configs", stem + ".yaml")) as f:
            cfg = yaml.safe_load(f)
        sched = cfg["exp_params"].get("schedules", {}) or {}
        keys = sorted(sched.keys())
        vals = np.zeros((len(keys), 201), dtype=np.float64)
        for e in range(201):
            row = interp(sched, e)
            assert sorted(row.keys()) == keys
            vals[:, e] = [row[k] for k in keys]
        out[f"{stem}_sched_keys"] = np.array(keys)
        out[f"{stem}_sched_vals"] = vals
        ep = cfg["exp_params"]
        # torch's own schedulers with the YAML's parameters (experiment.py:176-197): lr and Adam beta1 per step / epoch
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=float(ep["LR"]))
        epochs, spe = 6, 11
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=float(ep["LR"]), epochs=epochs, steps_per_epoch=spe,
                                                  pct_start=float(ep.get("onecycle_pct_start", 0.15)), anneal_strategy="cos",
                                                  div_factor=float(ep.get("onecycle_div_factor", 25.0)),
                                                  final_div_factor=float(ep.get("onecycle_final_div", 1500.0)))
        lrs, b1 = [], []
        for _ in range(epochs * spe):
            lrs.append(opt.param_groups[0]["lr"])
            b1.append(opt.param_groups[0]["betas"][0])
            opt.step()
            sch.step()
        out[f"{stem}_onecycle"] = np.array([lrs, b1], dtype=np.float64)
        out[f"{stem}_onecycle_cfg"] = np.array([float(ep["LR"]), epochs, spe, float(ep.get("onecycle_pct_start", 0.15)),
                                                float(ep.get("onecycle_div_factor", 25.0)),
                                                float(ep.get("onecycle_final_div", 1500.0))])
        opt = torch.optim.AdamW([p], lr=float(ep["LR"]))
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=12, eta_min=float(ep["LR"]) * 1e-6)
        lrs = []
        for _ in range(12):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"{stem}_cosine"] = np.array(lrs, dtype=np.float64)
    g = torch.Generator().manual_seed(4711)
    lens = [5, 17, 1, 9, 17, 12]
    items = [torch.randn(n, 6, generator=g) for n in lens]
    xb, mb = collate([t.clone() for t in items])
    out["collate_lens"] = np.array(lens)
    out["collate_x"] = np_(xb)
    out["collate_mask"] = np_(mb)
    try:
        collate([])
        out["collate_empty_raises"] = 0
    except RuntimeError:
        out["collate_empty_raises"] = 1
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: schedules of both YAMLs (epochs 0-200), pad_collate, OneCycle / cosine traces")


# ------------------------------------------------------------------------------------------
# 6. CurveDataset (dataset.py:54-139) on the file formats it accepts: np.save'd dict (.npy), .npz, a file holding NaN / Inf;
#    and the two shape errors.  The class is cut out of the reference source with ast (its module imports Lightning).
# ------------------------------------------------------------------------------------------
def dataset_case(name="dataset"):
    import tempfile
    from pathlib import Path
    from typing import Optional
    from torch.utils.data import Dataset
    RefDS = _extract_function(os.path.join(REF, "dataset.py"), "CurveDataset",
                              {"Dataset": Dataset, "Optional": Optional, "Path": Path, "os": os, "np": np, "torch": torch})
    rs = np.random.RandomState(77)
    lens = [7, 12, 5]
    curves = []
    for i, n in enumerate(lens):
        xyz = (rs.randn(n, 3) * 4.0 + np.array([10.0, -3.0, 0.5]) * (i + 1)).astype(np.float32)
        ss = np.eye(3, dtype=np.float32)[rs.randint(0, 3, n)]
        curves.append((xyz, ss))
    curves[2][0][1, 2] = np.nan                    # sanitised to 0 after centring (dataset.py:135-137)
    curves[2][0][3, 0] = np.inf
    out = {"lens": np.array(lens)}
    with tempfile.TemporaryDirectory() as tmp:
        names = ["a.npy", "b.npz", "c.npy"]
        np.save(os.path.join(tmp, names[0]), {"curve_coords": curves[0][0], "ss_one_hot": curves[0][1]}, allow_pickle=True)
        np.savez(os.path.join(tmp, names[1]), curve_coords=curves[1][0], ss_one_hot=curves[1][1])
        np.save(os.path.join(tmp, names[2]), {"curve_coords": curves[2][0], "ss_one_hot": curves[2][1]}, allow_pickle=True)
        np.save(os.path.join(tmp, "bad_xyz.npy"), {"curve_coords": curves[0][0][:, :2], "ss_one_hot": curves[0][1]}, allow_pickle=True)
        np.save(os.path.join(tmp, "bad_ss.npy"), {"curve_coords": curves[0][0], "ss_one_hot": curves[0][1][:-1]}, allow_pickle=True)
        lst = os.path.join(tmp, "list.txt")
        open(lst, "w").write("\n".join(names) + "\n\n")
        ds = RefDS(tmp, list_path=lst, train=True)
        assert len(ds) == 3
        for i in range(3):
            out[f"in_xyz_{i}"], out[f"in_ss_{i}"] = curves[i]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[f"item_{i}"] = np_(ds[i])
        for j, bad in enumerate(["bad_xyz.npy", "bad_ss.npy"]):
            open(lst, "w").write(bad + "\n")
            try:
                RefDS(tmp, list_file=lst, train=False)[0]
                out[f"bad_{j}_raises"] = 0
            except ValueError:
                out[f"bad_{j}_raises"] = 1
        open(lst, "w").write("\n")
        try:
            RefDS(tmp, list_path=lst)
            out["empty_list_raises"] = 0
        except FileNotFoundError:
            out["empty_list_raises"] = 1
        try:
            RefDS(tmp)
            out["no_list_raises"] = 0
        except ValueError:
            out["no_list_raises"] = 1
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"[golden] {name}: CurveDataset on .npy dict / .npz / NaN file, shape errors")


if __name__ == "__main__":
    only = set(sys.argv[1:])
    want = lambda n: (not only) or (n in only)
    if want("vq"):
        vq_case("vq_k512_d64_fresh", 16, 64, 512, 64, 1, seed=11, steps=2)
        vq_case("vq_k512_d64_cinit", 16, 64, 512, 64, 1, seed=12, steps=2, centroid_init=True, scale=0.2)
        vq_case("vq_k512_d64_eval", 16, 64, 512, 64, 1, seed=13, steps=1, train=False)
        vq_case("vq_k512_d64_ties", 4, 64, 512, 64, 1, seed=14, steps=1, dup_codes=True, centroid_init=True)
        vq_case("vq_k8192_d256", 32, 64, 8192, 256, 1, seed=15, steps=1, centroid_init=True)
        vq_case("vq_rvq4_k64_d32", 8, 64, 64, 32, 4, seed=16, steps=2, centroid_init=True)
        vq_case("vq_rvq4_k1024_d512", 4, 64, 1024, 512, 4, seed=17, steps=1, centroid_init=True)
        vq_case("vq_tiny_store", 2, 8, 16, 8, 1, seed=18, steps=1, centroid_init=True, store_inputs=True)
        vq_case("vq_k512_d64_masked", 16, 64, 512, 64, 1, seed=19, steps=3, centroid_init=True, scale=0.2, row_mask=True)
        vq_case("vq_rvq4_k64_d32_masked", 8, 64, 64, 32, 4, seed=20, steps=3, centroid_init=True, row_mask=True)
    if want("model"):
        model_case("model_small_vq_full", G.SMALL_VQ, 6, 24, 21, False, G.BASE_LOSS_WEIGHTS, steps=2)
        model_case("model_small_vq_ragged", G.SMALL_VQ, 5, 37, 22, True, G.ALL_LOSS_WEIGHTS, smooth=True, steps=2)
        model_case("model_small_rvq_ragged", G.SMALL_RVQ, 4, 32, 23, True, G.ALL_LOSS_WEIGHTS, smooth=True, steps=3)
        model_case("model_small_ae", G.SMALL_AE, 8, 40, 24, True, dict(ss_weight=0.6, xyz_tv_lambda=0.006), clip=1.0)
    if want("extra"):
        model_case("model_small_softvq", dict(G.SMALL_VQ, soft_vq_use=True, soft_vq_tau_start=2.0, soft_vq_tau_end=0.5,
                                              soft_vq_tau_warm_steps=10, soft_vq_alpha_warm_steps=20),
                   5, 24, 41, True, G.BASE_LOSS_WEIGHTS, eval_too=False, start_steps=5)
        model_case("model_small_uent", dict(G.SMALL_VQ, usage_entropy_lambda=0.05), 5, 24, 42, True, G.BASE_LOSS_WEIGHTS,
                   eval_too=False)
        loss_case("loss_datastats", 4, 40, 43, True, G.ALL_LOSS_WEIGHTS, G.SMALL_VQ,
                  data_stats=([0.3, -0.2, 0.1], [1.7, 0.9, 1.3]))
    if want("c2"):
        model_case("model_c2_b2", G.C2_MODEL, 2, 64, 25, False,
                   dict(G.BASE_LOSS_WEIGHTS, xyz_tv_lambda=0.0008), clip=3.0, lr=2e-4, wd=0.008,
                   full_grads=False, eval_too=False)
    if want("loss"):
        loss_case("loss_all_ragged", 5, 48, 31, True, G.ALL_LOSS_WEIGHTS, dict(G.SMALL_VQ, usage_entropy_lambda=0.01))
        loss_case("loss_all_full", 4, 64, 32, False, G.ALL_LOSS_WEIGHTS, G.SMALL_VQ)
        loss_case("loss_short", 3, 9, 33, False, G.ALL_LOSS_WEIGHTS, G.SMALL_VQ)
    if want("harness"):
        harness_case()
    if want("dataset"):
        dataset_case()
    if want("init"):
        init_case("init_small_vq_seed1265", G.SMALL_VQ, 1265)
        init_case("init_c2_seed1265", G.C2_MODEL, 1265)
        init_case("init_small_ae_seed7", G.SMALL_AE, 7)
