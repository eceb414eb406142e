# -*- coding: utf-8 -*-
"""CPU suite for the host side: C-ABI consistency (header == sources == ctypes table == exported symbols),
the model's drop-in surface (keys, seeded init, loud failure without a GPU), harness logic (schedules, LR
policies, checkpoints, warm start) and an end-to-end plumbing run of config C1's harness with a CPU stand-in
model (the oracle) -- the product model itself never runs on a CPU."""
import ctypes
import glob
import os
import re
import sys

import numpy as np
import pytest
import torch
import yaml

import gen_inputs as G
from gen_inputs import O
from conftest import load_golden, PKG, REPO
from abi_util import header_protos


# ---------------------------------------------------------------------------------------------- ABI
def _src_protos():
    def code(args):
        c = ""
        for a in args.split(","):
            a = a.strip()
            if not a or a == "void":
                continue
            if "*" in a or a.startswith("hipStream_t"):
                c += "p"
            elif a.startswith("long long"):
                c += "l"
            elif a.startswith("unsigned"):
                c += "u"
            elif a.startswith("float"):
                c += "f"
            elif a.startswith("int"):
                c += "i"
            else:
                raise ValueError(a)
        return c
    out = {}
    for f in glob.glob(os.path.join(PKG, "csrc", "*.hip")):
        for m in re.finditer(r'extern "C"\s+(?:int|const char\*|void)\s+(vqh_\w+)\s*\(([^)]*)\)\s*\{', open(f).read()):
            out[m.group(1)] = code(m.group(2))
    return out


def test_header_sources_and_ctypes_table_agree():
    from vqvae_hip import lib
    hdr, src = header_protos(), _src_protos()
    for name, sig in hdr.items():
        assert name in src, f"{name} declared in include/vqvae_hip.h but not defined"
        assert src[name] == sig, f"{name}: header {sig} vs source {src[name]}"
        if name in lib._PROTOS:
            assert lib._PROTOS[name] == sig, f"{name}: ctypes table {lib._PROTOS[name]} vs header {sig}"
    assert set(lib.EXPORTS) == set(hdr), set(lib.EXPORTS) ^ set(hdr)


def test_shared_library_loads_and_exports_every_declared_symbol():
    from vqvae_hip import lib
    L = lib.lib()                      # dlopen only: no kernel is launched without a GPU
    for name in header_protos():
        assert hasattr(L, name), name
    assert L.vqh_abi_version() == 1
    assert isinstance(L.vqh_last_error(), bytes)


# ---------------------------------------------------------------------------------------------- model surface
def test_model_keys_and_seeded_init_equal_the_reference():
    from models import vae_models, BaseVAE
    for name, cfg, seed in (("init_small_vq_seed1265", G.SMALL_VQ, 1265), ("init_small_ae_seed7", G.SMALL_AE, 7)):
        g = load_golden(name)
        torch.manual_seed(seed)
        m = vae_models["VQVAE"](**cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == [str(k) for k in g["keys"]]
        assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
        for k, want in zip(g["keys"], g["sums"]):
            assert abs(G.checksum(sd[str(k)]) - float(want)) <= 1e-6 * max(1.0, abs(float(want))), k
        assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    assert issubclass(BaseVAE, torch.nn.Module)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_model_fails_loudly_without_gpu():
    from models import vae_models
    from vqvae_hip.lib import VqhError
    m = vae_models["VQVAE"](**G.SMALL_VQ)
    with pytest.raises(VqhError):
        m(torch.zeros(1, 8, 6), torch.ones(1, 8, dtype=torch.bool))
    with pytest.raises(VqhError):
        m.decode(torch.zeros(1, 8, 16))


def test_constructor_swallows_unknown_kwargs_and_validates_centroids():
    from models import vae_models
    m = vae_models["VQVAE"](name="anything", **G.SMALL_VQ)
    with pytest.raises(ValueError):
        m.init_codebook_from_centroids(torch.zeros(3, 16))
    with pytest.raises(ValueError):
        m.init_codebook_from_centroids(torch.zeros(2, 16, 8))
    m.init_codebook_from_centroids(torch.ones(32, 16))
    assert float(m.quantizer.ema_cluster_size.min()) == 1.0 and float(m.quantizer.embedding.mean()) == 1.0
    m.beta = 0.125
    assert m.quantizer.beta == 0.125


# ---------------------------------------------------------------------------------------------- harness logic
def test_interpolate_schedule_known_answers():
    from experiment import interpolate_schedule
    s = {"a": [[0, 0.0], [10, 1.0], [20, 3.0]], "b": [[5, 2.0]], "empty": []}
    assert interpolate_schedule(s, 0) == {"a": 0.0, "b": 2.0}
    assert interpolate_schedule(s, 5)["a"] == pytest.approx(0.5)
    assert interpolate_schedule(s, 10)["a"] == pytest.approx(1.0)
    assert interpolate_schedule(s, 15)["a"] == pytest.approx(2.0)
    assert interpolate_schedule(s, 20)["a"] == 3.0 and interpolate_schedule(s, 999)["a"] == 3.0
    assert interpolate_schedule(s, 7)["b"] == 2.0
    assert interpolate_schedule({}, 3) == {} and interpolate_schedule(None, 3) == {}
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "stage2_vq.yaml")))
    v = interpolate_schedule(cfg["exp_params"]["schedules"], 100)
    assert v["beta"] == pytest.approx(0.005 + (100 - 80) / 60 * 0.001)
    assert v["LR"] == pytest.approx(0.0002 - (100 - 30) / 150 * 0.0001)


def test_lr_policies_match_torch_schedulers():
    from experiment import LRPolicy
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=0.004)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.004, epochs=3, steps_per_epoch=7, pct_start=0.12, anneal_strategy="cos",
                                              div_factor=20.0, final_div_factor=5000.0)
    pol = LRPolicy("onecycle", 0.004, max_epochs=3, steps_per_epoch=7, pct_start=0.12, div_factor=20.0, final_div=5000.0)
    for _ in range(21):
        lr, b1 = pol.current()
        assert lr == pytest.approx(opt.param_groups[0]["lr"], rel=1e-9)
        assert b1 == pytest.approx(opt.param_groups[0]["betas"][0], rel=1e-9)
        opt.step()
        if _ < 20:
            sch.step()
        pol.on_step()
    opt = torch.optim.AdamW([p], lr=0.01)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=9, eta_min=0.01 * 1e-6)
    pol = LRPolicy("cosine", 0.01, max_epochs=9)
    for _ in range(9):
        assert pol.current()[0] == pytest.approx(opt.param_groups[0]["lr"], rel=1e-6)
        opt.step(); sch.step(); pol.on_epoch()


def test_pad_collate_and_synthetic_dataset():
    from dataset import pad_collate, SyntheticCurveDataset
    ds = SyntheticCurveDataset(5, max_len=12, min_len=4, seed=3)
    items = [ds[i] for i in range(5)]
    x, mask = pad_collate(items)
    assert x.shape == (5, max(t.shape[0] for t in items), 6) and mask.dtype == torch.bool
    for i, t in enumerate(items):
        assert int(mask[i].sum()) == t.shape[0] and torch.equal(x[i, :t.shape[0]], t) and float(x[i, t.shape[0]:].abs().sum()) == 0
        assert float(t[:, :3].mean(0).abs().max()) < 1e-5 and torch.all(t[:, 3:].sum(-1) == 1)
    assert torch.equal(ds[2], ds[2])
    with pytest.raises(RuntimeError):
        pad_collate([])


def test_harness_matches_reference_generated_fixture():
    """tests/golden/harness.npz was recorded by make_golden.py from the reference's OWN interpolate_schedule
    (experiment.py:14-34) and pad_collate (dataset.py:30-49) -- cut out of the source with ast, no Lightning import --
    on the reference's own YAMLs, plus torch's OneCycleLR / CosineAnnealingLR with the YAMLs' parameters
    (experiment.py:176-197).  This repo's host mirror must reproduce all of it."""
    from experiment import interpolate_schedule, LRPolicy
    from dataset import pad_collate
    g = load_golden("harness")
    for stem in ("stage1_ae", "stage2_vq"):
        cfg = yaml.safe_load(open(os.path.join(PKG, "configs", stem + ".yaml")))
        sched = cfg["exp_params"].get("schedules", {}) or {}
        keys = [str(k) for k in g[f"{stem}_sched_keys"]]
        assert sorted(sched.keys()) == keys
        for e in range(201):
            row = interpolate_schedule(sched, e)
            assert sorted(row.keys()) == keys
            got = np.array([row[k] for k in keys])
            assert np.array_equal(got, g[f"{stem}_sched_vals"][:, e]), (stem, e)
        lr0, epochs, spe, pct, div, fdiv = g[f"{stem}_onecycle_cfg"]
        assert float(cfg["exp_params"]["LR"]) == lr0
        pol = LRPolicy("onecycle", lr0, max_epochs=int(epochs), steps_per_epoch=int(spe), pct_start=pct, div_factor=div,
                       final_div=fdiv)
        for i in range(int(epochs * spe)):
            lr, b1 = pol.current()
            assert lr == pytest.approx(g[f"{stem}_onecycle"][0, i], rel=1e-12, abs=1e-18)
            assert b1 == pytest.approx(g[f"{stem}_onecycle"][1, i], rel=1e-12)
            pol.on_step()
        pol = LRPolicy("cosine", lr0, max_epochs=12)
        for i in range(12):
            assert pol.current()[0] == pytest.approx(g[f"{stem}_cosine"][i], rel=1e-9)
            pol.on_epoch()
    gen = torch.Generator().manual_seed(4711)
    items = [torch.randn(int(n), 6, generator=gen) for n in g["collate_lens"]]
    x, mask = pad_collate(items)
    assert torch.equal(x, torch.from_numpy(g["collate_x"])) and torch.equal(mask, torch.from_numpy(g["collate_mask"]))
    assert x.dtype == torch.float32 and mask.dtype == torch.bool and int(g["collate_empty_raises"]) == 1


class _CpuStandIn:
    """Test-only stand-in with the harness-facing interface of models.VQVAE, computing with the oracle.
    Exists so the harness (schedules, LR policy, checkpoints, epoch hooks) can be exercised without a GPU."""

    def __init__(self, **mp):
        self.cfg = O.make_cfg(**mp)
        self.sd = O.attach_grads(O.random_state(self.cfg, 5), self.cfg)
        self.orc = O.OracleVQVAE(self.sd, drop_scale=0.0, **mp)
        self.opt = torch.optim.AdamW(self.orc.params(), lr=1e-3)
        self.use_vq = self.cfg["use_vq"]
        self.quantizer = None
        self.label_smoothing = self.usage_entropy_lambda = 0.0
        self._beta = 0.0
        self.sums = torch.zeros(24)
        self.training = True
        self.steps = 0

    beta = property(lambda s: s._beta, lambda s, v: setattr(s, "_beta", float(v)))

    def to(self, *_a, **_k): return self
    def train(self, mode=True): self.training = mode; self.orc.training = mode; return self
    def eval(self): return self.train(False)
    def state_dict(self): return {k: v.detach() for k, v in self.sd.items()}
    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            for k, v in sd.items():
                self.sd[k].copy_(v)
        return [], []
    def metric_names(self):
        from vqvae_hip.engine import METRIC_KEYS
        return list(METRIC_KEYS)
    def metric_sums(self): return self.sums
    def reset_metric_sums(self): self.sums.zero_()
    def _vec(self, ld):
        return torch.tensor([float(ld.get(k, 0.0)) for k in self.metric_names()])
    def train_step(self, x, mask, weights, lr, wd, clip, betas=(0.9, 0.999), use_graph=True):
        for g in self.opt.param_groups:
            g["lr"], g["weight_decay"], g["betas"] = lr, wd, betas
        ld, _, _ = self.orc.train_step(x, mask, self.opt, clip, weights)
        self.steps += 1
        v = self._vec(ld); self.sums += v
        return v
    def eval_step(self, x, mask, weights):
        with torch.no_grad():
            out = self.orc.forward(x, mask)
            return self._vec(self.orc.loss_function(*out, **weights))


def test_c1_harness_plumbing_two_epochs_on_cpu(tmp_path):
    """Config C1 (stage1_ae.yaml harness, synthetic 64-long curves, bs=8, 2 epochs) with the CPU stand-in."""
    from experiment import VQVAEExperiment
    from trainer import Trainer, ModelCheckpoint
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "stage1_ae.yaml")))
    mp = dict(cfg["model_params"], hidden_dim=64, num_heads=4, tokenizer_heads=4, num_layers=1, code_dim=16, latent_tokens=8,
              max_seq_len=64, print_init=False)
    dp = dict(cfg["data_params"], train_batch_size=8, val_batch_size=8, num_workers=0, pin_memory=False,
              synthetic={"n": 32, "n_val": 8, "max_len": 64, "min_len": 40, "seed": 1})
    exp = VQVAEExperiment(mp, cfg["exp_params"], dp, model_cls=_CpuStandIn)
    tr = Trainer(max_epochs=2, gradient_clip_val=1.0, accelerator="cpu", limit_val_batches=1.0,
                 callbacks=[ModelCheckpoint(str(tmp_path), filename="epoch{epoch:03d}", every_n_epochs=1)])
    tr.fit(exp)
    assert exp.model.steps == 8 and exp.global_step == 8 and exp.lr_policy.kind == "onecycle"
    assert exp.current_weights["ss_weight"] == pytest.approx(0.60 + 0.25 / 8)          # schedule at epoch 1
    assert exp.current_weights["xyz_tv_lambda"] == pytest.approx(0.006 + 0.0004)
    assert "epoch/loss" in exp.logged and np.isfinite(exp.logged["epoch/loss"])
    ck = torch.load(os.path.join(str(tmp_path), "last.ckpt"), map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and all(k.startswith("model.") for k in ck["state_dict"])
    assert set(k[6:] for k in ck["state_dict"]) == set(exp.model.state_dict().keys())
    # resume: epoch counter, LR policy position and weights come back
    exp2 = VQVAEExperiment(mp, cfg["exp_params"], dp, model_cls=_CpuStandIn)
    tr2 = Trainer(max_epochs=3, gradient_clip_val=1.0, accelerator="cpu", limit_val_batches=0)
    tr2.fit(exp2, ckpt_path=os.path.join(str(tmp_path), "last.ckpt"))
    assert exp2.model.steps == 4 and exp2.lr_policy.step_num == 12 and tr2.current_epoch == 2


def test_warm_start_filter_drops_quantizer_and_shape_mismatches():
    from experiment import VQVAEExperiment
    cand = {"model.a": torch.zeros(2), "model.quantizer.embedding": torch.zeros(3), "model.b": torch.zeros(5), "model.zz": torch.zeros(1)}
    st = VQVAEExperiment._strip_model_prefix(cand)
    kept, sp, ss = VQVAEExperiment._filter_state_dict_for_warmstart(st, {"a": torch.zeros(2), "b": torch.zeros(4), "quantizer.embedding": torch.zeros(3)})
    assert list(kept) == ["a"] and sp == ["quantizer.embedding"] and ss == ["b"]


def test_backward_phase_buckets_cover_every_parameter():
    """The flat gradient buffer is laid out by backward phase (engine.bwd_phases: one per decoder layer, tokenizer, SS
    encoder + fusion, one per geometry-encoder layer) so that each phase's gradients form one contiguous all-reduce
    bucket: every parameter of the shipped configurations must belong to exactly one phase, and the bucket that is
    exposed before the optimizer (the last one) must stay below 10 % of the gradient bytes at the true model width."""
    from vqvae_hip.engine import bwd_phases, _bwd_phase
    import gen_inputs as G
    from gen_inputs import O
    for cfg_kw in (G.C2_MODEL, G.SMALL_RVQ, G.SMALL_AE):
        cfg = O.make_cfg(**cfg_kw)
        shapes = O.param_shapes(cfg)
        names = list(shapes)
        PH = bwd_phases(cfg["num_layers"])
        assert len(PH) == 2 * cfg["num_layers"] + 2
        phases = [_bwd_phase(n, PH) for n in names]
        assert set(phases) == set(range(len(PH)))
        for n, ph in zip(names, phases):
            assert sum(n.startswith(p) for p in PH[ph]) == 1
            assert all(not n.startswith(p) for j, pr in enumerate(PH) if j != ph for p in pr), n
        if cfg_kw is G.C2_MODEL:
            size = [0] * len(PH)
            for n, ph in zip(names, phases):
                size[ph] += int(np.prod(shapes[n]))
            assert size[-1] / sum(size) < 0.10, size


# ---------------------------------------------------------------------------------------------- round 3: real-data path
def test_curve_dataset_matches_reference_on_npy_npz_and_bad_files(tmp_path):
    """CurveDataset (reference dataset.py:54-139) on an np.save'd dict, an .npz and a file with NaN / Inf; the two shape
    errors and the list-file errors.  Expected items come from the reference's own class (tests/golden/dataset.npz)."""
    from dataset import CurveDataset, pad_collate
    g = load_golden("dataset")
    tmp = str(tmp_path)
    names = ["a.npy", "b.npz", "c.npy"]
    cur = [(g[f"in_xyz_{i}"], g[f"in_ss_{i}"]) for i in range(3)]
    np.save(os.path.join(tmp, names[0]), {"curve_coords": cur[0][0], "ss_one_hot": cur[0][1]}, allow_pickle=True)
    np.savez(os.path.join(tmp, names[1]), curve_coords=cur[1][0], ss_one_hot=cur[1][1])
    np.save(os.path.join(tmp, names[2]), {"curve_coords": cur[2][0], "ss_one_hot": cur[2][1]}, allow_pickle=True)
    np.save(os.path.join(tmp, "bad_xyz.npy"), {"curve_coords": cur[0][0][:, :2], "ss_one_hot": cur[0][1]}, allow_pickle=True)
    np.save(os.path.join(tmp, "bad_ss.npy"), {"curve_coords": cur[0][0], "ss_one_hot": cur[0][1][:-1]}, allow_pickle=True)
    lst = os.path.join(tmp, "list.txt")
    open(lst, "w").write("\n".join(names) + "\n\n")
    ds = CurveDataset(tmp, list_path=lst, train=True)
    assert len(ds) == 3
    items = []
    for i in range(3):
        with np.errstate(all="ignore"):
            it = ds[i]
        assert it.dtype == torch.float32 and tuple(it.shape) == (int(g["lens"][i]), 6)
        assert np.array_equal(it.numpy(), g[f"item_{i}"]), f"item {i} differs from the reference's CurveDataset"
        items.append(it)
    x, mask = pad_collate(items)
    assert tuple(x.shape) == (3, 12, 6) and mask.sum(1).tolist() == [7, 12, 5]
    for j, bad in enumerate(["bad_xyz.npy", "bad_ss.npy"]):
        open(lst, "w").write(bad + "\n")
        assert int(g[f"bad_{j}_raises"]) == 1
        with pytest.raises(ValueError):
            CurveDataset(tmp, list_file=lst, train=False)[0]
    open(lst, "w").write("\n")
    assert int(g["empty_list_raises"]) == 1 and int(g["no_list_raises"]) == 1
    with pytest.raises(FileNotFoundError):
        CurveDataset(tmp, list_path=lst)
    with pytest.raises(ValueError):
        CurveDataset(tmp)


def test_length_buckets():
    """bucket_length: L <= 350 collapses to 11 padded lengths (32 ... 320, 350), all within VQH_MAX_ARENAS' default, and at the
    reference's train_batch_size 128 every bucket keeps B*L on the 256-row GEMM tile."""
    from vqvae_hip.engine import bucket_length
    buckets = sorted({bucket_length(L, 32, 350) for L in range(1, 351)})
    assert buckets == list(range(32, 321, 32)) + [350]
    assert all(bucket_length(L, 32, 350) >= L for L in range(1, 351))
    assert all((128 * b) % 256 == 0 for b in buckets)
    assert bucket_length(37, 1, 350) == 37 and bucket_length(37, 0, 350) == 37          # bucketing off
    assert bucket_length(40, 32, 40) == 40 and bucket_length(33, 32, 40) == 40         # capped at max_seq_len
    assert bucket_length(400, 32, 350) == 400                                          # never shorter than the batch


def test_get_epoch_stats_dict_matches_reference():
    """VectorQuantizerEMA.get_epoch_stats() (models/vq_vae.py:118-164) is host arithmetic over the accumulators: fed the
    reference's recorded _ep_usage / _ep_cnt it must return the reference's dict."""
    from models.vq_vae import VectorQuantizerEMA
    for name in ("vq_k512_d64_fresh", "vq_rvq4_k64_d32_masked", "vq_k512_d64_eval"):
        g = load_golden(name)
        q = VectorQuantizerEMA(int(g["K_per"]), int(g["D"]), num_quantizers=int(g["Q"]), print_init=False)
        s = int(g["steps"]) - 1
        q._ep_usage.copy_(torch.from_numpy(g[f"ep_usage_{s}"]))
        q._ep_cnt.copy_(torch.from_numpy(np.asarray(g[f"ep_cnt_{s}"]).reshape(1)))
        es = q.get_epoch_stats()
        got = [es["perplexity"], es["dead_ratio"], es["n_positions"], es["margin_mean"], es["qe_mean"], es["qe_p90"]]
        assert np.allclose(np.array(got, dtype=np.float64), g[f"epstats_{s}"], rtol=1e-6, atol=1e-9), (name, got, g[f"epstats_{s}"])
        q.reset_epoch_stats()
        assert q.get_epoch_stats()["n_positions"] == 0


def test_devices_resolution_and_single_command_launch(tmp_path):
    """trainer_params.devices drives the rank count like Lightning's `devices: N, strategy: ddp` (reference run.py:191-218):
    `python run.py -c cfg` with devices: 2 starts 2 child ranks itself.  Without a GPU the children fail loudly -- which is
    the proof that they were started (the parent never touches the GPU)."""
    import subprocess
    import run as run_mod
    assert run_mod.resolve_devices({"devices": 1}) == 1
    assert run_mod.resolve_devices({"devices": [0, 1, 2]}) == 3
    assert run_mod.resolve_devices({"devices": "2"}) == 2
    assert run_mod.resolve_devices({}) == 1
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "stage2_vq.yaml")))
    cfg["model_params"].update(hidden_dim=64, num_layers=1, num_heads=4, tokenizer_heads=4, tokenizer_layers=1, max_seq_len=48,
                               code_dim=16, latent_tokens=8, codebook_size=32, num_quantizers=1)
    cfg["data_params"].update(train_batch_size=8, val_batch_size=8, num_workers=0, pin_memory=False,
                              synthetic={"n": 32, "n_val": 8, "max_len": 40, "min_len": 20, "seed": 3})
    cfg["exp_params"].update(checkpoint_dir=str(tmp_path / "ck"))
    cfg["logging_params"] = {"save_dir": str(tmp_path / "logs"), "name": "t"}
    cfg["trainer_params"].update(max_epochs=1, devices=2)
    path = str(tmp_path / "ddp.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PYTHONPATH=PKG, VQH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(PKG, "run.py"), "-c", path], cwd=PKG, env=env, capture_output=True,
                       text=True, timeout=600)
    assert "[Launch] starting 2 ranks" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)
    # bench.py: --gpus must agree with the launcher's world size (it used to be parsed and ignored)
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env1, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_vq_nearest_form_and_workspace_decisions():
    """Host-side dispatch of vqh_vq_nearest (no kernel runs): which score kernel a (shape, workspace) pair gets and how much
    workspace the fastest form wants.  The plane-tensor form needs whole 256-row / 128-code tiles, D in {128, 256, 512}, at
    least 96 workgroups and the extra 1.5 (R + K) D + K + 6 ns R floats; with the documented minimum the round-2 forms run."""
    import ctypes
    from vqvae_hip import lib
    L = lib.lib()
    big = 1 << 40
    assert L.vqh_vq_nearest_form(262144, 8192, 256, big) == 3
    assert L.vqh_vq_nearest_form(8192, 1024, 512, big) == 3          # stage2_vq.yaml at B = 128: D = 512 has no register form
    assert L.vqh_vq_nearest_form(262144, 8192, 256, 48 << 20) == 2   # workspace of round 2: the register-resident split form
    assert L.vqh_vq_nearest_form(65536, 512, 64, big) == 2           # D = 64: too few K-steps per tile
    assert L.vqh_vq_nearest_form(3000, 1000, 32, big) == 1           # fp32 MFMA, code tiles through LDS
    assert L.vqh_vq_nearest_form(3000, 1000, 24, big) == 0           # D / 8 not a power of two: per-wave gather
    assert L.vqh_vq_nearest_form(4096, 4096, 512, big) == 3          # 16 row tiles x 8 code ranges = 128 workgroups
    assert L.vqh_vq_nearest_form(1024, 4096, 512, big) == 0          # 4 row tiles: per-wave gather
    assert L.vqh_vq_nearest_form(262144 + 8, 8192, 256, big) == 2    # ragged row count
    need = lib.vq_nearest_workspace(262144, 8192, 256)
    base = lib.vq_nearest_workspace(262144 + 8, 8192, 256)           # same shape class without the plane-tensor extra
    extra = 3 * 262144 * 256 // 2 + 3 * 8192 * 256 // 2 + 8192 + 6 * 8 * 262144 + 16
    assert need - extra > 0 and abs((need - extra) - base) < 64
    assert L.vqh_vq_nearest_form(262144, 8192, 256, need) == 3 and L.vqh_vq_nearest_form(262144, 8192, 256, need - 1) == 2
    old = L.vqh_vq_set_flags(32)                                     # A/B switch: never the plane-tensor form
    try:
        assert L.vqh_vq_nearest_form(262144, 8192, 256, big) == 2
        assert lib.vq_nearest_workspace(262144, 8192, 256) == need - extra
    finally:
        L.vqh_vq_set_flags(old)
