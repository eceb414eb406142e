# -*- coding: utf-8 -*-
"""-m gpu: the engine's world_size>1 training-step path (graph split around the single all-reduce, deferred EMA
refresh, 1/world folded into the clip coefficient) with 2 processes sharing cuda:0 over gloo.  After 3 steps
(eager, capture, replay) the weights and the codebook must equal a single-process run on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

import gen_inputs as G

pytestmark = pytest.mark.gpu
CFG = dict(G.SMALL_VQ)
B, LQ, SEED, STEPS = 8, 24, 515, 3
W = dict(G.BASE_LOSS_WEIGHTS, xyz_tv_lambda=0.001)


def _run(x, mask, use_graph=True, CFG=CFG):
    from models import vae_models
    m = vae_models["VQVAE"](**CFG)
    m.load_state_dict(G.model_state(CFG, SEED), strict=True)
    m = m.to("cuda:0").train()
    m.training_steps = 1
    eng = m._engine()
    eng.drop_scale = 0.0
    for _ in range(STEPS):
        m.train_step(x, mask, W, 1e-3, 0.01, 1.0, use_graph=use_graph)
    torch.cuda.synchronize()
    return ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, eng.metrics.cpu().clone())


def _worker(r, world, port, q, cfg_name="vq"):
    CFG = dict(G.SMALL_VQ) if cfg_name == "vq" else dict(G.SMALL_RVQ) if cfg_name == "rvq" else dict(G.SMALL_VQ, usage_entropy_lambda=0.05)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "pytorch-vae_amd"), os.path.join(here, "golden"), here):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=r, world_size=world)
    from vqvae_hip.parallel import shard_bounds
    x, mask = G.curve_batch(B, LQ, SEED + 1, ragged=False)
    lo, hi = shard_bounds(B)
    sd, met = _run(x[lo:hi].cuda(), mask[lo:hi].cuda(), CFG=CFG)
    if r == 0:
        q.put({k: v.numpy() for k, v in sd.items()})      # by value: the producer may exit before the parent reads
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("cfg_name", ["vq", "rvq", "uent"])
def test_two_rank_train_steps_equal_single_process(cfg_name):
    """vq: one level, EMA statistics deferred behind the last gradient bucket.  rvq / uent: the statistics are reduced
    inside the forward (per level / before the regulariser reads the refreshed table): the captured step is split into
    graph segments around those collectives as well (round 1 ran these configurations without graphs)."""
    CFG = dict(G.SMALL_VQ) if cfg_name == "vq" else dict(G.SMALL_RVQ) if cfg_name == "rvq" else dict(G.SMALL_VQ, usage_entropy_lambda=0.05)
    port = 29600 + (os.getpid() % 2000) + {"vq": 0, "rvq": 1, "uent": 2}[cfg_name]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, cfg_name)) for r in range(2)]
    for p in procs:
        p.start()
    sd2 = {k: torch.from_numpy(v) for k, v in q.get(timeout=600).items()}
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    x, mask = G.curve_batch(B, LQ, SEED + 1, ragged=False)
    sd1, _ = _run(x.cuda(), mask.cuda(), CFG=CFG)
    sd1e, _ = _run(x.cuda(), mask.cuda(), use_graph=False, CFG=CFG)
    for k in sd1:
        if k.startswith("quantizer._ep_"):      # epoch usage diagnostics are per-rank (local shard) by design
            continue
        a, b, c = sd1[k].double(), sd2[k].double(), sd1e[k].double()
        assert float((a - c).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), f"graph vs eager differ at {k}"
        # first Adam steps move every weight by ~lr*sign(g): round-off-level gradients may flip, hence the lr-scale bound
        assert float((a - b).abs().max()) <= 2.1e-3 * STEPS, k
        assert float((a - b).abs().median()) <= 2e-5, k
    assert torch.equal(sd1["quantizer.ema_cluster_size"], sd2["quantizer.ema_cluster_size"]) or \
        float((sd1["quantizer.ema_cluster_size"] - sd2["quantizer.ema_cluster_size"]).abs().max()) < 1e-5


def _rccl_selftest_worker(port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "pytorch-vae_amd"), os.path.join(here, "golden"), here):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x, mask = G.curve_batch(B, LQ, SEED + 1, ragged=False)
    os.environ["VQH_DP_SELFTEST"] = "0"
    sd_plain, met_plain = _run(x.cuda(), mask.cuda())
    os.environ["VQH_DP_SELFTEST"] = "1"          # same step through the data-parallel form: 5 graph segments + RCCL buckets
    sd_dp, met_dp = _run(x.cuda(), mask.cuda())
    sd_dpe, _ = _run(x.cuda(), mask.cuda(), use_graph=False)
    bad = [k for k in sd_plain if not torch.equal(sd_plain[k], sd_dp[k]) or not torch.equal(sd_plain[k], sd_dpe[k])]
    q.put((bad, bool(torch.equal(met_plain, met_dp))))
    torch.distributed.destroy_process_group()


def test_rccl_bucketed_step_on_one_rank_equals_plain_step():
    """The RCCL code path itself (async all-reduce per backward phase between graph replays, wait before the optimizer
    graph) on a ONE-rank nccl group: a one-rank sum is the identity, so weights after 3 steps must be bit-identical
    to the single-graph step.  (Two RCCL ranks cannot share one GPU; the N>1 arithmetic is covered above with gloo.)"""
    port = 29700 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_selftest_worker, args=(port, q))
    p.start()
    bad, met_equal = q.get(timeout=600)
    p.join(timeout=600)
    assert p.exitcode == 0
    assert not bad, f"tensors differ between the plain and the bucketed-RCCL step: {bad[:5]}"
    assert met_equal
