# -*- coding: utf-8 -*-
"""
bench.py -- training throughput of the VQ-VAE hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one full training step of config C2 (stage2_vq.yaml model with
num_quantizers=1, codebook K=512, D=64, N=64 latent tokens) on a per-rank batch of 256 synthetic
curve tensors [256, 64, 6] already resident in HBM: forward (dropout 0.1 active) + 24-term loss +
backward + RCCL all-reduce (N>1) + global-norm clip + AdamW + EMA codebook refresh, all fp32.
Prints ONE JSON line (rank 0) with the whole-job samples/s, the roofline of the dominant kernel
(the fp32-MFMA GEMM kernel with the largest share, measured live with HIP events) and a CPU baseline (the oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-vae_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import yaml  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
GFLOP_PER_SAMPLE_C2 = 16.846           # matmul FLOPs fwd+bwd per sample, config C2
BENCH_EPOCH = 100                      # loss weights = stage2 schedules evaluated at this epoch (all terms on)


def c2_setup():
    from experiment import interpolate_schedule
    cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "stage2_vq.yaml")))
    mp = dict(cfg["model_params"])
    mp.update(num_quantizers=1, codebook_size=512, code_dim=64, latent_tokens=64, reinit_dead_codes=False,
              print_init=False)
    ep = cfg["exp_params"]
    sched = interpolate_schedule(ep["schedules"], BENCH_EPOCH)
    keys = ["ss_weight", "bond_length_weight", "bond_angle_weight", "xyz_tv_lambda", "dir_weight", "dih_weight",
            "rmsd_weight", "pdm_weight", "win_kabsch_weight", "kappa_weight", "tau_weight", "lr_pdm_weight",
            "pdm_window", "win_kabsch_size", "win_kabsch_stride", "lr_min_sep", "lr_stride", "lr_max_offsets"]
    weights = {k: float(sched.get(k, ep.get(k, 0.0))) for k in keys}
    hp = dict(lr=float(sched.get("LR", ep["LR"])), wd=float(ep["weight_decay"]),
              clip=float(cfg["trainer_params"]["gradient_clip_val"]), beta=float(sched.get("beta", mp["beta"])),
              seed=int(ep["manual_seed"]))
    return mp, weights, hp


def synthetic_batch(B, L, seed, device):
    import gen_inputs as G
    x, mask = G.curve_batch(B, L, seed, ragged=False)
    return x.to(device), mask.to(device)


def cpu_baseline(mp, weights, hp, B_cpu=32, L=64, steps=8):
    """The oracle (kind 'port': our CPU restatement, verified against the reference by the golden vectors)
    timed on the host cores for the same model / step, on a bounded sample of the workload."""
    import gen_inputs as G
    from gen_inputs import O
    ncores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    torch.manual_seed(hp["seed"])
    cfg = O.make_cfg(**mp)
    sd = O.attach_grads(O.random_state(cfg, hp["seed"]), cfg)
    orc = O.OracleVQVAE(sd, drop_scale=1.0, **mp)
    orc.beta = hp["beta"]
    orc.training_steps = 1
    opt = torch.optim.AdamW(orc.params(), lr=hp["lr"], weight_decay=hp["wd"])
    x, mask = G.curve_batch(B_cpu, L, 7, ragged=False)
    orc.train_step(x, mask, opt, hp["clip"], weights)          # warm-up
    t0 = time.time()
    for _ in range(steps):
        orc.train_step(x, mask, opt, hp["clip"], weights)
    dt = (time.time() - t0) / steps
    return {"value": round(B_cpu / dt, 3), "unit": "images/s", "cores": ncores, "kind": "port",
            "sample": f"oracle train step (dropout 0.1, AdamW+clip), B={B_cpu} L={L}, {steps} timed steps after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-rank batch (weak scaling)")
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = torch.distributed
    selftest = os.environ.get("VQH_DP_SELFTEST") == "1"      # one-rank RCCL group: exercises the N>1 code path on one GPU
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from models import vae_models
    from vqvae_hip import lib as L
    if os.environ.get("VQH_GEMM_FLAGS"):                # tuning / diagnostic bits of vqh_gemm_set_flags (A/B runs)
        L.lib().vqh_gemm_set_flags(int(os.environ["VQH_GEMM_FLAGS"]))
    mp, weights, hp = c2_setup()
    torch.manual_seed(hp["seed"])                        # same initial weights on every rank (DDP semantics)
    model = vae_models["VQVAE"](**mp).to(dev).train()
    model.beta = hp["beta"]
    eng = model._engine()
    eng.rng[0] = hp["seed"] + 1000 * rank                # decorrelated dropout per rank
    B, Lq = args.batch, args.seq
    x, mask = synthetic_batch(B, Lq, 1000 + rank, dev)

    def step():
        return eng.train_step(x, mask, weights, hp["lr"], hp["wd"], hp["clip"], use_graph=not args.no_graph)

    for _ in range(max(args.warmup, 3)):                 # >= 3: eager warm-up, capture, first replay
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    metrics = eng.metrics_dict(weights)

    # ---- roofline of the dominant kernel: HIP events around every GEMM main-kernel launch of 2 eager steps, recorded
    # by the library on the launch stream and keyed by kernel instantiation (the names rocprofv3 prints) -------------
    roof = None
    prof = {}

    def two_eager_steps():
        for _ in range(2):      # every rank runs them (they contain the collective); only rank 0 reports
            eng.train_step(x, mask, weights, hp["lr"], hp["wd"], hp["clip"], use_graph=False)
    prof = L.gemm_profile(two_eager_steps)
    if rank == 0:
        tot_t = sum(v[1] for v in prof.values())
        tot_f = sum(v[2] for v in prof.values())
        dom = max(prof, key=lambda k: prof[k][1])            # the instantiation with the largest summed time
        n, t, f = prof[dom]
        kname = L.gemm_kernel_name(dom)
        traffic = None
        try:   # HBM bytes per launch from the separate rocprofv3 --pmc passes committed under profiles/
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_c2_pmc_traffic.json")))
            for pk, d in pm.items():
                if kname in pk and "hbm_bytes_per_launch_corrected" in d:
                    traffic = d["hbm_bytes_per_launch_corrected"]
        except Exception:
            traffic = None
        per_kernel = {L.gemm_kernel_name(k):
                      {"launches_per_step": v[0] // 2, "avg_launch_us": round(v[1] / v[0] * 1e6, 2),
                       "achieved": round(v[2] / v[1] / 1e12, 2)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(f / t / 1e12, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(f / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "launches_per_step": n // 2, "avg_launch_us": round(t / n * 1e6, 2),
                "gflop_per_launch": round(f / n / 1e9, 3),
                "all_gemm_kernels": {"achieved": round(tot_f / tot_t / 1e12, 2), "time_ms_per_step": round(tot_t / 2 * 1e3, 3),
                                     "gflop_per_step": round(tot_f / 2 / 1e9, 1), "per_kernel": per_kernel},
                "step_level": {"gflop_per_sample": GFLOP_PER_SAMPLE_C2,
                               "achieved": round(value / world * GFLOP_PER_SAMPLE_C2 / 1e3, 2),
                               "frac": round(value / world * GFLOP_PER_SAMPLE_C2 / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(mp, weights, hp)

    if rank == 0:
        out = {"metric": "train images/sec @64x64x3 bs256 (curve tensors [B,64,6], SURVEY.md s0)", "value": round(value, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "C2: stage2_vq.yaml model with num_quantizers=1 K=512 D=64 N=64; full train step "
                                      f"(fwd+loss+bwd+clip+AdamW+EMA), dropout 0.1, loss weights at epoch {BENCH_EPOCH}",
                          "per_gpu_batch": B, "global_batch": B * world, "seq_len": Lq, "parallelism": f"dp{world}",
                          "hipgraph": not args.no_graph, "params": int(sum(p.numel() for p in model.parameters()))},
               "loss": round(metrics["loss"], 6), "vq_loss": round(metrics["VQ_Loss"], 8),
               "recon_loss_xyz": round(metrics["Reconstruction_Loss_XYZ"], 6),
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1 or selftest:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
