# -*- coding: utf-8 -*-
"""Data-parallel plumbing of the training step: one process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The step has ONE kind of exchange: sum all-reduces over the flat buffer [gradients | EMA statistics], issued as
2 * num_layers + 2 buckets that follow the backward phases (engine.bwd_phases: one per decoder layer, the tokenizer, the
SS encoder + fusion MLP, one per geometry-encoder layer; the statistics ride behind the last one), so that each bucket
travels while the next phase computes and only the last one (geometry layer 0 + input_proj + statistics: 7 % of the bytes
at C2) is exposed:
  * gradients: DDP semantics (mean over ranks) -- the 1/world factor is folded into the clip coefficient
    that the fused AdamW kernel already multiplies into every gradient (hyper[8]), so no extra pass;
  * EMA statistics cnt[K] | sum[K,D]: wanted as SUMS over ranks, so that an N-rank step equals the
    single-process step on the concatenated batch (the reference's DDP never reduces
    them and broadcasts rank 0's codebook instead -- a documented, deliberate difference).
The payload for config C2 is 172.5 MB + 0.13 MB in 10 buckets of 12-25 MB: on xGMI (point-to-point links,
7 x ~153 GB/s per GPU) large messages are bandwidth-bound per link, many small ones latency-bound, so the buckets
follow the backward phases instead of a fixed small size."""
import os

import torch


def dp_active():
    """True when the step must run its data-parallel form (graph segments + bucketed all-reduce).  VQH_DP_SELFTEST=1
    also selects it for an initialised ONE-rank group, so that the RCCL code path can be exercised on a one-GPU box."""
    d = torch.distributed
    if not (d.is_available() and d.is_initialized()):
        return False
    return d.get_world_size() > 1 or os.environ.get("VQH_DP_SELFTEST") == "1"


def world_size():
    d = torch.distributed
    return d.get_world_size() if (d.is_available() and d.is_initialized()) else 1


def rank():
    d = torch.distributed
    return d.get_rank() if (d.is_available() and d.is_initialized()) else 0


def shard_bounds(n, r=None, w=None):
    """Contiguous, equal shard [lo, hi) of n samples for rank r of w (n must divide evenly: weak scaling)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    if n % w:
        raise ValueError(f"global batch {n} is not divisible by world size {w}")
    per = n // w
    return r * per, (r + 1) * per


def allreduce_flat(flat, include_stats=True, n_grad=None):
    """Sum-all-reduce `flat` (or only its first n_grad entries) in place; returns the gradient scale 1/world."""
    w = world_size()
    if w > 1:
        torch.distributed.all_reduce(flat if (include_stats or n_grad is None) else flat[:n_grad])
    return 1.0 / w


def allreduce_async(view):
    """Start an in-place sum all-reduce of `view` (a contiguous slice of the flat buffer) and return its Work handle
    (None for a single process).  With the nccl backend the collective is ordered behind the work already queued on
    the current stream and runs on RCCL's own stream; handle.wait() makes the current stream wait for it."""
    if dp_active() and view.numel() > 0:
        return torch.distributed.all_reduce(view, async_op=True)
    return None
