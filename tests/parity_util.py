# -*- coding: utf-8 -*-
"""Tolerance rules shared by the parity tests.

BASELINE.json:north_star: code indices bit-exact; losses and reconstructions within 1e-5 relative in fp32.
SURVEY.md 7.2 ("fp64 arbiter"): twelve pre-LN transformer blocks evaluated with a different summation order cannot all
land inside 1e-5 of an fp32 reference whose OWN round-off is larger than that for some entries (a regulariser that is a
difference of two O(1) numbers, a gradient whose true value is zero).  So every fixture also records the same step
re-evaluated in fp64 from the reference's fp32 state, and a value is accepted when

        |got - ref32|  <=  max(1e-5 * |ref32|, 4 * |ref32 - ref64|)  +  1e-8

-- i.e. 1e-5 relative wherever the reference itself is good to 1e-5, and never further from the reference than four
times the reference's own distance to the exact result.  For tensors the rule is applied in max-norm with |ref32|
= max|ref32| (relative to the tensor's scale)."""
import numpy as np
import torch

REL = 1e-5
ARB = 4.0
FLOOR = 1e-8


def scalar_tol(ref32, ref64=None, rel=REL, floor=FLOOR):
    t = rel * abs(float(ref32))
    if ref64 is not None:
        t = max(t, ARB * abs(float(ref32) - float(ref64)))
    return t + floor


def assert_scalar(got, ref32, ref64=None, what="", rel=REL):
    got = float(got.detach()) if hasattr(got, "detach") else float(got)
    ref32 = float(ref32)
    tol = scalar_tol(ref32, ref64, rel)
    assert abs(got - ref32) <= tol, f"{what}: got {got!r} vs reference {ref32!r} (|diff| {abs(got - ref32):.3e} > tol {tol:.3e})"


def assert_losses(got, keys, v32, v64=None, what=""):
    """got: mapping name -> value (tensor or float); keys/v32/v64: arrays from a fixture."""
    for i, k in enumerate(keys):
        assert_scalar(got[str(k)], v32[i], None if v64 is None else v64[i], f"{what} loss[{k}]")


def _t(a):
    if torch.is_tensor(a):
        return a.detach().double().cpu()
    return torch.as_tensor(np.asarray(a)).double()


def assert_tensor(got, ref32, err64=None, what="", rel=REL, floor=FLOOR):
    """max|got - ref32| <= max(rel * max|ref32|, 4 * err64) + floor, err64 = max|ref32 - ref64| from the fixture."""
    got, ref32 = _t(got), _t(ref32)
    assert got.shape == ref32.shape or got.numel() == ref32.numel(), f"{what}: shape {tuple(got.shape)} vs {tuple(ref32.shape)}"
    err = float((got.reshape(-1) - ref32.reshape(-1)).abs().max()) if got.numel() else 0.0
    scale = float(ref32.abs().max()) if ref32.numel() else 0.0
    tol = rel * scale
    if err64 is not None:
        tol = max(tol, ARB * float(err64))
    tol += floor
    assert err <= tol, f"{what}: max|diff| {err:.3e} > tol {tol:.3e} (scale {scale:.3e}, err64 {err64})"
    return err


def assert_norm_close(got, ref32, ref64=None, what="", rel=REL, floor=1e-12):
    """L2 form of the arbiter rule, for quantities where single elements are legitimately noise (post-optimizer weights:
    Adam's first updates are ~lr*sign(g), so an element whose gradient is round-off may move by +-lr)."""
    got, ref32 = _t(got).reshape(-1), _t(ref32).reshape(-1)
    err = float((got - ref32).norm())
    tol = rel * float(ref32.norm())
    if ref64 is not None:
        tol = max(tol, ARB * float((_t(ref64).reshape(-1) - ref32).norm()))
    tol += floor
    assert err <= tol, f"{what}: ||diff|| {err:.3e} > tol {tol:.3e}"
    return err
