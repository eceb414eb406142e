# -*- coding: utf-8 -*-
"""GPU parity suite (-m gpu): the HIP path, called through the C ABI, against
  * the committed golden vectors recorded from the real reference (tests/golden/*.npz), and
  * the CPU oracle (oracle/vqvae_oracle.py) on the same seeded inputs.
Tolerances (north_star): code indices bit-exact; losses / reconstructions / gradients within 1e-5 relative in fp32,
with the reference's own fp32 round-off arbitrated by an fp64 re-evaluation recorded in the fixtures
(tests/parity_util.py: |got - ref32| <= max(1e-5 |ref32|, 4 |ref32 - ref64|) + 1e-8)."""
import math
import os

import numpy as np
import pytest
import torch

import gen_inputs as G
from gen_inputs import O
from conftest import load_golden
from parity_util import assert_losses, assert_tensor, assert_scalar

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _hip():
    from vqvae_hip import lib
    lib.require_gpu()
    return lib


def _model(cfg_kw, sd0):
    from models import vae_models
    m = vae_models["VQVAE"](**cfg_kw)
    missing, unexpected = m.load_state_dict(sd0, strict=True)
    m = m.to(DEV)
    eng = m._engine()
    eng.drop_scale = 0.0
    return m, eng


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


# ------------------------------------------------------------------------------------------------
# GEMM + epilogues
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("akc,bkc,M,N,K", [(1, 1, 257, 96, 130), (1, 0, 200, 64, 77), (0, 0, 64, 48, 1000),
                                            (0, 1, 33, 65, 129), (1, 1, 1024, 512, 512), (1, 1, 111, 6, 64), (1, 1, 5, 3, 0),
                                            # skinny shapes (reconstruction heads): streaming kernels behind the same entry point
                                            (1, 1, 1000, 3, 512), (1, 1, 257, 8, 64), (1, 0, 1000, 512, 3), (1, 0, 77, 64, 8),
                                            # evenly tiled shapes: the 256x128 LDS-DMA kernel, all four operand layouts,
                                            # one / several K-steps, odd K-step counts
                                            (1, 1, 4096, 768, 160), (1, 0, 4096, 768, 160), (0, 1, 4096, 768, 96), (0, 0, 4096, 768, 32),
                                            (1, 1, 8192, 512, 512), (1, 0, 2560, 1280, 64), (0, 0, 3072, 1024, 224)])
def test_gemm_layouts(akc, bkc, M, N, K):
    L = _hip()
    torch.manual_seed(M + N + K)
    A = torch.randn((M, K) if akc else (K, M), device=DEV)
    B = torch.randn((N, K) if bkc else (K, N), device=DEV)
    bias = torch.randn(N, device=DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(1 << 22, device=DEV)
    L.gemm(akc, bkc, M, N, K, A, max(1, A.stride(0)), B, max(1, B.stride(0)), C, N, bias=bias, ws=ws)
    ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double() + bias.double()
    assert rel(C, ref) < 2e-6


@pytest.mark.parametrize("rows,n_out,k_in", [(1000, 96, 130), (4096, 512, 256), (1000, 3, 512), (300, 6, 64), (16384, 3, 512),
                                              (16384, 512, 512), (8192, 1536, 512), (4000, 256, 128)])
def test_gemm_wgrad_weight_and_bias_gradients(rows, n_out, k_in):
    """dW = dY^T X and db = column sums of dY in one entry point (split-K slabs, fused row sums, skinny outputs), with
    and without accumulation into existing gradients."""
    L = _hip()
    torch.manual_seed(rows + n_out)
    dY, X = torch.randn(rows, n_out, device=DEV), torch.randn(rows, k_in, device=DEV)
    ws = torch.empty(1 << 23, device=DEV)
    dW, db = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
    L.call("vqh_gemm_wgrad", rows, n_out, k_in, dY, n_out, X, k_in, dW, k_in, db, 0.0, ws, ws.numel())
    ref_w, ref_b = dY.double().t() @ X.double(), dY.double().sum(0)
    assert rel(dW, ref_w) < 3e-6 and rel(db, ref_b) < 3e-6
    L.call("vqh_gemm_wgrad", rows, n_out, k_in, dY, n_out, X, k_in, dW, k_in, db, 1.0, ws, ws.numel())     # accumulate
    assert rel(dW, 2 * ref_w) < 3e-6 and rel(db, 2 * ref_b) < 3e-6
    dW2 = torch.empty_like(dW)
    L.call("vqh_gemm_wgrad", rows, n_out, k_in, dY, n_out, X, k_in, dW2, k_in, None, 0.0, ws, ws.numel())  # no bias grad
    assert rel(dW2, ref_w) < 3e-6


def test_grouped_weight_gradients_match_fp64():
    """vqh_gemm_wgrad_group: the weight-gradient products of one layer in ONE launch with a common K-chunk (evenly tiled
    products grouped, the others executed one by one), against fp64; bit 8 of the GEMM flags runs everything one by one."""
    L = _hip()
    torch.manual_seed(11)
    rows = 4096
    shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048), (64, 512), (3, 512), (1024, 512), (512, 1024), (256, 128), (512, 512)]
    ws = torch.empty(48 << 20, device=DEV)
    for flags in (1, 1 | 256):
        old = L.lib().vqh_gemm_set_flags(flags)
        try:
            items, refs = [], []
            for i, (n_out, k_in) in enumerate(shapes):
                r = rows if i != 6 else 1024                      # one product with a shorter reduction
                dY, X = torch.randn(r, n_out, device=DEV), torch.randn(r, k_in, device=DEV)
                dW = torch.full((n_out, k_in), float("nan"), device=DEV)
                db = torch.full((n_out,), float("nan"), device=DEV) if i % 3 != 2 else None
                items.append((dY, n_out, X, k_in, r, dW, db))
                refs.append((dY.double().t() @ X.double(), dY.double().sum(0)))
            L.wgrad_group(items, ws)
            torch.cuda.synchronize()
            for (dY, _, X, _, r, dW, db), (rw, rb) in zip(items, refs):
                assert rel(dW, rw) < 3e-6, dW.shape
                if db is not None:
                    assert rel(db, rb) < 3e-6, dW.shape
        finally:
            L.lib().vqh_gemm_set_flags(old)


def test_gemm_accumulate_into_output():
    """beta = 1 (second head's input gradient added to the first, decode_bwd) on the MFMA and the skinny path."""
    L = _hip()
    torch.manual_seed(5)
    for M, N, K in ((500, 512, 3), (300, 256, 96)):
        A, B = torch.randn(M, K, device=DEV), torch.randn(K, N, device=DEV)
        C = torch.randn(M, N, device=DEV)
        ref = C.double() + A.double() @ B.double()
        L.gemm(1, 0, M, N, K, A, K, B, N, C, N, beta=1.0)
        assert rel(C, ref) < 2e-6


def test_gemm_epilogues_and_dropout_determinism():
    L = _hip()
    torch.manual_seed(1)
    M, N, K = 300, 256, 128
    X, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 8, torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV)
    lin = X.double() @ W.double().t() + b.double()
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_GELU, aux_out=aux, ldaux=N)
    assert rel(aux, lin) < 2e-6 and rel(out, torch.nn.functional.gelu(lin)) < 2e-6
    L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_SIGMOID)
    assert rel(out, torch.sigmoid(lin)) < 2e-6
    L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_DROP_RESID, aux_in=R, ldaux=N)
    assert rel(out, lin + R.double()) < 2e-6
    L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_MUL_GELUGRAD, aux_in=R, ldaux=N)
    rd = R.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(rd).sum().backward()
    assert rel(out, (X.double() @ W.double().t()).cpu() * rd.grad) < 5e-6
    # dropout: mean keep rate, scale, and forward/backward mask agreement through vqh_dropout_bwd
    rng = torch.tensor([1234, 7], device=DEV, dtype=torch.int64)
    p = 0.1
    L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_DROP_RESID, aux_in=torch.zeros_like(R), ldaux=N,
           rng=rng, site=5, p=p)
    ones, keep = torch.ones(M * N, device=DEV), torch.empty(M * N, device=DEV)
    L.call("vqh_dropout_bwd", ones, keep, M * N, rng, 5, p)
    keep = keep.view(M, N)
    assert abs(float((keep == 0).float().mean()) - p) < 0.01
    assert rel(out, lin.float().to(DEV) * keep) < 2e-6
    out2 = torch.empty_like(out)
    L.gemm(1, 1, M, N, K, X, K, W, K, out2, N, bias=b, mode=L.EPI_RELU_DROP, rng=rng, site=5, p=p)
    assert rel(out2, torch.relu(lin).float().to(DEV) * keep) < 2e-6
    rng2 = torch.tensor([1234, 8], device=DEV, dtype=torch.int64)   # next step -> different mask
    L.call("vqh_dropout_bwd", ones, out2.view(-1), M * N, rng2, 5, p)
    assert float((out2 != keep).float().mean()) > 0.05


@pytest.mark.parametrize("flags", [1, 1 | 128])
def test_gemm_epilogues_on_evenly_tiled_shapes(flags):
    """The fused epilogues on a shape the 256x128 LDS-DMA kernel takes (flags bit 7 forces the 128x128 kernel: both must
    give the same answers, dropout masks included)."""
    L = _hip()
    torch.manual_seed(7)
    M, N, K = 2560, 1280, 96
    X, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 8, torch.randn(N, device=DEV)
    Wt = W.t().contiguous()
    R = torch.randn(M, N, device=DEV)
    lin = X.double() @ W.double().t() + b.double()
    rng = torch.tensor([4321, 9], device=DEV, dtype=torch.int64)
    old = L.lib().vqh_gemm_set_flags(flags)
    try:
        out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b)
        assert rel(out, lin) < 2e-6
        L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_GELU, aux_out=aux, ldaux=N)
        assert rel(aux, lin) < 2e-6 and rel(out, torch.nn.functional.gelu(lin)) < 2e-6
        L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_SIGMOID)
        assert rel(out, torch.sigmoid(lin)) < 2e-6
        p = 0.1
        ones, keep = torch.ones(M * N, device=DEV), torch.empty(M * N, device=DEV)
        L.call("vqh_dropout_bwd", ones, keep, M * N, rng, 5, p)
        keep = keep.view(M, N).double()
        L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_DROP_RESID, aux_in=R, ldaux=N, rng=rng, site=5, p=p)
        assert rel(out, lin * keep + R.double()) < 2e-6
        L.gemm(1, 1, M, N, K, X, K, W, K, out, N, bias=b, mode=L.EPI_RELU_DROP, rng=rng, site=5, p=p)
        assert rel(out, torch.relu(lin) * keep) < 2e-6
        # dgrad forms: dX[M, K'] = dY[M, N'] . W[N', K'] with the masks of the forward
        dY = torch.randn(M, K, device=DEV)                      # reuse sizes: [M, K] . Wt^T?  -> out2 [M, N] = dY . Wt[K, N]
        out2 = torch.empty(M, N, device=DEV)
        prod = dY.double() @ Wt.double()
        L.gemm(1, 0, M, N, K, dY, K, Wt, N, out2, N, mode=L.EPI_MUL_POSMASK, aux_in=R, ldaux=N, p=p)
        scale = 1.0 / (1.0 - round(p * 65536) / 65536.0)
        assert rel(out2, prod * (R.double() > 0) * scale) < 2e-6
        L.gemm(1, 0, M, N, K, dY, K, Wt, N, out2, N, mode=L.EPI_MUL_GELUGRAD, aux_in=R, ldaux=N)
        rd = R.double().cpu().requires_grad_(True)
        torch.nn.functional.gelu(rd).sum().backward()
        assert rel(out2, prod.cpu() * rd.grad) < 5e-6
        C0 = torch.randn(M, N, device=DEV)
        C1 = C0.clone()
        L.gemm(1, 0, M, N, K, dY, K, Wt, N, C1, N, beta=1.0)
        assert rel(C1, C0.double() + prod) < 2e-6
    finally:
        L.lib().vqh_gemm_set_flags(old)


def test_layernorm_forward_backward():
    L = _hip()
    torch.manual_seed(2)
    for rows, H in ((37, 64), (1000, 512), (5, 200)):
        x = torch.randn(rows, H, device=DEV) * 3 + 1
        w, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
        dy = torch.randn(rows, H, device=DEV)
        y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        L.call("vqh_layernorm_fwd", x, H, w, b, y, H, mean, rstd, rows, H, 1e-5)
        xr, wr, br = (t.double().cpu().requires_grad_(True) for t in (x, w, b))
        yr = torch.nn.functional.layer_norm(xr, (H,), wr, br, 1e-5)
        yr.backward(dy.double().cpu())
        assert rel(y, yr) < 2e-6
        dx, dw, db = torch.ones_like(x), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
        ws = torch.empty(1 << 21, device=DEV)
        L.call("vqh_layernorm_bwd", dy, H, x, H, w, mean, rstd, dx, H, 1, dw, db, 0.0, rows, H, None, None, 0, 0.0, ws, ws.numel())
        assert rel(dx - 1.0, xr.grad) < 5e-6 and rel(dw, wr.grad) < 5e-6 and rel(db, br.grad) < 5e-6
        # folded dropout backward: second output = dx * keep-mask(site), identical to the stand-alone pass
        rng = torch.tensor([77, 3], device=DEV, dtype=torch.int64)
        dx2, dxd = torch.zeros_like(x), torch.full_like(x, float("nan"))
        L.call("vqh_layernorm_bwd", dy, H, x, H, w, mean, rstd, dx2, H, 0, dw, db, 0.0, rows, H, dxd, rng, 9, 0.2, ws, ws.numel())
        want = torch.empty_like(x)
        L.call("vqh_dropout_bwd", dx2, want, rows * H, rng, 9, 0.2)
        assert rel(dx2, xr.grad) < 5e-6 and torch.equal(dxd, want)
        assert 0.1 < float((dxd == 0).float().mean()) < 0.3


@pytest.mark.parametrize("B,nh,T,S,dh,self_attn,ragged", [(3, 4, 24, 24, 16, True, True), (2, 8, 64, 64, 64, True, False),
                                                          (2, 8, 70, 33, 64, False, True), (2, 2, 8, 100, 32, False, True),
                                                          (2, 4, 50, 64, 32, False, True), (3, 2, 64, 17, 64, False, True),
                                                          (2, 2, 33, 64, 16, False, False),
                                                          # T or S >= 128: the 4-wave general kernels, several K/V chunks
                                                          (2, 4, 160, 200, 64, False, True), (1, 2, 256, 256, 32, True, True),
                                                          (2, 2, 130, 70, 16, False, False),
                                                          # head dim 64, beyond 64 queries or keys: the bf16x3 kernels (attention_x3.inc)
                                                          (2, 8, 350, 350, 64, True, True), (2, 8, 64, 350, 64, False, True),
                                                          (1, 8, 350, 64, 64, False, False), (2, 2, 129, 65, 64, False, True)])
def test_attention_forward_backward(B, nh, T, S, dh, self_attn, ragged):
    L = _hip()
    torch.manual_seed(T * S + dh)
    E = nh * dh
    q = torch.randn(B, T, E, device=DEV)
    k = torch.randn(B, S, E, device=DEV)
    v = torch.randn(B, S, E, device=DEV)
    do = torch.randn(B, T, E, device=DEV)
    valid = torch.ones(B, S, dtype=torch.bool, device=DEV)
    if ragged:
        lens = torch.randint(max(1, S // 2), S + 1, (B,))
        valid = (torch.arange(S)[None] < lens[:, None]).to(DEV)
    o, lse = torch.empty(B, T, E, device=DEV), torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, valid, B, nh, T, S, dh, 0, None, 0, 0.0)
    qd, kd, vd = (t.double().cpu().requires_grad_(True) for t in (q, k, v))
    sc = (qd.view(B, T, nh, dh).transpose(1, 2) @ kd.view(B, S, nh, dh).transpose(1, 2).transpose(-1, -2)) / math.sqrt(dh)
    sc = sc.masked_fill(~valid.cpu()[:, None, None, :], float("-inf"))
    ref = (torch.softmax(sc, -1) @ vd.view(B, S, nh, dh).transpose(1, 2)).transpose(1, 2).reshape(B, T, E)
    ref.backward(do.double().cpu())
    assert rel(o, ref) < 3e-6
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dsum = torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, valid, B, nh, T, S, dh, 0, None, 0, 0.0)
    assert rel(dq, qd.grad) < 5e-6 and rel(dk, kd.grad) < 5e-6 and rel(dv, vd.grad) < 5e-6


@pytest.mark.parametrize("T,S,dh", [(40, 40, 16), (160, 136, 64), (72, 200, 32)])
def test_attention_dropout_consistent_between_forward_and_backward(T, S, dh):
    """With dropout on, backward must regenerate the forward mask: check d(sum(O*dO))/dV numerically-free via linearity:
    O is linear in V, so O(V) . dO == V . dV for the same mask."""
    L = _hip()
    torch.manual_seed(3)
    B, nh = 2, 4
    E = nh * dh
    q, do = torch.randn(B, T, E, device=DEV), torch.randn(B, T, E, device=DEV)
    k, v = torch.randn(B, S, E, device=DEV), torch.randn(B, S, E, device=DEV)
    rng = torch.tensor([99, 3], device=DEV, dtype=torch.int64)
    o, lse = torch.empty(B, T, E, device=DEV), torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, None, B, nh, T, S, dh, 0, rng, 11, 0.25)
    dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, None, B, nh, T, S, dh, 0, rng, 11, 0.25)
    lhs, rhs = float((o.double() * do.double()).sum()), float((v.double() * dv.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)
    o2 = torch.empty_like(o)
    L.call("vqh_attn_fwd", q, E, k, E, v, E, o2, E, lse, None, B, nh, T, S, dh, 0, None, 0, 0.0)
    assert float((o - o2).abs().max()) > 1e-3          # dropout really changed the output


@pytest.mark.parametrize("T,S,dh,p", [(64, 64, 64, 0.1), (40, 64, 32, 0.25), (64, 24, 16, 0.0)])
def test_attention_short_sequence_kernels_equal_general_kernels(T, S, dh, p):
    """T,S <= 64 take fused single-workgroup kernels; they must reproduce the general flash kernels (same dropout
    masks, same masking) to rounding."""
    L = _hip()
    torch.manual_seed(T + S + dh)
    B, nh = 3, 4
    E = nh * dh
    q, do = torch.randn(B, T, E, device=DEV), torch.randn(B, T, E, device=DEV)
    k, v = torch.randn(B, S, E, device=DEV), torch.randn(B, S, E, device=DEV)
    lens = torch.randint(max(1, S // 2), S + 1, (B,))
    valid = (torch.arange(S)[None] < lens[:, None]).to(DEV)
    rng = torch.tensor([7, 5], device=DEV, dtype=torch.int64)
    res = []
    for flags in (0, 1, 2):          # quartered short-sequence backward, general kernels, round-1 short-sequence backward
        old = L.lib().vqh_attn_set_flags(flags)
        try:
            o, lse = torch.empty(B, T, E, device=DEV), torch.empty(B * nh * T, device=DEV)
            L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, valid, B, nh, T, S, dh, 0, rng, 5, p)
            dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=DEV)
            L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, valid, B, nh, T, S, dh, 0,
                   rng, 5, p)
            torch.cuda.synchronize()
        finally:
            L.lib().vqh_attn_set_flags(old)
        res.append((o, lse, dq, dk, dv))
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert rel(x, y) < 2e-6


@pytest.mark.parametrize("T,S,p", [(350, 350, 0.1), (64, 350, 0.1), (200, 96, 0.0), (130, 257, 0.25), (64, 64, 0.1), (40, 64, 0.25),
                                   (64, 17, 0.0), (33, 31, 0.1), (7, 5, 0.1)])
def test_attention_x3_kernels_equal_the_fp32_mfma_kernels(T, S, p):
    """Head dim 64 runs on the bf16 matrix pipes from exact 3-way operand splits (attention_x3.inc: the general kernels beyond 64
    queries / keys, the fused short-sequence backward up to 64);
    vqh_attn_set_flags bit 2 keeps the native fp32 MFMA kernels.  Same masking, the SAME dropout masks (a dropped probability is
    dropped in both), results equal to fp32 round-off, and both as close to an fp64 evaluation."""
    L = _hip()
    torch.manual_seed(T + S)
    B, nh, dh = 2, 8, 64
    E = nh * dh
    q, do = torch.randn(B, T, E, device=DEV), torch.randn(B, T, E, device=DEV)
    k, v = torch.randn(B, S, E, device=DEV), torch.randn(B, S, E, device=DEV)
    lens = torch.randint(max(1, S // 2), S + 1, (B,))
    valid = (torch.arange(S)[None] < lens[:, None]).to(DEV)
    rng = torch.tensor([17, 9], device=DEV, dtype=torch.int64)
    res = []
    for flags in (0, 4):
        old = L.lib().vqh_attn_set_flags(flags)
        try:
            o, lse = torch.empty(B, T, E, device=DEV), torch.empty(B * nh * T, device=DEV)
            L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, valid, B, nh, T, S, dh, 0, rng, 5, p)
            dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=DEV)
            L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, valid, B, nh, T, S, dh, 0, rng, 5, p)
            torch.cuda.synchronize()
        finally:
            L.lib().vqh_attn_set_flags(old)
        res.append((o, lse, dq, dk, dv))
    for x, y in zip(res[0], res[1]):
        assert rel(x, y) < 2e-6


def test_attention_dropout_gradients_match_autograd_with_extracted_mask():
    """Recover the kernel's dropout mask (V = identity makes O = P*keep), then check dQ/dK/dV against torch autograd
    evaluated with that explicit mask: forward, dQ kernel and dK/dV kernel must all regenerate the same mask."""
    L = _hip()
    torch.manual_seed(4)
    B, nh, T, S, dh, p = 2, 2, 40, 16, 16, 0.3
    E = nh * dh
    q, k, do = torch.randn(B, T, E, device=DEV), torch.randn(B, S, E, device=DEV), torch.randn(B, T, E, device=DEV)
    rng = torch.tensor([7, 5], device=DEV, dtype=torch.int64)
    eye = torch.eye(S, dh, device=DEV).repeat(1, nh)[None].repeat(B, 1, 1).contiguous()      # V[b, s, head*dh + d] = (s == d)
    o, o0, lse = torch.empty(B, T, E, device=DEV), torch.empty(B, T, E, device=DEV), torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_fwd", q, E, k, E, eye, E, o, E, lse, None, B, nh, T, S, dh, 0, rng, 3, p)
    L.call("vqh_attn_fwd", q, E, k, E, eye, E, o0, E, lse, None, B, nh, T, S, dh, 0, None, 0, 0.0)
    keep = (o / o0).view(B, T, nh, dh).transpose(1, 2)[..., :S]                                # [B, nh, T, S] in {0, 1/(1-p)}
    assert abs(float((keep == 0).float().mean()) - p) < 0.05
    assert float(((keep - 1 / (1 - p)).abs() < 1e-4).float().mean() + (keep == 0).float().mean()) > 0.999
    v = torch.randn(B, S, E, device=DEV)
    L.call("vqh_attn_fwd", q, E, k, E, v, E, o, E, lse, None, B, nh, T, S, dh, 0, rng, 3, p)
    dq, dk, dv, dsum = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty(B * nh * T, device=DEV)
    L.call("vqh_attn_bwd", q, E, k, E, v, E, o, E, lse, do, E, dsum, dq, E, dk, E, dv, E, None, B, nh, T, S, dh, 0, rng, 3, p)
    qd, kd, vd = (t.double().cpu().requires_grad_(True) for t in (q, k, v))
    sc = (qd.view(B, T, nh, dh).transpose(1, 2) @ kd.view(B, S, nh, dh).transpose(1, 2).transpose(-1, -2)) / math.sqrt(dh)
    pm = torch.softmax(sc, -1) * keep.double().cpu().round(decimals=6)
    ref = (pm @ vd.view(B, S, nh, dh).transpose(1, 2)).transpose(1, 2).reshape(B, T, E)
    ref.backward(do.double().cpu())
    assert rel(o, ref) < 1e-5
    assert rel(dq, qd.grad) < 2e-5 and rel(dk, kd.grad) < 2e-5 and rel(dv, vd.grad) < 2e-5


# ------------------------------------------------------------------------------------------------
# quantizer against the golden vectors recorded from the reference
# ------------------------------------------------------------------------------------------------
from test_oracle import VQ_CASES, vq_setup, vq_masks, MODEL_CASES, EXTRA_CASES, model_inputs  # noqa: E402


@pytest.mark.parametrize("name", VQ_CASES)
def test_quantizer_matches_reference_golden(name):
    from models.vq_vae import VQVAE
    g = load_golden(name)
    B, M, K_per, D, Q, steps, zs, emb0 = vq_setup(g)
    m = VQVAE(hidden_dim=64, num_heads=4, tokenizer_heads=4, codebook_size=K_per, code_dim=D, num_quantizers=Q,
              latent_tokens=M, use_vq=True, reinit_dead_codes=False, print_init=False).to(DEV)
    q = m.quantizer
    q.embedding.copy_(emb0.to(DEV))
    if int(g["centroid_init"]):
        q.ema_embedding.copy_(emb0.to(DEV))
        q.ema_cluster_size.fill_(1.0)
    q.train(bool(int(g["train"])))
    masks = vq_masks(g)
    for s in range(steps):
        z = zs[s].view(B, M, D).to(DEV)
        zst, zq, idx, st = q(z, do_ema_update=True, allow_reinit=False, mask=None if masks[s] is None else masks[s].to(DEV))
        got = idx.reshape(-1).cpu().numpy().astype(np.int32)
        want = g[f"idx_{s}"]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{bad.size} index mismatches at rows {bad[:8]}"
        assert rel(st, g[f"stats_{s}"]) < 1e-5
        assert rel(zq.reshape(-1, D)[:8], g[f"zq_head_{s}"]) < 1e-6
        assert float((zst - (z + (zq - z))).abs().max()) == 0.0
        assert rel(q.ema_cluster_size, g[f"ecs_{s}"]) < 2e-6
        if f"emb_{s}" in g:
            assert rel(q.embedding, g[f"emb_{s}"]) < 2e-6 and rel(q.ema_embedding, g[f"eemb_{s}"]) < 2e-6
        else:
            assert rel(q.embedding[:16], g[f"emb_head_{s}"]) < 2e-6
        assert rel(q._ep_usage, g[f"ep_usage_{s}"]) == 0.0 and float(q._ep_cnt) == float(np.asarray(g[f"ep_cnt_{s}"]).reshape(-1)[0])
        # the dict get_epoch_stats() returns (models/vq_vae.py:118-164), against the reference's own call on the same state
        es, want_es = q.get_epoch_stats(), g[f"epstats_{s}"]
        got_es = [es["perplexity"], es["dead_ratio"], es["n_positions"], es["margin_mean"], es["qe_mean"], es["qe_p90"]]
        assert torch.equal(es["usage_hist"], torch.from_numpy(g[f"ep_usage_{s}"]))
        for name_es, a, b in zip(("perplexity", "dead_ratio", "n_positions", "margin_mean", "qe_mean", "qe_p90"), got_es, want_es):
            assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(b))), (name_es, a, b)


def test_quantizer_bit_exact_vs_oracle_at_c2_shape():
    """R=16384, K=512, D=64 (config C2): indices bit-exact against the oracle on the same seeded input."""
    from models.vq_vae import VQVAE
    z, emb = G.vq_inputs(16384, 512, 64, 4242)
    m = VQVAE(hidden_dim=64, num_heads=4, tokenizer_heads=4, codebook_size=512, code_dim=64, latent_tokens=64,
              reinit_dead_codes=False, print_init=False).to(DEV)
    m.quantizer.embedding.copy_(emb.to(DEV))
    m.quantizer.eval()
    _, _, idx, _ = m.quantizer(z.view(256, 64, 64).to(DEV))
    d = (z.double() ** 2).sum(1, keepdim=True) - 2 * z.double() @ emb.double().t() + (emb.double() ** 2).sum(1)[None]
    assert torch.equal(idx.reshape(-1).cpu(), d.argmin(1))


# ------------------------------------------------------------------------------------------------
# loss function alone (all 19+5 terms, forward value and input gradients)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cfg_kw", [("loss_all_ragged", dict(G.SMALL_VQ, usage_entropy_lambda=0.01)),
                                         ("loss_all_full", G.SMALL_VQ), ("loss_short", G.SMALL_VQ),
                                         ("loss_datastats", G.SMALL_VQ)])
def test_loss_function_matches_reference_golden(name, cfg_kw):
    g = load_golden(name)
    sd0 = G.model_state(cfg_kw, int(g["seed"]))
    m, eng = _model(cfg_kw, sd0)
    m.train()
    if "stats_std" in g:
        m.set_data_stats(torch.from_numpy(g["stats_mean"]), torch.from_numpy(g["stats_std"]))
    weights = {k: float(v) for k, v in zip(g["weights_keys"], g["weights_vals"])}
    x, mask = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    rec, ze, zq = (torch.from_numpy(g[k]).to(DEV) for k in ("recons", "ze", "zq"))
    B, Nt = ze.shape[:2]
    for tag, mk in (("m", mask), ("nomask", None)):
        # perplexity / dead ratio travel through vq_pack, like the reference's call (models/vq_vae.py:1097)
        pack = (zq, ze, torch.zeros(B, Nt, dtype=torch.long, device=DEV), torch.tensor(3.0, device=DEV), torch.tensor(0.5, device=DEV))
        ld = m.loss_function(rec, x, pack, mk, **weights)
        assert_losses(ld, g[f"{tag}_loss_keys"], g[f"{tag}_loss_vals"], g[f"{tag}_loss_vals64"], f"{name}/{tag}")
        assert_tensor(eng.ctx["d_rec"].view(B, -1, 6), g[f"{tag}_d_recons"], g[f"{tag}_d_recons_err64"], f"{name}/{tag} d_recons")
        assert_tensor(eng.ctx["d_ze"].view(B, Nt, -1), g[f"{tag}_d_ze"], g[f"{tag}_d_ze_err64"], f"{name}/{tag} d_ze")


# ------------------------------------------------------------------------------------------------
# whole model: forward + loss + backward (+ AdamW) against the reference golden vectors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cfg_kw,_r", MODEL_CASES + EXTRA_CASES)
def test_train_step_matches_reference_golden(name, cfg_kw, _r):
    """The decomposed drop-in surface: forward -> loss_function -> loss.backward() -> clip + AdamW."""
    g = load_golden(name)
    batches, sd0, weights = model_inputs(g, cfg_kw)
    m, eng = _model(cfg_kw, sd0)
    m.train()
    m.training_steps = int(g["start_steps"]) if "start_steps" in g else 1
    x, mask = batches[0]
    out = m(x.to(DEV), mask.to(DEV))
    ld = m.loss_function(*out, **weights)
    assert_tensor(out[0], g["recons_0"], g["recons_err64_0"], "recons")
    assert_tensor(out[2][1], g["z_e_0"], g["z_e_err64_0"], "z_e")
    if m.use_vq:
        assert np.array_equal(out[2][2].reshape(-1).cpu().numpy().astype(np.int32), g["idx_0"]), "code indices must be bit-exact"
    assert_losses(ld, g["loss_keys_0"], g["loss_vals_0"], g["loss_vals64_0"], name)
    m.backward()
    eng.set_hyper(float(g["lr"]), float(g["wd"]), float(g["clip"]))
    eng.optimizer_step()        # clips the flat gradient in place (like clip_grad_norm_) then AdamW
    torch.cuda.synchronize()
    assert_scalar(eng.norm[0], g["grad_norm_0"], g["grad_norm64_0"], "total gradient norm")
    pnames = [str(k) for k in g["param_names"]]
    # per-tensor gradient norms (post-clip, like the fixture): ||g_hip|| within the arbiter rule of ||g_ref||; the
    # reference's own distance to fp64 is an upper bound of |  ||g32|| - ||g64||  | via the max-error * sqrt(n)
    for i, k in enumerate(pnames):
        n32, n64 = float(g["gradnorm_each_0"][i]), float(g["gradnorm_each64_0"][i])
        got = float(eng.G[k].norm())
        tol = max(1e-5 * n32, 4.0 * abs(n32 - n64)) + 1e-5 * float(g["grad_maxabs_each_0"].max()) * 1e-2 + 1e-12
        assert abs(got - n32) <= tol, f"||grad {k}||: {got} vs {n32} (tol {tol:.3e})"
    for key in g:
        if key.startswith("grad_0::"):
            k = key.split("::")[1]
            assert_tensor(eng.G[k], g[key], float(g["grad_maxerr64_each_0"][pnames.index(k)]), f"grad {k}")
    if m.use_vq:
        assert_tensor(m.quantizer.embedding, g["q_emb_0"], None, "codebook")
        assert_tensor(m.quantizer.ema_cluster_size, g["q_ecs_0"], None, "ema_cluster_size")


def test_layer0_shared_projection_equals_per_sample_path():
    """The first decoder / tokenizer layer projects its batch-invariant queries once (engine.share_layer0); the
    per-sample path (what the reference computes) must give the same losses and gradients."""
    name, cfg_kw, _r = MODEL_CASES[0]
    g = load_golden(name)
    batches, sd0, weights = model_inputs(g, cfg_kw)
    res = []
    for share in (True, False):
        m, eng = _model(cfg_kw, sd0)
        eng.share_layer0 = share
        m.train()
        x, mask = batches[0]
        ld = m.loss_function(*m(x.to(DEV), mask.to(DEV)), **weights)
        m.backward()
        torch.cuda.synchronize()
        res.append((float(ld["loss"]), eng.flat_g.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    assert rel(res[0][1], res[1][1]) < 1e-4


@pytest.mark.parametrize("width", ["small_rvq", "c2"])
def test_folded_dropout_backward_equals_standalone_pass(width):
    """With dropout ON, the LayerNorm backward of each block also writes dres * mask(next block's dropout site); the
    gradients must be bit-identical to the path that runs the stand-alone dropout-backward kernel per site (a wrong
    site pairing would silently train on wrong gradients; the golden vectors run with dropout off)."""
    if width == "small_rvq":
        name, cfg_kw, _r = MODEL_CASES[2]        # residual VQ, ragged
        g = load_golden(name)
        batches, sd0, weights = model_inputs(g, cfg_kw)
        x, mask = batches[0]
    else:                                        # true width (H=512, 4+2+4 layers), B=3, L=64
        cfg_kw, weights = dict(G.C2_MODEL), dict(G.BASE_LOSS_WEIGHTS)
        sd0 = G.model_state(cfg_kw, 77)
        x, mask = G.curve_batch(3, 64, 78, ragged=True)
    res = []
    for fold in (True, False):
        m, eng = _model(cfg_kw, sd0)
        eng.drop_scale = 1.0                  # real dropout (p = 0.1 / tokenizer_dropout)
        eng.fold_dropout_bwd = fold
        eng.rng[0] = 4242
        m.train()
        ld = m.loss_function(*m(x.to(DEV), mask.to(DEV)), **weights)
        m.backward()
        torch.cuda.synchronize()
        res.append((float(ld["loss"]), eng.flat_g.clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]), float((res[0][1] - res[1][1]).abs().max())


@pytest.mark.parametrize("name,cfg_kw,_r", MODEL_CASES)
def test_eval_forward_decode_match_reference_golden(name, cfg_kw, _r):
    g = load_golden(name)
    batches, sd0, weights = model_inputs(g, cfg_kw)
    m, eng = _model(cfg_kw, sd0)
    m.eval()
    x, mask = batches[0]
    with torch.no_grad():
        out = m(x.to(DEV), mask.to(DEV))
        ld = m.loss_function(*out, **weights)
    assert_tensor(out[0], g["eval_recons"], g["eval_recons_err64"], "eval recons")
    if m.use_vq:
        assert np.array_equal(out[2][2].reshape(-1).cpu().numpy().astype(np.int32), g["eval_idx"])
    assert_losses(ld, g["eval_loss_keys"], g["eval_loss_vals"], g["eval_loss_vals64"], f"{name} eval")
    assert_tensor(m.decode(out[2][0], mask.to(DEV)), g["eval_decode"], g["eval_recons_err64"], "decode")
    # encode / _tokenize_to_codes entry points used by the reference's scripts
    hf, hg, hs = m.encode(x.to(DEV), mask.to(DEV))
    z = m._tokenize_to_codes(hf, mask.to(DEV))
    assert rel(z, out[2][1]) < 1e-6


def test_true_width_c2_model_b2_matches_reference_golden():
    g = load_golden("model_c2_b2")
    batches, sd0, weights = model_inputs(g, G.C2_MODEL)
    m, eng = _model(G.C2_MODEL, sd0)
    m.train()
    m.training_steps = 1
    x, mask = batches[0]
    out = m(x.to(DEV), mask.to(DEV))
    ld = m.loss_function(*out, **weights)
    assert_tensor(out[0], g["recons_0"], g["recons_err64_0"], "recons")
    assert_tensor(out[2][1], g["z_e_0"], g["z_e_err64_0"], "z_e")
    assert np.array_equal(out[2][2].reshape(-1).cpu().numpy().astype(np.int32), g["idx_0"])
    assert_losses(ld, g["loss_keys_0"], g["loss_vals_0"], g["loss_vals64_0"], "c2_b2")
    m.backward()
    torch.cuda.synchronize()
    pnames = [str(k) for k in g["param_names"]]
    cl = min(1.0, float(g["clip"]) / (float(g["grad_norm_0"]) + 1e-6))
    for i, k in enumerate(pnames):
        n32, n64 = float(g["gradnorm_each_0"][i]), float(g["gradnorm_each64_0"][i])
        got = float(eng.G[k].norm()) * cl
        # ||g32 - g64|| <= sqrt(n) * max-error: the fp32 reference's own distance to the exact gradient norm
        noise = float(g["grad_maxerr64_each_0"][i]) * float(np.sqrt(eng.G[k].numel()))
        assert abs(got - n32) <= max(1e-5 * n32, 4.0 * abs(n32 - n64), noise) + 1e-12, (k, got, n32, n64)


def test_input_augmentation_kernel_and_plumbing():
    """vqh_augment against the reference formula (_random_rotation :331-345, :775-792) with the same draws; and the
    model-level plumbing: the target stays un-augmented, SS channels untouched, rotations preserve distances."""
    L = _hip()
    torch.manual_seed(11)
    B, Lq = 5, 17
    x = torch.randn(B, Lq, 6, device=DEV)
    u, t = torch.rand(B, 3, device=DEV), torch.randn(B, 3, device=DEV) * 0.02
    noise = torch.randn(B, Lq, 3, device=DEV) * 0.1
    out = torch.empty_like(x)
    L.call("vqh_augment", x, u, t, noise, out, B, Lq)
    u1, u2, u3 = u[:, 0].double(), u[:, 1].double(), u[:, 2].double()
    qx, qy = torch.sqrt(1 - u1) * torch.sin(2 * math.pi * u2), torch.sqrt(1 - u1) * torch.cos(2 * math.pi * u2)
    qz, qw = torch.sqrt(u1) * torch.sin(2 * math.pi * u3), torch.sqrt(u1) * torch.cos(2 * math.pi * u3)
    R = torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                     2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                     2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], -1).view(B, 3, 3)
    want = torch.einsum("bij,blj->bli", R, x[..., :3].double()) + t.double()[:, None] + noise.double()
    assert rel(out[..., :3], want) < 2e-6 and torch.equal(out[..., 3:], x[..., 3:])
    from models.vq_vae import VQVAE
    cfg = dict(G.SMALL_VQ, rigid_aug_prob=1.0, max_noise_std=0.0)
    m = VQVAE(**cfg)
    m.load_state_dict(G.model_state(G.SMALL_VQ, 3), strict=True)
    m = m.to(DEV).train()
    eng = m._engine()
    eng.drop_scale = 0.0
    xb, mask = G.curve_batch(4, 20, 12, ragged=False)
    outp = m(xb.to(DEV), mask.to(DEV))
    assert torch.equal(outp[1].cpu(), xb)                               # target is the ORIGINAL input
    xa = eng.buf["in.x_aug"].cpu()
    d0 = (xb[:, 1:, :3] - xb[:, :-1, :3]).norm(dim=-1)
    d1 = (xa[:, 1:, :3] - xa[:, :-1, :3]).norm(dim=-1)
    assert float((d0 - d1).abs().max()) < 1e-4 and float((xa[..., :3] - xb[..., :3]).abs().max()) > 0.1
    m.eval()
    eng.buf.pop("in.x_aug")
    m(xb.to(DEV), mask.to(DEV))
    assert "in.x_aug" not in eng.buf                                    # no augmentation in eval mode


def test_dead_code_reinit_semantics():
    """models/vq_vae.py:91-107: codes with batch usage <= threshold are re-seeded from encoder rows (embedding and
    ema_embedding equal to one of the z_e rows, ema_cluster_size = 1); live codes are untouched; gated by
    step % 500 == 0 and step >= max(freeze, 800)."""
    from models.vq_vae import VQVAE
    cfg = dict(G.SMALL_VQ, reinit_dead_codes=True, reinit_prob=1.0, dead_usage_threshold=0)
    m = VQVAE(**cfg)
    m.load_state_dict(G.model_state(cfg, 9), strict=True)
    m = m.to(DEV).train()
    eng = m._engine()
    eng.drop_scale = 0.0
    x, mask = G.curve_batch(6, 24, 10, ragged=True)
    m.training_steps = 998                      # forward makes it 999: no re-init
    m(x.to(DEV), mask.to(DEV))
    before = m.quantizer.embedding.clone()
    m(x.to(DEV), mask.to(DEV))                  # step 1000: trigger
    usage = eng.buf["vq.usage"].clone()
    dead = usage <= 0
    assert int(dead.sum()) > 0
    ze = eng.buf["tok.z_e"]
    q = m.quantizer
    for k in torch.nonzero(dead).flatten().tolist():
        hit = (ze == q.embedding[k]).all(dim=1)
        assert bool(hit.any()) and torch.equal(q.embedding[k], q.ema_embedding[k]) and float(q.ema_cluster_size[k]) == 1.0
    m2 = VQVAE(**dict(cfg, reinit_dead_codes=False))
    m2.load_state_dict(G.model_state(cfg, 9), strict=True)
    m2 = m2.to(DEV).train()
    m2._engine().drop_scale = 0.0
    m2.training_steps = 998
    m2(x.to(DEV), mask.to(DEV)); m2(x.to(DEV), mask.to(DEV))
    live = ~dead
    assert torch.equal(q.embedding[live], m2.quantizer.embedding[live])      # live codes identical to a run without re-init
    assert not torch.equal(q.embedding[dead], m2.quantizer.embedding[dead])


def test_adamw_and_clip_match_torch():
    L = _hip()
    torch.manual_seed(5)
    n = 100003
    p0, g0 = torch.randn(n), torch.randn(n) * 3
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=0.01)
    P, Gd = p0.clone().to(DEV), torch.empty(n, device=DEV)
    Mm, V = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper, norm = torch.zeros(9, device=DEV), torch.zeros(2, device=DEV)
    ws = torch.empty(1024, device=DEV, dtype=torch.float64)
    for t in range(1, 4):
        gt = g0 * t
        ref_p.grad = gt.clone()
        gn = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        Gd.copy_(gt)
        hyper.copy_(torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 1 - 0.9 ** t, 1 - 0.999 ** t, 1.0]))
        L.call("vqh_grad_norm", Gd, n, hyper, norm, ws)
        L.call("vqh_adamw_step", P, Gd, Mm, V, n, hyper, norm)
        assert abs(float(norm[0]) - float(gn)) < 1e-5 * float(gn)
        assert float((P.cpu() - ref_p.detach()).abs().max()) < 2e-6
