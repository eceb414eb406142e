# -*- coding: utf-8 -*-
"""Plane-tensor GEMM (csrc/gemm_p3.inc: pre-split bf16 planes, LDS-DMA staging, v_mfma_f32_16x16x32_bf16) through the C ABI.

  * vqh_p3_split is EXACT: h + m + l == x bit for bit over the guaranteed domain (2^-100 .. 3.38e38), and a product with an
    identity matrix returns the other operand bit for bit in every layout;
  * on random data (normal, wide log-normal, all-positive) the result is as close to fp64 as the native fp32 MFMA tile;
  * every fused epilogue agrees with vqh_gemm's (same dropout masks: the hash is keyed by element index, not by kernel);
    plane outputs hold exactly the fp32 value the epilogue computed;
  * the grouped weight gradient (incl. the bias gradient computed by MFMAs against a ones fragment) agrees with the fp32 entry.
Reference call sites: every nn.Linear of models/vq_vae.py (forward :458-473, :525-528 and their autograd transposes)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-vae_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NATIVE = 512


def _hip():
    from vqvae_hip import lib as L
    L.lib()
    return L


def _rand(shape, seed, kind="normal"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(shape, generator=g)
    if kind == "wide":
        a = a * torch.exp(4.0 * torch.randn(shape, generator=g))
    elif kind == "positive":
        a = a.abs() + 1.0
    return a.to(DEV)


def _rms(c, ref):
    c, ref = c.double(), ref.double()
    return float(((c - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


def test_split_is_exact_over_the_guaranteed_domain():
    L = _hip()
    g = torch.Generator(device="cpu").manual_seed(1)
    R, Cc = 300, 256
    mant = torch.randn(R, Cc, generator=g)
    expo = torch.randint(-100, 126, (R, Cc), generator=g).float()
    X = (mant * torch.exp2(expo)).clamp(-3.38e38, 3.38e38).to(DEV)
    X[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.38e38, -3.38e38, 2.0 ** -100, -(2.0 ** -100)], device=DEV)
    P = L.p3_split(X)
    assert P.shape == (R, Cc // 32, 3, 32)
    assert torch.equal(L.p3_to_float(P), X)
    # a strided view (column slice of a wider matrix) splits like its contiguous copy
    W = torch.randn(64, 512, device=DEV)
    assert torch.equal(L.p3_split(W[:, 128:384]), L.p3_split(W[:, 128:384].contiguous()))


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 0)])
def test_identity_product_returns_the_operand_bit_for_bit(akc, bkc):
    L = _hip()
    M = N = K = 512
    g = torch.Generator(device="cpu").manual_seed(3)
    X = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-60, 60, (M, K), generator=g).float())).to(DEV)
    eye = torch.eye(K, device=DEV)
    ws = torch.empty(1 << 22, device=DEV)
    A = X if akc else X.t().contiguous()
    C = torch.full((M, N), float("nan"), device=DEV)
    L.gemm_p3(akc, bkc, M, N, K, L.p3_split(A), L.p3_split(eye), C, N, ws=ws)
    assert torch.equal(C, X), f"A operand not reproduced: {int((C != X).sum())} of {C.numel()} differ"
    Bm = X.t().contiguous() if bkc else X
    Ai = eye
    C = torch.full((M, N), float("nan"), device=DEV)
    L.gemm_p3(akc, bkc, M, N, K, L.p3_split(Ai), L.p3_split(Bm), C, N, ws=ws)
    assert torch.equal(C, X), f"B operand not reproduced: {int((C != X).sum())} of {C.numel()} differ"


@pytest.mark.parametrize("kind", ["normal", "wide", "positive"])
@pytest.mark.parametrize("akc,bkc,M,N,K", [(1, 1, 4096, 512, 512), (1, 1, 2048, 512, 2048), (1, 0, 4096, 512, 2048), (1, 1, 512, 128, 32),
                                            (0, 0, 2048, 512, 16384), (0, 0, 512, 512, 4096), (1, 0, 1024, 256, 64), (1, 1, 256, 128, 96)])
def test_p3_is_as_close_to_fp64_as_the_fp32_mfma(kind, akc, bkc, M, N, K):
    L = _hip()
    A = _rand((M, K) if akc else (K, M), M + N + K, kind)
    B = _rand((N, K) if bkc else (K, N), M + N + K + 1, kind) * (1e-6 if kind == "wide" else 1.0)
    ref = (A if akc else A.t()).double() @ (B.t() if bkc else B).double()
    ws = torch.empty(1 << 24, device=DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    L.gemm_p3(akc, bkc, M, N, K, L.p3_split(A), L.p3_split(B), C, N, ws=ws)
    old = L.lib().vqh_gemm_set_flags(1)
    L.lib().vqh_gemm_set_flags(old | NATIVE)
    try:
        Cn = torch.full((M, N), float("nan"), device=DEV)
        L.gemm(akc, bkc, M, N, K, A, A.stride(0), B, B.stride(0), Cn, N, ws=ws)
    finally:
        L.lib().vqh_gemm_set_flags(old)
    e_p3, e_nat = _rms(C, ref), _rms(Cn, ref)
    assert e_nat < 5e-6, e_nat
    assert e_p3 <= max(1.25 * e_nat, 2.0 ** -23), (e_p3, e_nat)
    absdot = (A if akc else A.t()).double().abs() @ (B.t() if bkc else B).double().abs()
    assert float(((C.double() - ref).abs() / absdot).max()) < 3e-6


MODES = dict(LINEAR=0, RELU_DROP=1, GELU=2, DROP_RESID=3, SIGMOID=4, MUL_POSMASK=5, MUL_GELUGRAD=6, MUL_SIGGRAD=7)


@pytest.mark.parametrize("mode", ["LINEAR", "RELU_DROP", "GELU", "DROP_RESID", "SIGMOID", "MUL_GELUGRAD", "MUL_SIGGRAD"])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_epilogues_agree_with_the_fp32_entry_point(mode, p):
    """Same epilogue arithmetic and the SAME dropout mask as vqh_gemm (the keep/drop hash is a function of the element index);
    the plane output reproduces the fp32 output bit for bit."""
    L = _hip()
    md = MODES[mode]
    akc, bkc = (1, 1) if md <= 4 else (1, 0)
    M, N, K = 1024, 512, 256
    A = _rand((M, K), 11)
    B = _rand((N, K) if bkc else (K, N), 12) * 0.05
    bias = _rand((N,), 13) if md <= 4 else None
    aux = _rand((M, N), 14)
    if mode == "MUL_SIGGRAD":
        aux = torch.sigmoid(aux)
    rng = torch.tensor([1234, 7], device=DEV, dtype=torch.int64)
    ws = torch.empty(1 << 22, device=DEV)
    if p > 0.0 and mode not in ("RELU_DROP", "DROP_RESID"):
        pytest.skip("no dropout in this epilogue")
    C0, a0 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
    L.gemm(akc, bkc, M, N, K, A, K, B, B.stride(0), C0, N, bias=bias, mode=md, aux_in=aux if md in (3, 6, 7) else None,
           aux_out=a0 if md == 2 else None, ldaux=N, rng=rng, site=5, p=p, ws=ws)
    C1, a1 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
    Cp = L.p3_empty(M, N, DEV)
    bits = torch.zeros(M * N // 32, device=DEV, dtype=torch.int32)
    L.gemm_p3(akc, bkc, M, N, K, L.p3_split(A), L.p3_split(B), C1, N, Cp=Cp, bias=bias, mode=md, aux_in=aux if md in (3, 6, 7) else None,
              aux_out=a1 if md == 2 else None, ldaux=N, sign_bits=bits if md == 1 else None, rng=rng, site=5, p=p, ws=ws)
    assert torch.equal(C1 == 0, C0 == 0) or mode not in ("RELU_DROP",), "dropout / ReLU masks differ"
    scale = float(C0.abs().max())
    assert float((C1 - C0).abs().max()) <= 2e-6 * scale, float((C1 - C0).abs().max()) / scale
    if md == 2:
        assert float((a1 - a0).abs().max()) <= 2e-6 * float(a0.abs().max())
    assert torch.equal(L.p3_to_float(Cp), C1), "plane output != fp32 output"
    if p > 0.0:
        drop_frac = float((C1 == 0).float().mean()) if mode == "RELU_DROP" else None
        if drop_frac is not None:
            assert 0.5 < drop_frac < 0.6            # relu zeroes ~half, dropout 10 % of the rest


def test_relu_sign_bits_drive_the_posmask_backward():
    """FFN backward: d(pre-activation) = (dY . W2) * (f1 > 0 ? 1 / (1 - p) : 0).  The fp32 path reads the saved activation f1; the
    plane path reads the sign bits the forward epilogue wrote (same tile geometry in both GEMMs)."""
    L = _hip()
    rows, H, F, p = 1024, 256, 512, 0.1
    X, W1, b1 = _rand((rows, H), 21), _rand((F, H), 22) * 0.1, _rand((F,), 23) * 0.1
    dY, W2 = _rand((rows, H), 24), _rand((H, F), 25) * 0.1
    rng = torch.tensor([99, 3], device=DEV, dtype=torch.int64)
    ws = torch.empty(1 << 22, device=DEV)
    f1 = torch.empty(rows, F, device=DEV)
    L.gemm(1, 1, rows, F, H, X, H, W1, H, f1, F, bias=b1, mode=1, rng=rng, site=9, p=p, ws=ws)
    dpre0 = torch.empty(rows, F, device=DEV)
    L.gemm(1, 0, rows, F, H, dY, H, W2, F, dpre0, F, mode=5, aux_in=f1, ldaux=F, p=p, ws=ws)
    f1p = L.p3_empty(rows, F, DEV)
    bits = torch.zeros(rows * F // 32, device=DEV, dtype=torch.int32)
    L.gemm_p3(1, 1, rows, F, H, L.p3_split(X), L.p3_split(W1), None, 0, Cp=f1p, bias=b1, mode=1, sign_bits=bits, rng=rng, site=9, p=p, ws=ws)
    assert torch.equal(L.p3_to_float(f1p) > 0, f1 > 0)
    dprep = L.p3_empty(rows, F, DEV)
    L.gemm_p3(1, 0, rows, F, H, L.p3_split(dY), L.p3_split(W2), None, 0, Cp=dprep, mode=5, sign_bits=bits, p=p, ws=ws)
    got = L.p3_to_float(dprep)
    assert torch.equal(got == 0, dpre0 == 0)
    assert float((got - dpre0).abs().max()) <= 2e-6 * float(dpre0.abs().max())


def test_grouped_weight_gradient_on_planes():
    L = _hip()
    rows = 4096
    shapes = [(512, 512), (1536, 512), (2048, 512), (512, 2048)]
    ws = torch.empty(48 << 20, device=DEV)
    items32, itemsp3, outs = [], [], []
    for i, (n_out, k_in) in enumerate(shapes):
        dY, X = _rand((rows, n_out), 40 + i), _rand((rows, k_in), 50 + i)
        g0, b0 = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
        g1, b1 = torch.full((n_out, k_in), float("nan"), device=DEV), torch.full((n_out,), float("nan"), device=DEV)
        items32.append((dY, n_out, X, k_in, rows, g0, b0))
        dYp, Xp = L.p3_split(dY), L.p3_split(X)
        itemsp3.append((dYp, L.p3_pitch(dYp), Xp, L.p3_pitch(Xp), rows, g1, b1))
        ref = dY.double().t() @ X.double()
        outs.append((g0, b0, g1, b1, ref, dY.double().sum(0), dYp, Xp))
    L.wgrad_group(items32, ws)
    L.wgrad_group_p3(itemsp3, ws)
    torch.cuda.synchronize()
    for g0, b0, g1, b1, ref, bref, _, _ in outs:
        assert _rms(g1, ref) <= max(1.25 * _rms(g0, ref), 2.0 ** -23)
        assert float((b1.double() - bref).abs().max()) <= 2e-6 * float(bref.abs().max()) + 1e-4
        assert float((b0 - b1).abs().max()) <= 3e-6 * float(b0.abs().max()) + 1e-4


def test_p3_small_magnitudes_and_ineligible_shapes():
    L = _hip()
    assert L.gemm_p3_eligible(256, 128, 32) and not L.gemm_p3_eligible(128, 128, 32) and not L.gemm_p3_eligible(256, 64, 32)
    assert not L.gemm_p3_eligible(256, 128, 48)
    M, N, K = 512, 256, 256
    B = _rand((N, K), 61)
    for log2s, bound in ((-100, None), (-108, 2.0 ** -8), (-120, 2.0 ** -8)):
        A = _rand((M, K), 60) * 2.0 ** log2s
        ref = A.double() @ B.double().t()
        mag = A.double().abs() @ B.double().abs().t()
        C = torch.empty(M, N, device=DEV)
        L.gemm_p3(1, 1, M, N, K, L.p3_split(A), L.p3_split(B), C, N)
        e = float(((C.double() - ref).abs() / mag).max())
        assert torch.isfinite(C).all()
        assert e <= (bound if bound is not None else 2.0 ** -21), (log2s, e)
    with pytest.raises(L.VqhError):
        L.gemm_p3(1, 1, 128, 128, 32, L.p3_split(_rand((128, 32), 1)), L.p3_split(_rand((128, 32), 2)), torch.empty(128, 128, device=DEV), 128)


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(512, 256, 256), (1024, 512, 1280), (256, 256, 4096)])
def test_gemm_p3_on_stage_images_equals_plane_tensors(akc, bkc, M, N, K):
    """Operands as STAGE IMAGES (pitch 0: 256-row tiles of [16 rows][64 B] blocks, every LDS-DMA instruction reads 1 KB of
    consecutive memory) give bit-identical results to the row-pitched plane tensors: same values, same order of operations.
    All three operand layouts of the step (forward, dgrad, weight gradient), with and without split-K."""
    L = _hip()
    g = torch.Generator(device="cpu").manual_seed(M + N + K + akc * 2 + bkc)
    A = torch.randn((M, K) if akc else (K, M), generator=g).to(DEV)
    B = (torch.randn((N, K) if bkc else (K, N), generator=g) / K ** 0.5).to(DEV)
    ws = torch.empty(8 << 20, device=DEV)
    Ap, Bp = L.p3_split(A), L.p3_split(B)
    Ai, Bi = L.p3_image(A), L.p3_image(B)
    want = torch.empty(M, N, device=DEV)
    L.gemm_p3(akc, bkc, M, N, K, Ap, Bp, want, N, ws=ws)
    for (a, pa), (b, pb) in (((Ai, 0), (Bi, 0)), ((Ai, 0), (Bp, None)), ((Ap, None), (Bi, 0))):
        got = torch.full((M, N), float("nan"), device=DEV)
        L.gemm_p3(akc, bkc, M, N, K, a, b, got, N, ws=ws, pitch_a=pa, pitch_b=pb)
        torch.cuda.synchronize()
        assert torch.equal(got, want), float((got - want).abs().max())
    ref = (A.double() if akc else A.double().t()) @ (B.double().t() if bkc else B.double())
    assert float((want.double() - ref).abs().max() / ref.abs().max()) < 2e-6
