# -*- coding: utf-8 -*-
"""ctypes binding of libvqvae_hip.so (the C ABI declared in include/vqvae_hip.h).

There is NO fallback: if the shared library is missing, or no GPU is present when a kernel is
called, callers get a loud VqhError.  PyTorch is used only for device memory (tensors), the current
HIP stream and torch.distributed; every arithmetic kernel of the training step lives in the library."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqvae_hip.so")
_lib = None


class VqhError(RuntimeError):
    pass


# i=int  f=float  p=pointer  l=long long  u=unsigned
_PROTOS = {
    "vqh_rng_advance": "pp",
    "vqh_memset": "pilp",
    "vqh_gemm": "iiiiipipipipippifpufplp",
    "vqh_gemm_wgrad": "iiipipipipfplp",
    "vqh_gemm_wgrad_group": "ipplp",
    "vqh_p3_split": "pipliip",
    "vqh_p3_split_multi": "ipp",
    "vqh_gemm_p3": "iiiiiplplpiplpippipfpufplp",
    "vqh_gemm_p3_wgrad_group": "ipplp",
    "vqh_layernorm_fwd": "pipppippiifp",
    "vqh_layernorm_bwd": "pipippppiippfiippufplp",
    "vqh_reduce_slabs": "pillpfp",
    "vqh_colsum": "piiipfplp",
    "vqh_embed_fwd": "piippppiiipufp",
    "vqh_embed_bwd": "ppiippfiipufplp",
    "vqh_bcast_rows": "pppilp",
    "vqh_dropout_bwd": "pplpufp",
    "vqh_add": "ppplp",
    "vqh_copy2d": "pipiiip",
    "vqh_sigmoid_bwd": "ppplp",
    "vqh_augment": "pppppiip",
    "vqh_softmax_rows": "pipfiip",
    "vqh_softmax_bwd_colgrad": "pipiip",
    "vqh_usage_entropy_finish": "piifppiip",
    "vqh_vq_mix": "pppfplp",
    "vqh_attn_fwd": "pipipipippiiiiiipufp",
    "vqh_attn_bwd": "pipipipippippipipipiiiiiipufp",
    "vqh_vq_nearest": "pipipiiiifplp",
    "vqh_vq_gather": "pipipippiip",
    "vqh_vq_finish": "pipippiip",
    "vqh_vq_mask_ids": "pppip",
    "vqh_vq_segment_sum": "pipiiiippplp",
    "vqh_vq_ema_apply": "pppppiifffp",
    "vqh_vq_usage_stats": "pifpppp",
    "vqh_row_sqnorm": "piiipfp",
    "vqh_vq_reinit": "pfppipppiip",
    "vqh_loss_fwd_bwd": "pppipppiiipiiippppppplp",
    "vqh_grad_norm": "plpppp",
    "vqh_adamw_step": "pppplppp",
}
_CT = {"i": C.c_int, "f": C.c_float, "p": C.c_void_p, "l": C.c_longlong, "u": C.c_uint}
EXPORTS = ["vqh_last_error", "vqh_abi_version", "vqh_gemm_p3_eligible", "vqh_vq_nearest_form", "vqh_vq_nearest_workspace", "vqh_gemm_set_flags", "vqh_attn_set_flags", "vqh_gemm_profile_begin", "vqh_gemm_profile_end",
           "vqh_vq_set_flags", "vqh_vq_profile_begin", "vqh_vq_profile_end"] + list(_PROTOS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise VqhError(f"HIP extension not built: {LIB_PATH} is missing "
                           f"(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                           f"There is no CPU fallback for the product path.")
        L = C.CDLL(LIB_PATH)
        L.vqh_last_error.restype = C.c_char_p
        L.vqh_abi_version.restype = C.c_int
        L.vqh_gemm_p3_eligible.argtypes = [C.c_int, C.c_int, C.c_int]
        L.vqh_gemm_p3_eligible.restype = C.c_int
        L.vqh_vq_nearest_form.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong]
        L.vqh_vq_nearest_form.restype = C.c_int
        L.vqh_vq_nearest_workspace.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
        L.vqh_vq_nearest_workspace.restype = C.c_int
        L.vqh_gemm_profile_end.argtypes = [C.c_void_p]
        L.vqh_vq_profile_end.argtypes = [C.c_void_p]
        for name, sig in _PROTOS.items():
            fn = getattr(L, name)
            fn.argtypes = [_CT[c] for c in sig]
            fn.restype = C.c_int
        # process-wide A/B switches of the kernel families (see include/vqvae_hip.h), e.g. VQH_GEMM_FLAGS=513 runs every
        # large GEMM tile on the native fp32 MFMA instead of the split-operand bf16 tiles, VQH_VQ_FLAGS=2 the same for VQ scores
        # The bits that produce WRONG results (timing-only: skip stores / loads) are refused here and masked off by the
        # product build of the library itself (-DVQH_DIAG labs only); every override is announced once on stderr.
        for env, setter, bad in (("VQH_GEMM_FLAGS", "vqh_gemm_set_flags", 2 | 4), ("VQH_VQ_FLAGS", "vqh_vq_set_flags", 0),
                                 ("VQH_ATTN_FLAGS", "vqh_attn_set_flags", 0)):
            if os.environ.get(env):
                val = int(os.environ[env])
                if val & bad:
                    raise VqhError(f"{env}={val}: bits {val & bad} are result-corrupting timing diagnostics, not available "
                                   f"in the product library")
                getattr(L, setter)(val)
                import sys
                print(f"[vqvae_hip] {env}={val} applied ({setter}): kernel A/B override for this process", file=sys.stderr)
        _lib = L
    return _lib


def vq_nearest_workspace(R, K, D):
    """Workspace floats with which vqh_vq_nearest takes its fastest form for the shape."""
    out = C.c_longlong(0)
    rc = lib().vqh_vq_nearest_workspace(int(R), int(K), int(D), C.byref(out))
    if rc != 0:
        raise VqhError(f"vqh_vq_nearest_workspace failed (rc={rc}): {lib().vqh_last_error().decode()}")
    return int(out.value)


def require_gpu():
    if not torch.cuda.is_available():
        raise VqhError("vqvae_hip needs an MI355X (gfx950) GPU: the product path has no CPU fallback")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an entry point on the current torch stream (appended as the last argument)."""
    conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
    rc = getattr(lib(), name)(*conv, _stream())
    if rc != 0:
        raise VqhError(f"{name} failed (rc={rc}): {lib().vqh_last_error().decode()}")


# epilogue modes of vqh_gemm
EPI_LINEAR, EPI_RELU_DROP, EPI_GELU, EPI_DROP_RESID, EPI_SIGMOID, EPI_MUL_POSMASK, EPI_MUL_GELUGRAD, EPI_MUL_SIGGRAD = range(8)


def gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, Cout, ldc, bias=None, mode=EPI_LINEAR, aux_in=None, aux_out=None,
         ldaux=0, beta=0.0, rng=None, site=0, p=0.0, ws=None):
    _gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, Cout, ldc, bias, mode, aux_in, aux_out, ldaux, beta, rng, site, p, ws)


GEMM_PROF_FAMILIES = 7
GEMM_FLAG_NATIVE_F32 = 512          # vqh_gemm_set_flags bit: large tiles on v_mfma_f32_32x32x2_f32 instead of the bf16x3 split


def gemm_profile(fn):
    """Run fn() with the library's per-launch GEMM timing on; returns {(family, a_kc, b_kc, MODE): (launches, seconds, flops)}
    keyed like the kernels' template arguments: family 0 = gemm_f32_mfma<a_kc, b_kc, 32, MODE> (128x128 tile),
    family 1 = gemm_f32_dma<a_kc, b_kc, MODE> (256x128 tile, LDS-DMA, native fp32 MFMA), family 2 = gemm_f32_dma_group (grouped
    weight gradients), families 3 / 4 = gemm_f32_x3<...> / gemm_f32_x3_group: the same tiles on the bf16 matrix pipes (exact 3-way
    operand split, 6 products; the default)."""
    import ctypes
    lib().vqh_gemm_profile_begin()
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        out = (ctypes.c_double * (GEMM_PROF_FAMILIES * 4 * 9 * 3))()
        rc = lib().vqh_gemm_profile_end(ctypes.cast(out, ctypes.c_void_p))
    if rc != 0:
        raise VqhError(f"vqh_gemm_profile_end failed: {lib().vqh_last_error().decode()}")
    res = {}
    for fam in range(GEMM_PROF_FAMILIES):
        for lay in range(4):
            for mc in range(9):
                o = ((fam * 4 + lay) * 9 + mc) * 3
                n, t, f = out[o:o + 3]
                if n > 0:
                    res[(fam, lay >> 1, lay & 1, mc - 1)] = (int(n), t, f)
    return res


def vq_profile(fn):
    """Run fn() with per-launch timing of the nearest-neighbour main kernel on; returns (launches, seconds, flops)."""
    import ctypes
    lib().vqh_vq_profile_begin()
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        out = (ctypes.c_double * 3)()
        rc = lib().vqh_vq_profile_end(ctypes.cast(out, ctypes.c_void_p))
    if rc != 0:
        raise VqhError(f"vqh_vq_profile_end failed: {lib().vqh_last_error().decode()}")
    return int(out[0]), out[1], out[2]


def gemm_kernel_name(key):
    fam, a_kc, b_kc, mode = key
    tf = lambda b: "true" if b else "false"
    if fam == 0:
        return "gemm_f32_mfma<%s, %s, 32, %d>" % (tf(a_kc), tf(b_kc), mode)
    if fam == 2:
        return "gemm_f32_dma_group"
    if fam == 4:
        return "gemm_f32_x3_group"
    if fam == 3:
        return "gemm_f32_x3<%s, %s, %d>" % (tf(a_kc), tf(b_kc), mode)
    return "gemm_f32_dma<%s, %s, %d>" % (tf(a_kc), tf(b_kc), mode)


def gemm_kernel_is_x3(key):
    return key[0] in (3, 4)


class WgradT(C.Structure):
    """vqh_wgrad_t of include/vqvae_hip.h"""
    _fields_ = [("rows", C.c_int), ("n_out", C.c_int), ("k_in", C.c_int), ("dY", C.c_void_p), ("lddy", C.c_int),
                ("X", C.c_void_p), ("ldx", C.c_int), ("dW", C.c_void_p), ("lddw", C.c_int), ("db", C.c_void_p)]


def wgrad_group(items, ws):
    """items: [(dY, lddy, X, ldx, rows, dW [n_out, k_in], db or None)] -> one grouped weight-gradient launch."""
    n = len(items)
    if n == 0:
        return
    arr = (WgradT * n)()
    for i, (dy, lddy, x, ldx, rows, gW, gb) in enumerate(items):
        n_out, k_in = gW.shape
        arr[i] = WgradT(int(rows), int(n_out), int(k_in), dy.data_ptr(), int(lddy), x.data_ptr(), int(ldx), gW.data_ptr(),
                        int(gW.stride(0)), _p(gb))
    call("vqh_gemm_wgrad_group", n, C.cast(arr, C.c_void_p), ws, ws.numel())


def _gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, Cout, ldc, bias, mode, aux_in, aux_out, ldaux, beta, rng, site, p, ws):
    call("vqh_gemm", int(a_kc), int(b_kc), M, N, K, _p(A), lda, _p(B), ldb, _p(Cout), ldc, _p(bias), mode,
         _p(aux_in), _p(aux_out), ldaux, float(beta), _p(rng), site, float(p), _p(ws),
         (ws.numel() if ws is not None else 0))


# ---------------------------------------------------------------------------------------------------------------------
# plane tensors (include/vqvae_hip.h "P3"): an fp32 matrix [R, C] as its exact 3-way bf16 split, [R][C/32][3][32] bf16
# ---------------------------------------------------------------------------------------------------------------------
class P3ItemT(C.Structure):
    """vqh_p3_item_t"""
    _fields_ = [("X", C.c_void_p), ("P", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ldx", C.c_longlong),
                ("pitch_bytes", C.c_longlong)]


class WgradP3T(C.Structure):
    """vqh_wgrad_p3_t"""
    _fields_ = [("rows", C.c_int), ("n_out", C.c_int), ("k_in", C.c_int), ("dYp", C.c_void_p), ("pitch_dy", C.c_longlong),
                ("Xp", C.c_void_p), ("pitch_x", C.c_longlong), ("dW", C.c_void_p), ("lddw", C.c_int), ("db", C.c_void_p)]


def p3_empty(rows, cols, device):
    """Uninitialised plane tensor for an fp32 [rows, cols] matrix (cols % 32 == 0): int16 [rows, cols // 32, 3, 32]."""
    if cols % 32:
        raise VqhError(f"plane tensors need cols % 32 == 0 (got {cols})")
    return torch.empty(rows, cols // 32, 3, 32, device=device, dtype=torch.int16)


def p3_pitch(P):
    return int(P.stride(0)) * 2


def p3_split(X, P=None):
    """fp32 matrix (2-D, row stride % 4 == 0) -> its plane tensor (generic converter kernel)."""
    rows, cols = X.shape
    if P is None:
        P = p3_empty(rows, cols, X.device)
    call("vqh_p3_split", X, int(X.stride(0)), P, p3_pitch(P), int(rows), int(cols))
    return P


def p3_image(X, P=None):
    """fp32 matrix [rows % 256 == 0, cols % 32 == 0] -> its STAGE IMAGES (pitch 0 in the vqh_gemm_p3* calls): int16 tensor of
    rows * cols * 3 elements laid out [rows / 256][cols / 32][16 row blocks][3 planes][64 lanes x 16 B]."""
    rows, cols = X.shape
    if rows % 256 or cols % 32:
        raise VqhError(f"stage images need rows % 256 == 0 and cols % 32 == 0 (got {rows} x {cols})")
    if P is None:
        P = torch.empty(rows * cols * 3, device=X.device, dtype=torch.int16)
    call("vqh_p3_split", X, int(X.stride(0)), P, 0, int(rows), int(cols))
    return P


def p3_to_float(P):
    """Plane tensor -> fp32 matrix h + m + l (host-side check helper; exact)."""
    f = (P.to(torch.int32) << 16).view(torch.float32)          # bf16 bits -> fp32
    rows, nb = P.shape[0], P.shape[1]
    return (f[:, :, 0] + f[:, :, 1] + f[:, :, 2]).reshape(rows, nb * 32)


def gemm_p3_eligible(M, N, K):
    return bool(lib().vqh_gemm_p3_eligible(int(M), int(N), int(K)))


def gemm_p3(a_kc, b_kc, M, N, K, Ap, Bp, Cout=None, ldc=0, Cp=None, bias=None, mode=EPI_LINEAR, aux_in=None, aux_out=None,
            ldaux=0, sign_bits=None, beta=0.0, rng=None, site=0, p=0.0, ws=None, pitch_a=None, pitch_b=None, pitch_c=None):
    call("vqh_gemm_p3", int(a_kc), int(b_kc), M, N, K, Ap, pitch_a if pitch_a is not None else p3_pitch(Ap), Bp,
         pitch_b if pitch_b is not None else p3_pitch(Bp), _p(Cout), int(ldc), _p(Cp),
         (pitch_c if pitch_c is not None else (p3_pitch(Cp) if Cp is not None else 0)), _p(bias), mode, _p(aux_in), _p(aux_out),
         int(ldaux), _p(sign_bits), float(beta), _p(rng), site, float(p), _p(ws), (ws.numel() if ws is not None else 0))


def wgrad_group_p3(items, ws):
    """items: [(dYp, pitch_dy, Xp, pitch_x, rows, dW [n_out, k_in], db or None)] -> one grouped launch on plane operands."""
    n = len(items)
    if n == 0:
        return
    arr = (WgradP3T * n)()
    for i, (dyp, pdy, xp, px, rows, gW, gb) in enumerate(items):
        n_out, k_in = gW.shape
        arr[i] = WgradP3T(int(rows), int(n_out), int(k_in), dyp.data_ptr() if torch.is_tensor(dyp) else int(dyp), int(pdy),
                          xp.data_ptr() if torch.is_tensor(xp) else int(xp), int(px), gW.data_ptr(), int(gW.stride(0)), _p(gb))
    call("vqh_gemm_p3_wgrad_group", n, C.cast(arr, C.c_void_p), ws, ws.numel())
