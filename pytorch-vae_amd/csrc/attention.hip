// Multi-head attention forward / backward in fp32 on v_mfma_f32_32x32x2_f32 (gfx950), flash style:
// the [T,S] score matrix is never written to memory; backward recomputes P from Q, K and the saved
// log-sum-exp.  Restates torch.nn.MultiheadAttention's core as the reference uses it
// (/root/reference/models/vq_vae.py:300,319 tokenizer cross-attention; nn.TransformerEncoderLayer /
// DecoderLayer self- and cross-attention built at :458-473, :525-528): scores scaled by 1/sqrt(dh),
// additive -inf key-padding mask, softmax over keys, dropout on the probabilities, P.V.
//
// One wave = one (batch, head, 32-row tile).  Orientation trick: the score tile is computed
// TRANSPOSED, S^T[key][query] = K.Q^T, so that in the MFMA accumulator a lane owns one QUERY column
// and its registers run over KEYS.  Then
//   * softmax max / sum over keys are register reductions + one cross-half shuffle,
//   * the probabilities are already laid out as the B operand (k = key) of the next product
//     O^T[d][query] += V^T[d][key] . P^T[key][query]  -- no LDS, no lane movement.
// Operands come straight from global/L2 (tiles are a few KB and shared by neighbouring waves);
// the 16-byte fragment loads use the same k-permutation for both operands (see gemm.hip).
#include "common.h"

namespace {

__device__ __forceinline__ int kmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x4 ld4_guard(const float* p, bool ok) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? *reinterpret_cast<const f32x4*>(p) : z;
}

struct AttnArgs {
    const float* Q; int ldq;
    const float* K; int ldk;
    const float* V; int ldv;
    float* O; int ldo;            // fwd out / bwd in
    float* LSE;                   // [B, nh, T]
    const unsigned char* kvalid;  // [B, S] 1 = attend, 0 = padded key; null = all valid
    int B, nh, T, S;
    float scale;
    DropCfg drop;
    // backward
    const float* dO; int lddo;
    float* Dsum;                  // [B, nh, T] rowsum(dO * O)
    float* dQ; int lddq;
    float* dK; int lddk;
    float* dV; int lddv;
};

template <int DH>
__global__ __launch_bounds__(64, 2) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, hh = blockIdx.y, b = blockIdx.z;
    const int q = q0 + l31;
    const float* Qp = a.Q + ((size_t)b * a.T + q) * a.ldq + hh * DH;
    const float* Kb = a.K + (size_t)b * a.S * a.ldk + hh * DH;
    const float* Vb = a.V + (size_t)b * a.S * a.ldv + hh * DH;
    const unsigned char* kv = a.kvalid ? a.kvalid + (size_t)b * a.S : nullptr;
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }

    f32x4 qf[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        qf[t] = ld4_guard(Qp + 8 * t + 4 * h, q < a.T);
        qf[t] *= a.scale;
    }
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    for (int s0 = 0; s0 < a.S; s0 += 32) {
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const int key_l = s0 + l31;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const f32x4 kf = ld4_guard(Kb + (size_t)key_l * a.ldk + 8 * t + 4 * h, key_l < a.S);
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], sacc, 0, 0, 0);
        }
        float mx = -INFINITY;
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = s0 + kmap(r, h);
            ok[r] = (key < a.S) && (!kv || kv[key]);
            if (ok[r]) mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float corr = (m_new == -INFINITY) ? 1.f : __expf(m - m_new);
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = ok[r] ? __expf(sacc[r] - m_new) : 0.f;
            psum += p[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        lsum = lsum * corr + psum;
        m = m_new;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= corr;
        if (a.drop.p > 0.f) {
            const unsigned long long base = (((unsigned long long)b * a.nh + hh) * a.T + q) * (unsigned long long)a.S;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] *= drop1(a.drop, seed, step, base + (unsigned long long)(s0 + kmap(r, h)));
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int dcol = d * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = s0 + kmap(r, h);
                const float vv = (key < a.S && dcol < DH) ? Vb[(size_t)key * a.ldv + dcol] : 0.f;
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, p[r], o[d], 0, 0, 0);
            }
        }
    }
    if (q < a.T) {
        const float inv = 1.f / lsum;
        float* Op = a.O + ((size_t)b * a.T + q) * a.ldo + hh * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = d * 32 + kmap(r, h);
                if (dd < DH) Op[dd] = o[d][r] * inv;
            }
        if (h == 0 && a.LSE) a.LSE[((size_t)b * a.nh + hh) * a.T + q] = m + __logf(lsum);
    }
}

// dQ (and D = rowsum(dO*O)) : one wave per 32 queries, loop over key tiles.
template <int DH>
__global__ __launch_bounds__(64, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, hh = blockIdx.y, b = blockIdx.z;
    const int q = q0 + l31;
    const bool qok = q < a.T;
    const float* Qp = a.Q + ((size_t)b * a.T + q) * a.ldq + hh * DH;
    const float* dOp = a.dO + ((size_t)b * a.T + q) * a.lddo + hh * DH;
    const float* Op = a.O + ((size_t)b * a.T + q) * a.ldo + hh * DH;
    const float* Kb = a.K + (size_t)b * a.S * a.ldk + hh * DH;
    const float* Vb = a.V + (size_t)b * a.S * a.ldv + hh * DH;
    const unsigned char* kv = a.kvalid ? a.kvalid + (size_t)b * a.S : nullptr;
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }

    f32x4 qf[NG], dof[NG];
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        qf[t] = ld4_guard(Qp + 8 * t + 4 * h, qok);
        qf[t] *= a.scale;
        dof[t] = ld4_guard(dOp + 8 * t + 4 * h, qok);
        const f32x4 of = ld4_guard(Op + 8 * t + 4 * h, qok);
        dsum += dof[t][0] * of[0] + dof[t][1] * of[1] + dof[t][2] * of[2] + dof[t][3] * of[3];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const size_t rowid = ((size_t)b * a.nh + hh) * a.T + q;
    if (qok && h == 0) a.Dsum[rowid] = dsum;
    const float lse = qok ? a.LSE[rowid] : 0.f;

    f32x16 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    for (int s0 = 0; s0 < a.S; s0 += 32) {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        const int key_l = s0 + l31;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const f32x4 kf = ld4_guard(Kb + (size_t)key_l * a.ldk + 8 * t + 4 * h, key_l < a.S);
            const f32x4 vf = ld4_guard(Vb + (size_t)key_l * a.ldv + 8 * t + 4 * h, key_l < a.S);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], dof[t][j], dpacc, 0, 0, 0);
            }
        }
        float ds[16];
        const unsigned long long base = (((unsigned long long)b * a.nh + hh) * a.T + q) * (unsigned long long)a.S;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = s0 + kmap(r, h);
            const bool ok = (key < a.S) && (!kv || kv[key]);
            const float p = ok ? __expf(sacc[r] - lse) : 0.f;
            float keep = 1.f;
            if (a.drop.p > 0.f) keep = drop1(a.drop, seed, step, base + (unsigned long long)key);
            ds[r] = p * (dpacc[r] * keep - dsum);
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int dcol = d * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = s0 + kmap(r, h);
                const float kk = (key < a.S && dcol < DH) ? Kb[(size_t)key * a.ldk + dcol] : 0.f;
                dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, ds[r], dq[d], 0, 0, 0);
            }
        }
    }
    if (qok) {
        float* dQp = a.dQ + ((size_t)b * a.T + q) * a.lddq + hh * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = d * 32 + kmap(r, h);
                if (dd < DH) dQp[dd] = dq[d][r] * a.scale;
            }
    }
}

// dK, dV : one wave per 32 keys, loop over query tiles (needs Dsum from the dQ kernel).
template <int DH>
__global__ __launch_bounds__(64, 1) void attn_bwd_dkv_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int s0 = blockIdx.x * 32, hh = blockIdx.y, b = blockIdx.z;
    const int key = s0 + l31;
    const float* Qb = a.Q + (size_t)b * a.T * a.ldq + hh * DH;
    const float* dOb = a.dO + (size_t)b * a.T * a.lddo + hh * DH;
    const float* Kp = a.K + ((size_t)b * a.S + key) * a.ldk + hh * DH;
    const float* Vp = a.V + ((size_t)b * a.S + key) * a.ldv + hh * DH;
    const bool kin = key < a.S;
    const bool kok = kin && (!a.kvalid || a.kvalid[(size_t)b * a.S + key]);
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }

    f32x4 kf[NG], vf[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        kf[t] = ld4_guard(Kp + 8 * t + 4 * h, kin);
        vf[t] = ld4_guard(Vp + 8 * t + 4 * h, kin);
    }
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

    const size_t rowbase = ((size_t)b * a.nh + hh) * a.T;
    for (int q0 = 0; q0 < a.T; q0 += 32) {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
        const int q_l = q0 + l31;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            f32x4 qf = ld4_guard(Qb + (size_t)q_l * a.ldq + 8 * t + 4 * h, q_l < a.T);
            qf *= a.scale;
            const f32x4 dof = ld4_guard(dOb + (size_t)q_l * a.lddo + 8 * t + 4 * h, q_l < a.T);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j], kf[t][j], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[j], vf[t][j], dpacc, 0, 0, 0);
            }
        }
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + kmap(r, h);
            const bool ok = kok && (q < a.T);
            float lse = 0.f, dsum = 0.f;
            if (q < a.T) { lse = a.LSE[rowbase + q]; dsum = a.Dsum[rowbase + q]; }
            const float p = ok ? __expf(sacc[r] - lse) : 0.f;
            float keep = 1.f;
            if (a.drop.p > 0.f && ok)
                keep = drop1(a.drop, seed, step, ((unsigned long long)(rowbase + q)) * (unsigned long long)a.S + key);
            pd[r] = p * keep;
            ds[r] = p * (dpacc[r] * keep - dsum);
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int dcol = d * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + kmap(r, h);
                const bool in = (q < a.T) && (dcol < DH);
                const float dov = in ? dOb[(size_t)q * a.lddo + dcol] : 0.f;
                const float qv = in ? Qb[(size_t)q * a.ldq + dcol] : 0.f;
                dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(dov, pd[r], dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, ds[r], dk[d], 0, 0, 0);
            }
        }
    }
    if (kin) {
        float* dKp = a.dK + ((size_t)b * a.S + key) * a.lddk + hh * DH;
        float* dVp = a.dV + ((size_t)b * a.S + key) * a.lddv + hh * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = d * 32 + kmap(r, h);
                if (dd < DH) {
                    dKp[dd] = dk[d][r] * a.scale;
                    dVp[dd] = dv[d][r];
                }
            }
    }
}

bool aligned16(const void* p, int ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0); }

}  // namespace

extern "C" int vqh_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                            int ldo, float* LSE, const unsigned char* kvalid, int B, int nh, int T, int S, int dh,
                            const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                            hipStream_t stream) {
    VQH_CHECK_ARG(B >= 0 && nh > 0 && T >= 0 && S >= 0, "vqh_attn_fwd: bad shape");
    VQH_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "vqh_attn_fwd: head dim must be 16/32/64");
    if (B == 0 || T == 0) return VQH_OK;
    VQH_CHECK_ARG(Q && K && V && O, "vqh_attn_fwd: null pointer");
    VQH_CHECK_ARG(aligned16(Q, ldq) && aligned16(K, ldk) && aligned16(V, ldv), "vqh_attn_fwd: operands must be 16-byte aligned");
    VQH_CHECK_ARG(drop_p == 0.f || rng_state, "vqh_attn_fwd: dropout needs rng_state");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE;
    a.kvalid = kvalid; a.B = B; a.nh = nh; a.T = T; a.S = S; a.scale = 1.0f / sqrtf((float)dh);
    a.drop = DropCfg{rng_state, drop_site, drop_p, 1.f / (1.f - drop_p)};
    dim3 grid((T + 31) / 32, nh, B);
    switch (dh) {
        case 16: hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, dim3(64), 0, stream, a); break;
        case 32: hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(64), 0, stream, a); break;
        default: hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(64), 0, stream, a); break;
    }
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                            const float* O, int ldo, const float* LSE, const float* dO, int lddo, float* Dsum,
                            float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                            const unsigned char* kvalid, int B, int nh, int T, int S, int dh,
                            const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                            hipStream_t stream) {
    VQH_CHECK_ARG(B >= 0 && nh > 0 && T >= 0 && S >= 0, "vqh_attn_bwd: bad shape");
    VQH_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "vqh_attn_bwd: head dim must be 16/32/64");
    if (B == 0 || T == 0 || S == 0) return VQH_OK;
    VQH_CHECK_ARG(Q && K && V && O && LSE && dO && Dsum && dQ && dK && dV, "vqh_attn_bwd: null pointer");
    VQH_CHECK_ARG(aligned16(Q, ldq) && aligned16(K, ldk) && aligned16(V, ldv) && aligned16(O, ldo) && aligned16(dO, lddo),
                  "vqh_attn_bwd: operands must be 16-byte aligned");
    VQH_CHECK_ARG(drop_p == 0.f || rng_state, "vqh_attn_bwd: dropout needs rng_state");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = const_cast<float*>(O); a.ldo = ldo;
    a.LSE = const_cast<float*>(LSE); a.kvalid = kvalid; a.B = B; a.nh = nh; a.T = T; a.S = S;
    a.scale = 1.0f / sqrtf((float)dh);
    a.drop = DropCfg{rng_state, drop_site, drop_p, 1.f / (1.f - drop_p)};
    a.dO = dO; a.lddo = lddo; a.Dsum = Dsum; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
    dim3 gq((T + 31) / 32, nh, B), gk((S + 31) / 32, nh, B);
    switch (dh) {
        case 16:
            hipLaunchKernelGGL(attn_bwd_dq_kernel<16>, gq, dim3(64), 0, stream, a);
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<16>, gk, dim3(64), 0, stream, a);
            break;
        case 32:
            hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, gq, dim3(64), 0, stream, a);
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, gk, dim3(64), 0, stream, a);
            break;
        default:
            hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(64), 0, stream, a);
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, gk, dim3(64), 0, stream, a);
            break;
    }
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
