// Multi-head attention forward / backward in fp32 on v_mfma_f32_32x32x2_f32 (gfx950), flash style:
// the [T,S] score matrix is never written to memory; backward recomputes P from Q, K and the saved
// log-sum-exp.  Restates torch.nn.MultiheadAttention's core as the reference uses it
// (/root/reference/models/vq_vae.py:300,319 tokenizer cross-attention; nn.TransformerEncoderLayer /
// DecoderLayer self- and cross-attention built at :458-473, :525-528): scores scaled by 1/sqrt(dh),
// additive -inf key-padding mask, softmax over keys, dropout on the probabilities, P.V.
//
// One wave = one (batch, head, 32-row tile); two waves per workgroup share LDS-staged operand tiles.  Orientation trick: the score tile is computed
// TRANSPOSED, S^T[key][query] = K.Q^T, so that in the MFMA accumulator a lane owns one QUERY column
// and its registers run over KEYS.  Then
//   * softmax max / sum over keys are register reductions + one cross-half shuffle,
//   * the probabilities are already laid out as the B operand (k = key) of the next product
//     O^T[d][query] += V^T[d][key] . P^T[key][query]  -- no LDS, no lane movement.
// The streamed operand (K/V, or Q/dO in the key-side backward kernel) is staged 64 rows at a time through LDS
// with coalesced 16-byte loads; the 16-byte fragment reads use the same k-permutation for both MFMA operands
// (see gemm.hip).
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
    long long qbs, kbs, vbs;      // batch strides (elements) of the Q / K / V INPUTS; 0 = one copy shared by every sample
    float scale;
    DropCfg drop;
    // backward
    const float* dO; int lddo;
    float* Dsum;                  // [B, nh, T] rowsum(dO * O)
    float* dQ; int lddq;
    float* dK; int lddk;
    float* dV; int lddv;
};

// ---- attention dropout stream ----------------------------------------------------------------------
// Attention probabilities use the counter hash of common.h keyed per row: element (row = (b*nh+head)*T + q, key)
// keeps iff its 16-bit half of mix32(rowkey(row) ^ (key/2)*GOLDEN) >= thr, with rowkey = mix32 chain over
// (seed, step, site, row).  Any kernel can evaluate any (row, key) directly, so the forward, dQ and dK/dV kernels
// regenerate identical masks whatever their register layout.
__device__ __forceinline__ unsigned attn_rowkey(const DropCfg& d, unsigned long long seed, unsigned long long step,
                                                unsigned long long row) {
    unsigned k = mix32((unsigned)seed ^ (d.site * 0x9E3779B9U));
    k = mix32(k ^ (unsigned)(seed >> 32) ^ ((unsigned)step * 0x85EBCA6BU));
    k = mix32(k ^ (unsigned)(step >> 32) ^ (unsigned)(row >> 32));
    return mix32(k ^ (unsigned)row);
}
// keys 2j and 2j+1 of a row share one hash (16-bit draws, common.h); a lane's accumulator registers 4g..4g+3 hold 4
// consecutive keys, so the query-side kernels evaluate two hashes per four probabilities
__device__ __forceinline__ float attn_keep(const DropCfg& d, unsigned rowkey, int key) {
    const unsigned bits = mix32(rowkey ^ ((unsigned)(key >> 1) * 0x9E3779B9U));
    const unsigned draw = (key & 1) ? (bits >> 16) : (bits & 0xffffu);
    return (draw >= d.thr) ? d.scale : 0.f;
}

// ---- LDS staging -------------------------------------------------------------------------------
// A block = 2 waves = 64 rows of one (batch, head).  The other operand (K/V for the query-side kernels, Q/dO
// for the key-side kernel) is staged 64 rows at a time into LDS with coalesced 16-byte loads issued in bulk,
// then every MFMA fragment comes from LDS:  [row][DH+4] floats (row stride == 4 mod 64 dwords: the 16-lane
// groups of ds_read_b128 cover all 64 banks; the transposed operand reads 32 consecutive dwords per half).
// Loads are issued as one batch per thread, branch-free: rows beyond the end are read from the last valid row (clamped
// address) and zeroed by a select.  With a guard around each load the compiler emitted one load -> s_waitcnt -> LDS store
// round trip per row group, i.e. 4-8 serial memory latencies in front of every tile.
template <int DH, int NT>
struct RowBatch {
    static constexpr int C4 = DH / 4, N = (64 * C4 + NT - 1) / NT;
    f32x4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int row0, int nrows_total, int tid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int i = tid + j * NT, r = row0 + i / C4, c = (i % C4) * 4;
            const int rr = max(min(r, nrows_total - 1), 0);
            v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)rr * ld + c);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int i = tid + j * NT, r = row0 + i / C4;
            const bool in = (r < nrows_total) && (i < 64 * C4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = in ? v[j][e] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ dst, int tid) const {
        constexpr int LD = DH + 4;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int i = tid + j * NT, r = i / C4, c = (i % C4) * 4;
            if (i < 64 * C4) *reinterpret_cast<f32x4*>(dst + r * LD + c) = v[j];
        }
    }
};

template <int DH, int NT>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, int ld, int row0,
                                           int nrows_total, int tid) {
    RowBatch<DH, NT> b;
    b.load(src, ld, row0, nrows_total, tid);
    b.store(dst, tid);
}

// two operands with the same row range (K and V, Q and dO): both batches in flight before the first LDS store
template <int DH, int NT>
__device__ __forceinline__ void stage_rows2(float* __restrict__ dst0, const float* __restrict__ src0, int ld0,
                                            float* __restrict__ dst1, const float* __restrict__ src1, int ld1, int row0,
                                            int nrows_total, int tid) {
    RowBatch<DH, NT> b0, b1;
    b0.load(src0, ld0, row0, nrows_total, tid);
    b1.load(src1, ld1, row0, nrows_total, tid);
    b0.store(dst0, tid);
    b1.store(dst1, tid);
}

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32, LD = DH + 4;
    // K/V chunks of 64 keys in LDS.  PF (4-wave workgroups = long sequences): two stages; the next chunk is fetched into
    // registers before the current one is consumed and stored behind it, one barrier per chunk instead of two.
    extern __shared__ __attribute__((aligned(16))) float kv_smem[];
    constexpr bool PF = (NW == 4);
#define KST(s_) (kv_smem + (s_) * 2 * 64 * LD)
#define VST(s_) (kv_smem + (s_) * 2 * 64 * LD + 64 * LD)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32, hh = blockIdx.y, b = blockIdx.z;
    const int q = q0 + l31;
    const bool active = q0 < a.T;
    const float* Qp = a.Q + (size_t)b * a.qbs + (size_t)q * a.ldq + hh * DH;
    const float* Kb = a.K + (size_t)b * a.kbs + hh * DH;
    const float* Vb = a.V + (size_t)b * a.vbs + hh * DH;
    const unsigned char* kv = a.kvalid ? a.kvalid + (size_t)b * a.S : nullptr;
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }

    const unsigned rowkey = (a.drop.p > 0.f) ? attn_rowkey(a.drop, seed, step, ((unsigned long long)b * a.nh + hh) * a.T + q) : 0u;
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

    RowBatch<DH, 64 * NW> kreg, vreg;
    auto gload = [&](int c0) {
        kreg.load(Kb, a.ldk, c0, a.S, tid);
        vreg.load(Vb, a.ldv, c0, a.S, tid);
    };
    auto lstore = [&](float* Kd, float* Vd) {
        kreg.store(Kd, tid);
        vreg.store(Vd, tid);
    };
    if (PF) {
        gload(0);
        lstore(KST(0), VST(0));
        __syncthreads();
    }
    for (int c0 = 0, it = 0; c0 < a.S; c0 += 64, ++it) {
        const float* Ks = KST(0);
        const float* Vs = VST(0);
        const bool more = c0 + 64 < a.S;
        if (PF) {
            Ks = KST(it & 1);
            Vs = VST(it & 1);
            if (more) gload(c0 + 64);
        } else {
            __syncthreads();
            stage_rows2<DH, 64 * NW>(KST(0), Kb, a.ldk, VST(0), Vb, a.ldv, c0, a.S, tid);
            __syncthreads();
        }
        if (active) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int s0 = c0 + sub * 32;
            if (s0 >= a.S) break;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float* kr = Ks + (sub * 32 + l31) * LD + 4 * h;
            f32x4 kfr[NG];                                  // all fragment reads first: one LDS wait per chain, not per MFMA
#pragma unroll
            for (int t = 0; t < NG; ++t) kfr[t] = *reinterpret_cast<const f32x4*>(kr + 8 * t);
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kfr[t][j], qf[t][j], sacc, 0, 0, 0);
            // validity of the tile's 32 keys as a bit mask: one coalesced byte load + ballot, no divergent branches
            const int keyl = s0 + l31;
            const unsigned char vb = (kv && keyl < a.S) ? kv[keyl] : (unsigned char)1;
            const unsigned int vmask = (unsigned int)__ballot((keyl < a.S) && vb != 0);
            float mx = -INFINITY;
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ok[r] = (vmask >> kmap(r, h)) & 1u;
                mx = ok[r] ? fmaxf(mx, sacc[r]) : mx;
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
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] *= attn_keep(a.drop, rowkey, s0 + kmap(r, h));
            }
            float vv[ND][16];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dcol = d * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[d][r] = (dcol < DH) ? Vs[(sub * 32 + kmap(r, h)) * LD + dcol] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < ND; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[d][r], p[r], o[d], 0, 0, 0);
        }
            }
        if (PF) {
            if (more) lstore(KST((it & 1) ^ 1), VST((it & 1) ^ 1));
            __syncthreads();
        }
    }
#undef KST
#undef VST

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

// dQ (and D = rowsum(dO*O)) : one wave per 32 queries (2 per block), K/V staged through LDS.
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32, LD = DH + 4;
    // K/V chunks of 64 keys in LDS.  PF (4-wave workgroups = long sequences): two stages; the next chunk is fetched into
    // registers before the current one is consumed and stored behind it, one barrier per chunk instead of two.
    extern __shared__ __attribute__((aligned(16))) float kv_smem[];
    constexpr bool PF = (NW == 4);
#define KST(s_) (kv_smem + (s_) * 2 * 64 * LD)
#define VST(s_) (kv_smem + (s_) * 2 * 64 * LD + 64 * LD)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32, hh = blockIdx.y, b = blockIdx.z;
    const int q = q0 + l31;
    const bool qok = q < a.T, active = q0 < a.T;
    const float* Qp = a.Q + (size_t)b * a.qbs + (size_t)q * a.ldq + hh * DH;
    const float* dOp = a.dO + ((size_t)b * a.T + q) * a.lddo + hh * DH;
    const float* Op = a.O + ((size_t)b * a.T + q) * a.ldo + hh * DH;
    const float* Kb = a.K + (size_t)b * a.kbs + hh * DH;
    const float* Vb = a.V + (size_t)b * a.vbs + hh * DH;
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
    const unsigned rowkey = (a.drop.p > 0.f) ? attn_rowkey(a.drop, seed, step, (unsigned long long)rowid) : 0u;
    if (qok && h == 0) a.Dsum[rowid] = dsum;
    const float lse = qok ? a.LSE[rowid] : 0.f;

    f32x16 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    RowBatch<DH, 64 * NW> kreg, vreg;
    auto gload = [&](int c0) {
        kreg.load(Kb, a.ldk, c0, a.S, tid);
        vreg.load(Vb, a.ldv, c0, a.S, tid);
    };
    auto lstore = [&](float* Kd, float* Vd) {
        kreg.store(Kd, tid);
        vreg.store(Vd, tid);
    };
    if (PF) {
        gload(0);
        lstore(KST(0), VST(0));
        __syncthreads();
    }
    for (int c0 = 0, it = 0; c0 < a.S; c0 += 64, ++it) {
        const float* Ks = KST(0);
        const float* Vs = VST(0);
        const bool more = c0 + 64 < a.S;
        if (PF) {
            Ks = KST(it & 1);
            Vs = VST(it & 1);
            if (more) gload(c0 + 64);
        } else {
            __syncthreads();
            stage_rows2<DH, 64 * NW>(KST(0), Kb, a.ldk, VST(0), Vb, a.ldv, c0, a.S, tid);
            __syncthreads();
        }
        if (active) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int s0 = c0 + sub * 32;
            if (s0 >= a.S) break;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            const float* kr = Ks + (sub * 32 + l31) * LD + 4 * h;
            const float* vr = Vs + (sub * 32 + l31) * LD + 4 * h;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + 8 * t);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(vr + 8 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], dof[t][j], dpacc, 0, 0, 0);
                }
            }
            float ds[16], keep[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) keep[r] = (a.drop.p > 0.f) ? attn_keep(a.drop, rowkey, s0 + kmap(r, h)) : 1.f;
            const int keyl = s0 + l31;
            const unsigned char vb = (kv && keyl < a.S) ? kv[keyl] : (unsigned char)1;
            const unsigned int vmask = (unsigned int)__ballot((keyl < a.S) && vb != 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vmask >> kmap(r, h)) & 1u;
                const float pe = __expf(fminf(sacc[r] - lse, 80.f));
                const float p = ok ? pe : 0.f;
                ds[r] = p * (dpacc[r] * keep[r] - dsum);
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dcol = d * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float kk = (dcol < DH) ? Ks[(sub * 32 + kmap(r, h)) * LD + dcol] : 0.f;
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, ds[r], dq[d], 0, 0, 0);
                }
            }
        }
            }
        if (PF) {
            if (more) lstore(KST((it & 1) ^ 1), VST((it & 1) ^ 1));
            __syncthreads();
        }
    }
#undef KST
#undef VST

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

// dK, dV : one wave per 32 keys (2 per block); Q / dO / LSE / D of 64 queries at a time staged through LDS.
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32, LD = DH + 4;
    __shared__ __attribute__((aligned(16))) float Qs[64 * LD];
    __shared__ __attribute__((aligned(16))) float Os[64 * LD];
    __shared__ float Ls[64], Ds[64];
    __shared__ unsigned Rk[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int s0 = blockIdx.x * (32 * NW) + wave * 32, hh = blockIdx.y, b = blockIdx.z;
    const int key = s0 + l31;
    const bool active = s0 < a.S;
    const float* Qb = a.Q + (size_t)b * a.qbs + hh * DH;
    const float* dOb = a.dO + (size_t)b * a.T * a.lddo + hh * DH;
    const float* Kp = a.K + (size_t)b * a.kbs + (size_t)key * a.ldk + hh * DH;
    const float* Vp = a.V + (size_t)b * a.vbs + (size_t)key * a.ldv + hh * DH;
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
    for (int c0 = 0; c0 < a.T; c0 += 64) {
        __syncthreads();
        stage_rows<DH, 64 * NW>(Qs, Qb, a.ldq, c0, a.T, tid);          // one batch at a time: this kernel has no registers to spare
        stage_rows<DH, 64 * NW>(Os, dOb, a.lddo, c0, a.T, tid);
        if (tid < 64) {
            const int qq = c0 + tid;
            Ls[tid] = (qq < a.T) ? a.LSE[rowbase + qq] : 0.f;
            Ds[tid] = (qq < a.T) ? a.Dsum[rowbase + qq] : 0.f;
            Rk[tid] = (a.drop.p > 0.f) ? attn_rowkey(a.drop, seed, step, (unsigned long long)(rowbase + qq)) : 0u;
        }
        __syncthreads();
        if (!active) continue;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int q0 = c0 + sub * 32;
            if (q0 >= a.T) break;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            const float* qr = Qs + (sub * 32 + l31) * LD + 4 * h;
            const float* orow = Os + (sub * 32 + l31) * LD + 4 * h;
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                f32x4 qf = *reinterpret_cast<const f32x4*>(qr + 8 * t);
                qf *= a.scale;
                const f32x4 dof = *reinterpret_cast<const f32x4*>(orow + 8 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j], kf[t][j], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[j], vf[t][j], dpacc, 0, 0, 0);
                }
            }
            float pd[16], ds[16], keep[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                keep[r] = (a.drop.p > 0.f) ? attn_keep(a.drop, Rk[sub * 32 + kmap(r, h)], key) : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = sub * 32 + kmap(r, h);
                const int q = c0 + ql;
                const bool ok = kok && (q < a.T);
                const float pe = __expf(fminf(sacc[r] - Ls[ql], 80.f));   // evaluated unconditionally: no divergent branch
                const float p = ok ? pe : 0.f;
                pd[r] = p * keep[r];
                ds[r] = p * (dpacc[r] * keep[r] - Ds[ql]);
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dcol = d * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + kmap(r, h);
                    const float dov = (dcol < DH) ? Os[ql * LD + dcol] : 0.f;
                    const float qv = (dcol < DH) ? Qs[ql * LD + dcol] : 0.f;
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(dov, pd[r], dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, ds[r], dk[d], 0, 0, 0);
                }
            }
        }
    }
    if (kin) {
        float* dKp = a.dK + ((size_t)b * a.S + key) * a.lddk + hh * DH;
        float* dVp = a.dV + ((size_t)b * a.S + key) * a.lddv + hh * DH;
        const bool vec = ((a.lddk | a.lddv) & 3) == 0 && (((reinterpret_cast<uintptr_t>(a.dK) | reinterpret_cast<uintptr_t>(a.dV)) & 15) == 0);
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;          // registers 4g..4g+3 = 4 consecutive columns
                if (dd >= DH) continue;
                f32x4 kk, vv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { kk[e] = dk[d][4 * g4 + e] * a.scale; vv[e] = dv[d][4 * g4 + e]; }
                if (vec) {
                    *reinterpret_cast<f32x4*>(dKp + dd) = kk;
                    *reinterpret_cast<f32x4*>(dVp + dd) = vv;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dKp[dd + e] = kk[e]; dVp[dd + e] = vv[e]; }
                }
            }
    }
}

#include "attention_small4.inc"
#include "attention_x3.inc"

// ---- short sequences: forward -----------------------------------------------------------------------
// T <= 64 and S <= 64: one workgroup (2 waves) per (batch, head); Q, K, V staged once with coalesced 16-byte loads,
// O leaves through LDS as whole rows (the general kernel's per-lane row-strided 4-byte stores touch 32-64 lines per
// instruction).  Same arithmetic, masking and dropout stream as attn_fwd_kernel.
template <int DH>
__global__ __launch_bounds__(128, 2) void attn_fwd_small_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32, LD = DH + 4, C4 = DH / 4;
    __shared__ __attribute__((aligned(16))) float Qs[64 * LD];
    __shared__ __attribute__((aligned(16))) float Ks[64 * LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z;
    const float* Qb = a.Q + (size_t)b * a.qbs + hh * DH;
    const float* Kb = a.K + (size_t)b * a.kbs + hh * DH;
    const float* Vb = a.V + (size_t)b * a.vbs + hh * DH;
    const unsigned char* kv = a.kvalid ? a.kvalid + (size_t)b * a.S : nullptr;
    const size_t rowbase = ((size_t)b * a.nh + hh) * a.T;
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }
    {
        RowBatch<DH, 128> bq, bk, bv;          // 24 loads per lane in flight, one wait
        bq.load(Qb, a.ldq, 0, a.T, tid);
        bk.load(Kb, a.ldk, 0, a.S, tid);
        bv.load(Vb, a.ldv, 0, a.S, tid);
        bq.store(Qs, tid);
        bk.store(Ks, tid);
        bv.store(Vs, tid);
    }
    __syncthreads();
    const int q = wave * 32 + l31;
    if (wave * 32 < a.T) {
        const unsigned rowkey = (a.drop.p > 0.f) ? attn_rowkey(a.drop, seed, step, (unsigned long long)(rowbase + q)) : 0u;
        f32x4 qf[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            qf[t] = *reinterpret_cast<const f32x4*>(Qs + q * LD + 8 * t + 4 * h);
            qf[t] *= a.scale;
        }
        f32x16 o[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        // All S <= 64 keys are in LDS: one softmax pass (no running-max rescale), and both 32-key score tiles in flight as
        // independent MFMA chains so the exp / dropout VALU of one can issue under the MFMAs of the other.
        f32x16 sacc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
        {
            f32x4 kfr[2][NG];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int t = 0; t < NG; ++t)
                    kfr[sub][t] = *reinterpret_cast<const f32x4*>(Ks + (sub * 32 + l31) * LD + 4 * h + 8 * t);
#pragma unroll
            for (int t = 0; t < NG; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
                        sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kfr[sub][t][j], qf[t][j], sacc[sub], 0, 0, 0);
        }
        unsigned int vmask[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int keyl = sub * 32 + l31;
            const unsigned char vb = (kv && keyl < a.S) ? kv[keyl] : (unsigned char)1;
            vmask[sub] = (unsigned int)__ballot((keyl < a.S) && vb != 0);      // keys beyond S (zero rows in LDS) are masked
        }
        float m = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = ((vmask[sub] >> kmap(r, h)) & 1u) ? fmaxf(m, sacc[sub][r]) : m;
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float p[2][16];
        float lsum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[sub][r] = ((vmask[sub] >> kmap(r, h)) & 1u) ? __expf(sacc[sub][r] - m) : 0.f;
                lsum += p[sub][r];
            }
        lsum += __shfl_xor(lsum, 32, 64);
        if (a.drop.p > 0.f) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[sub][r] *= attn_keep(a.drop, rowkey, sub * 32 + kmap(r, h));
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float vv[ND][16];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dcol = d * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[d][r] = (dcol < DH) ? Vs[(sub * 32 + kmap(r, h)) * LD + dcol] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < ND; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[d][r], p[sub][r], o[d], 0, 0, 0);
        }
        // O^T accumulator -> this wave's own rows of Qs (only this wave read them, and only into qf above)
        const float inv = 1.f / lsum;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;
                if (dd >= DH) continue;
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = o[d][4 * g4 + e] * inv;
                *reinterpret_cast<f32x4*>(Qs + q * LD + dd) = x;
            }
        if (h == 0 && a.LSE && q < a.T) a.LSE[rowbase + q] = m + __logf(lsum);
    }
    __syncthreads();
    float* Ob = a.O + (size_t)b * a.T * a.ldo + hh * DH;
#pragma unroll
    for (int i = tid; i < 64 * C4; i += 128) {
        const int r = i / C4, c = (i % C4) * 4;
        if (r < a.T) *reinterpret_cast<f32x4*>(Ob + (size_t)r * a.ldo + c) = *reinterpret_cast<const f32x4*>(Qs + r * LD + c);
    }
}

// ---- short sequences: fused backward ------------------------------------------------------------
// T <= 64 and S <= 64 (every attention of the 64-residue / 64-token configs): one workgroup owns a whole
// (batch, head).  Q, K, V and dO are staged into LDS ONCE with coalesced 16-byte loads (D = rowsum(dO*O) is reduced on
// the way in), waves 0-1 produce dQ for 32 queries each while waves 2-3 produce dK/dV for 32 keys each, and the three
// results leave through LDS as whole rows.  Against the two general kernels this halves the operand reads (both read
// Q, K, V, dO) and replaces their per-lane row-strided 4-byte accesses by 16-byte coalesced ones: 268 MB instead of
// ~400 MB per C2 call, 195 us -> see DESIGN.md.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(const AttnArgs a) {
    constexpr int NG = DH / 8, ND = (DH + 31) / 32, LD = DH + 4, C4 = DH / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;
    float* Ks = Qs + 64 * LD;
    float* Vs = Ks + 64 * LD;
    float* Os = Vs + 64 * LD;              // dO
    float* Ls = Os + 64 * LD;
    float* Ds = Ls + 64;
    unsigned* Rk = reinterpret_cast<unsigned*>(Ds + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int hh = blockIdx.y, b = blockIdx.z;
    const float* Qb = a.Q + (size_t)b * a.qbs + hh * DH;
    const float* Kb = a.K + (size_t)b * a.kbs + hh * DH;
    const float* Vb = a.V + (size_t)b * a.vbs + hh * DH;
    const float* dOb = a.dO + (size_t)b * a.T * a.lddo + hh * DH;
    const float* Ob = a.O + (size_t)b * a.T * a.ldo + hh * DH;
    const unsigned char* kv = a.kvalid ? a.kvalid + (size_t)b * a.S : nullptr;
    const size_t rowbase = ((size_t)b * a.nh + hh) * a.T;
    unsigned long long seed = 0, step = 0;
    if (a.drop.p > 0.f) { seed = a.drop.rng_state[0]; step = a.drop.rng_state[1]; }

    // ---- stage operands (rows beyond T / S are zero) and reduce D: 20 loads per lane in flight, one wait
    {
        RowBatch<DH, 256> bq, bdo, bo, bk, bv;
        bq.load(Qb, a.ldq, 0, a.T, tid);
        bdo.load(dOb, a.lddo, 0, a.T, tid);
        bo.load(Ob, a.ldo, 0, a.T, tid);
        bk.load(Kb, a.ldk, 0, a.S, tid);
        bv.load(Vb, a.ldv, 0, a.S, tid);
        bq.store(Qs, tid);
        bk.store(Ks, tid);
        bv.store(Vs, tid);
        bdo.store(Os, tid);
#pragma unroll
        for (int j = 0; j < RowBatch<DH, 256>::N; ++j) {
            const int i = tid + j * 256, r = i / C4;
            float dsum = bdo.v[j][0] * bo.v[j][0] + bdo.v[j][1] * bo.v[j][1] + bdo.v[j][2] * bo.v[j][2] + bdo.v[j][3] * bo.v[j][3];
#pragma unroll
            for (int off = C4 / 2; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
            if ((i % C4) == 0 && i < 64 * C4) Ds[r] = dsum;
        }
    }
    if (tid < 64) {
        Ls[tid] = (tid < a.T) ? a.LSE[rowbase + tid] : 0.f;
        Rk[tid] = (a.drop.p > 0.f) ? attn_rowkey(a.drop, seed, step, (unsigned long long)(rowbase + tid)) : 0u;
    }
    __syncthreads();

    f32x16 acc0[ND], acc1[ND];             // query waves: acc0 = dQ^T ; key waves: acc0 = dK^T, acc1 = dV^T
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[d][r] = 0.f; acc1[d][r] = 0.f; }

    if (wave < 2) {
        // ---- dQ for queries 32*wave .. +31 (the general dq kernel with every operand in LDS)
        const int q = wave * 32 + l31;
        if (wave * 32 < a.T) {
            f32x4 qf[NG], dof[NG];
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                qf[t] = *reinterpret_cast<const f32x4*>(Qs + q * LD + 8 * t + 4 * h);
                qf[t] *= a.scale;
                dof[t] = *reinterpret_cast<const f32x4*>(Os + q * LD + 8 * t + 4 * h);
            }
            const float dsum = Ds[q], lse = Ls[q];
            const unsigned rowkey = Rk[q];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int s0 = sub * 32;
                if (s0 >= a.S) break;
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
                const float* kr = Ks + (s0 + l31) * LD + 4 * h;
                const float* vr = Vs + (s0 + l31) * LD + 4 * h;
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + 8 * t);
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(vr + 8 * t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[t][j], sacc, 0, 0, 0);
                        dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], dof[t][j], dpacc, 0, 0, 0);
                    }
                }
                float ds[16], keep[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) keep[r] = (a.drop.p > 0.f) ? attn_keep(a.drop, rowkey, s0 + kmap(r, h)) : 1.f;
                const int keyl = s0 + l31;
                const unsigned char vb = (kv && keyl < a.S) ? kv[keyl] : (unsigned char)1;
                const unsigned int vmask = (unsigned int)__ballot((keyl < a.S) && vb != 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = (vmask >> kmap(r, h)) & 1u;
                    const float pe = __expf(fminf(sacc[r] - lse, 80.f));
                    const float p = ok ? pe : 0.f;
                    ds[r] = p * (dpacc[r] * keep[r] - dsum);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const int dcol = d * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float kk = (dcol < DH) ? Ks[(s0 + kmap(r, h)) * LD + dcol] : 0.f;
                        acc0[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, ds[r], acc0[d], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        // ---- dK, dV for keys 32*(wave-2) .. +31 (the general dkv kernel with every operand in LDS)
        const int key = (wave - 2) * 32 + l31;
        if ((wave - 2) * 32 < a.S) {
            const bool kin = key < a.S;
            const bool kok = kin && (!kv || kv[key]);
            f32x4 kf[NG], vf[NG];
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                kf[t] = *reinterpret_cast<const f32x4*>(Ks + key * LD + 8 * t + 4 * h);
                vf[t] = *reinterpret_cast<const f32x4*>(Vs + key * LD + 8 * t + 4 * h);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int q0 = sub * 32;
                if (q0 >= a.T) break;
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
                const float* qr = Qs + (q0 + l31) * LD + 4 * h;
                const float* orow = Os + (q0 + l31) * LD + 4 * h;
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    f32x4 qf = *reinterpret_cast<const f32x4*>(qr + 8 * t);
                    qf *= a.scale;
                    const f32x4 dof = *reinterpret_cast<const f32x4*>(orow + 8 * t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j], kf[t][j], sacc, 0, 0, 0);
                        dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[j], vf[t][j], dpacc, 0, 0, 0);
                    }
                }
                float pd[16], ds[16], keep[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    keep[r] = (a.drop.p > 0.f) ? attn_keep(a.drop, Rk[q0 + kmap(r, h)], key) : 1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = q0 + kmap(r, h);
                    const bool ok = kok && (ql < a.T);
                    const float pe = __expf(fminf(sacc[r] - Ls[ql], 80.f));
                    const float p = ok ? pe : 0.f;
                    pd[r] = p * keep[r];
                    ds[r] = p * (dpacc[r] * keep[r] - Ds[ql]);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const int dcol = d * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = q0 + kmap(r, h);
                        const float dov = (dcol < DH) ? Os[ql * LD + dcol] : 0.f;
                        const float qv = (dcol < DH) ? Qs[ql * LD + dcol] : 0.f;
                        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(dov, pd[r], acc1[d], 0, 0, 0);
                        acc0[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, ds[r], acc0[d], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();                       // every wave is done reading the staged operands

    // ---- results -> LDS as [row][d] (registers 4g..4g+3 of an accumulator are 4 consecutive d), then whole rows out
    {
        const int row = (wave & 1) * 32 + l31;
        float* dst0 = (wave < 2 ? Qs : Ks) + row * LD;
        float* dst1 = Vs + row * LD;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;
                if (dd >= DH) continue;
                f32x4 x0, x1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { x0[e] = acc0[d][4 * g4 + e] * a.scale; x1[e] = acc1[d][4 * g4 + e]; }
                *reinterpret_cast<f32x4*>(dst0 + dd) = x0;
                if (wave >= 2) *reinterpret_cast<f32x4*>(dst1 + dd) = x1;
            }
    }
    __syncthreads();
    float* dQb = a.dQ + (size_t)b * a.T * a.lddq + hh * DH;
    float* dKb = a.dK + (size_t)b * a.S * a.lddk + hh * DH;
    float* dVb = a.dV + (size_t)b * a.S * a.lddv + hh * DH;
#pragma unroll
    for (int i = tid; i < 64 * C4; i += 256) {
        const int r = i / C4, c = (i % C4) * 4;
        if (r < a.T) *reinterpret_cast<f32x4*>(dQb + (size_t)r * a.lddq + c) = *reinterpret_cast<const f32x4*>(Qs + r * LD + c);
        if (r < a.S) {
            *reinterpret_cast<f32x4*>(dKb + (size_t)r * a.lddk + c) = *reinterpret_cast<const f32x4*>(Ks + r * LD + c);
            *reinterpret_cast<f32x4*>(dVb + (size_t)r * a.lddv + c) = *reinterpret_cast<const f32x4*>(Vs + r * LD + c);
        }
    }
}

bool aligned16(const void* p, int ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0); }

// query-side general kernels: K/V stages in dynamic LDS (two stages for the 4-wave form: 70 KB at DH = 64)
template <int DH, int NW>
int launch_kv(void (*kernel)(const AttnArgs), dim3 grid, const AttnArgs& a, hipStream_t stream) {
    const size_t smem = (size_t)(NW == 4 ? 2 : 1) * 2 * 64 * (DH + 4) * sizeof(float);
    static const void* configured[4] = {nullptr, nullptr, nullptr, nullptr};     // per <DH, NW>: forward and dQ kernels
    const void* fn = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (const void* c : configured) seen = seen || (c == fn);
    if (!seen) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            vqh_set_error(hipGetErrorString(e));
            return VQH_ERR_LAUNCH;
        }
        for (const void*& c : configured)
            if (!c) { c = fn; break; }
    }
    hipLaunchKernelGGL(kernel, grid, dim3(64 * NW), smem, stream, a);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern int g_attn_flags;

template <int DH>
int launch_bwd_small4(const AttnArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    constexpr int LD = DH + 4;
    constexpr int VT = (64 * LD > 64 * 68) ? 64 * LD : 64 * 68;
    const size_t smem = (size_t)(3 * 64 * LD + VT + 3 * 64) * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small4_kernel<DH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            vqh_set_error(hipGetErrorString(e));
            return VQH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_small4_kernel<DH>), dim3(1, a.nh, a.B), dim3(256), smem, stream, a);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

template <int DH>
int launch_bwd_small(const AttnArgs& a, hipStream_t stream) {
    if (!(g_attn_flags & 2)) return launch_bwd_small4<DH>(a, stream);      // bit 1: the two-group kernel of round 1 (A/B runs)
    static bool attr_set = false;
    const size_t smem = (size_t)(4 * 64 * (DH + 4) + 3 * 64) * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small_kernel<DH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            vqh_set_error(hipGetErrorString(e));
            return VQH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_small_kernel<DH>), dim3(1, a.nh, a.B), dim3(256), smem, stream, a);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

int g_attn_flags = 0;

}  // namespace

// diagnostic switch (tests / probes): bit 0 = never take the short-sequence fused kernels
extern "C" int vqh_attn_set_flags(int flags) { const int old = g_attn_flags; g_attn_flags = flags; return old; }

extern "C" int vqh_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                            int ldo, float* LSE, const unsigned char* kvalid, int B, int nh, int T, int S, int dh,
                            int qkv_shared, const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                            hipStream_t stream) {
    VQH_CHECK_ARG(B >= 0 && nh > 0 && T >= 0 && S >= 0, "vqh_attn_fwd: bad shape");
    VQH_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "vqh_attn_fwd: head dim must be 16/32/64");
    if (B == 0 || T == 0) return VQH_OK;
    VQH_CHECK_ARG(S > 0, "vqh_attn_fwd: no keys");
    VQH_CHECK_ARG(Q && K && V && O, "vqh_attn_fwd: null pointer");
    VQH_CHECK_ARG(aligned16(Q, ldq) && aligned16(K, ldk) && aligned16(V, ldv), "vqh_attn_fwd: operands must be 16-byte aligned");
    VQH_CHECK_ARG(drop_p == 0.f || rng_state, "vqh_attn_fwd: dropout needs rng_state");
    AttnArgs a{};
    a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.LSE = LSE;
    a.kvalid = kvalid; a.B = B; a.nh = nh; a.T = T; a.S = S; a.scale = 1.0f / sqrtf((float)dh);
    a.qbs = (qkv_shared & 1) ? 0 : (long long)T * ldq; a.kbs = (qkv_shared & 2) ? 0 : (long long)S * ldk; a.vbs = (qkv_shared & 2) ? 0 : (long long)S * ldv;
    a.drop = make_drop(rng_state, drop_site, drop_p);
    if (T <= 64 && S <= 64 && aligned16(O, ldo) && !(g_attn_flags & 1)) {
        dim3 grid(1, nh, B);
        if (dh == 64 && !(g_attn_flags & (2 | 4))) {           // bf16 matrix pipes on exactly split operands (attention_x3.inc)
            static bool attr_set = false;
            const int smem = 2 * ax::IMG + 2 * 128 * 4;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_small_x3_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
                attr_set = true;
            }
            hipLaunchKernelGGL(attn_fwd_small_x3_kernel, grid, dim3(256), smem, stream, a);
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
        if (!(g_attn_flags & 2)) {                   // quartered form: 4 waves, one 32 x 32 quarter each
            switch (dh) {
                case 16: hipLaunchKernelGGL((attn_fwd_small4_kernel<16>), grid, dim3(256), 0, stream, a); break;
                case 32: hipLaunchKernelGGL((attn_fwd_small4_kernel<32>), grid, dim3(256), 0, stream, a); break;
                default: hipLaunchKernelGGL((attn_fwd_small4_kernel<64>), grid, dim3(256), 0, stream, a); break;
            }
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
        switch (dh) {
            case 16: hipLaunchKernelGGL((attn_fwd_small_kernel<16>), grid, dim3(128), 0, stream, a); break;
            case 32: hipLaunchKernelGGL((attn_fwd_small_kernel<32>), grid, dim3(128), 0, stream, a); break;
            default: hipLaunchKernelGGL((attn_fwd_small_kernel<64>), grid, dim3(128), 0, stream, a); break;
        }
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
    // 4 waves (128 queries) share each staged K/V chunk when there are enough queries per (batch, head)
    const int NWq = (T >= 128) ? 4 : 2;
    dim3 grid((T + 32 * NWq - 1) / (32 * NWq), nh, B);
    if (dh == 64 && !(g_attn_flags & 4) && aligned16(O, ldo)) {     // bf16 matrix pipes on exactly split operands (attention_x3.inc)
        static bool attr_x3[2] = {false, false};
        const void* fn = (NWq == 4) ? reinterpret_cast<const void*>(&attn_fwd_x3_kernel<4>) : reinterpret_cast<const void*>(&attn_fwd_x3_kernel<2>);
        if (!attr_x3[NWq == 4]) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ax::IMG);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_x3[NWq == 4] = true;
        }
        if (NWq == 4) hipLaunchKernelGGL((attn_fwd_x3_kernel<4>), grid, dim3(256), 2 * ax::IMG, stream, a);
        else hipLaunchKernelGGL((attn_fwd_x3_kernel<2>), grid, dim3(128), 2 * ax::IMG, stream, a);
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
#define FWD(DH_)                                                                                              \
    if (NWq == 4) rc = launch_kv<DH_, 4>(attn_fwd_kernel<DH_, 4>, grid, a, stream);                            \
    else rc = launch_kv<DH_, 2>(attn_fwd_kernel<DH_, 2>, grid, a, stream)
    int rc = VQH_OK;
    switch (dh) {
        case 16: FWD(16); break;
        case 32: FWD(32); break;
        default: FWD(64); break;
    }
#undef FWD
    return rc;
}

extern "C" int vqh_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                            const float* O, int ldo, const float* LSE, const float* dO, int lddo, float* Dsum,
                            float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                            const unsigned char* kvalid, int B, int nh, int T, int S, int dh, int qkv_shared,
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
    a.qbs = (qkv_shared & 1) ? 0 : (long long)T * ldq; a.kbs = (qkv_shared & 2) ? 0 : (long long)S * ldk; a.vbs = (qkv_shared & 2) ? 0 : (long long)S * ldv;
    a.drop = make_drop(rng_state, drop_site, drop_p);
    a.dO = dO; a.lddo = lddo; a.Dsum = Dsum; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
    if (T <= 64 && S <= 64 && aligned16(dQ, lddq) && aligned16(dK, lddk) && aligned16(dV, lddv) && !(g_attn_flags & 1)) {
        int rc = VQH_OK;
        if (dh == 64 && !(g_attn_flags & (2 | 4))) {           // bf16 matrix pipes on exactly split operands (attention_x3.inc)
            static bool attr_set = false;
            const int smem = 2 * 64 * (ax::DH + 4) * 4 + 64 * (ax::DH + 4) * 4;       // three fp32 result tiles > two images + row scalars
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small_x3_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
                attr_set = true;
            }
            hipLaunchKernelGGL(attn_bwd_small_x3_kernel, dim3(1, nh, B), dim3(256), smem, stream, a);
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
        switch (dh) {
            case 16: rc = launch_bwd_small<16>(a, stream); break;
            case 32: rc = launch_bwd_small<32>(a, stream); break;
            default: rc = launch_bwd_small<64>(a, stream); break;
        }
        return rc;
    }
    const int NWq = (T >= 128) ? 4 : 2, NWk = (S >= 128) ? 4 : 2;
    dim3 gq((T + 32 * NWq - 1) / (32 * NWq), nh, B), gk((S + 32 * NWk - 1) / (32 * NWk), nh, B);
    if (dh == 64 && !(g_attn_flags & 4) && aligned16(dQ, lddq) && aligned16(dK, lddk) && aligned16(dV, lddv)) {
        // bf16 matrix pipes on exactly split operands (attention_x3.inc)
        static bool attr_b[4] = {false, false, false, false};
        const int smem_q = 2 * ax::IMG, smem_k = 2 * ax::IMG + 3 * 64 * 4;
        const void* fq = (NWq == 4) ? reinterpret_cast<const void*>(&attn_bwd_dq_x3_kernel<4>) : reinterpret_cast<const void*>(&attn_bwd_dq_x3_kernel<2>);
        const void* fk = (NWk == 4) ? reinterpret_cast<const void*>(&attn_bwd_dkv_x3_kernel<4>) : reinterpret_cast<const void*>(&attn_bwd_dkv_x3_kernel<2>);
        if (!attr_b[NWq == 4]) {
            hipError_t e = hipFuncSetAttribute(fq, hipFuncAttributeMaxDynamicSharedMemorySize, smem_q);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_b[NWq == 4] = true;
        }
        if (!attr_b[2 + (NWk == 4)]) {
            hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, smem_k);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_b[2 + (NWk == 4)] = true;
        }
        if (NWq == 4) hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<4>), gq, dim3(256), smem_q, stream, a);
        else hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<2>), gq, dim3(128), smem_q, stream, a);
        if (NWk == 4) hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<4>), gk, dim3(256), smem_k, stream, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<2>), gk, dim3(128), smem_k, stream, a);
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
#define BWD(DH_)                                                                                          \
    if (NWq == 4) rc = launch_kv<DH_, 4>(attn_bwd_dq_kernel<DH_, 4>, gq, a, stream);                       \
    else rc = launch_kv<DH_, 2>(attn_bwd_dq_kernel<DH_, 2>, gq, a, stream);                                \
    if (rc != VQH_OK) return rc;                                                                          \
    if (NWk == 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH_, 4>), gk, dim3(256), 0, stream, a);          \
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH_, 2>), gk, dim3(128), 0, stream, a)
    int rc = VQH_OK;
    switch (dh) {
        case 16: BWD(16); break;
        case 32: BWD(32); break;
        default: BWD(64); break;
    }
#undef BWD
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
