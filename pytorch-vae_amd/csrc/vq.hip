// EMA vector quantizer kernels (reference: VectorQuantizerEMA, /root/reference/models/vq_vae.py:19-283).
//
//   vqh_vq_nearest     argmin_k ||z_r - e_k||^2  (:183-188 / :238-244), first-minimum tie rule.
//                      fp32 MFMA scores e.e - 2 z.e with top-2 tracking per row; rows whose top-2 gap is
//                      inside the fp32 noise band (or exactly tied) are re-evaluated exactly in fp64 with
//                      the direct sum (z-e)^2 form, lowest index winning ties.  The R x K distance matrix
//                      is never materialised (the reference writes it twice: distances + one-hot).
//   vqh_vq_gather      z_q = E[idx]; z_st = z + (z_q - z) (:189,:199); residual for the next RVQ level (:258)
//   vqh_vq_segment_sum cnt[k] = #rows with idx==k, sum[k,:] = sum of those rows (:81-83, without the
//                      dense one-hot GEMM); deterministic: one block per code, fixed reduction order.
//   vqh_vq_ema_apply   ema <- d*ema + (1-d)*stat; E <- ema_emb / (ema_cnt + eps) for the WHOLE table (:85-89)
//   vqh_vq_usage_stats perplexity / dead ratio / epoch accumulators (:201-220, :265-278)
#include "common.h"
#include <vector>

namespace {

__device__ __forceinline__ int kmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void row_sqnorm_kernel(const float* __restrict__ X, int ld, int rows, int D, float* __restrict__ out,
                                  float scale = 1.f) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float v = X[(size_t)row * ld + i];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s * scale;
}

// ---- duplicate codes --------------------------------------------------------------------------------
// Identical codebook rows (e.g. the hundreds of all-zero rows a fresh EMA table has after its first refresh) give
// bit-identical scores, i.e. exact ties that torch.argmin resolves to the lowest index.  They must not count as
// "ambiguous" (the refinement re-reads the whole table per flagged row).  canon[k] = lowest index j <= k whose row
// equals row k; a candidate only competes for "second best" if its canonical index differs from the best's.
__global__ void code_hash_kernel(const float* __restrict__ E, int lde, int K, int D, unsigned long long* __restrict__ hash) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    unsigned long long h = 0;
    for (int i = lane; i < D; i += 64) {
        float v = E[(size_t)k * lde + i];
        if (v == 0.f) v = 0.f;                                    // -0 and +0 score identically
        unsigned x = __float_as_uint(v) ^ ((unsigned)i * 0x9E3779B9U);
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        h += ((unsigned long long)x << 17) ^ (unsigned long long)(x * 0x85EBCA6BU);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if (lane == 0) hash[k] = h;
}
// canon[k] in O(K): an open-addressing table keyed by the 64-bit row hash collects, per distinct hash, the LOWEST code
// index (atomicMin: the result does not depend on the insertion order); a wave per code then verifies that row k really
// equals that row (a hash collision of different rows leaves canon[k] = k, which is merely conservative: such a row
// pair would be flagged ambiguous and settled exactly by the refinement).  Round 1 scanned all j < k per code (O(K^2):
// 0.9 ms at K = 8192, every step).
constexpr unsigned long long HASH_EMPTY = ~0ull;
__global__ void canon_init_kernel(unsigned long long* __restrict__ keys, int* __restrict__ minidx, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) { keys[i] = HASH_EMPTY; minidx[i] = 0x7fffffff; }
}
__global__ void canon_insert_kernel(const unsigned long long* __restrict__ hash, int K, unsigned long long* __restrict__ keys,
                                    int* __restrict__ minidx, int T) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    unsigned long long hk = hash[k];
    if (hk == HASH_EMPTY) hk = 0x1234567ull;
    unsigned slot = (unsigned)(hk ^ (hk >> 29)) & (unsigned)(T - 1);
    for (int probe = 0; probe < T; ++probe) {
        const unsigned long long prev = atomicCAS(&keys[slot], HASH_EMPTY, hk);
        if (prev == HASH_EMPTY || prev == hk) { atomicMin(&minidx[slot], k); return; }
        slot = (slot + 1) & (unsigned)(T - 1);
    }
}
__global__ __launch_bounds__(256) void canon_lookup_kernel(const unsigned long long* __restrict__ hash,
                                                           const float* __restrict__ E, int lde, int K, int D,
                                                           const unsigned long long* __restrict__ keys,
                                                           const int* __restrict__ minidx, int T, int* __restrict__ canon) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    unsigned long long hk = hash[k];
    if (hk == HASH_EMPTY) hk = 0x1234567ull;
    unsigned slot = (unsigned)(hk ^ (hk >> 29)) & (unsigned)(T - 1);
    int c = k;
    for (int probe = 0; probe < T; ++probe) {
        const unsigned long long key = keys[slot];
        if (key == hk) { c = minidx[slot]; break; }
        if (key == HASH_EMPTY) break;
        slot = (slot + 1) & (unsigned)(T - 1);
    }
    if (c != k) {                                    // wave-uniform: verify the rows are really identical
        bool same = true;
        for (int i = lane; i < D; i += 64) same = same && (E[(size_t)c * lde + i] == E[(size_t)k * lde + i]);
        if (!__all(same)) c = k;
    }
    if (lane == 0) canon[k] = c;
}

// one wave per (32 rows, code range): codes in tiles of 32, D multiple of 8.  blockIdx.y selects a contiguous range
// of codes so that small R with large K still fills the chip; partial (best, second, index) triples are merged in
// code order by vq_combine_kernel (ties -> lower index, like torch.argmin).
__global__ __launch_bounds__(64, 2) void vq_nearest_kernel(const float* __restrict__ Z, int ldz,
                                                           const float* __restrict__ E, int lde,
                                                           const float* __restrict__ enorm,
                                                           const int* __restrict__ canon, float* __restrict__ pbest,
                                                           float* __restrict__ psecond, int* __restrict__ pidx, int R,
                                                           int K, int D, int kchunk) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * 32;
    const int row = r0 + l31;
    const bool rok = row < R;
    const float* zp = Z + (size_t)row * ldz;
    float best = INFINITY, second = INFINITY;
    int bidx = 0x7fffffff, bcan = -1;
    const int ng = D / 8;
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);
    for (int c0 = kbeg; c0 < kend; c0 += 32) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int code_l = c0 + l31;
        const float* ep = E + (size_t)code_l * lde;
        for (int t = 0; t < ng; ++t) {
            f32x4 ef = {0.f, 0.f, 0.f, 0.f}, zf = {0.f, 0.f, 0.f, 0.f};
            if (code_l < kend) ef = *reinterpret_cast<const f32x4*>(ep + 8 * t + 4 * h);
            if (rok) zf = *reinterpret_cast<const f32x4*>(zp + 8 * t + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[j], zf[j], acc, 0, 0, 0);
        }
        // acc[r] = dot(E[c0 + kmap(r,h)], Z[row])
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = c0 + kmap(r, h);
            if (code < kend) {
                const float d = enorm[code] - 2.f * acc[r];
                if (d < best) {
                    second = best;
                    best = d;
                    bidx = code;
                    bcan = canon[code];
                } else if (d < second) {
                    if (!(d == best && canon[code] == bcan)) second = d;    // an exact duplicate of the best is not a rival
                }
            }
        }
    }
    // combine the two lane halves (same row, disjoint code subsets)
    const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64), oc = __shfl_xor(bcan, 32, 64);
    const bool dup = (ob == best) && (oc == bcan);
    float fb, fs;
    int fi;
    if (ob < best || (ob == best && oi < bidx)) {
        fb = ob; fi = oi; fs = dup ? fminf(second, os) : fminf(best, os);
    } else {
        fb = best; fi = bidx; fs = dup ? fminf(second, os) : fminf(ob, second);
    }
    if (rok && h == 0) {
        const size_t o = (size_t)blockIdx.y * R + row;
        pbest[o] = fb;
        psecond[o] = fs;
        pidx[o] = fi;
    }
}

// LDS-staged form (D = 8 * NT8 <= 256): a workgroup of 4 waves owns 128 rows; every wave keeps ITS 32 rows of Z in
// registers for the whole kernel (D/2 VGPRs per lane, the k-permuted B operand) and the codebook streams through LDS in
// tiles of 64 codes shared by the 4 waves (register-staged, double buffered, padded rows: conflict-free ds_read_b128) --
// the codebook is fetched once per 128 rows instead of once per 32, and never from a per-lane global gather.  The code
// norms and canonical indices of a tile ride along in LDS.  Scores, top-2 tracking and tie rules are EXACTLY those of
// vq_nearest_kernel (same MFMA order over k, same comparisons), so both kernels return identical partials.
template <int NT8>
__global__ __launch_bounds__(256, 2) void vq_nearest_lds_kernel(const float* __restrict__ Z, int ldz,
                                                                const float* __restrict__ E, int lde,
                                                                const float* __restrict__ enorm,
                                                                const int* __restrict__ canon, float* __restrict__ pbest,
                                                                float* __restrict__ psecond, int* __restrict__ pidx, int R,
                                                                int K, int kchunk) {
    constexpr int D = 8 * NT8, LDT = D + 4;
    constexpr int CT = (NT8 >= 32) ? 32 : 64;            // codes per LDS tile (D = 256: 32, or the staging registers spill)
    constexpr int TILE_F = CT * LDT;                          // floats per code tile
    constexpr int CHUNKS = CT * (D / 4);                      // 16-byte chunks per tile
    constexpr int LPT = (CHUNKS + 255) / 256;                 // chunk loads per thread
    extern __shared__ __attribute__((aligned(16))) float vsm[];
    auto tileb = [&](int b) { return vsm + b * TILE_F; };
    auto nrm = [&](int b) { return vsm + 2 * TILE_F + b * CT; };
    auto can = [&](int b) { return reinterpret_cast<int*>(vsm + 2 * TILE_F + 2 * CT) + b * CT; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + l31;
    const bool rok = row < R;
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);

    f32x4 zf[NT8];
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
        zf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rok) zf[t] = *reinterpret_cast<const f32x4*>(Z + (size_t)row * ldz + 8 * t + 4 * h);
    }
    f32x4 stg[LPT];
    float stg_n = 0.f;
    int stg_c = 0;
    auto load_tile = [&](int c0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = tid + 256 * i;
            const int code = c0 + ch / (D / 4);
            stg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < CHUNKS && code < kend) stg[i] = *reinterpret_cast<const f32x4*>(E + (size_t)code * lde + (ch % (D / 4)) * 4);
        }
        if (tid < CT) {
            const int code = c0 + tid;
            stg_n = (code < kend) ? enorm[code] : 0.f;
            stg_c = (code < kend) ? canon[code] : -1;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = tid + 256 * i;
            if (ch < CHUNKS) *reinterpret_cast<f32x4*>(tileb(buf) + (ch / (D / 4)) * LDT + (ch % (D / 4)) * 4) = stg[i];
        }
        if (tid < CT) { nrm(buf)[tid] = stg_n; can(buf)[tid] = stg_c; }
    };

    float best = INFINITY, second = INFINITY;
    int bidx = 0x7fffffff, bcan = -1;
    load_tile(kbeg);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = kbeg; c0 < kend; c0 += CT) {
        const bool more = c0 + CT < kend;
        if (more) load_tile(c0 + CT);
        const float* tb = tileb(buf);
        const float* nb = nrm(buf);
        const int* cb = can(buf);
#pragma unroll
        for (int blk = 0; blk < CT / 32; ++blk) {
            if (c0 + blk * 32 >= kend) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ep = tb + (blk * 32 + l31) * LDT + 4 * h;
#pragma unroll
            for (int t = 0; t < NT8; ++t) {
                const f32x4 ef = *reinterpret_cast<const f32x4*>(ep + 8 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[j], zf[t][j], acc, 0, 0, 0);
            }
            // acc[r] = dot(E[c0 + 32 blk + kmap(r,h)], Z[row])
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = blk * 32 + kmap(r, h);
                const int code = c0 + cl;
                if (code < kend) {
                    const float d = nb[cl] - 2.f * acc[r];
                    if (d < best) {
                        second = best;
                        best = d;
                        bidx = code;
                        bcan = cb[cl];
                    } else if (d < second) {
                        if (!(d == best && cb[cl] == bcan)) second = d;    // an exact duplicate of the best is not a rival
                    }
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // combine the two lane halves (same row, disjoint code subsets)
    const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64), oc = __shfl_xor(bcan, 32, 64);
    const bool dup = (ob == best) && (oc == bcan);
    float fb, fs;
    int fi;
    if (ob < best || (ob == best && oi < bidx)) {
        fb = ob; fi = oi; fs = dup ? fminf(second, os) : fminf(best, os);
    } else {
        fb = best; fi = bidx; fs = dup ? fminf(second, os) : fminf(ob, second);
    }
    if (rok && h == 0) {
        const size_t o = (size_t)blockIdx.y * R + row;
        pbest[o] = fb;
        psecond[o] = fs;
        pidx[o] = fi;
    }
}

// ---- split-operand form (D = 16 * NT16 in {64, 128, 256}) ------------------------------------------------------------------
// The scores E . Z^T on the bf16 matrix pipes from EXACT three-way splits of both operands (the scheme of gemm_dma.inc's
// x3 tiles: a = h + m + l, six products, fp32 accumulation; as close to fp64 as the fp32 MFMA).  Here the split costs
// nothing inside the loop: the codebook is split ONCE per call by vq_split_codebook_kernel into rows of [h | m | l | pad]
// (6 D + 16 bytes: the pad makes the 32 rows of a fragment read hit 16 different 16-byte bank slots), so a code tile goes
// to LDS as a plain copy, and a wave splits its 32 rows of Z once into registers (3 D / 4 VGPRs per lane).  Per 32-code
// block: D / 16 * 6 MFMAs of 32 cycles instead of D / 2 of 64.  Top-2 tracking, tie rules, the noise band and the fp64
// re-evaluation of ambiguous rows are those of the fp32 kernels (the band is 30x the error of either arithmetic).
typedef unsigned vq_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 vq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vq_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void vq_split_pair(float a0, float a1, unsigned& hp, unsigned& mp, unsigned& lp) {
    const vq_bf16x2 h2 = {(__bf16)a0, (__bf16)a1};
    const float r0 = a0 - (float)h2[0], r1 = a1 - (float)h2[1];
    const vq_bf16x2 m2 = {(__bf16)r0, (__bf16)r1};
    const float s0 = r0 - (float)m2[0], s1 = r1 - (float)m2[1];
    const vq_bf16x2 l2 = {(__bf16)s0, (__bf16)s1};
    hp = __builtin_bit_cast(unsigned, h2);
    mp = __builtin_bit_cast(unsigned, m2);
    lp = __builtin_bit_cast(unsigned, l2);
}

// Ex[k] = [h plane: D bf16 | m plane | l plane | 16 bytes pad]; one thread per pair of elements
__global__ void vq_split_codebook_kernel(const float* __restrict__ E, int lde, int K, int D, unsigned char* __restrict__ Ex) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int half = D / 2;
    if (i >= (long long)K * half) return;
    const int k = (int)(i / half), c = (int)(i % half) * 2;
    const float a0 = E[(size_t)k * lde + c], a1 = E[(size_t)k * lde + c + 1];
    unsigned hp, mp, lp;
    vq_split_pair(a0, a1, hp, mp, lp);
    unsigned char* row = Ex + (size_t)k * (6 * D + 16);
    *reinterpret_cast<unsigned*>(row + 2 * c) = hp;
    *reinterpret_cast<unsigned*>(row + 2 * D + 2 * c) = mp;
    *reinterpret_cast<unsigned*>(row + 4 * D + 2 * c) = lp;
}

template <int NT16>
__global__ __launch_bounds__(256, (NT16 >= 16) ? 1 : 2) void vq_nearest_x3_kernel(const float* __restrict__ Z, int ldz,
                                                                                 const unsigned char* __restrict__ Ex,
                                                                                 const float* __restrict__ enorm,
                                                                                 const int* __restrict__ canon,
                                                                                 float* __restrict__ pbest,
                                                                                 float* __restrict__ psecond,
                                                                                 int* __restrict__ pidx, int R, int K, int kchunk) {
    constexpr int D = 16 * NT16, ROWB = 6 * D + 16;
    constexpr int CT = (NT16 >= 16) ? 32 : 64;                // codes per LDS tile
    constexpr int TILE_B = CT * ROWB;                         // bytes per code tile
    constexpr int CHUNKS = TILE_B / 16;                       // 16-byte chunks per tile
    constexpr int LPT = (CHUNKS + 255) / 256;                 // chunk loads per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    auto tileb = [&](int b) { return xsm + b * TILE_B; };
    auto nrm = [&](int b) { return reinterpret_cast<float*>(xsm + 2 * TILE_B) + b * CT; };
    auto can = [&](int b) { return reinterpret_cast<int*>(xsm + 2 * TILE_B + 2 * CT * 4) + b * CT; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + l31;
    const bool rok = row < R;
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);

    // this lane's share of its row of Z as three bf16 planes: group g holds k = 16 g + 8 h .. + 7
    vq_u32x4 zh[NT16], zm[NT16], zl[NT16];
#pragma unroll
    for (int g = 0; g < NT16; ++g) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (rok) {
            a = *reinterpret_cast<const f32x4*>(Z + (size_t)row * ldz + 16 * g + 8 * h);
            b = *reinterpret_cast<const f32x4*>(Z + (size_t)row * ldz + 16 * g + 8 * h + 4);
        }
        unsigned ph[4], pm[4], pl[4];
        vq_split_pair(a[0], a[1], ph[0], pm[0], pl[0]);
        vq_split_pair(a[2], a[3], ph[1], pm[1], pl[1]);
        vq_split_pair(b[0], b[1], ph[2], pm[2], pl[2]);
        vq_split_pair(b[2], b[3], ph[3], pm[3], pl[3]);
        zh[g] = vq_u32x4{ph[0], ph[1], ph[2], ph[3]};
        zm[g] = vq_u32x4{pm[0], pm[1], pm[2], pm[3]};
        zl[g] = vq_u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
    vq_u32x4 stg[LPT];
    float stg_n = 0.f;
    int stg_c = 0;
    auto load_tile = [&](int c0) {
        const int ncodes = min(CT, kend - c0);                // codes of this tile that exist (rows beyond stay stale: never scored)
        const unsigned char* src = Ex + (size_t)c0 * ROWB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = tid + 256 * i;
            stg[i] = vq_u32x4{0u, 0u, 0u, 0u};
            if (ch < CHUNKS && ch * 16 < ncodes * ROWB) stg[i] = *reinterpret_cast<const vq_u32x4*>(src + (size_t)ch * 16);
        }
        if (tid < CT) {
            const int code = c0 + tid;
            stg_n = (code < kend) ? enorm[code] : 0.f;
            stg_c = (code < kend) ? canon[code] : -1;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = tid + 256 * i;
            if (ch < CHUNKS) *reinterpret_cast<vq_u32x4*>(tileb(buf) + (size_t)ch * 16) = stg[i];
        }
        if (tid < CT) { nrm(buf)[tid] = stg_n; can(buf)[tid] = stg_c; }
    };

    float best = INFINITY, second = INFINITY;
    int bidx = 0x7fffffff, bcan = -1;
    load_tile(kbeg);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = kbeg; c0 < kend; c0 += CT) {
        const bool more = c0 + CT < kend;
        if (more) load_tile(c0 + CT);
        const unsigned char* tb = tileb(buf);
        const float* nb = nrm(buf);
        const int* cb = can(buf);
#pragma unroll
        for (int blk = 0; blk < CT / 32; ++blk) {
            if (c0 + blk * 32 >= kend) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const unsigned char* ep = tb + (size_t)(blk * 32 + l31) * ROWB + 16 * h;
            // the three plane fragments of group g + 1 are requested between the six MFMAs of group g (at D = 256 the
            // 192 registers of Z planes leave the compiler no room to keep both sets apart: it issues the reads late and
            // half of the LDS latency stays exposed; Z planes in AGPRs through inline-asm MFMAs were tried and dropped)
            vq_u32x4 ef[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ef[0][pl] = *reinterpret_cast<const vq_u32x4*>(ep + pl * 2 * D);
#pragma unroll
            for (int g = 0; g < NT16; ++g) {
                if (g + 1 < NT16) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ef[(g + 1) & 1][pl] = *reinterpret_cast<const vq_u32x4*>(ep + pl * 2 * D + 32 * (g + 1));
                }
                const vq_bf16x8 eh = __builtin_bit_cast(vq_bf16x8, ef[g & 1][0]), em = __builtin_bit_cast(vq_bf16x8, ef[g & 1][1]),
                                el = __builtin_bit_cast(vq_bf16x8, ef[g & 1][2]);
                const vq_bf16x8 bh = __builtin_bit_cast(vq_bf16x8, zh[g]), bm = __builtin_bit_cast(vq_bf16x8, zm[g]),
                                bl = __builtin_bit_cast(vq_bf16x8, zl[g]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(el, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(em, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(em, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, bh, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < NT16; ++g)
#pragma unroll
                for (int m6 = 0; m6 < 6; ++m6) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m6 < 3 && g + 1 < NT16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            // acc[r] = dot(E[c0 + 32 blk + kmap(r,h)], Z[row])
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = blk * 32 + kmap(r, h);
                const int code = c0 + cl;
                if (code < kend) {
                    const float d = nb[cl] - 2.f * acc[r];
                    if (d < best) {
                        second = best;
                        best = d;
                        bidx = code;
                        bcan = cb[cl];
                    } else if (d < second) {
                        if (!(d == best && cb[cl] == bcan)) second = d;    // an exact duplicate of the best is not a rival
                    }
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // combine the two lane halves (same row, disjoint code subsets)
    const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64), oc = __shfl_xor(bcan, 32, 64);
    const bool dup = (ob == best) && (oc == bcan);
    float fb, fs;
    int fi;
    if (ob < best || (ob == best && oi < bidx)) {
        fb = ob; fi = oi; fs = dup ? fminf(second, os) : fminf(best, os);
    } else {
        fb = best; fi = bidx; fs = dup ? fminf(second, os) : fminf(ob, second);
    }
    if (rok && h == 0) {
        const size_t o = (size_t)blockIdx.y * R + row;
        pbest[o] = fb;
        psecond[o] = fs;
        pidx[o] = fi;
    }
}

// start values of the plane-tensor score kernel (gemm_p3.inc, EPI_NEAREST): -|e|^2 / 2, or -inf for a code that is an exact copy
// of a lower one (it can neither win -- the lower index has the same score -- nor count as a rival)
__global__ void vq_p3_init_kernel(const float* __restrict__ enorm, const int* __restrict__ canon, int K, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = (canon[k] == k) ? -0.5f * enorm[k] : -INFINITY;
}

// merge the per-range partials (ascending code ranges), write the index and flag rows whose top-2 gap is inside the
// fp32 noise band; bestval keeps the winning score for the refinement's pre-filter.
__global__ void vq_combine_kernel(const float* __restrict__ pbest, const float* __restrict__ psecond,
                                  const int* __restrict__ pidx, const int* __restrict__ canon, int nsplit,
                                  const float* __restrict__ znorm,
                                  long long* __restrict__ idx_out, int idx_offset, unsigned char* __restrict__ ambiguous,
                                  float* __restrict__ bestval, int R, float rel_tol) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float fb = INFINITY, fs = INFINITY;
    int fi = 0x7fffffff;
    for (int s = 0; s < nsplit; ++s) {
        const size_t o = (size_t)s * R + row;
        const float b = pbest[o], sc = psecond[o];
        const int i = pidx[o];
        if (i == 0x7fffffff) continue;
        const bool dup = (fi != 0x7fffffff) && (b == fb) && (canon[i] == canon[fi]);
        if (b < fb || (b == fb && i < fi)) {
            fs = dup ? fminf(fs, sc) : fminf(fb, fminf(fs, sc));
            fb = b;
            fi = i;
        } else {
            fs = dup ? fminf(fs, sc) : fminf(fs, b);
        }
    }
    // a row of NaN (or Inf - Inf) scores never satisfies d < best: torch.argmin returns index 0 for an all-NaN row
    // (/root/reference/models/vq_vae.py:188), and so does this; the loss then goes NaN instead of the gather faulting
    if (fi == 0x7fffffff) fi = 0;
    idx_out[row] = (long long)fi + idx_offset;
    const float scale = znorm[row] + fabsf(fb) + 1e-30f;
    ambiguous[row] = ((fs - fb) <= rel_tol * scale) ? 1 : 0;
    bestval[row] = fb;
}

// exact re-evaluation of flagged rows, lowest index wins ties.  One wave per row.  Every code is first scored with the
// fp32 direct form sum (z-e)^2 (4 independent accumulators); only codes within the noise band of the provisional
// best are re-evaluated in fp64, so the cost is an fp32 scan plus a handful of fp64 rows.
__global__ __launch_bounds__(256) void vq_refine_kernel(const float* __restrict__ Z, int ldz,
                                                        const float* __restrict__ E, int lde,
                                                        long long* __restrict__ idx_out, int idx_offset,
                                                        const unsigned char* __restrict__ ambiguous,
                                                        const float* __restrict__ znorm,
                                                        const float* __restrict__ bestval, int R, int K, int D,
                                                        float rel_tol) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R || !ambiguous[row]) return;
    const float* zp = Z + (size_t)row * ldz;
    const float zn = znorm[row];
    const float scale = zn + fabsf(bestval[row]) + 1e-30f;
    const float thr = (zn + bestval[row]) + 4.f * rel_tol * scale;      // provisional best distance + a generous band
    double best = 1e300;
    int bidx = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float* ep = E + (size_t)k * lde;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int i = 0; i < D; i += 4) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(zp + i), ev = *reinterpret_cast<const f32x4*>(ep + i);
            const f32x4 d = zv - ev;
            s0 += d[0] * d[0]; s1 += d[1] * d[1]; s2 += d[2] * d[2]; s3 += d[3] * d[3];
        }
        if ((s0 + s1) + (s2 + s3) <= thr) {
            double s = 0.0;
            for (int i = 0; i < D; ++i) {
                const double d = (double)zp[i] - (double)ep[i];
                s += d * d;
            }
            if (s < best) { best = s; bidx = k; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0 && bidx != 0x7fffffff) idx_out[row] = (long long)bidx + idx_offset;
}

// ---- chip-wide refinement ----------------------------------------------------------------------------------------------
// vq_refine_kernel gives every flagged row ONE wave that scans all K codes (K * D * 4 bytes through one wave: 8 MB and
// 1.1 ms per call at K = 8192, D = 256, whatever the number of flagged rows).  Here the flagged rows are compacted into a
// list and the (row, 256-code chunk) pairs are spread over the whole chip; a second small kernel merges a row's partials
// (lowest index wins ties, like torch.argmin).  Same arithmetic and the same candidate rule as vq_refine_kernel, so both give
// identical indices.  Rows beyond the partial buffer's capacity (a pathological table) fall back to the full scan.
constexpr int REFINE_CPW = 256;            // codes per work item (4 per lane)
__global__ void vq_amb_compact_kernel(const unsigned char* __restrict__ ambiguous, int R, int* __restrict__ count,
                                      int* __restrict__ list) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R && ambiguous[r]) list[atomicAdd(count, 1)] = r;      // order is irrelevant: rows are independent
}

__device__ __forceinline__ void refine_score_codes(const float* __restrict__ zp, const float* __restrict__ E, int lde, int D, int kbeg,
                                                   int kend, int lane, float thr, double& best, int& bidx) {
    for (int k = kbeg + lane; k < kend; k += 64) {
        const float* ep = E + (size_t)k * lde;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int i = 0; i < D; i += 4) {
            const f32x4 zv = *reinterpret_cast<const f32x4*>(zp + i), ev = *reinterpret_cast<const f32x4*>(ep + i);
            const f32x4 d = zv - ev;
            s0 += d[0] * d[0]; s1 += d[1] * d[1]; s2 += d[2] * d[2]; s3 += d[3] * d[3];
        }
        if ((s0 + s1) + (s2 + s3) <= thr) {
            double s = 0.0;
            for (int i = 0; i < D; ++i) {
                const double d = (double)zp[i] - (double)ep[i];
                s += d * d;
            }
            if (s < best || (s == best && k < bidx)) { best = s; bidx = k; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
}

__global__ __launch_bounds__(256) void vq_refine_items_kernel(const float* __restrict__ Z, int ldz, const float* __restrict__ E, int lde,
                                                              const int* __restrict__ count, const int* __restrict__ list,
                                                              const float* __restrict__ znorm, const float* __restrict__ bestval,
                                                              int K, int D, int CH, int cap, float rel_tol,
                                                              double* __restrict__ pdist, int* __restrict__ pidx) {
    const int lane = threadIdx.x & 63;
    const int nrows = min(*count, cap);
    const long long items = (long long)nrows * CH, nwaves = (long long)gridDim.x * 4;
    for (long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); w < items; w += nwaves) {
        const int li = (int)(w / CH), ch = (int)(w % CH);
        const int row = list[li];
        const float zn = znorm[row];
        const float scale = zn + fabsf(bestval[row]) + 1e-30f;
        const float thr = (zn + bestval[row]) + 4.f * rel_tol * scale;
        double best = 1e300;
        int bidx = 0x7fffffff;
        refine_score_codes(Z + (size_t)row * ldz, E, lde, D, ch * REFINE_CPW, min(K, (ch + 1) * REFINE_CPW), lane, thr, best, bidx);
        if (lane == 0) { pdist[w] = best; pidx[w] = bidx; }
    }
}

// one wave per listed row: merge its CH partials (or, beyond the buffer's capacity, scan the whole table)
__global__ __launch_bounds__(256) void vq_refine_merge_kernel(const float* __restrict__ Z, int ldz, const float* __restrict__ E, int lde,
                                                              const int* __restrict__ count, const int* __restrict__ list,
                                                              const float* __restrict__ znorm, const float* __restrict__ bestval,
                                                              int K, int D, int CH, int cap, float rel_tol,
                                                              const double* __restrict__ pdist, const int* __restrict__ pidx,
                                                              long long* __restrict__ idx_out, int idx_offset) {
    const int lane = threadIdx.x & 63;
    const int n = *count;
    for (int li = blockIdx.x * 4 + (threadIdx.x >> 6); li < n; li += gridDim.x * 4) {
        const int row = list[li];
        double best = 1e300;
        int bidx = 0x7fffffff;
        if (li < cap) {
            for (int c = lane; c < CH; c += 64) {
                const double d = pdist[(size_t)li * CH + c];
                const int i = pidx[(size_t)li * CH + c];
                if (d < best || (d == best && i < bidx)) { best = d; bidx = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bidx, o, 64);
                if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            }
        } else {
            const float zn = znorm[row];
            const float scale = zn + fabsf(bestval[row]) + 1e-30f;
            const float thr = (zn + bestval[row]) + 4.f * rel_tol * scale;
            refine_score_codes(Z + (size_t)row * ldz, E, lde, D, 0, K, lane, thr, best, bidx);
        }
        if (lane == 0 && bidx != 0x7fffffff) idx_out[row] = (long long)bidx + idx_offset;
    }
}

// z_q = E[idx - idx_offset]; z_st = z + (z_q - z); zq_acc (+)= z_q ; res_out = res_in - z_q
__global__ void vq_gather_kernel(const float* __restrict__ E, int lde, const long long* __restrict__ idx,
                                 int idx_offset, const float* __restrict__ rows_in, int ldr,
                                 float* __restrict__ zq_level, float* __restrict__ res_out, int R, int D) {
    const long long total = (long long)R * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / D), c = (int)(e % D);
        const float q = E[(size_t)(idx[r] - idx_offset) * lde + c];
        if (zq_level) zq_level[e] = q;
        if (res_out) res_out[e] = __fsub_rn(rows_in[(size_t)r * ldr + c], q);
    }
}

// z_q_total = sum over levels (in level order), z_st = z_e + (z_q_total - z_e)
__global__ void vq_finish_kernel(const float* __restrict__ zq_levels, int Q, const float* __restrict__ ze, int ldz,
                                 float* __restrict__ zq, float* __restrict__ zst, int R, int D) {
    const long long total = (long long)R * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / D), c = (int)(e % D);
        float s = zq_levels[e];
        for (int l = 1; l < Q; ++l) s = __fadd_rn(s, zq_levels[(size_t)l * total + e]);
        zq[e] = s;
        const float z = ze[(size_t)r * ldz + c];
        zst[e] = __fadd_rn(z, __fsub_rn(s, z));
    }
}

// block per code k in [k0, k0+Kn): cnt[k], sum[k,:] over rows with idx[r] == k   (deterministic)
template <int VPT>
__global__ __launch_bounds__(256) void vq_segment_sum_kernel(const float* __restrict__ rows, int ldr,
                                                             const long long* __restrict__ idx, int R, int D, int k0,
                                                             float* __restrict__ cnt, float* __restrict__ sum) {
    __shared__ float red[4][64 * VPT];
    __shared__ int cred[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = k0 + blockIdx.x;
    float acc[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) acc[j] = 0.f;
    int c = 0;
    const int per = (R + 3) / 4;
    const int rbeg = wave * per, rend = min(R, rbeg + per);
    for (int base = rbeg; base < rend; base += 64) {
        const int r = base + lane;
        const bool match = (r < rend) && (idx[r] == (long long)k);
        unsigned long long mask = __ballot(match);
        c += __popcll(mask);
        while (mask) {      // up to 4 matching rows per trip so their loads are in flight together
            int bits[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bits[u] = mask ? __ffsll((long long)mask) - 1 : -1;
                if (mask) mask &= mask - 1;
            }
            float v[4][VPT];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int col = lane + 64 * j;
                    v[u][j] = (bits[u] >= 0 && col < D) ? rows[(size_t)(base + bits[u]) * ldr + col] : 0.f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VPT; ++j) acc[j] += v[u][j];
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) red[wave][lane + 64 * j] = acc[j];
    if (lane == 0) cred[wave] = c;
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256)
        sum[(size_t)k * D + col] = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
    if (threadIdx.x == 0) cnt[k] = (float)(cred[0] + cred[1] + cred[2] + cred[3]);
}

// Small tables (Kn*D <= 32768 floats): one block per 64 rows accumulates into an LDS-resident table.
// Wave w owns the codes with (code & 3) == w and visits the chunk's rows in ascending order, so every table
// entry has exactly one writer and a fixed summation order: deterministic, and independent of how skewed the
// code usage is (a collapsed codebook no longer serialises one block).  Partials: part[blk][Kn*D + Kn].
__global__ __launch_bounds__(256) void vq_segment_table_kernel(const float* __restrict__ rows, int ldr,
                                                               const long long* __restrict__ idx, int R, int D, int k0,
                                                               int Kn, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float table[];   // [Kn*D] sums, then [Kn] counts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = Kn * D;
    for (int i = threadIdx.x; i < n + Kn; i += 256) table[i] = 0.f;
    const int r0 = blockIdx.x * 64;
    const int nrows = min(64, R - r0);
    const int mycode = (lane < nrows) ? (int)(idx[r0 + lane] - k0) : -1;
    __syncthreads();
    for (int i0 = 0; i0 < nrows; i0 += 8) {
        float v[8];
        int code[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            code[u] = __shfl(mycode, min(i0 + u, 63), 64);
            if (i0 + u >= nrows) code[u] = -1;
        }
        for (int col = lane; col < D; col += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (code[u] >= 0 && (code[u] & 3) == wave) ? rows[(size_t)(r0 + i0 + u) * ldr + col] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (code[u] >= 0 && (code[u] & 3) == wave) table[code[u] * D + col] += v[u];
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (code[u] >= 0 && (code[u] & 3) == wave) table[n + code[u]] += 1.f;
        }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * (n + Kn);
    for (int i = threadIdx.x; i < n + Kn; i += 256) out[i] = table[i];
}

// Large tables (Kn * D floats beyond the LDS): the per-code block of vq_segment_sum_kernel reads ALL R indices (Kn blocks x R
// x 8 bytes: 17 GB of L2 traffic at Kn = 8192, R = 262144 -- 1.36 ms).  Here a workgroup owns (row chunk c, code range g):
// it scans the chunk's indices once, and accumulates the rows whose code falls into ITS range of KR codes into an LDS table
// of KR * D floats.  Every row is fetched exactly once over the whole grid (it matches one range), the index array is read
// once per range, and the result is deterministic: wave w owns the codes with (code & 15) == w and visits the chunk's rows in
// ascending order, the chunks' partial tables are summed in chunk order by vq_table_reduce_kernel.
// part[c][Kn * D + Kn]: sums then counts of chunk c.
__global__ __launch_bounds__(1024) void vq_segment_range_kernel(const float* __restrict__ rows, int ldr, const long long* __restrict__ idx,
                                                                int R, int D, int k0, int Kn, int KR, int rows_per_chunk,
                                                                float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float table[];   // [KR * D] sums, [KR] counts, then the waves' row lists
    // 16 waves: wave w owns the codes with (code & 15) == w.  The scan only COLLECTS the wave's rows (ascending) in a small LDS
    // list; the list is then gathered 8 rows at a time, so 16 x 8 KB are in flight per CU (matches are sparse -- one row in
    // 64 falls into a range -- and a gather per match would cost one dependent HBM round trip each).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x, c = blockIdx.y;
    const int klo = g * KR, kn = min(KR, Kn - klo);                  // this block's codes: klo .. klo + kn (relative to k0)
    const int n = kn * D;
    for (int i = threadIdx.x; i < KR * D + KR; i += 1024) table[i] = 0.f;
    int* lrow = reinterpret_cast<int*>(table + (size_t)KR * D + KR) + wave * 256;   // [128] rows, [128] codes
    int* lcod = lrow + 128;
    const int r_beg = c * rows_per_chunk, r_end = min(R, r_beg + rows_per_chunk);
    __syncthreads();
    const int vpl = D >> 8;                                          // float4 per lane and row when D % 256 == 0
    int cnt = 0;                                                     // entries in this wave's list (wave-uniform)
    auto flush = [&]() {
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            int rr[8], cd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < cnt;
                rr[u] = on ? lrow[i0 + u] : -1;
                cd[u] = on ? lcod[i0 + u] : 0;
            }
            if ((D & 255) == 0) {
                for (int q = 0; q < vpl; ++q) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = (rr[u] >= 0) ? *reinterpret_cast<const f32x4*>(rows + (size_t)rr[u] * ldr + q * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (rr[u] >= 0) {
                            f32x4* t = reinterpret_cast<f32x4*>(table + (size_t)cd[u] * D + q * 256 + lane * 4);
                            *t = *t + v[u];
                        }
                }
            } else {
                for (int col = lane; col < D; col += 64) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (rr[u] >= 0) ? rows[(size_t)rr[u] * ldr + col] : 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (rr[u] >= 0) table[(size_t)cd[u] * D + col] += v[u];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rr[u] >= 0) table[(size_t)KR * D + cd[u]] += 1.f;
            }
        }
        cnt = 0;
    };
    // the indices of 8 row groups are requested together (one dependent L2 round trip per group made the scan the bottleneck)
    for (int base8 = r_beg; base8 < r_end; base8 += 512) {
        long long ci8[8];
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) {
            const int r = base8 + 64 * u8 + lane;
            ci8[u8] = (r < r_end) ? idx[r] : -1;
        }
#pragma unroll
        for (int u8 = 0; u8 < 8; ++u8) {
            const long long ci = ci8[u8] - k0 - klo;
            const int code = (ci8[u8] >= 0 && ci >= 0 && ci < kn) ? (int)ci : -1;
            const bool mine = code >= 0 && (code & 15) == wave;
            const unsigned long long m = __ballot(mine);
            if (m) {
                if (mine) {
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    lrow[pos] = base8 + 64 * u8 + lane;
                    lcod[pos] = code;
                }
                cnt += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (cnt >= 64) {                                   // a group adds at most 64 entries: the list (128) cannot overflow
                    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the list entries have landed
                    flush();
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    flush();
    __syncthreads();
    float* out = part + (size_t)c * ((size_t)Kn * D + Kn);
    for (int i = threadIdx.x; i < n; i += 1024) out[(size_t)klo * D + i] = table[i];
    for (int i = threadIdx.x; i < kn; i += 1024) out[(size_t)Kn * D + klo + i] = table[(size_t)KR * D + i];
}

// ---- sorted form: stable radix sort of the rows by code, then a segmented sum over the sorted list ------------------------------
// Both table forms above serialise on a skewed code distribution (early in training most rows share a few codes: one wave of
// one workgroup then adds thousands of rows, 0.6 ms per RVQ level at R = 8192).  Sorted by code (stable: rows of a code stay in
// ascending order), the sums are a segmented reduction over a list: every wave takes SEG consecutive entries whatever the
// distribution, interior runs are written directly, the runs cut by a wave boundary go through a small partial buffer that a
// fix-up kernel adds in wave order.  Deterministic, no table, no per-chunk partial tables; cost = the row bytes once.
// Sort: LSD radix, 7 bits per pass (1-3 passes for K <= 2^21), ranks by one ballot per bucket (no atomics):
//   pass kernels: (1) per 64-row group: bucket counts [group][128] + each row's rank inside its group and bucket,
//                 (2) per bucket: exclusive scan of the counts down the groups + bucket totals,
//                 (3) scatter to base(bucket) + prefix(group, bucket) + rank.
constexpr int SORT_BITS = 7, SORT_NB = 1 << SORT_BITS;
__global__ __launch_bounds__(256) void vq_sort_hist_kernel(const long long* __restrict__ idx, const int* __restrict__ key_in, int R, int k0,
                                                           int Kn, int shift, int* __restrict__ key_out0,
                                                           int* __restrict__ counts, unsigned char* __restrict__ rank) {
    const int lane = threadIdx.x & 63, group = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = group * 64 + lane;
    if (group * 64 >= R) return;
    int key = Kn;                                                   // invalid / masked rows sort behind every code
    if (r < R) {
        if (key_in) key = key_in[r];
        else {
            const long long c = idx[r] - k0;
            key = (c >= 0 && c < Kn) ? (int)c : Kn;
            key_out0[r] = key;
        }
    }
    const int d = (r < R) ? ((key >> shift) & (SORT_NB - 1)) : -1;
    int myrank = 0, c0 = 0, c1 = 0;
#pragma unroll 8
    for (int b = 0; b < SORT_NB; ++b) {
        const unsigned long long m = __ballot(d == b);
        if (d == b) myrank = __popcll(m & ((1ull << lane) - 1ull));
        const int n = __popcll(m);
        if ((b & 63) == lane) { if (b < 64) c0 = n; else c1 = n; }
    }
    counts[(size_t)group * SORT_NB + lane] = c0;
    counts[(size_t)group * SORT_NB + 64 + lane] = c1;
    if (r < R) rank[r] = (unsigned char)myrank;
}

// one workgroup per bucket: exclusive prefix of counts[g][b] over the groups g (in place) and totals[b]
__global__ __launch_bounds__(1024) void vq_sort_scan_kernel(int* __restrict__ counts, int G, int* __restrict__ totals) {
    __shared__ int part[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (G + 1023) / 1024;
    const int g0 = t * per, g1 = min(G, g0 + per);
    int s = 0;
    for (int g = g0; g < g1; ++g) s += counts[(size_t)g * SORT_NB + b];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                           // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                                          // exclusive prefix of this thread's slice
    for (int g = g0; g < g1; ++g) {
        const int c = counts[(size_t)g * SORT_NB + b];
        counts[(size_t)g * SORT_NB + b] = run;
        run += c;
    }
    if (t == 1023) totals[b] = part[1023];
}

__global__ __launch_bounds__(256) void vq_sort_scatter_kernel(const int* __restrict__ key_in, const int* __restrict__ row_in, int R, int shift,
                                                              const int* __restrict__ counts, const int* __restrict__ totals,
                                                              const unsigned char* __restrict__ rank, int* __restrict__ key_out,
                                                              int* __restrict__ row_out) {
    __shared__ int base[SORT_NB];
    if (threadIdx.x < 64) {                                         // exclusive scan of the 128 bucket totals (one wave)
        const int a = totals[2 * threadIdx.x], b2 = totals[2 * threadIdx.x + 1];
        int incl = a + b2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += v;
        }
        base[2 * threadIdx.x] = incl - a - b2;
        base[2 * threadIdx.x + 1] = incl - b2;
    }
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int key = key_in[r];
    const int d = (key >> shift) & (SORT_NB - 1);
    const int pos = base[d] + counts[(size_t)(r >> 6) * SORT_NB + d] + (int)rank[r];
    key_out[pos] = key;
    row_out[pos] = row_in ? row_in[r] : r;
}

// Segmented sum over the sorted list.  One wave per SEG entries; lane l holds columns {4 l + 256 q .. + 3} (D % 4 == 0, D <= 1024).
//   interior runs -> sum / cnt directly;  first and last run of the wave -> partial slots 2 w and 2 w + 1 (key, count, vector)
constexpr int SEG = 256;
template <int NQ>
__global__ __launch_bounds__(256) void vq_seg_reduce_kernel(const float* __restrict__ rows, int ldr, const int* __restrict__ skey,
                                                            const int* __restrict__ srow, int R, int D, int k0, int Kn,
                                                            float* __restrict__ cnt, float* __restrict__ sum, int* __restrict__ pkey,
                                                            float* __restrict__ pcnt, float* __restrict__ pvec) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int e0 = w * SEG, e1 = min(R, e0 + SEG);
    if (e0 >= R) return;
    const int prev_key = (e0 > 0) ? skey[e0 - 1] : -1;
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = -1, n = 0, nruns = 0;
    if (lane == 0) { pkey[2 * w] = -1; pkey[2 * w + 1] = -1; }
    // end of a run: a run that continues from the previous wave (the wave's first run, same key as the entry before e0) or into
    // the next wave (`cont`) is a partial; everything else is final
    auto flush = [&](bool cont) {
        if (cur < 0 || cur >= Kn) return;                           // nothing yet / the invalid bucket
        const bool is_first = (nruns == 0) && (cur == prev_key);
        if (is_first || cont) {
            const int slot = is_first ? 2 * w : 2 * w + 1;
            if (lane == 0) { pkey[slot] = cur; pcnt[slot] = (float)n; }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (4 * lane + 256 * q < D) *reinterpret_cast<f32x4*>(pvec + (size_t)slot * D + 4 * lane + 256 * q) = acc[q];
        } else {
            if (lane == 0) cnt[k0 + cur] = (float)n;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (4 * lane + 256 * q < D) *reinterpret_cast<f32x4*>(sum + (size_t)(k0 + cur) * D + 4 * lane + 256 * q) = acc[q];
        }
    };
    for (int base = e0; base < e1; base += 64) {
        const int e = base + lane;
        const int mykey = (e < e1) ? skey[e] : -1;
        const int myrow = (e < e1) ? srow[e] : 0;
        const int cntb = min(64, e1 - base);
        for (int i0 = 0; i0 < cntb; i0 += 8) {
            f32x4 v[8][NQ];
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < cntb;
                kk[u] = on ? __shfl(mykey, i0 + u, 64) : -2;
                const int rr = __shfl(myrow, on ? i0 + u : 0, 64);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    v[u][q] = (on && kk[u] < Kn && 4 * lane + 256 * q < D) ? *reinterpret_cast<const f32x4*>(rows + (size_t)rr * ldr + 4 * lane + 256 * q)
                                                                          : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kk[u] == -2) continue;
                if (kk[u] != cur) {
                    flush(false);
                    if (cur >= 0) ++nruns;
                    cur = kk[u]; n = 0;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] += v[u][q];
                ++n;
            }
        }
    }
    // the wave's last run continues into the next wave iff that wave starts with the same key
    flush((e1 < R) && (skey[e1] == cur));
}

// partial slots in order 0, 1, 2, ...: slots with the same key are consecutive (sorted list); the first slot of such a run adds
// them in slot order and writes the code's final sum and count.  One wave per slot.
template <int NQ>
__global__ __launch_bounds__(256) void vq_seg_fixup_kernel(const int* __restrict__ pkey, const float* __restrict__ pcnt,
                                                           const float* __restrict__ pvec, int nslots, int D, int k0,
                                                           float* __restrict__ cnt, float* __restrict__ sum) {
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslots) return;
    const int key = pkey[s];
    if (key < 0) return;
    // previous valid slot with the same key -> not the head of the run
    for (int t = s - 1; t >= 0; --t) {
        const int k = pkey[t];
        if (k == key) return;
        if (k >= 0) break;
    }
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float n = 0.f;
    for (int t = s; t < nslots; ++t) {
        const int k = pkey[t];
        if (k < 0) continue;
        if (k != key) break;
        n += pcnt[t];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (4 * lane + 256 * q < D) acc[q] += *reinterpret_cast<const f32x4*>(pvec + (size_t)t * D + 4 * lane + 256 * q);
    }
    if (lane == 0) cnt[k0 + key] = n;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (4 * lane + 256 * q < D) *reinterpret_cast<f32x4*>(sum + (size_t)(k0 + key) * D + 4 * lane + 256 * q) = acc[q];
}

__global__ __launch_bounds__(1024) void vq_table_reduce_kernel(const float* __restrict__ part, int S, long long stride,
                                                               long long n, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < n)
        for (int k = ty; k < S; k += 16) s += part[(size_t)k * stride + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][tx];
        out[i] = t;
    }
}

// whole-table EMA refresh, arithmetic order of the reference (mul_, add_ of a scaled stat, divide)
__global__ void vq_ema_apply_kernel(const float* __restrict__ cnt, const float* __restrict__ sum,
                                    float* __restrict__ ema_cnt, float* __restrict__ ema_emb, float* __restrict__ emb,
                                    int K, int D, float decay, float one_minus_decay, float eps) {
    const long long total = (long long)K * D;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e / D);
        const float c_new = __fadd_rn(__fmul_rn(ema_cnt[k], decay), __fmul_rn(cnt[k], one_minus_decay));
        const float e_new = __fadd_rn(__fmul_rn(ema_emb[e], decay), __fmul_rn(sum[e], one_minus_decay));
        ema_emb[e] = e_new;
        emb[e] = __fdiv_rn(e_new, __fadd_rn(c_new, eps));
    }
}
__global__ void vq_ema_cnt_kernel(const float* __restrict__ cnt, float* __restrict__ ema_cnt, int K, float decay,
                                  float one_minus_decay) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) ema_cnt[k] = __fadd_rn(__fmul_rn(ema_cnt[k], decay), __fmul_rn(cnt[k], one_minus_decay));
}

// usage[K] (float counts) -> stats[0] = perplexity, stats[1] = dead ratio; ep_usage += usage; ep_cnt += n
__global__ __launch_bounds__(256) void vq_usage_stats_kernel(const float* __restrict__ usage, int K, float n_positions,
                                                             float* __restrict__ ep_usage, float* __restrict__ ep_cnt,
                                                             float* __restrict__ stats) {
    __shared__ float red[3][4];
    float tot = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) tot += usage[k];
    tot = wave_sum(tot);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = tot;
    __syncthreads();
    tot = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float den = fmaxf(tot, 1.f);
    float ent = 0.f, dead = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float u = usage[k];
        const float p = u / den;
        if (p > 0.f) ent += p * logf(p);
        if (u == 0.f) dead += 1.f;
        if (ep_usage) ep_usage[k] += u;
    }
    ent = wave_sum(ent);
    dead = wave_sum(dead);
    if ((threadIdx.x & 63) == 0) { red[1][threadIdx.x >> 6] = ent; red[2][threadIdx.x >> 6] = dead; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ent = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        dead = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        stats[0] = (tot > 0.f) ? expf(-ent) : 0.f;
        stats[1] = dead / (float)K;
        if (ep_cnt) ep_cnt[0] += n_positions;
    }
}

// dead-code re-seeding (reference _maybe_reinit_dead_codes, models/vq_vae.py:91-107): every code whose batch usage
// is <= threshold takes a random encoder row: embedding = ema_embedding = z_e[pick[k]], ema_cluster_size = 1
__global__ void vq_reinit_kernel(const float* __restrict__ usage, float threshold, const long long* __restrict__ pick,
                                 const float* __restrict__ rows, int ldr, float* __restrict__ emb,
                                 float* __restrict__ ema_emb, float* __restrict__ ema_cnt, int K, int D) {
    const int k = blockIdx.x;
    if (k >= K || !(usage[k] <= threshold)) return;
    const float* src = rows + (size_t)pick[k] * ldr;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float v = src[c];
        emb[(size_t)k * D + c] = v;
        ema_emb[(size_t)k * D + c] = v;
    }
    if (threadIdx.x == 0) ema_cnt[k] = 1.f;
}

inline int blocks_for(long long n) {
    long long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace

// Live timing of the nearest-neighbour main kernel (bench.py --vq-only), like vqh_gemm_profile_*: begin(), eager calls,
// end(out[3]) = {launches, kernel seconds, sum of 2*R*K*D}.
namespace {
struct VqProfRec { double flops; hipEvent_t e0, e1; };
bool g_vq_prof_on = false;
std::vector<VqProfRec> g_vq_prof;
int g_vq_flags = 0;             // bit 0: force the per-wave global-gather kernel (round-1 form); bit 1: fp32 MFMA instead of the split-operand kernel
}  // namespace
extern "C" int vqh_vq_set_flags(int flags) { const int old = g_vq_flags; g_vq_flags = flags; return old; }
extern "C" int vqh_vq_profile_begin(void) {
    g_vq_prof.clear();
    g_vq_prof_on = true;
    return VQH_OK;
}
extern "C" int vqh_vq_profile_end(double* out) {
    g_vq_prof_on = false;
    int rc = VQH_OK;
    if (out) out[0] = out[1] = out[2] = 0.0;
    for (VqProfRec& r : g_vq_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
            vqh_set_error("vqh_vq_profile_end: event query failed");
            rc = VQH_ERR_LAUNCH;
        } else if (out) {
            out[0] += 1.0;
            out[1] += (double)ms * 1e-3;
            out[2] += r.flops;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_vq_prof.clear();
    return rc;
}

// workspace (floats): hash table (3 T, T = power of two >= 2K) + K hashes (2K) + canon (K) + K code norms + R row norms +
// R best scores + nsplit*R*(best, second, index) + R/4 flag bytes + the split codebook + the refinement's list of flagged rows
// (R + 4) and partials (3 * min(R, 16384) * ceil(K / 256))
extern "C" int vqh_p3_split(const float* X, int ldx, void* P, long long pitch_bytes, int rows, int cols, hipStream_t stream);
extern "C" int vqh_internal_nearest_p3(const float* Z, int ldz, const float* E, int lde, void* Zp, void* Ep, const float* nb_init, float* pbest,
                                       float* psecond, int* pidx, int R, int K, int D, int nsplit, hipStream_t stream);

struct VqNearestPlan {
    bool lds_form, x3_form;
    long long ex_floats, refine_floats, need;
    int rows_per_block, row_blocks, nsplit, kchunk, T, CH, cap;
};
static VqNearestPlan vq_nearest_plan(int R, int K, int D) {
    // LDS-staged kernel: 128 rows per workgroup; D/8 in {1,2,4,8,16,32}
    const int nt8 = D / 8;
    const bool lds_form = !(g_vq_flags & 1) && D <= 256 && (nt8 & (nt8 - 1)) == 0;
    // split-operand (bf16 pipes) form for D in {64, 128, 256}; bit 1 of the flags keeps the fp32 MFMA kernel (A/B runs, tests)
    const bool x3_form = lds_form && !(g_vq_flags & 2) && (D == 64 || D == 128 || D == 256);
    const long long ex_floats = x3_form ? ((long long)K * (6 * D + 16) + 3) / 4 : 0;
    const int rows_per_block = lds_form ? 128 : 32;
    const int row_blocks = (R + rows_per_block - 1) / rows_per_block;
    const int target = lds_form ? 512 : 2048;    // workgroups wanted (2 x 256 CUs / ~2048 single waves)
    int nsplit = 1;                              // split the code range when there are few rows
    if (row_blocks < target / 2) {
        nsplit = (target + row_blocks - 1) / row_blocks;
        const int max_split = (K + 255) / 256;   // at least 256 codes per range
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    }
    int kchunk = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
    nsplit = (K + kchunk - 1) / kchunk;
    int T = 64;
    while (T < 2 * K) T <<= 1;
    // chip-wide refinement (vq_refine_items_kernel): list of flagged rows + partials for up to `cap` of them
    const int CH = (K + REFINE_CPW - 1) / REFINE_CPW;
    const int cap = R < 16384 ? R : 16384;
    const long long refine_floats = 4 + (long long)R + 3LL * cap * CH + 2;
    const long long need = 3LL * T + 4LL * K + 2LL * R + 3LL * nsplit * R + (R + 3) / 4 + 8 + ex_floats + 4 + refine_floats;
    return VqNearestPlan{lds_form, x3_form, ex_floats, refine_floats, need, rows_per_block, row_blocks, nsplit, kchunk, T, CH, cap};
}
// Extra workspace floats of the plane-tensor form and its code split, 0 when the shape is not taken by it: whole 256-row and
// 128-code tiles, D = 128 / 256 / 512, at least 96 workgroups after splitting the code range (each keeps >= 16 K-steps).
static long long vq_p3_extra_floats(int R, int K, int D, int* ns_out) {
    if ((g_vq_flags & (1 | 2 | 32)) || !(D == 128 || D == 256 || D == 512) || R <= 0 || (R % 256) != 0 || (K % 128) != 0) return 0;
    // the code range is split 8 ways where possible, not only for few rows: the workgroups of one row tile share its Z planes
    // through one XCD's L2 (vq_nearest_p3_kernel)
    const int mt = R / 256, tn = K / 128, nk = D / 32;
    int ns = 1;
    while (ns < 8 && (tn % (2 * ns)) == 0 && (tn / (2 * ns)) * nk >= 16) ns *= 2;
    if (mt * ns < 96 || tn / ns > 32) return 0;
    if (ns_out) *ns_out = ns;
    return 3LL * R * D / 2 + 3LL * K * D / 2 + K + 6LL * ns * R + 16;
}
// 3 = plane-tensor form, 2 = register-resident split form, 1 = fp32 MFMA with the codebook through LDS, 0 = per-wave gather
extern "C" int vqh_vq_nearest_form(int R, int K, int D, long long workspace_floats) {
    if (R <= 0 || K <= 0 || D <= 0 || (D % 8) != 0) return 0;
    const VqNearestPlan pl = vq_nearest_plan(R, K, D);
    const long long extra = vq_p3_extra_floats(R, K, D, nullptr);
    if (extra > 0 && extra + pl.need <= workspace_floats) return 3;
    return pl.x3_form ? 2 : pl.lds_form ? 1 : 0;
}

// workspace floats with which vqh_vq_nearest takes its fastest form for the shape (>= the documented minimum)
extern "C" int vqh_vq_nearest_workspace(int R, int K, int D, long long* floats_out) {
    VQH_CHECK_ARG(floats_out && R >= 0 && K > 0 && D > 0 && (D % 8) == 0, "vqh_vq_nearest_workspace: bad argument");
    const VqNearestPlan pl = vq_nearest_plan(R > 0 ? R : 1, K, D);
    *floats_out = pl.need + vq_p3_extra_floats(R, K, D, nullptr);
    return VQH_OK;
}

extern "C" int vqh_vq_nearest(const float* Z, int ldz, const float* E, int lde, long long* idx_out, int idx_offset,
                              int R, int K, int D, float rel_tol, float* workspace, long long workspace_floats,
                              hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && K > 0 && D > 0 && (D % 8) == 0, "vqh_vq_nearest: D must be a positive multiple of 8");
    if (R == 0) return VQH_OK;
    VQH_CHECK_ARG(Z && E && idx_out && workspace, "vqh_vq_nearest: null pointer");
    VQH_CHECK_ARG(((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(E)) & 15) == 0 && (ldz & 3) == 0 && (lde & 3) == 0,
                  "vqh_vq_nearest: operands must be 16-byte aligned");
    const VqNearestPlan pl = vq_nearest_plan(R, K, D);
    const bool lds_form = pl.lds_form, x3_form = pl.x3_form;
    const int nt8 = D / 8;
    const long long ex_floats = pl.ex_floats, need = pl.need;
    const int row_blocks = pl.row_blocks, nsplit = pl.nsplit, kchunk = pl.kchunk, T = pl.T, CH = pl.CH, cap = pl.cap;
    VQH_CHECK_ARG(need <= workspace_floats, "vqh_vq_nearest: workspace too small");
    // Plane-tensor form (D = 128 / 256, whole 256-row and 128-code tiles): Z and the codebook are split once into bf16 plane
    // tensors and scored by the LDS-DMA fed 256 x 128 tile loop of gemm_p3.inc with a running top-2 instead of an output
    // (vq_nearest_p3_kernel).  It needs 1.5 (R + K) D + K + 6 ns R more workspace floats and is taken when the caller gave them.
    int ns3 = 1;
    const long long p3_floats = vq_p3_extra_floats(R, K, D, &ns3);
    const bool p3_form = p3_floats > 0 && need + p3_floats <= workspace_floats;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);       // 2T floats, 8-byte aligned
    int* minidx = reinterpret_cast<int*>(workspace + 2 * (size_t)T);
    unsigned long long* hash = reinterpret_cast<unsigned long long*>(workspace + 3 * (size_t)T);   // 2K floats
    int* canon = reinterpret_cast<int*>(workspace + 3 * (size_t)T + 2 * (size_t)K);
    float* enorm = workspace + 3 * (size_t)T + 3 * (size_t)K;
    float* znorm = enorm + K;
    float* bestval = znorm + R;
    float* pbest = bestval + R;
    float* psecond = pbest + (size_t)nsplit * R;
    int* pidx = reinterpret_cast<int*>(psecond + (size_t)nsplit * R);
    unsigned char* amb = reinterpret_cast<unsigned char*>(pidx + (size_t)nsplit * R);
    // pre-split codebook rows, 16-byte aligned, behind the flag bytes
    unsigned char* Ex = reinterpret_cast<unsigned char*>(
        (reinterpret_cast<uintptr_t>(amb + (size_t)((R + 3) / 4) * 4) + 15) & ~(uintptr_t)15);
    if (x3_form) {
        const long long pairs = (long long)K * (D / 2);
        hipLaunchKernelGGL(vq_split_codebook_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, E, lde, K, D, Ex);
    }
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, E, lde, K, D, enorm, 1.f);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, Z, ldz, R, D, znorm, 1.f);
    hipLaunchKernelGGL(code_hash_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, E, lde, K, D, hash);
    hipLaunchKernelGGL(canon_init_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, keys, minidx, T);
    hipLaunchKernelGGL(canon_insert_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, hash, K, keys, minidx, T);
    hipLaunchKernelGGL(canon_lookup_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, hash, E, lde, K, D, keys, minidx, T, canon);
    VqProfRec rec{};
    if (g_vq_prof_on) {
        rec.flops = 2.0 * R * (double)K * D;
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess ||
            hipEventRecord(rec.e0, stream) != hipSuccess) {
            vqh_set_error("vqh_vq_nearest: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
    }
    int n_partials = nsplit;
    if (p3_form) {
        float* base = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace + need) + 15) & ~(uintptr_t)15);
        char* Zp = reinterpret_cast<char*>(base);
        char* Ep = Zp + (size_t)R * D * 6;
        float* nb_init = reinterpret_cast<float*>(Ep + (size_t)K * D * 6);
        n_partials = 2 * ns3;
        pbest = nb_init + K;
        psecond = pbest + (size_t)n_partials * R;
        pidx = reinterpret_cast<int*>(psecond + (size_t)n_partials * R);
        hipLaunchKernelGGL(vq_p3_init_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, enorm, canon, K, nb_init);
        const int rc = vqh_internal_nearest_p3(Z, ldz, E, lde, Zp, Ep, nb_init, pbest, psecond, pidx, R, K, D, ns3, stream);
        if (rc != VQH_OK) return rc;
    } else if (x3_form) {
        const int nt16 = D / 16;
        const int ct = (nt16 >= 16) ? 32 : 64;
        const size_t smem = (size_t)2 * ct * (6 * D + 16) + (size_t)4 * ct * sizeof(float);
#define VQ_X3(N)                                                                                                         \
    case N: {                                                                                                            \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_nearest_x3_kernel<N>),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }                         \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL(vq_nearest_x3_kernel<N>, dim3(row_blocks, nsplit), dim3(256), smem, stream, Z, ldz, Ex, enorm, \
                           canon, pbest, psecond, pidx, R, K, kchunk);                                                   \
    } break
        switch (nt16) { VQ_X3(4); VQ_X3(8); VQ_X3(16); default: break; }
#undef VQ_X3
    } else if (lds_form) {
        const int ct = (nt8 >= 32) ? 32 : 64;
        const size_t smem = (size_t)(2 * ct * (D + 4) + 4 * ct) * sizeof(float);
#define VQ_LDS(N)                                                                                                        \
    case N: {                                                                                                            \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_nearest_lds_kernel<N>),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }                         \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL(vq_nearest_lds_kernel<N>, dim3(row_blocks, nsplit), dim3(256), smem, stream, Z, ldz, E, lde,  \
                           enorm, canon, pbest, psecond, pidx, R, K, kchunk);                                            \
    } break
        switch (nt8) { VQ_LDS(1); VQ_LDS(2); VQ_LDS(4); VQ_LDS(8); VQ_LDS(16); VQ_LDS(32); default: break; }
#undef VQ_LDS
    } else {
        hipLaunchKernelGGL(vq_nearest_kernel, dim3(row_blocks, nsplit), dim3(64), 0, stream, Z, ldz, E, lde, enorm, canon, pbest,
                           psecond, pidx, R, K, D, kchunk);
    }
    if (g_vq_prof_on) {
        if (hipEventRecord(rec.e1, stream) != hipSuccess) {
            vqh_set_error("vqh_vq_nearest: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
        g_vq_prof.push_back(rec);
    }
    hipLaunchKernelGGL(vq_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, pbest, psecond, pidx, canon, n_partials, znorm,
                       idx_out, idx_offset, amb, bestval, R, rel_tol);
    if (g_vq_flags & 8) {                  // A/B: the one-wave-per-row refinement
        hipLaunchKernelGGL(vq_refine_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, Z, ldz, E, lde, idx_out, idx_offset, amb,
                           znorm, bestval, R, K, D, rel_tol);
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
    {
        unsigned char* after = Ex + (size_t)ex_floats * 4;
        int* rcount = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(after) + 15) & ~(uintptr_t)15);
        int* rlist = rcount + 4;
        double* pdist = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(rlist + R) + 7) & ~(uintptr_t)7);
        int* pidx2 = reinterpret_cast<int*>(pdist + (size_t)cap * CH);
        hipError_t e = hipMemsetAsync(rcount, 0, sizeof(int), stream);
        if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
        hipLaunchKernelGGL(vq_amb_compact_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, amb, R, rcount, rlist);
        hipLaunchKernelGGL(vq_refine_items_kernel, dim3(1024), dim3(256), 0, stream, Z, ldz, E, lde, rcount, rlist, znorm, bestval, K, D,
                           CH, cap, rel_tol, pdist, pidx2);
        hipLaunchKernelGGL(vq_refine_merge_kernel, dim3(256), dim3(256), 0, stream, Z, ldz, E, lde, rcount, rlist, znorm, bestval, K, D, CH,
                           cap, rel_tol, pdist, pidx2, idx_out, idx_offset);
    }
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_gather(const float* E, int lde, const long long* idx, int idx_offset, const float* rows_in,
                             int ldr, float* zq_level, float* res_out, int R, int D, hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && D > 0, "vqh_vq_gather: bad shape");
    if (R == 0) return VQH_OK;
    VQH_CHECK_ARG(E && idx && (zq_level || res_out) && (!res_out || rows_in), "vqh_vq_gather: null pointer");
    hipLaunchKernelGGL(vq_gather_kernel, dim3(blocks_for((long long)R * D)), dim3(256), 0, stream, E, lde, idx, idx_offset,
                       rows_in, ldr, zq_level, res_out, R, D);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_finish(const float* zq_levels, int Q, const float* ze, int ldz, float* zq, float* zst, int R, int D,
                             hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && D > 0 && Q >= 1, "vqh_vq_finish: bad shape");
    if (R == 0) return VQH_OK;
    VQH_CHECK_ARG(zq_levels && ze && zq && zst, "vqh_vq_finish: null pointer");
    hipLaunchKernelGGL(vq_finish_kernel, dim3(blocks_for((long long)R * D)), dim3(256), 0, stream, zq_levels, Q, ze, ldz, zq,
                       zst, R, D);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

// cnt[k0..k0+Kn), sum[k0..k0+Kn, :] are overwritten.
// out[r] = valid[r] ? idx[r] : -1   (-1 matches no code in the statistics kernels: VectorQuantizerEMA(mask=...), :192-205)
__global__ void vq_mask_ids_kernel(const long long* __restrict__ idx, const unsigned char* __restrict__ valid,
                                   long long* __restrict__ out, int R) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) out[r] = valid[r] ? idx[r] : -1ll;
}

extern "C" int vqh_vq_mask_ids(const long long* idx, const unsigned char* valid, long long* out, int R, hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0, "vqh_vq_mask_ids: bad shape");
    if (R == 0) return VQH_OK;
    VQH_CHECK_ARG(idx && valid && out, "vqh_vq_mask_ids: null pointer");
    hipLaunchKernelGGL(vq_mask_ids_kernel, dim3(blocks_for(R)), dim3(256), 0, stream, idx, valid, out, R);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_segment_sum(const float* rows, int ldr, const long long* idx, int R, int D, int k0, int Kn,
                                  float* cnt, float* sum, float* workspace, long long workspace_floats,
                                  hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && D > 0 && D <= 1024 && Kn >= 0 && k0 >= 0, "vqh_vq_segment_sum: bad shape (D <= 1024)");
    if (Kn == 0) return VQH_OK;
    VQH_CHECK_ARG(rows && idx && cnt && sum, "vqh_vq_segment_sum: null pointer");
    const long long tbl = (long long)Kn * D + Kn;
    const int nblk = (R + 63) / 64;
    if (workspace && tbl <= 36864 && nblk > 0 && tbl * nblk <= workspace_floats) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_segment_table_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_set = true;
        }
        hipLaunchKernelGGL(vq_segment_table_kernel, dim3(nblk), dim3(256), (size_t)tbl * sizeof(float), stream, rows, ldr,
                           idx, R, D, k0, Kn, workspace);
        const long long n = (long long)Kn * D;
        hipLaunchKernelGGL(vq_table_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, stream, workspace, nblk,
                           tbl, n, sum + (size_t)k0 * D);
        hipLaunchKernelGGL(vq_table_reduce_kernel, dim3((Kn + 63) / 64), dim3(1024), 0, stream, workspace + n, nblk, tbl,
                           (long long)Kn, cnt + k0);
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
    // large tables, default: stable sort by code + segmented sum (see vq_sort_* / vq_seg_reduce_kernel)
    if (workspace && !(g_vq_flags & (4 | 16)) && (D & 3) == 0 && D <= 1024 && R >= 1 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0 &&
        (ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(sum) & 15) == 0) {
        const int G = (R + 63) / 64, nwaves = (R + SEG - 1) / SEG, nslots = 2 * nwaves;
        const long long Rp = ((long long)R + 3) / 4 * 4;
        // workspace: 4 int arrays [Rp] (keys / rows, ping-pong) + counts [G][128] + totals [128] + ranks (Rp bytes) + partial slots
        const long long need_i = 4 * Rp + (long long)G * SORT_NB + SORT_NB + Rp / 4 + 2LL * nslots + (long long)nslots * D + 8;
        if (need_i <= workspace_floats) {
            int* keyA = reinterpret_cast<int*>(workspace);
            int* keyB = keyA + Rp;
            int* rowA = keyB + Rp;
            int* rowB = rowA + Rp;
            int* counts = rowB + Rp;
            int* totals = counts + (size_t)G * SORT_NB;
            unsigned char* rank = reinterpret_cast<unsigned char*>(totals + SORT_NB);
            int* pkey = reinterpret_cast<int*>(rank + Rp);
            float* pcnt = reinterpret_cast<float*>(pkey + nslots);
            float* pvec = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(pcnt + nslots) + 15) & ~(uintptr_t)15);
            int nbits = 1;
            while ((1 << nbits) <= Kn) ++nbits;                      // keys 0 .. Kn (Kn = invalid) need nbits bits
            const int passes = (nbits + SORT_BITS - 1) / SORT_BITS;
            hipError_t e = hipMemsetAsync(cnt + k0, 0, sizeof(float) * (size_t)Kn, stream);
            if (e == hipSuccess) e = hipMemsetAsync(sum + (size_t)k0 * D, 0, sizeof(float) * (size_t)Kn * D, stream);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            const int* kin = nullptr;
            const int* rin = nullptr;
            int* kout = keyB;
            int* rout = rowB;
            int* kcur = keyA;                                       // pass 0 writes the extracted keys here
            for (int p = 0; p < passes; ++p) {
                hipLaunchKernelGGL(vq_sort_hist_kernel, dim3((G + 3) / 4), dim3(256), 0, stream, idx, kin, R, k0, Kn, p * SORT_BITS, kcur,
                                   counts, rank);
                hipLaunchKernelGGL(vq_sort_scan_kernel, dim3(SORT_NB), dim3(1024), 0, stream, counts, G, totals);
                hipLaunchKernelGGL(vq_sort_scatter_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, p == 0 ? kcur : kin, rin, R,
                                   p * SORT_BITS, counts, totals, rank, kout, rout);
                kin = kout; rin = rout;
                kout = (kout == keyB) ? keyA : keyB;
                rout = (rout == rowB) ? rowA : rowB;
            }
            const int nq = (D + 255) / 256;
#define SEGR(N)                                                                                                                   \
    case N:                                                                                                                       \
        hipLaunchKernelGGL((vq_seg_reduce_kernel<N>), dim3((nwaves + 3) / 4), dim3(256), 0, stream, rows, ldr, kin, rin, R, D, k0, Kn, \
                           cnt, sum, pkey, pcnt, pvec);                                                                           \
        hipLaunchKernelGGL((vq_seg_fixup_kernel<N>), dim3((nslots + 3) / 4), dim3(256), 0, stream, pkey, pcnt, pvec, nslots, D, k0, cnt, \
                           sum);                                                                                                  \
        break
            switch (nq) { SEGR(1); SEGR(2); SEGR(3); default: SEGR(4); }
#undef SEGR
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
    }
    // (row chunk, code range) workgroups with an LDS table per range (see vq_segment_range_kernel): flag bit 4 selects it (A/B)
    if (workspace && !(g_vq_flags & 4) && D >= 8 && D <= 1024 && R >= 4096) {
        int KR = 1;
        while (KR * 2 * (long long)(D + 1) <= 32768 && KR * 2 <= Kn) KR *= 2;       // table <= 128 KB
        const int ranges = (Kn + KR - 1) / KR;
        int chunks = (1024 + ranges - 1) / ranges;                                  // ~1024 workgroups
        const int max_chunks = (R + 2047) / 2048;                                   // >= 2048 rows per chunk
        if (chunks > max_chunks) chunks = max_chunks;
        while (chunks > 1 && (long long)chunks * tbl > workspace_floats) --chunks;
        if (chunks >= 1 && (long long)chunks * tbl <= workspace_floats && (reinterpret_cast<uintptr_t>(rows) & 15) == 0 && (ldr & 3) == 0) {
            const int rpc = ((R + chunks - 1) / chunks + 63) / 64 * 64;
            chunks = (R + rpc - 1) / rpc;
            static bool attr_set2 = false;
            if (!attr_set2) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_segment_range_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
                attr_set2 = true;
            }
            hipLaunchKernelGGL(vq_segment_range_kernel, dim3(ranges, chunks), dim3(1024), (size_t)(KR * (D + 1) + 16 * 256) * sizeof(float), stream,
                               rows, ldr, idx, R, D, k0, Kn, KR, rpc, workspace);
            const long long n = (long long)Kn * D;
            hipLaunchKernelGGL(vq_table_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, stream, workspace, chunks, tbl, n,
                               sum + (size_t)k0 * D);
            hipLaunchKernelGGL(vq_table_reduce_kernel, dim3((Kn + 63) / 64), dim3(1024), 0, stream, workspace + n, chunks, tbl,
                               (long long)Kn, cnt + k0);
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
    }
#define SEG(V) hipLaunchKernelGGL((vq_segment_sum_kernel<V>), dim3(Kn), dim3(256), 0, stream, rows, ldr, idx, R, D, k0, cnt, sum)
    if (D <= 64) SEG(1);
    else if (D <= 128) SEG(2);
    else if (D <= 256) SEG(4);
    else if (D <= 512) SEG(8);
    else SEG(16);
#undef SEG
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_ema_apply(const float* cnt, const float* sum, float* ema_cnt, float* ema_emb, float* emb, int K,
                                int D, float decay, float one_minus_decay, float eps, hipStream_t stream) {
    VQH_CHECK_ARG(K > 0 && D > 0, "vqh_vq_ema_apply: bad shape");
    VQH_CHECK_ARG(cnt && sum && ema_cnt && ema_emb && emb, "vqh_vq_ema_apply: null pointer");
    // embeddings first (they read the OLD ema_cnt and recompute the new one), then the counts
    hipLaunchKernelGGL(vq_ema_apply_kernel, dim3(blocks_for((long long)K * D)), dim3(256), 0, stream, cnt, sum, ema_cnt,
                       ema_emb, emb, K, D, decay, one_minus_decay, eps);
    hipLaunchKernelGGL(vq_ema_cnt_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, cnt, ema_cnt, K, decay,
                       one_minus_decay);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_usage_stats(const float* usage, int K, float n_positions, float* ep_usage, float* ep_cnt,
                                  float* stats, hipStream_t stream) {
    VQH_CHECK_ARG(K > 0 && usage && stats, "vqh_vq_usage_stats: bad argument");
    hipLaunchKernelGGL(vq_usage_stats_kernel, dim3(1), dim3(256), 0, stream, usage, K, n_positions, ep_usage, ep_cnt, stats);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_reinit(const float* usage, float threshold, const long long* pick, const float* rows, int ldr,
                             float* emb, float* ema_emb, float* ema_cnt, int K, int D, hipStream_t stream) {
    VQH_CHECK_ARG(K > 0 && D > 0 && usage && pick && rows && emb && ema_emb && ema_cnt, "vqh_vq_reinit: bad argument");
    hipLaunchKernelGGL(vq_reinit_kernel, dim3(K), dim3(64), 0, stream, usage, threshold, pick, rows, ldr, emb, ema_emb,
                       ema_cnt, K, D);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

// out[r] = scale * ||X[r,:]||^2   (soft-VQ logits need -||e_k||^2 / tau as a column bias, reference :838-840)
extern "C" int vqh_row_sqnorm(const float* X, int ld, int rows, int D, float* out, float scale, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && D > 0 && ld >= D && X && out, "vqh_row_sqnorm: bad argument");
    if (rows == 0) return VQH_OK;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, X, ld, rows, D, out, scale);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
