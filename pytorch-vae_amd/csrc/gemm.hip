// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 155 TF/s chip peak).
//
//   C[M,N] = epilogue( opA(A)[M,K] . opB(B)[K,N] )
//
// One kernel template covers the three products of a Linear layer (reference call sites: every
// nn.Linear / MHA projection of /root/reference/models/vq_vae.py:455-533):
//   forward  Y  = X . W^T      : A k-contiguous [M,K], B k-contiguous [N,K]   (A_KC, B_KC)
//   dgrad    dX = dY . W       : A k-contiguous [M,K], B stored [K,N]          (A_KC, !B_KC)
//   wgrad    dW = dY^T . X     : A stored [K,M],       B stored [K,N]          (!A_KC, !B_KC), split-K
//
// Block = 256 threads = 4 waves (2x2), tile 128x128x32, each wave 64x64 = 2x2 MFMA tiles of 32x32.
// Operand tiles are staged global -> registers -> LDS (double buffered, one barrier per K-step; the
// next tile's global loads are issued before the MFMA block so HBM/L2 latency hides under it).
// LDS images:
//   k-contiguous source : [row][BK+4] floats; a lane reads 4 consecutive k with one ds_read_b128
//                         (row stride 36 dwords => the 16-lane groups of ds_read_b128 hit 64
//                         distinct banks); lane half h owns k = 8t+4h+j, the same permutation of
//                         the reduction index is used for A and B so the sum is unchanged.
//   row-contiguous source: [k][rows] floats; a lane reads [8t+4h+j][row0 + lane&31] with
//                         ds_read_b32 (32 consecutive dwords per half: conflict free).
#include "common.h"
#include <type_traits>
#include <utility>
#include <vector>
#include <algorithm>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int BK = 32;                   // granularity of split-K chunks (both kernel variants divide it)
template <int BKT> struct Tile {
    static constexpr int KC_LD = BKT + 4;           // padded row of a k-contiguous LDS image
    static constexpr int A_TILE = BM * KC_LD;       // floats per operand tile buffer (upper bound of both layouts)
    static constexpr int NLD = BKT / 8;             // float4 loads per thread and operand tile
};

enum EpiMode {
    EPI_LINEAR = 0,        // v = acc + bias
    EPI_RELU_DROP = 1,     // v = dropout(relu(acc + bias))
    EPI_GELU = 2,          // aux_out = acc + bias ; v = gelu(aux_out)
    EPI_DROP_RESID = 3,    // v = aux_in + dropout(acc + bias)
    EPI_SIGMOID = 4,       // v = sigmoid(acc + bias)
    EPI_MUL_POSMASK = 5,   // v = acc * (aux_in > 0 ? drop.scale : 0)      (relu+dropout backward)
    EPI_MUL_GELUGRAD = 6,  // v = acc * gelu'(aux_in)
    EPI_MUL_SIGGRAD = 7,   // v = acc * aux_in * (1 - aux_in)
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int kchunk;            // K range handled by one blockIdx.z (multiple of BK)
    int vecA, vecB;        // 16-byte vector loads legal for this operand
    int vecC;              // 16-byte epilogue accesses legal (C, aux, bias aligned; ldc, ldaux multiples of 4)
    float* ws;             // split-K slabs [gridDim.z][M][N] (raw partial sums) or null
    float* rowsum;         // optional: rowsum[m] = sum_k opA(A)[m,k]  (bias gradient riding on the weight-gradient GEMM;
                           // only for the [K,M]-stored A layout). With split-K: partials at rowsum_ws[z*M + m].
    float* rowsum_ws;
    // epilogue
    int mode;
    const float* bias;
    const float* aux_in;
    float* aux_out;
    int ldaux;
    float beta;
    DropCfg drop;
    int flags;             // tuning/diagnostic knobs (vqh_gemm_set_flags): 1 = XCD-aware tile order,
                           // 2 = skip epilogue stores (timing only), 4 = skip global loads after the first tile (timing only)
};

__device__ __forceinline__ float epilogue_value(const GemmArgs& g, float acc, int row, int col, unsigned dkey) {
    float v = acc;
    if (g.mode <= EPI_SIGMOID) {
        if (g.bias) v += g.bias[col];
        switch (g.mode) {
            case EPI_RELU_DROP:
                v = fmaxf(v, 0.f);
                if (g.drop.p > 0.f) v *= drop_keep(g.drop, dkey, (unsigned long long)row * g.N + col);
                break;
            case EPI_GELU:
                g.aux_out[(size_t)row * g.ldaux + col] = v;
                v = gelu_erf(v);
                break;
            case EPI_DROP_RESID:
                if (g.drop.p > 0.f) v *= drop_keep(g.drop, dkey, (unsigned long long)row * g.N + col);
                v += g.aux_in[(size_t)row * g.ldaux + col];
                break;
            case EPI_SIGMOID:
                v = 1.f / (1.f + __expf(-v));
                break;
            default:
                break;
        }
    } else {
        const float a = g.aux_in[(size_t)row * g.ldaux + col];
        if (g.mode == EPI_MUL_POSMASK) v = (a > 0.f) ? v * g.drop.scale : 0.f;
        else if (g.mode == EPI_MUL_GELUGRAD) v *= gelu_erf_grad(a);
        else v *= a * (1.f - a);
    }
    return v;
}

// Load this thread's share (4 x float4) of one operand tile into registers.
//   KC  : tile element (r, k) lives at src[(row0 + r) * ld + k]
//   !KC : tile element (r, k) lives at src[k * ld + row0 + r]
template <bool KC, int BKT>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int ld, int rows, int row0, int k0,
                                          int kend, bool vec, int tid, f32x4 (&regs)[BKT / 8]) {
    constexpr int CPR = BKT / 4;            // float4 chunks per k-contiguous row
    constexpr int RPI = 256 / CPR;          // rows covered per pass
#pragma unroll
    for (int i = 0; i < BKT / 8; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int r = row0 + (tid / CPR) + RPI * i;
            const int k = k0 + (tid % CPR) * 4;
            if (r < rows) {
                const float* p = src + (size_t)r * ld + k;
                if (vec && k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) v[e] = p[e];
                }
            }
        } else {
            const int k = k0 + (tid >> 5) + 8 * i;
            const int r = row0 + (tid & 31) * 4;
            if (k < kend) {
                const float* p = src + (size_t)k * ld + r;
                if (vec && r + 3 < rows) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (r + e < rows) v[e] = p[e];
                }
            }
        }
        regs[i] = v;
    }
}

// Same thread->element map as load_tile, for tiles that are completely inside the matrix and 16-byte loadable
// (the overwhelmingly common case): no per-lane predicates, no scalar fallback -> straight-line loads.
template <bool KC, int BKT>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ src, int ld, int row0, int k0, int tid,
                                               f32x4 (&regs)[BKT / 8]) {
    constexpr int CPR = BKT / 4, RPI = 256 / CPR;
    if (KC) {
        const float* p = src + (size_t)(row0 + tid / CPR) * ld + k0 + (tid % CPR) * 4;
#pragma unroll
        for (int i = 0; i < BKT / 8; ++i) regs[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(RPI * i) * ld);
    } else {
        const float* p = src + (size_t)(k0 + (tid >> 5)) * ld + row0 + (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < BKT / 8; ++i) regs[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(8 * i) * ld);
    }
}

template <bool KC, int BKT>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, int tid, const f32x4 (&regs)[BKT / 8]) {
    constexpr int CPR = BKT / 4, RPI = 256 / CPR, KC_LD = BKT + 4;
#pragma unroll
    for (int i = 0; i < BKT / 8; ++i) {
        if (KC) {
            const int r = (tid / CPR) + RPI * i;
            *reinterpret_cast<f32x4*>(lds + r * KC_LD + (tid % CPR) * 4) = regs[i];
        } else {
            const int k = (tid >> 5) + 8 * i;
            *reinterpret_cast<f32x4*>(lds + k * BM + (tid & 31) * 4) = regs[i];
        }
    }
}

// Fragment for MFMA group t (k = 8t .. 8t+7): 4 values per lane, value j belongs to k = 8t + 4h + j.
template <bool KC, int BKT>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ lds, int row, int t, int h) {
    constexpr int KC_LD = BKT + 4;
    if (KC) {
        return *reinterpret_cast<const f32x4*>(lds + row * KC_LD + 8 * t + 4 * h);
    } else {
        f32x4 v;
        const float* p = lds + (8 * t + 4 * h) * BM + row;
        v[0] = p[0];
        v[1] = p[BM];
        v[2] = p[2 * BM];
        v[3] = p[3 * BM];
        return v;
    }
}

// MODE >= 0: epilogue specialised at compile time (small enough to unroll fully, which the aux prefetch needs: its 16
// registers are indexed by the unrolled loop counters); MODE < 0: generic kernel, epilogue selected by g.mode at run
// time and unrolled 4x (a fully unrolled generic epilogue cost every GEMM ~9 %: 292 -> 320 us on 16384x2048x512).
template <bool A_KC, bool B_KC, int BKT, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma(const GemmArgs g) {
    constexpr int A_TILE = Tile<BKT>::A_TILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // stage s: A tile at smem + 2*s*A_TILE, B tile right behind it
#define AS(s) (smem + (s) * 2 * A_TILE)
#define BS(s) (smem + (s) * 2 * A_TILE + A_TILE)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int tiles_n = (g.N + BN - 1) / BN;
    int tile = blockIdx.x, zsplit = blockIdx.z;
    if (g.flags & 1) {
        // Workgroups are dealt round-robin to the 8 XCDs in linear-id order (x fastest, then z), so ids b and b+8
        // share an XCD.  Give each XCD a contiguous run of (split, tile) pairs: tiles that share A panels / weight
        // columns -- and, for split-K, all tiles of one K-chunk -- then hit the same private L2 (bijective for any grid).
        const int nx = gridDim.x, nt = nx * gridDim.z, lin = blockIdx.z * nx + blockIdx.x;
        const int q = nt >> 3, r = nt & 7, x = lin & 7, j = lin >> 3;
        const int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        tile = v % nx;
        zsplit = v / nx;
    }
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int kbeg = zsplit * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[BKT / 8], rb[BKT / 8];
    const int nk = (kend - kbeg + BKT - 1) / BKT;
    const bool do_rowsum = (g.rowsum != nullptr) && (n0 == 0);     // one column of tiles carries the row sums
    float rs_acc = 0.f;
    const bool inA = g.vecA && (m0 + BM <= g.M), inB = g.vecB && (n0 + BN <= g.N);   // wave-uniform
#define LOAD_AB(k0_)                                                                                          \
    {                                                                                                         \
        const bool fullk = ((k0_) + BKT <= kend);                                                             \
        if (inA && fullk) load_tile_fast<A_KC, BKT>(g.A, g.lda, m0, (k0_), tid, ra);                          \
        else load_tile<A_KC, BKT>(g.A, g.lda, g.M, m0, (k0_), kend, g.vecA, tid, ra);                         \
        if (inB && fullk) load_tile_fast<B_KC, BKT>(g.B, g.ldb, n0, (k0_), tid, rb);                          \
        else load_tile<B_KC, BKT>(g.B, g.ldb, g.N, n0, (k0_), kend, g.vecB, tid, rb);                         \
    }
    // The epilogue's second input (residual stream / saved activation: 64 floats per lane, laid out as the epilogue
    // reads them) is fetched two K-steps before the end so its HBM latency hides under the last MFMAs instead of
    // stalling the epilogue (+15 % on the FFN dgrad, +8 % on K=512 out-projections when loaded in place).
    const bool pre_aux = (MODE == EPI_DROP_RESID || MODE > EPI_SIGMOID) && g.aux_in && !g.ws && g.vecC && nk > 0 &&
                         (m0 + BM <= g.M) && (n0 + BN <= g.N) && !(g.flags & 16);
    const int kpre = nk >= 2 ? nk - 2 : 0;
    f32x4 auxr[16];

    // FASTL: both operand tiles in bounds, 16-byte addressable and K a multiple of the K-step (block-uniform): a
    // branch-free steady state (unconditional fast loads / LDS stores, last step peeled).  Everything else takes the
    // checked loop.  Keeping the two apart (like the epilogue) keeps the hot loop compact and contiguous.
    auto mainloop = [&](auto fast_c) {
        constexpr bool FASTL = decltype(fast_c)::value;
        auto load_ab = [&](int k0) {
            if (FASTL) {
                load_tile_fast<A_KC, BKT>(g.A, g.lda, m0, k0, tid, ra);
                load_tile_fast<B_KC, BKT>(g.B, g.ldb, n0, k0, tid, rb);
            } else if (!(g.flags & 4)) {
                LOAD_AB(k0)
            }
        };
        auto prefetch_aux = [&]() {
            const float* ap = g.aux_in + (size_t)(m0 + wm * 64 + (lane >> 4)) * g.ldaux + n0 + wn * 64 + (lane & 15) * 4;
#pragma unroll
            for (int q = 0; q < 16; ++q) auxr[q] = *reinterpret_cast<const f32x4*>(ap + (size_t)(q * 4) * g.ldaux);
        };
        auto compute = [&](int cur) {
            const float* a_s = AS(cur);
            const float* b_s = BS(cur);
#pragma unroll
            for (int t = 0; t < BKT / 8; ++t) {
                f32x4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = read_frag<A_KC, BKT>(a_s, wm * 64 + i * 32 + l31, t, h);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_KC, BKT>(b_s, wn * 64 + j * 32 + l31, t, h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
            }
            if (!A_KC && do_rowsum && tid < BM) {
                // A image is [k][m]: column tid of the tile summed over this K-step (conflict-free: 32 consecutive dwords)
#pragma unroll
                for (int k = 0; k < BKT; ++k) rs_acc += a_s[k * BM + tid];
            }
        };
        auto read_frags = [&](int cur, int t, f32x4 (&fa)[2], f32x4 (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = read_frag<A_KC, BKT>(AS(cur), wm * 64 + i * 32 + l31, t, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = read_frag<B_KC, BKT>(BS(cur), wn * 64 + j * 32 + l31, t, h);
        };
        auto mfma_step = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        };
        auto rowsum_step = [&](int cur) {
            if (!A_KC && do_rowsum && tid < BM) {
                const float* a_s = AS(cur);
#pragma unroll
                for (int k = 0; k < BKT; ++k) rs_acc += a_s[k * BM + tid];
            }
        };
        if (nk > 0) {
            if (FASTL) {
                load_ab(kbeg);
            } else {
                LOAD_AB(kbeg)
            }
            store_tile<A_KC, BKT>(AS(0), tid, ra);
            store_tile<B_KC, BKT>(BS(0), tid, rb);
        }
        __syncthreads();
        if (FASTL && !(g.flags & 32)) {
            // Software pipeline across the barrier: the fragments of the LAST sub-step of tile kt are already in
            // registers when the next tile is written to LDS and the block synchronises, and the first fragments of
            // tile kt+1 are requested right after the barrier, so barrier skew and LDS latency sit under 16 MFMAs
            // instead of in front of the next tile's first one.
            constexpr int NT = BKT / 8;
            f32x4 fa[2][2], fb[2][2];
            read_frags(0, 0, fa[0], fb[0]);
            for (int kt = 0; kt + 1 < nk; ++kt) {
                const int cur = kt & 1;
                load_ab(kbeg + (kt + 1) * BKT);
                if (pre_aux && kt == kpre) prefetch_aux();
#pragma unroll
                for (int t = 0; t + 1 < NT; ++t) {
                    read_frags(cur, t + 1, fa[(t + 1) & 1], fb[(t + 1) & 1]);
                    mfma_step(fa[t & 1], fb[t & 1]);
                }
                rowsum_step(cur);
                store_tile<A_KC, BKT>(AS(cur ^ 1), tid, ra);
                store_tile<B_KC, BKT>(BS(cur ^ 1), tid, rb);
                __syncthreads();
                read_frags(cur ^ 1, 0, fa[0], fb[0]);
                __builtin_amdgcn_sched_barrier(0);      // or the scheduler hoists these MFMAs back above the barrier
                mfma_step(fa[(NT - 1) & 1], fb[(NT - 1) & 1]);
            }
            {
                const int cur = (nk - 1) & 1;
                if (pre_aux && nk == 1) prefetch_aux();
#pragma unroll
                for (int t = 0; t + 1 < NT; ++t) {
                    read_frags(cur, t + 1, fa[(t + 1) & 1], fb[(t + 1) & 1]);
                    mfma_step(fa[t & 1], fb[t & 1]);
                }
                rowsum_step(cur);
                mfma_step(fa[(NT - 1) & 1], fb[(NT - 1) & 1]);
                __syncthreads();
            }
            return;
        }
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int cur = kt & 1;
            load_ab(kbeg + (kt + 1) * BKT);
            if (pre_aux && kt == kpre) prefetch_aux();
            compute(cur);
            store_tile<A_KC, BKT>(AS(cur ^ 1), tid, ra);
            store_tile<B_KC, BKT>(BS(cur ^ 1), tid, rb);
            __syncthreads();
        }
        if (nk > 0) {
            if (pre_aux && nk == 1) prefetch_aux();
            compute((nk - 1) & 1);
            __syncthreads();
        }
    };
    if (nk > 0 && inA && inB && ((kend - kbeg) % BKT == 0) && !(g.flags & 4)) mainloop(std::true_type{});
    else mainloop(std::false_type{});
    if (!A_KC && do_rowsum && tid < BM && m0 + tid < g.M) {
        if (g.rowsum_ws) g.rowsum_ws[(size_t)zsplit * g.M + m0 + tid] = rs_acc;
        else g.rowsum[m0 + tid] = (g.beta != 0.f) ? g.beta * g.rowsum[m0 + tid] + rs_acc : rs_acc;
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    // Accumulator layout: lane owns column (lane&31), rows (r&3) + 8*(r>>2) + 4*h of each 32x32 tile, i.e.
    // 4-byte accesses 128 B apart.  Stage the wave's 64x64 result through its (now idle) share of the LDS
    // and let each lane handle 4 consecutive columns: 16-byte global stores / aux loads, 16 per lane
    // instead of 64 (the scattered form ran at ~1 TB/s and cost ~30 % of a K=512 GEMM).
    unsigned dkey = 0;
    if (g.drop.p > 0.f && g.drop.rng_state) dkey = drop_key(g.drop, g.drop.rng_state[0], g.drop.rng_state[1]);
    if (g.flags & 2) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) g.C[0] = s;   // keep the accumulators live
        return;
    }
    constexpr int EP_LD = 64;                         // 32-row half tile per wave: 8 KiB, conflict-free both ways
    float* ep = smem + wave * (32 * EP_LD);
    float* slab = g.ws ? g.ws + (size_t)zsplit * g.M * g.N : nullptr;
    const int c4 = (lane & 15) * 4;
    const int col = n0 + wn * 64 + c4;
    const bool vecC = slab ? ((g.N & 3) == 0) : g.vecC;
    // Whole tile in bounds and 16-byte addressable (block-uniform): vector-only epilogue, no per-element fallback
    // code in the unrolled copies (the mixed form made the kernels 80-144 KB, most of it never executed).
    // The generic kernel (MODE < 0) keeps a fast path for the linear epilogue only; the fused modes that matter have
    // specialised kernels, the rest (and ragged tiles) take the compact per-element loop.
    const bool tile_fast = vecC && (m0 + BM <= g.M) && (n0 + BN <= g.N) && (MODE >= 0 || g.mode == EPI_LINEAR);
    auto epilogue = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        constexpr int EPI_UNROLL = FAST ? (MODE >= 0 ? 8 : 4) : 1;
        const int mode = (MODE >= 0) ? MODE : (FAST ? (int)EPI_LINEAR : g.mode);
    #pragma unroll
        for (int half = 0; half < 2; ++half) {
        __builtin_amdgcn_wave_barrier();
    #pragma unroll
        for (int j = 0; j < 2; ++j)
    #pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EP_LD + j * 32 + l31] = acc[half][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
    #pragma unroll EPI_UNROLL
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + (lane >> 4);
            const int row = m0 + wm * 64 + half * 32 + rl;
            if (!FAST && (row >= g.M || col >= g.N)) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + rl * EP_LD + c4);
            if (slab) {
                float* dst = slab + (size_t)row * g.N + col;
                if (FAST || (vecC && col + 3 < g.N)) *reinterpret_cast<f32x4*>(dst) = v;
                else
                    for (int e = 0; e < 4; ++e) if (col + e < g.N) dst[e] = v[e];
                continue;
            }
            const bool full = FAST || (vecC && (col + 3 < g.N));
            float* cptr = g.C + (size_t)row * g.ldc + col;
            if (full) {
                // ---- vector path: 4 consecutive columns
                if (mode <= EPI_SIGMOID) {
                    if (g.bias) { const f32x4 bb = *reinterpret_cast<const f32x4*>(g.bias + col); v += bb; }
                    if (mode == EPI_RELU_DROP || mode == EPI_DROP_RESID) {
                        if (mode == EPI_RELU_DROP)
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        if (g.drop.p > 0.f) {
                            float f[4];
                            drop4(g.drop, dkey, ((unsigned long long)row * g.N + col) >> 2, f);
                            for (int e = 0; e < 4; ++e) v[e] *= f[e];
                        }
                        if (mode == EPI_DROP_RESID)
                            v += (FAST && pre_aux) ? auxr[half * 8 + it] : *reinterpret_cast<const f32x4*>(g.aux_in + (size_t)row * g.ldaux + col);
                    } else if (mode == EPI_GELU) {
                        *reinterpret_cast<f32x4*>(g.aux_out + (size_t)row * g.ldaux + col) = v;
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    } else if (mode == EPI_SIGMOID) {
                        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                    }
                } else {
                    const f32x4 a = (FAST && pre_aux) ? auxr[half * 8 + it] : *reinterpret_cast<const f32x4*>(g.aux_in + (size_t)row * g.ldaux + col);
                    if (mode == EPI_MUL_POSMASK)
                        for (int e = 0; e < 4; ++e) v[e] = (a[e] > 0.f) ? v[e] * g.drop.scale : 0.f;
                    else if (mode == EPI_MUL_GELUGRAD)
                        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(a[e]);
                    else
                        for (int e = 0; e < 4; ++e) v[e] *= a[e] * (1.f - a[e]);
                }
                if (g.beta != 0.f) { const f32x4 old = *reinterpret_cast<const f32x4*>(cptr); v += g.beta * old; }
                *reinterpret_cast<f32x4*>(cptr) = v;
            } else {
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= g.N) break;
                    float x = epilogue_value(g, v[e], row, col + e, dkey);
                    if (g.beta != 0.f) x += g.beta * cptr[e];
                    cptr[e] = x;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    };
    if (tile_fast) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// Sum split-K slabs, add bias, C = beta*C + sum (+ the row-sum partials of the fused bias gradient).
__global__ void splitk_reduce(const float* __restrict__ ws, int S, int M, int N, float* __restrict__ C, int ldc,
                              const float* __restrict__ bias, float beta, const float* __restrict__ rs_ws,
                              float* __restrict__ rowsum) {
    const size_t total = (size_t)M * N;
    if (rowsum) {      // fused bias gradient: M outputs spread over the first ceil(M/blockDim) blocks
        const int m = blockIdx.x * blockDim.x + threadIdx.x;
        if (m < M) {
            float s = 0.f;
            for (int z = 0; z < S; ++z) s += rs_ws[(size_t)z * M + m];
            rowsum[m] = (beta != 0.f) ? beta * rowsum[m] + s : s;
        }
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += ws[(size_t)z * total + e];
        const int row = (int)(e / N), col = (int)(e % N);
        if (bias) s += bias[col];
        float* c = C + (size_t)row * ldc + col;
        if (beta != 0.f) s += beta * (*c);
        *c = s;
    }
}

// 16-byte form (N % 4 == 0, aligned C / bias): one output quad per thread, slabs read 8 at a time so that 8 independent
// 16-byte loads are in flight per lane, added in slab order (same sums as the scalar kernel).
__global__ __launch_bounds__(256) void splitk_reduce_vec(const float* __restrict__ ws, int S, int M, int N,
                                                         float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                         float beta, const float* __restrict__ rs_ws,
                                                         float* __restrict__ rowsum) {
    const size_t total = (size_t)M * N, quads = total >> 2;
    if (rowsum) {
        const int m = blockIdx.x * blockDim.x + threadIdx.x;
        if (m < M) {
            float s = 0.f;
            for (int z = 0; z < S; ++z) s += rs_ws[(size_t)z * M + m];
            rowsum[m] = (beta != 0.f) ? beta * rowsum[m] + s : s;
        }
    }
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        const float* src = ws + q * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= S; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * total);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < S; ++z) s += *reinterpret_cast<const f32x4*>(src + (size_t)z * total);
        const size_t e = q * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        if (bias) s += *reinterpret_cast<const f32x4*>(bias + col);
        float* c = C + (size_t)row * ldc + col;
        if (beta != 0.f) s += beta * *reinterpret_cast<const f32x4*>(c);
        *reinterpret_cast<f32x4*>(c) = s;
    }
}

// The split-K reductions of a grouped weight-gradient launch in ONE kernel: blockIdx.y selects the product.
struct RedGroupArgs {
    int n;
    struct Item { const float* ws; int splits, M, N; float* C; int ldc; const float* rs_ws; float* rowsum; } r[8];
};
__global__ __launch_bounds__(256) void splitk_reduce_group(const RedGroupArgs G) {
    const RedGroupArgs::Item& it = G.r[blockIdx.y];
    const int S = it.splits, M = it.M, N = it.N;
    const size_t total = (size_t)M * N, quads = total >> 2;
    if (it.rowsum) {
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int z = 0; z < S; ++z) s += it.rs_ws[(size_t)z * M + m];
            it.rowsum[m] = s;
        }
    }
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        const float* src = it.ws + q * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= S; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * total);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < S; ++z) s += *reinterpret_cast<const f32x4*>(src + (size_t)z * total);
        const size_t e = q * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        *reinterpret_cast<f32x4*>(it.C + (size_t)row * it.ldc + col) = s;
    }
}

// ---- skinny shapes: the reconstruction heads (512 -> 3, models/vq_vae.py:763-764) ---------------------------------------
// A 128x128 MFMA tile with 3 live columns wastes 98 % of the matrix work (60-90 us per launch); these shapes are pure
// streaming of the wide operand (33 MB at C2) and run at HBM speed in simple kernels instead.  SK_MAX = largest small
// dimension handled.
constexpr int SK_MAX = 8;

// C[M,N] = A[M,K] . B[N,K]^T + bias (+ beta C), N <= SK_MAX: a wave per row, lanes over K, N wave reductions.
template <int N>     // compile-time small dimension: with a run-time bound every load sits behind its own branch
__global__ __launch_bounds__(256) void skinny_n_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ C, int ldc,
                                                       const float* __restrict__ bias, float beta, int M, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        float acc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = 0.f;
        const float* ar = A + (size_t)row * lda;
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ar + k);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(B + (size_t)n * ldb + k);
                    acc[n] += av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2] + av[3] * bv[3];
                }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
                float v = acc[n];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0) {
                    if (bias) v += bias[n];
                    float* c = C + (size_t)row * ldc + n;
                    *c = (beta != 0.f) ? beta * (*c) + v : v;
                }
            }
    }
}

// C[M,N] = A[M,K] . B[K,N] (+ beta C), K <= SK_MAX: one output quad per thread.
template <int K>
__global__ __launch_bounds__(256) void skinny_k_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ C, int ldc, float beta, int M, int N) {
    const int nq = N >> 2;
    const size_t total = (size_t)M * nq;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / nq), col = (int)(e % nq) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) v += A[(size_t)row * lda + k] * *reinterpret_cast<const f32x4*>(B + (size_t)k * ldb + col);
        float* c = C + (size_t)row * ldc + col;
        if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(c);
        *reinterpret_cast<f32x4*>(c) = v;
    }
}

// slab[z][m][n] = sum over the rows of chunk z of A[r,m] * B[r,n], rs[z][m] = sum A[r,m]   (M <= SK_MAX; A [K,M], B [K,N],
// N <= 1024); the chunks are summed by splitk_reduce like any split-K launch.  The 256 threads are 256/(N/4) row slices x
// N/4 column quads; the slices meet in LDS in a fixed order.
template <int M>
__global__ __launch_bounds__(256) void skinny_m_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ slab, float* __restrict__ rs, int N, int K,
                                                       int rows_per_block) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4 * M];           // [slice][m][N]
    __shared__ float red_rs[64][M];                                            // [slice][m] row sums of A
    const int r0 = blockIdx.x * rows_per_block, r1 = min(K, r0 + rows_per_block);
    const int nq = N >> 2, slices = 256 / nq;
    const int q = threadIdx.x % nq, sl = threadIdx.x / nq;
    f32x4 acc[M];
    float asum[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; asum[m] = 0.f; }
    if (sl < slices) {
        int r = r0 + sl;
        for (; r + 7 * slices < r1; r += 8 * slices) {        // 8 rows in flight
            f32x4 bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = *reinterpret_cast<const f32x4*>(B + (size_t)(r + u * slices) * ldb + q * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                        const float am = A[(size_t)(r + u * slices) * lda + m];
                        acc[m] += am * bv[u];
                        asum[m] += am;
                    }
        }
        for (; r < r1; r += slices) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(B + (size_t)r * ldb + q * 4);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                    const float am = A[(size_t)r * lda + m];
                    acc[m] += am * bv;
                    asum[m] += am;
                }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
                *reinterpret_cast<f32x4*>(red + ((size_t)sl * M + m) * N + q * 4) = acc[m];
                if (q == 0) red_rs[sl][m] = asum[m];
            }
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * M * N;
    for (int i = threadIdx.x; i < M * nq; i += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(red + (size_t)i * 4);
        for (int s2 = 1; s2 < slices; ++s2) v += *reinterpret_cast<const f32x4*>(red + (size_t)s2 * M * N + (size_t)i * 4);
        *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
    }
    if (rs && threadIdx.x < M) {
        float t = 0.f;
        for (int s2 = 0; s2 < slices; ++s2) t += red_rs[s2][threadIdx.x];
        rs[(size_t)blockIdx.x * M + threadIdx.x] = t;
    }
}

#include "gemm_dma.inc"
#include "gemm_p3.inc"

// ---- live per-kernel timing (bench.py): hipEvents around each main-kernel launch, on the launch stream ----------
struct ProfRec { int slot; double flops; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
constexpr int PROF_MODES = 9;                       // template MODE -1..7 -> column MODE+1
constexpr int PROF_FAMILIES = 7;                    // 0 = gemm_f32_mfma (128x128 tile), 1 = gemm_f32_dma (256x128 tile), 2 = gemm_f32_dma_group,
                                                    // 3 = gemm_f32_x3 (256x128 tile, bf16x3 split operands), 4 = gemm_f32_x3_group,
                                                    // 5 = gemm_p3 (pre-split plane operands, 16x16x32 MFMA), 6 = gemm_p3_group
inline int prof_slot(bool a_kc, bool b_kc, int mode_t, int family = 0) {
    return (family * 4 + (a_kc ? 2 : 0) + (b_kc ? 1 : 0)) * PROF_MODES + mode_t + 1;
}

template <bool A_KC, bool B_KC, int BKT, int MODE = -1>
int launch(const GemmArgs& g, int splits, hipStream_t stream) {
    static bool attr_set = false;
    const size_t smem = (size_t)4 * Tile<BKT>::A_TILE * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma<A_KC, B_KC, BKT, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            vqh_set_error(hipGetErrorString(e));
            return VQH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles, 1, splits);
    ProfRec rec{};
    if (g_prof_on) {
        rec.slot = prof_slot(A_KC, B_KC, MODE);
        rec.flops = 2.0 * g.M * g.N * g.K;
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess ||
            hipEventRecord(rec.e0, stream) != hipSuccess) {
            vqh_set_error("vqh_gemm: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL((gemm_f32_mfma<A_KC, B_KC, BKT, MODE>), grid, dim3(256), smem, stream, g);
    VQH_LAUNCH_CHECK();
    if (g_prof_on) {
        if (hipEventRecord(rec.e1, stream) != hipSuccess) {
            vqh_set_error("vqh_gemm: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
        g_prof.push_back(rec);
    }
    return VQH_OK;
}

template <bool A_KC, bool B_KC, int MODE, bool X3>
int launch_dma_impl(const GemmArgs& g, int splits, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = X3 ? &gemm_f32_x3<A_KC, B_KC, MODE> : &gemm_f32_dma<A_KC, B_KC, MODE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, dma::LDS_BYTES);
        if (e != hipSuccess) {
            vqh_set_error(hipGetErrorString(e));
            return VQH_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid((g.M / dma::TBM) * (g.N / dma::TBN), 1, splits);
    ProfRec rec{};
    if (g_prof_on) {
        rec.slot = prof_slot(A_KC, B_KC, MODE, X3 ? 3 : 1);
        rec.flops = 2.0 * g.M * g.N * g.K;
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess ||
            hipEventRecord(rec.e0, stream) != hipSuccess) {
            vqh_set_error("vqh_gemm: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), dma::LDS_BYTES, stream, g);
    VQH_LAUNCH_CHECK();
    if (g_prof_on) {
        if (hipEventRecord(rec.e1, stream) != hipSuccess) {
            vqh_set_error("vqh_gemm: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
        g_prof.push_back(rec);
    }
    return VQH_OK;
}

// g_gemm_flags bit 512 selects the native fp32 MFMA tile; the default is the bf16x3 split tile (same results to fp32 round-off)
template <bool A_KC, bool B_KC, int MODE = -1>
int launch_dma(const GemmArgs& g, int splits, hipStream_t stream) {
    if (g.flags & 512) return launch_dma_impl<A_KC, B_KC, MODE, false>(g, splits, stream);
    return launch_dma_impl<A_KC, B_KC, MODE, true>(g, splits, stream);
}

}  // namespace

// Timing of every GEMM main-kernel launch between begin and end (not under stream capture).  end() synchronises
// the recorded events and fills out[5][4][9][3]: per (kernel family 0 = gemm_f32_mfma / 1 = gemm_f32_dma / 2 = gemm_f32_dma_group / 3 = gemm_f32_x3 / 4 = gemm_f32_x3_group, operand layout
// a_kc*2+b_kc, template MODE+1) the number of
// launches, the summed kernel seconds and the summed 2*M*N*K.  The split-K reduce launch is outside the bracket.
extern "C" int vqh_gemm_profile_begin(void) {
    g_prof.clear();
    g_prof_on = true;
    return VQH_OK;
}
extern "C" int vqh_gemm_profile_end(double* out) {
    g_prof_on = false;
    if (out)
        for (int i = 0; i < PROF_FAMILIES * 4 * PROF_MODES * 3; ++i) out[i] = 0.0;
    int rc = VQH_OK;
    for (ProfRec& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
            vqh_set_error("vqh_gemm_profile_end: event query failed");
            rc = VQH_ERR_LAUNCH;
        } else if (out) {
            out[r.slot * 3 + 0] += 1.0;
            out[r.slot * 3 + 1] += (double)ms * 1e-3;
            out[r.slot * 3 + 2] += r.flops;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return rc;
}

extern "C" int vqh_reduce_slabs(const float* slabs, int S, long long stride, long long n, float* out, float beta,
                                hipStream_t stream);      // rowwise.hip

static int g_gemm_flags = 1;
extern "C" int vqh_gemm_set_flags(int flags) {
#ifndef VQH_DIAG
    flags &= ~(2 | 4);      // the timing-only bits that skip stores / loads exist in -DVQH_DIAG lab builds only
#endif
    const int old = g_gemm_flags; g_gemm_flags = flags; return old;
}

// C-ABI: see include/vqvae_hip.h for the contract.
static int gemm_impl(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B,
                     int ldb, float* C, int ldc, const float* bias, int mode, const float* aux_in, float* aux_out,
                     int ldaux, float beta, const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                     float* workspace, long long workspace_floats, float* rowsum, hipStream_t stream) {
    VQH_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "vqh_gemm: negative dimension");
    if (M == 0 || N == 0) return VQH_OK;
    VQH_CHECK_ARG(C && ((A && B) || K == 0), "vqh_gemm: null operand");
    VQH_CHECK_ARG(mode >= EPI_LINEAR && mode <= EPI_MUL_SIGGRAD, "vqh_gemm: unknown epilogue mode");
    VQH_CHECK_ARG(lda >= (a_kcontig ? K : M) && ldb >= (b_kcontig ? K : N) && ldc >= N, "vqh_gemm: leading dim too small");
    if (mode == EPI_GELU) VQH_CHECK_ARG(aux_out && ldaux >= N, "vqh_gemm: GELU epilogue needs aux_out");
    if (mode == EPI_DROP_RESID || mode >= EPI_MUL_POSMASK)
        VQH_CHECK_ARG(aux_in && ldaux >= N, "vqh_gemm: epilogue needs aux_in");
    VQH_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "vqh_gemm: dropout p out of range");
    if (drop_p > 0.f && (mode == EPI_RELU_DROP || mode == EPI_DROP_RESID))
        VQH_CHECK_ARG(rng_state != nullptr, "vqh_gemm: dropout needs rng_state");

    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    g.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    {
        uintptr_t al = reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(aux_in) |
                       reinterpret_cast<uintptr_t>(aux_out);
        g.vecC = ((al & 15) == 0 && (ldc & 3) == 0 && ((aux_in || aux_out) ? (ldaux & 3) == 0 : true) && (N & 3) == 0) ? 1 : 0;
    }
    g.mode = mode; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ldaux = ldaux; g.beta = beta;
    g.drop = make_drop(rng_state, drop_site, drop_p);
    g.ws = nullptr;
    g.rowsum = rowsum;
    g.rowsum_ws = nullptr;
    g.flags = g_gemm_flags;
    if (rowsum) VQH_CHECK_ARG(!a_kcontig && mode == EPI_LINEAR, "vqh_gemm_wgrad: row sums need the [K,M] A layout and a linear epilogue");

    // skinny shapes (reconstruction heads): streaming kernels instead of a 97 % empty MFMA tile
    if (!(g_gemm_flags & 64)) {
        auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
        if (mode == EPI_LINEAR && !rowsum && a_kcontig && b_kcontig && N <= SK_MAX && K >= 64 && (K & 3) == 0 &&
            (lda & 3) == 0 && (ldb & 3) == 0 && al16(A) && al16(B)) {
            int blocks = (M + 3) / 4;
            if (blocks > 2048) blocks = 2048;
#define SK_CASE(V) case V: hipLaunchKernelGGL(skinny_n_kernel<V>, dim3(blocks), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, bias, beta, M, K); break
            switch (N) { SK_CASE(1); SK_CASE(2); SK_CASE(3); SK_CASE(4); SK_CASE(5); SK_CASE(6); SK_CASE(7); default: SK_CASE(8); }
#undef SK_CASE
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
        if (mode == EPI_LINEAR && !rowsum && !bias && a_kcontig && !b_kcontig && K <= SK_MAX && K > 0 && N >= 64 &&
            (N & 3) == 0 && (ldb & 3) == 0 && (ldc & 3) == 0 && al16(B) && al16(C)) {
            const long long quads = (long long)M * (N >> 2);
            int blocks = (int)((quads + 255) / 256);
            if (blocks > 4096) blocks = 4096;
#define SK_CASE(V) case V: hipLaunchKernelGGL(skinny_k_kernel<V>, dim3(blocks), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, beta, M, N); break
            switch (K) { SK_CASE(1); SK_CASE(2); SK_CASE(3); SK_CASE(4); SK_CASE(5); SK_CASE(6); SK_CASE(7); default: SK_CASE(8); }
#undef SK_CASE
            VQH_LAUNCH_CHECK();
            return VQH_OK;
        }
        if (mode == EPI_LINEAR && !a_kcontig && !b_kcontig && M <= SK_MAX && N >= 64 && (N & 3) == 0 && (ldb & 3) == 0 &&
            (ldc & 3) == 0 && al16(B) && al16(C) && al16(workspace) && !bias && workspace && K >= 256 && N <= 1024 &&
            256 % (N >> 2) == 0) {
            const int rows_per_block = 64;
            const int nblk = (K + rows_per_block - 1) / rows_per_block;
            const long long per = (long long)M * N + (rowsum ? M : 0);
            if ((long long)nblk * per <= workspace_floats && ldc == N) {
                float* rs_ws = rowsum ? workspace + (size_t)nblk * M * N : nullptr;
#define SK_CASE(V) case V: hipLaunchKernelGGL(skinny_m_kernel<V>, dim3(nblk), dim3(256), 0, stream, A, lda, B, ldb, workspace, rs_ws, N, K, rows_per_block); break
                switch (M) { SK_CASE(1); SK_CASE(2); SK_CASE(3); SK_CASE(4); SK_CASE(5); SK_CASE(6); SK_CASE(7); default: SK_CASE(8); }
#undef SK_CASE
                VQH_LAUNCH_CHECK();
                // many thin slabs: the slab-parallel reducer of rowwise.hip (16 slab lanes per column) instead of splitk_reduce
                int rc2 = vqh_reduce_slabs(workspace, nblk, (long long)M * N, (long long)M * N, C, beta, stream);
                if (rc2 == VQH_OK && rowsum) rc2 = vqh_reduce_slabs(rs_ws, nblk, (long long)M, (long long)M, rowsum, beta, stream);
                return rc2;
            }
        }
    }

    // ---- large-tile LDS-DMA kernel (gemm_dma.inc) for everything that tiles evenly: all the Linear layers of the model
    if (!(g_gemm_flags & 128) && (M % dma::TBM) == 0 && (N % dma::TBN) == 0 && (K % dma::TBK) == 0 && K >= dma::TBK && g.vecA &&
        g.vecB && g.vecC && (!rowsum || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0)) {
        const int tiles_d = (M / dma::TBM) * (N / dma::TBN);
        int splits_d = 1;
        if (workspace && mode == EPI_LINEAR && tiles_d < 192 && K >= 8 * dma::TBK) {
            splits_d = 256 / tiles_d;                   // one resident workgroup per CU: a single round, no tail
            const int max_by_k = K / (4 * dma::TBK);
            if (splits_d > max_by_k) splits_d = max_by_k;
            const long long per = (long long)M * N + (rowsum ? M : 0);
            if ((long long)splits_d * per > workspace_floats) splits_d = (int)(workspace_floats / per);
            if (splits_d < 2) splits_d = 1;
        }
        int kchunk_d = ((K + splits_d - 1) / splits_d + dma::TBK - 1) / dma::TBK * dma::TBK;
        splits_d = (K + kchunk_d - 1) / kchunk_d;
        const bool pairs_fit = (long long)M * N / 2 < (1LL << 32);     // the kernel's 32-bit dropout pair index
        if (tiles_d * splits_d >= 96 && pairs_fit) {                 // enough workgroups to be worth a 256-row tile
            g.kchunk = kchunk_d;
            if (splits_d > 1) {
                g.ws = workspace;
                if (rowsum) g.rowsum_ws = workspace + (size_t)splits_d * M * N;
            }
            int rc;
            if (a_kcontig && b_kcontig) {
                if (splits_d == 1 && mode == EPI_DROP_RESID) rc = launch_dma<true, true, EPI_DROP_RESID>(g, splits_d, stream);
                else if (splits_d == 1 && mode == EPI_RELU_DROP) rc = launch_dma<true, true, EPI_RELU_DROP>(g, splits_d, stream);
                else if (splits_d == 1 && mode == EPI_GELU) rc = launch_dma<true, true, EPI_GELU>(g, splits_d, stream);
                else if (mode == EPI_LINEAR) rc = launch_dma<true, true, EPI_LINEAR>(g, splits_d, stream);
                else rc = launch_dma<true, true>(g, splits_d, stream);
            } else if (a_kcontig && !b_kcontig) {
                if (splits_d == 1 && mode == EPI_MUL_POSMASK) rc = launch_dma<true, false, EPI_MUL_POSMASK>(g, splits_d, stream);
                else if (splits_d == 1 && mode == EPI_MUL_GELUGRAD) rc = launch_dma<true, false, EPI_MUL_GELUGRAD>(g, splits_d, stream);
                else if (mode == EPI_LINEAR) rc = launch_dma<true, false, EPI_LINEAR>(g, splits_d, stream);
                else rc = launch_dma<true, false>(g, splits_d, stream);
            } else if (!a_kcontig && b_kcontig) {
                if (mode == EPI_LINEAR) rc = launch_dma<false, true, EPI_LINEAR>(g, splits_d, stream);
                else rc = launch_dma<false, true>(g, splits_d, stream);
            } else {
                if (mode == EPI_LINEAR) rc = launch_dma<false, false, EPI_LINEAR>(g, splits_d, stream);
                else rc = launch_dma<false, false>(g, splits_d, stream);
            }
            if (rc != VQH_OK) return rc;
            if (splits_d > 1) {
                const size_t total = (size_t)M * N;
                int blocks = (int)((total / 4 + 255) / 256);
                if (blocks > 4096) blocks = 4096;
                if (rowsum && blocks < (M + 255) / 256) blocks = (M + 255) / 256;
                hipLaunchKernelGGL(splitk_reduce_vec, dim3(blocks), dim3(256), 0, stream, workspace, splits_d, M, N, C, ldc, bias,
                                   beta, g.rowsum_ws, rowsum);
                VQH_LAUNCH_CHECK();
            }
            return VQH_OK;
        }
        g.ws = nullptr;
        g.rowsum_ws = nullptr;
    }

    // split-K when the output has too few tiles to fill 256 CUs (weight gradients: K = B*L rows).
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int splits = 1;
    if (workspace && mode == EPI_LINEAR && tiles < 256 && K >= 8 * BK) {
        splits = 512 / tiles;                       // 2 resident blocks per CU x 256 CUs: no tail round
        const int max_by_k = K / (4 * BK);
        if (splits > max_by_k) splits = max_by_k;
        const long long per = (long long)M * N + (rowsum ? M : 0);
        if ((long long)splits * per > workspace_floats) splits = (int)(workspace_floats / per);
        if (splits < 2) splits = 1;
    }
    int kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    if (kchunk <= 0) kchunk = BK;
    splits = (K + kchunk - 1) / kchunk;
    if (splits < 1) splits = 1;
    g.kchunk = kchunk;
    if (splits > 1) {
        g.ws = workspace;
        if (rowsum) g.rowsum_ws = workspace + (size_t)splits * M * N;
    }

    int rc;
    if (splits == 1 && g.vecC && a_kcontig && b_kcontig && mode == EPI_DROP_RESID) {
        rc = launch<true, true, 32, EPI_DROP_RESID>(g, splits, stream);       // out-projections / FFN2 (forward)
    } else if (splits == 1 && g.vecC && a_kcontig && b_kcontig && mode == EPI_RELU_DROP) {
        rc = launch<true, true, 32, EPI_RELU_DROP>(g, splits, stream);        // FFN1 (forward)
    } else if (splits == 1 && g.vecC && a_kcontig && b_kcontig && mode == EPI_GELU) {
        rc = launch<true, true, 32, EPI_GELU>(g, splits, stream);             // tokenizer FFN / fuse MLP (forward)
    } else if (splits == 1 && g.vecC && a_kcontig && !b_kcontig && mode == EPI_MUL_POSMASK) {
        rc = launch<true, false, 32, EPI_MUL_POSMASK>(g, splits, stream);     // FFN1 input gradient
    } else if (splits == 1 && g.vecC && a_kcontig && !b_kcontig && mode == EPI_MUL_GELUGRAD) {
        rc = launch<true, false, 32, EPI_MUL_GELUGRAD>(g, splits, stream);    // tokenizer FFN / fuse MLP input gradient
    } else {
        if (a_kcontig && b_kcontig) rc = launch<true, true, 32>(g, splits, stream);
        else if (a_kcontig && !b_kcontig) rc = launch<true, false, 32>(g, splits, stream);
        else if (!a_kcontig && b_kcontig) rc = launch<false, true, 32>(g, splits, stream);
        else rc = launch<false, false, 32>(g, splits, stream);
    }
    if (rc != VQH_OK) return rc;
    if (splits > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (rowsum && blocks < (M + 255) / 256) blocks = (M + 255) / 256;
        const bool vec = (N % 4 == 0) && (ldc % 4 == 0) &&
                         (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) |
                            reinterpret_cast<uintptr_t>(workspace)) & 15) == 0);
        if (vec) {
            blocks = (int)((total / 4 + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            if (rowsum && blocks < (M + 255) / 256) blocks = (M + 255) / 256;
            hipLaunchKernelGGL(splitk_reduce_vec, dim3(blocks), dim3(256), 0, stream, workspace, splits, M, N, C, ldc, bias,
                               beta, g.rowsum_ws, rowsum);
        } else
            hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, stream, workspace, splits, M, N, C, ldc, bias, beta,
                               g.rowsum_ws, rowsum);
        VQH_LAUNCH_CHECK();
    }
    return VQH_OK;
}

extern "C" int vqh_gemm(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B,
                        int ldb, float* C, int ldc, const float* bias, int mode, const float* aux_in, float* aux_out,
                        int ldaux, float beta, const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                        float* workspace, long long workspace_floats, hipStream_t stream) {
    return gemm_impl(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, C, ldc, bias, mode, aux_in, aux_out, ldaux, beta,
                     rng_state, drop_site, drop_p, workspace, workspace_floats, nullptr, stream);
}

// Weight + bias gradient of a Linear in one launch: dW[Nout,Kin] = dY[rows,Nout]^T . X[rows,Kin] and
// db[Nout] = column sums of dY (both written as beta*old + value).  dY is the [K,M]-stored A operand.
extern "C" int vqh_gemm_wgrad(int rows, int n_out, int k_in, const float* dY, int lddy, const float* X, int ldx, float* dW,
                              int lddw, float* db, float beta, float* workspace, long long workspace_floats,
                              hipStream_t stream) {
    if (rows > 0 && n_out > 0 && k_in == 0) return VQH_OK;
    return gemm_impl(0, 0, n_out, k_in, rows, dY, lddy, X, ldx, dW, lddw, nullptr, EPI_LINEAR, nullptr, nullptr, 0, beta,
                     nullptr, 0, 0.f, workspace, workspace_floats, db, stream);
}

// All weight-gradient products of one layer in one launch (see gemm_f32_dma_group).  Every problem is dW = dY^T X with
// db = column sums of dY, written with beta = 0.  Products that do not tile evenly run one by one through vqh_gemm_wgrad.
struct vqh_wgrad_t {
    int rows, n_out, k_in;
    const float* dY; int lddy;
    const float* X; int ldx;
    float* dW; int lddw;
    float* db;
};
extern "C" int vqh_gemm_wgrad_group(int n, const vqh_wgrad_t* pr, float* workspace, long long workspace_floats,
                                    hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && (n == 0 || pr), "vqh_gemm_wgrad_group: bad argument");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    std::vector<int> big;
    for (int i = 0; i < n; ++i) {
        const vqh_wgrad_t& w = pr[i];
        VQH_CHECK_ARG(w.rows >= 0 && w.n_out >= 0 && w.k_in >= 0, "vqh_gemm_wgrad_group: negative dimension");
        const bool ok = !(g_gemm_flags & (128 | 256)) && workspace && w.rows >= 8 * dma::TBK && (w.rows % dma::TBK) == 0 &&
                        (w.n_out % dma::TBM) == 0 && (w.k_in % dma::TBN) == 0 && w.n_out > 0 && w.k_in > 0 && al16(w.dY) &&
                        al16(w.X) && al16(w.dW) && al16(workspace) && (w.lddy & 3) == 0 && (w.ldx & 3) == 0 && (w.lddw & 3) == 0 &&
                        w.lddy >= w.n_out && w.ldx >= w.k_in && w.lddw >= w.k_in;
        if (ok) {
            big.push_back(i);
        } else {
            const int rc = vqh_gemm_wgrad(w.rows, w.n_out, w.k_in, w.dY, w.lddy, w.X, w.ldx, w.dW, w.lddw, w.db, 0.f, workspace,
                                          workspace_floats, stream);
            if (rc != VQH_OK) return rc;
        }
    }
    for (size_t c0 = 0; c0 < big.size(); c0 += GROUP_MAX) {
        const int m = (int)std::min<size_t>(GROUP_MAX, big.size() - c0);
        // common K-chunk: maximise (CU utilisation of the last round) x (share of a workgroup's time spent in K-steps)
        long long tiles_tot = 0;
        int kmax = 0;
        for (int j = 0; j < m; ++j) {
            const vqh_wgrad_t& w = pr[big[c0 + j]];
            tiles_tot += (long long)(w.n_out / dma::TBM) * (w.k_in / dma::TBN);
            kmax = std::max(kmax, w.rows);
        }
        int best_kc = kmax;
        double best = -1.0;
        for (int sp = 1; sp <= 64; ++sp) {
            const int kc = ((kmax + sp - 1) / sp + dma::TBK - 1) / dma::TBK * dma::TBK;
            if (kc < 16 * dma::TBK && sp > 1) break;
            long long wgs = 0, slab = 0;
            for (int j = 0; j < m; ++j) {
                const vqh_wgrad_t& w = pr[big[c0 + j]];
                const int splits = (w.rows + kc - 1) / kc;
                wgs += (long long)(w.n_out / dma::TBM) * (w.k_in / dma::TBN) * splits;
                if (splits > 1) slab += (long long)splits * ((long long)w.n_out * w.k_in + w.n_out);
            }
            if (slab > workspace_floats) continue;
            const double rounds = (double)((wgs + 255) / 256);
            const double util = (double)wgs / (256.0 * rounds);
            const double score = util * (double)kc / ((double)kc + 12.0 * dma::TBK);     // ~12 K-steps of fixed cost per workgroup
            if (score > best) { best = score; best_kc = kc; }
        }
        GemmGroupArgs G;
        G.n = m;
        int wg = 0;
        long long off = 0;
        double flops = 0.0;
        struct Red { float* ws; int splits, M, N; float* C; int ldc; float* rs_ws; float* rowsum; };
        Red red[GROUP_MAX];
        int nred = 0;
        for (int j = 0; j < m; ++j) {
            const vqh_wgrad_t& w = pr[big[c0 + j]];
            GemmArgs& g = G.p[j];
            g.A = w.dY; g.B = w.X; g.C = w.dW;
            g.M = w.n_out; g.N = w.k_in; g.K = w.rows;
            g.lda = w.lddy; g.ldb = w.ldx; g.ldc = w.lddw;
            g.kchunk = best_kc;
            g.vecA = g.vecB = g.vecC = 1;
            g.mode = EPI_LINEAR; g.bias = nullptr; g.aux_in = nullptr; g.aux_out = nullptr; g.ldaux = 0; g.beta = 0.f;
            g.drop = make_drop(nullptr, 0, 0.f);
            g.flags = g_gemm_flags;
            g.rowsum = w.db;
            const int splits = (w.rows + best_kc - 1) / best_kc;
            if (splits > 1) {
                g.ws = workspace + off;
                off += (long long)splits * w.n_out * w.k_in;
                g.rowsum_ws = w.db ? workspace + off : nullptr;
                if (w.db) off += (long long)splits * w.n_out;
                off = (off + 3) / 4 * 4;
                red[nred++] = Red{g.ws, splits, w.n_out, w.k_in, w.dW, w.lddw, g.rowsum_ws, w.db};
            } else {
                g.ws = nullptr;
                g.rowsum_ws = nullptr;
            }
            G.tiles[j] = (w.n_out / dma::TBM) * (w.k_in / dma::TBN);
            G.wg_begin[j] = wg;
            wg += G.tiles[j] * splits;
            flops += 2.0 * w.n_out * (double)w.k_in * w.rows;
        }
        for (int j = m; j <= GROUP_MAX; ++j) G.wg_begin[j] = wg;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_dma_group),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, dma::LDS_BYTES);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_x3_group), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        dma::LDS_BYTES);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_set = true;
        }
        const bool native = (g_gemm_flags & 512) != 0;
        ProfRec rec{};
        if (g_prof_on) {
            rec.slot = prof_slot(false, false, EPI_LINEAR, native ? 2 : 4);
            rec.flops = flops;
            if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess ||
                hipEventRecord(rec.e0, stream) != hipSuccess) {
                vqh_set_error("vqh_gemm_wgrad_group: profiling events failed");
                return VQH_ERR_LAUNCH;
            }
        }
        if (native) hipLaunchKernelGGL(gemm_f32_dma_group, dim3(wg), dim3(256), dma::LDS_BYTES, stream, G);
        else hipLaunchKernelGGL(gemm_f32_x3_group, dim3(wg), dim3(256), dma::LDS_BYTES, stream, G);
        VQH_LAUNCH_CHECK();
        if (g_prof_on) {
            if (hipEventRecord(rec.e1, stream) != hipSuccess) {
                vqh_set_error("vqh_gemm_wgrad_group: profiling events failed");
                return VQH_ERR_LAUNCH;
            }
            g_prof.push_back(rec);
        }
        if (nred > 0) {
            RedGroupArgs RG;
            RG.n = nred;
            int blocks = 1;
            for (int j = 0; j < nred; ++j) {
                const Red& r = red[j];
                RG.r[j] = RedGroupArgs::Item{r.ws, r.splits, r.M, r.N, r.C, r.ldc, r.rs_ws, r.rowsum};
                blocks = std::max(blocks, (int)(((size_t)r.M * r.N / 4 + 255) / 256));
            }
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(splitk_reduce_group, dim3(blocks, nred), dim3(256), 0, stream, RG);
        }
        VQH_LAUNCH_CHECK();
    }
    return VQH_OK;
}


// ================================================================================================================================
// Plane-tensor ("P3") entry points: see gemm_p3.inc and include/vqvae_hip.h
// ================================================================================================================================
extern "C" int vqh_p3_split(const float* X, int ldx, void* P, long long pitch_bytes, int rows, int cols, hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return VQH_OK;
    VQH_CHECK_ARG(X && P, "vqh_p3_split: null pointer");
    if (pitch_bytes == 0) {                 // stage images: [rows / 256][cols / 32][16 row blocks][3 planes][1 KB]
        VQH_CHECK_ARG((rows % 256) == 0 && (cols % 32) == 0 && ldx >= cols && (ldx & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(P)) & 15) == 0,
                      "vqh_p3_split: stage images need rows % 256 == 0, cols % 32 == 0, ldx % 4 == 0 and 16-byte alignment");
        const long long tot = (long long)rows * (cols >> 3);
        hipLaunchKernelGGL(p3_split_tiled_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 1 << 16)), dim3(256), 0, stream, X,
                           (long long)ldx, reinterpret_cast<char*>(P), rows, cols, 256);
        VQH_LAUNCH_CHECK();
        return VQH_OK;
    }
    VQH_CHECK_ARG((cols % 32) == 0 && ldx >= cols && (ldx & 3) == 0 && pitch_bytes >= (long long)cols * 6 && (pitch_bytes % 16) == 0 &&
                  ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(P)) & 15) == 0,
                  "vqh_p3_split: cols % 32, ldx % 4, 16-byte alignment and pitch >= 6 * cols required");
    const long long total = (long long)rows * (cols >> 3);
    int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(p3_split_kernel, dim3(blocks), dim3(256), 0, stream, X, (long long)ldx, reinterpret_cast<char*>(P), pitch_bytes, rows, cols);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

struct vqh_p3_item_t { const float* X; void* P; int rows, cols; long long ldx, pitch_bytes; };
extern "C" int vqh_p3_split_multi(int n, const vqh_p3_item_t* items, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && (n == 0 || items), "vqh_p3_split_multi: bad argument");
    for (int c0 = 0; c0 < n; c0 += P3_SPLIT_MAX) {
        P3SplitArgs S;
        S.n = std::min(P3_SPLIT_MAX, n - c0);
        long long most = 0;
        for (int i = 0; i < S.n; ++i) {
            const vqh_p3_item_t& it = items[c0 + i];
            VQH_CHECK_ARG(it.X && it.P && it.rows > 0 && it.cols > 0 && (it.cols % 32) == 0 && it.ldx >= it.cols && (it.ldx & 3) == 0 &&
                          it.pitch_bytes >= (long long)it.cols * 6 && (it.pitch_bytes % 16) == 0 &&
                          ((reinterpret_cast<uintptr_t>(it.X) | reinterpret_cast<uintptr_t>(it.P)) & 15) == 0,
                          "vqh_p3_split_multi: bad item (cols % 32, alignment, pitch)");
            S.it[i] = P3SplitItem{it.X, reinterpret_cast<char*>(it.P), it.rows, it.cols, it.ldx, it.pitch_bytes};
            most = std::max(most, (long long)it.rows * (it.cols >> 3));
        }
        int bx = (int)std::min<long long>((most + 255) / 256, 512);
        hipLaunchKernelGGL(p3_split_multi_kernel, dim3(bx, S.n), dim3(256), 0, stream, S);
        VQH_LAUNCH_CHECK();
    }
    return VQH_OK;
}

namespace {
inline int p3_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        else n = 256;
    }
    return n;
}
template <bool A_KC, bool B_KC, int MODE>
int launch_p3(const P3Args& g, int splits, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = &gemm_p3<A_KC, B_KC, MODE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, p3::LDS_BYTES);
        if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
        attr_set = true;
    }
    // persistent: at most one resident workgroup per CU (144 KB of LDS each), each walking its share of the tiles x splits
    const int n_items = (g.M / p3::TBM) * (g.N / p3::TBN) * splits;
    dim3 grid(std::min(n_items, p3_num_cus()));
    // the kernel's stage parity is static: a workgroup that walks several items needs an even number of K-steps per item
    if (((g.kchunk / p3::TBK) & 1) && n_items > (int)grid.x) grid.x = n_items;
    ProfRec rec{};
    if (g_prof_on) {
        rec.slot = prof_slot(A_KC, B_KC, MODE, 5);
        rec.flops = 2.0 * g.M * g.N * g.K;
        if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess || hipEventRecord(rec.e0, stream) != hipSuccess) {
            vqh_set_error("vqh_gemm_p3: profiling events failed");
            return VQH_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), p3::LDS_BYTES, stream, g, n_items);
    VQH_LAUNCH_CHECK();
    if (g_prof_on) {
        if (hipEventRecord(rec.e1, stream) != hipSuccess) { vqh_set_error("vqh_gemm_p3: profiling events failed"); return VQH_ERR_LAUNCH; }
        g_prof.push_back(rec);
    }
    return VQH_OK;
}
inline bool p3_tensor_ok(const void* P, long long pitch, int cols) {
    if (pitch == 0) return P && (reinterpret_cast<uintptr_t>(P) & 15) == 0;          // stage images (rows % 256 checked by the caller)
    return P && (reinterpret_cast<uintptr_t>(P) & 15) == 0 && (pitch % 16) == 0 && pitch >= (long long)cols * 6;
}
}  // namespace

// Internal (vq.hip): scores of R rows of Z against K codes on pre-split operands, top-2 partials per (slot, row).
// Z / E: fp32 [R][D] and [K][D], split here into stage images Zp (6 R D bytes) and Ep (6 K D bytes); nb_init[K]; partial arrays
// [2 * nsplit][R].  R % 256 == 0, K % (128 * nsplit) == 0, D % 64 == 0 (an even number of K-steps per tile), at most 32 code
// tiles per workgroup (their start values sit in the 16 KB of LDS behind the two stages).
extern "C" __attribute__((visibility("hidden"))) int vqh_internal_nearest_p3(const float* Z, int ldz, const float* E, int lde, void* Zp, void* Ep,
                                                                            const float* nb_init, float* pbest, float* psecond, int* pidx,
                                                                            int R, int K, int D, int nsplit, hipStream_t stream) {
    VQH_CHECK_ARG(R > 0 && K > 0 && nsplit > 0 && (R % p3::TBM) == 0 && (K % (p3::TBN * nsplit)) == 0 && (D % 64) == 0 &&
                      K / p3::TBN / nsplit <= 32,
                  "nearest_p3: shape not eligible");
    VQH_CHECK_ARG(Z && E && Zp && Ep && ((reinterpret_cast<uintptr_t>(Zp) | reinterpret_cast<uintptr_t>(Ep)) & 15) == 0 && nb_init && pbest &&
                      psecond && pidx, "nearest_p3: bad operand");
    {
        const long long tz = (long long)R * (D / 8), te = (long long)K * (D / 8);
        hipLaunchKernelGGL(p3_split_tiled_kernel, dim3((unsigned)std::min<long long>((tz + 255) / 256, 1 << 16)), dim3(256), 0, stream, Z,
                           (long long)ldz, static_cast<char*>(Zp), R, D, p3::TBM);
        hipLaunchKernelGGL(p3_split_tiled_kernel, dim3((unsigned)std::min<long long>((te + 255) / 256, 1 << 16)), dim3(256), 0, stream, E,
                           (long long)lde, static_cast<char*>(Ep), K, D, p3::TBN);
    }
    const long long pz = (long long)D * 6, pe = (long long)D * 6;
    const int lds_bytes = p3::LDS_BYTES + (K / p3::TBN / nsplit) * p3::TBN * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_nearest_p3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
        attr_set = true;
    }
    P3Args g{};
    g.A = static_cast<const char*>(Zp); g.B = static_cast<const char*>(Ep);
    g.pa = pz; g.pb = pe;
    g.M = R; g.N = K; g.K = D; g.kchunk = D;
    g.mode = EPI_NEAREST;
    g.nb_init = nb_init; g.nb_best = pbest; g.nb_second = psecond; g.nb_idx = pidx;
    const int per = K / p3::TBN / nsplit;
#ifdef P3_STAMPS
    static unsigned long long* stamp_buf = nullptr;
    const int nwg = (R / p3::TBM) * nsplit;
    if (!stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)65536 * 64);
    g.aux_out = reinterpret_cast<float*>(stamp_buf);
#endif
    hipLaunchKernelGGL(vq_nearest_p3_kernel, dim3((R / p3::TBM) * nsplit), dim3(512), lds_bytes, stream, g, per, nsplit);
    VQH_LAUNCH_CHECK();
#ifdef P3_STAMPS
    {
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> hb((size_t)nwg * 8);
        (void)hipMemcpy(hb.data(), stamp_buf, hb.size() * 8, hipMemcpyDeviceToHost);
        double tot = 0, real = 0, loop = 0, epi = 0, vm = 0, bar = 0, iss = 0, Q = 0;
        for (int w = 0; w < nwg; ++w) {
            tot += hb[w * 8 + 0]; real += hb[w * 8 + 1]; loop += hb[w * 8 + 2]; epi += (double)(hb[w * 8 + 3] >> 20); Q += (double)(hb[w * 8 + 3] & 0xfffff);
            vm += hb[w * 8 + 4]; bar += hb[w * 8 + 5]; iss += hb[w * 8 + 6];
        }
        fprintf(stderr, "[p3n stamps] R=%d K=%d D=%d ns=%d wgs=%d | per wg: total %.0f cyc (%.1f us real @100MHz ticks %.0f) loop %.0f epi %.0f | per K-step: loop %.0f epi %.0f | loader: vm-wait %.0f barrier %.0f issue %.0f per step\n",
                R, K, D, nsplit, nwg, tot / nwg, real / nwg / 100.0, real / nwg, loop / nwg, epi / nwg, loop / Q, epi / Q, vm / Q, bar / Q, iss / Q);
    }
#endif
    return VQH_OK;
}

// 1 when vqh_gemm_p3 accepts the shape (the caller keeps fp32 operands and vqh_gemm for everything else)
extern "C" int vqh_gemm_p3_eligible(int M, int N, int K) {
    return (M > 0 && N > 0 && K >= p3::TBK && (M % p3::TBM) == 0 && (N % p3::TBN) == 0 && (K % p3::TBK) == 0 &&
            (long long)M * N / 2 < (1LL << 32)) ? 1 : 0;
}

extern "C" int vqh_gemm_p3(int a_kcontig, int b_kcontig, int M, int N, int K, const void* Ap, long long pitch_a, const void* Bp,
                           long long pitch_b, float* C, int ldc, void* Cp, long long pitch_c, const float* bias, int mode,
                           const float* aux_in, float* aux_out, int ldaux, unsigned* sign_bits, float beta,
                           const unsigned long long* rng_state, unsigned drop_site, float drop_p, float* workspace,
                           long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(vqh_gemm_p3_eligible(M, N, K), "vqh_gemm_p3: shape not eligible (M % 256, N % 128, K % 32)");
    VQH_CHECK_ARG(mode >= EPI_LINEAR && mode <= EPI_MUL_SIGGRAD, "vqh_gemm_p3: unknown epilogue mode");
    VQH_CHECK_ARG(p3_tensor_ok(Ap, pitch_a, a_kcontig ? K : M) && p3_tensor_ok(Bp, pitch_b, b_kcontig ? K : N),
                  "vqh_gemm_p3: operand plane tensor (alignment / pitch)");
    // pitch 0 = stage images (256-row tiles): the operand's row count must be a multiple of 256
    VQH_CHECK_ARG(pitch_a != 0 || ((a_kcontig ? M : K) % 256) == 0, "vqh_gemm_p3: stage-image A needs rows % 256 == 0");
    VQH_CHECK_ARG(pitch_b != 0 || ((b_kcontig ? N : K) % 256) == 0, "vqh_gemm_p3: stage-image B needs rows % 256 == 0");
    VQH_CHECK_ARG(!Cp || pitch_c != 0, "vqh_gemm_p3: stage-image OUTPUT is not implemented");
    VQH_CHECK_ARG(C || Cp, "vqh_gemm_p3: no output");
    VQH_CHECK_ARG(!C || (ldc >= N && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0), "vqh_gemm_p3: C alignment / ldc");
    VQH_CHECK_ARG(!Cp || p3_tensor_ok(Cp, pitch_c, N), "vqh_gemm_p3: output plane tensor (alignment / pitch)");
    VQH_CHECK_ARG(!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0, "vqh_gemm_p3: bias alignment");
    if (mode == EPI_GELU) VQH_CHECK_ARG(aux_out && ldaux >= N && (ldaux & 3) == 0, "vqh_gemm_p3: GELU epilogue needs aux_out");
    if (mode == EPI_DROP_RESID || mode == EPI_MUL_GELUGRAD || mode == EPI_MUL_SIGGRAD)
        VQH_CHECK_ARG(aux_in && ldaux >= N && (ldaux & 3) == 0 && (reinterpret_cast<uintptr_t>(aux_in) & 15) == 0, "vqh_gemm_p3: epilogue needs aux_in");
    if (mode == EPI_MUL_POSMASK) VQH_CHECK_ARG(sign_bits && (reinterpret_cast<uintptr_t>(sign_bits) & 15) == 0, "vqh_gemm_p3: POSMASK needs the sign bits");
    VQH_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "vqh_gemm_p3: dropout p out of range");
    if (drop_p > 0.f && (mode == EPI_RELU_DROP || mode == EPI_DROP_RESID)) VQH_CHECK_ARG(rng_state != nullptr, "vqh_gemm_p3: dropout needs rng_state");
    VQH_CHECK_ARG(beta == 0.f || (C && !Cp), "vqh_gemm_p3: beta needs the fp32 output (and no plane output)");

    P3Args g{};
    g.A = reinterpret_cast<const char*>(Ap); g.B = reinterpret_cast<const char*>(Bp);
    g.pa = pitch_a; g.pb = pitch_b;
    g.C = C; g.ldc = ldc; g.Cp = reinterpret_cast<char*>(Cp); g.pc = pitch_c;
    g.M = M; g.N = N; g.K = K;
    g.mode = mode; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out; g.ldaux = ldaux;
    g.bits_out = (mode == EPI_RELU_DROP) ? sign_bits : nullptr;
    g.bits_in = (mode == EPI_MUL_POSMASK) ? sign_bits : nullptr;
    g.beta = beta;
    g.drop = make_drop(rng_state, drop_site, drop_p);
    g.ws = nullptr; g.rowsum = nullptr; g.rowsum_ws = nullptr;
    g.flags = g_gemm_flags;
    const int tiles = (M / p3::TBM) * (N / p3::TBN);
    int splits = 1;
    if (workspace && mode == EPI_LINEAR && !Cp && beta == 0.f && tiles < 192 && K >= 8 * p3::TBK) {
        splits = 256 / tiles;
        const int max_by_k = K / (4 * p3::TBK);
        if (splits > max_by_k) splits = max_by_k;
        if ((long long)splits * M * N > workspace_floats) splits = (int)(workspace_floats / ((long long)M * N));
        if (splits < 2) splits = 1;
    }
    int kchunk = ((K + splits - 1) / splits + p3::TBK - 1) / p3::TBK * p3::TBK;
    splits = (K + kchunk - 1) / kchunk;
    g.kchunk = kchunk;
    if (splits > 1) g.ws = workspace;
    int rc;
    const bool one = splits == 1;
    if (a_kcontig && b_kcontig) {
        if (one && mode == EPI_DROP_RESID) rc = launch_p3<true, true, EPI_DROP_RESID>(g, splits, stream);
        else if (one && mode == EPI_RELU_DROP) rc = launch_p3<true, true, EPI_RELU_DROP>(g, splits, stream);
        else if (one && mode == EPI_GELU) rc = launch_p3<true, true, EPI_GELU>(g, splits, stream);
        else if (mode == EPI_LINEAR) rc = launch_p3<true, true, EPI_LINEAR>(g, splits, stream);
        else rc = launch_p3<true, true, -1>(g, splits, stream);
    } else if (a_kcontig && !b_kcontig) {
        if (one && mode == EPI_MUL_POSMASK) rc = launch_p3<true, false, EPI_MUL_POSMASK>(g, splits, stream);
        else if (one && mode == EPI_MUL_GELUGRAD) rc = launch_p3<true, false, EPI_MUL_GELUGRAD>(g, splits, stream);
        else if (mode == EPI_LINEAR) rc = launch_p3<true, false, EPI_LINEAR>(g, splits, stream);
        else rc = launch_p3<true, false, -1>(g, splits, stream);
    } else if (!a_kcontig && !b_kcontig) {
        VQH_CHECK_ARG(mode == EPI_LINEAR, "vqh_gemm_p3: the transposed-A product supports the linear epilogue only");
        rc = launch_p3<false, false, EPI_LINEAR>(g, splits, stream);
    } else {
        vqh_set_error("vqh_gemm_p3: layout (a_kcontig = 0, b_kcontig = 1) is not built (no call site)");
        return VQH_ERR_ARG;
    }
    if (rc != VQH_OK) return rc;
    if (splits > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)((total / 4 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_vec, dim3(blocks), dim3(256), 0, stream, workspace, splits, M, N, C, ldc, bias, beta,
                           (const float*)nullptr, (float*)nullptr);
        VQH_LAUNCH_CHECK();
    }
    return VQH_OK;
}

// All weight-gradient products of one layer on plane operands: dW_p[n_out, k_in] = dY_p[rows, n_out]^T . X_p[rows, k_in], db_p =
// column sums of dY_p (MFMAs against a ones fragment), one grouped launch + one grouped split-K reduce.  Every product must tile
// evenly (n_out % 256, k_in % 128, rows % 32): the caller routes other shapes through the fp32 entry points.
struct vqh_wgrad_p3_t {
    int rows, n_out, k_in;
    const void* dYp; long long pitch_dy;
    const void* Xp; long long pitch_x;
    float* dW; int lddw;
    float* db;
};
extern "C" int vqh_gemm_p3_wgrad_group(int n, const vqh_wgrad_p3_t* pr, float* workspace, long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && (n == 0 || pr), "vqh_gemm_p3_wgrad_group: bad argument");
    VQH_CHECK_ARG(workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "vqh_gemm_p3_wgrad_group: workspace");
    for (int i = 0; i < n; ++i) {
        const vqh_wgrad_p3_t& w = pr[i];
        VQH_CHECK_ARG(w.rows >= p3::TBK && (w.rows % p3::TBK) == 0 && w.n_out > 0 && (w.n_out % p3::TBM) == 0 && w.k_in > 0 &&
                      (w.k_in % p3::TBN) == 0 && p3_tensor_ok(w.dYp, w.pitch_dy, w.n_out) && p3_tensor_ok(w.Xp, w.pitch_x, w.k_in) &&
                      w.dW && (reinterpret_cast<uintptr_t>(w.dW) & 15) == 0 && (w.lddw & 3) == 0 && w.lddw >= w.k_in,
                      "vqh_gemm_p3_wgrad_group: product does not tile evenly / bad operand");
    }
    for (int c0 = 0; c0 < n; c0 += GROUP_MAX) {
        const int m = std::min(GROUP_MAX, n - c0);
        int kmax = 0;
        for (int j = 0; j < m; ++j) kmax = std::max(kmax, pr[c0 + j].rows);
        int best_kc = kmax;
        double best = -1.0;
        for (int sp = 1; sp <= 64; ++sp) {
            const int kc = ((kmax + sp - 1) / sp + p3::TBK - 1) / p3::TBK * p3::TBK;
            if (kc < 16 * p3::TBK && sp > 1) break;
            long long wgs = 0, slab = 0;
            for (int j = 0; j < m; ++j) {
                const vqh_wgrad_p3_t& w = pr[c0 + j];
                const int splits = (w.rows + kc - 1) / kc;
                wgs += (long long)(w.n_out / p3::TBM) * (w.k_in / p3::TBN) * splits;
                if (splits > 1) slab += (long long)splits * ((long long)w.n_out * w.k_in + w.n_out) + 4;
            }
            if (slab > workspace_floats) continue;
            const double rounds = (double)((wgs + 255) / 256);
            const double util = (double)wgs / (256.0 * rounds);
            const double score = util * (double)kc / ((double)kc + 12.0 * p3::TBK);
            if (score > best) { best = score; best_kc = kc; }
        }
        VQH_CHECK_ARG(best >= 0.0, "vqh_gemm_p3_wgrad_group: workspace too small");
        P3GroupArgs G;
        G.n = m;
        int wg = 0;
        long long off = 0;
        double flops = 0.0;
        struct Red { float* ws; int splits, M, N; float* C; int ldc; float* rs_ws; float* rowsum; };
        Red red[GROUP_MAX];
        int nred = 0;
        for (int j = 0; j < m; ++j) {
            const vqh_wgrad_p3_t& w = pr[c0 + j];
            P3Args& g = G.p[j];
            g = P3Args{};
            g.A = reinterpret_cast<const char*>(w.dYp); g.B = reinterpret_cast<const char*>(w.Xp);
            g.pa = w.pitch_dy; g.pb = w.pitch_x;
            g.C = w.dW; g.ldc = w.lddw; g.Cp = nullptr; g.pc = 0;
            g.M = w.n_out; g.N = w.k_in; g.K = w.rows;
            g.kchunk = best_kc;
            g.mode = EPI_LINEAR; g.beta = 0.f;
            g.drop = make_drop(nullptr, 0, 0.f);
            g.flags = g_gemm_flags;
            g.rowsum = w.db;
            const int splits = (w.rows + best_kc - 1) / best_kc;
            if (splits > 1) {
                g.ws = workspace + off;
                off += (long long)splits * w.n_out * w.k_in;
                g.rowsum_ws = w.db ? workspace + off : nullptr;
                if (w.db) off += (long long)splits * w.n_out;
                off = (off + 3) / 4 * 4;
                red[nred++] = Red{g.ws, splits, w.n_out, w.k_in, w.dW, w.lddw, g.rowsum_ws, w.db};
            }
            G.wg_begin[j] = wg;
            wg += (w.n_out / p3::TBM) * (w.k_in / p3::TBN) * splits;
            flops += 2.0 * w.n_out * (double)w.k_in * w.rows;
        }
        for (int j = m; j <= GROUP_MAX; ++j) G.wg_begin[j] = wg;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p3_group), hipFuncAttributeMaxDynamicSharedMemorySize, p3::LDS_BYTES);
            if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
            attr_set = true;
        }
        ProfRec rec{};
        if (g_prof_on) {
            rec.slot = prof_slot(false, false, EPI_LINEAR, 6);
            rec.flops = flops;
            if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess || hipEventRecord(rec.e0, stream) != hipSuccess) {
                vqh_set_error("vqh_gemm_p3_wgrad_group: profiling events failed");
                return VQH_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(gemm_p3_group, dim3(wg), dim3(512), p3::LDS_BYTES, stream, G);
        VQH_LAUNCH_CHECK();
        if (g_prof_on) {
            if (hipEventRecord(rec.e1, stream) != hipSuccess) { vqh_set_error("vqh_gemm_p3_wgrad_group: profiling events failed"); return VQH_ERR_LAUNCH; }
            g_prof.push_back(rec);
        }
        if (nred > 0) {
            RedGroupArgs RG;
            RG.n = nred;
            int blocks = 1;
            for (int j = 0; j < nred; ++j) {
                const Red& r = red[j];
                RG.r[j] = RedGroupArgs::Item{r.ws, r.splits, r.M, r.N, r.C, r.ldc, r.rs_ws, r.rowsum};
                blocks = std::max(blocks, (int)(((size_t)r.M * r.N / 4 + 255) / 256));
            }
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(splitk_reduce_group, dim3(blocks, nred), dim3(256), 0, stream, RG);
            VQH_LAUNCH_CHECK();
        }
    }
    return VQH_OK;
}
