// dma lab (diagnostic, not product): who should issue the LDS-DMA instructions of the plane-tensor GEMM loop?
// gemm_p3 (csrc/gemm_p3.inc) measured ~4000 cycles per K-step against 3089 for fragment reads + MFMAs alone (tools/shape_lab.hip):
// 18 global_load_lds_dwordx4 per wave and K-step, issued by the wave that also issues the 192 MFMAs, stall that wave for tens
// of cycles each (MI355X_MICROARCH.md: "LDS-DMA piece issue cost 60 cycles among bare MFMAs").  Variants, same bytes and flops
// per K-step and workgroup (256 x 128 x 32 tile, 72 KB of planes, 768 MFMAs of 16x16x32), one barrier per K-step:
//   V4   4 waves x (128 x 64), every wave issues its 18 DMAs between its own MFMAs
//   V8   8 waves x (64 x 64), every wave issues 9 DMAs: two waves per SIMD cover each other's DMA stalls
//   V4L  4 compute waves x (128 x 64) + 4 loader waves that issue all DMAs (18 each) and nothing else
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_lab.hip -o tools/_build/dma_lab && tools/_build/dma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <utility>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STAGE = 73728, LDS_BYTES = 2 * STAGE, A_BYTES = 49152, UNIT = 1024;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void glds16_s(const char* ubase, unsigned voff, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(ubase), "s"(lds_byte_addr_uniform) : "memory");
}

// VAR 0 = V4, 1 = V8, 2 = V4L
template <int VAR>
__global__ __launch_bounds__(VAR == 0 ? 256 : 512) void lab_kernel(const char* __restrict__ src, size_t src_bytes, float* __restrict__ out,
                                                                   int ksteps, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned voff = (unsigned)((lane >> 2) * 12288 + (lane & 3) * 16);          // 16 rows of a 12 KB-pitch plane tensor
    constexpr int NW = (VAR == 0) ? 4 : 8;
    constexpr bool LOADER_SPLIT = (VAR == 2);
    const bool is_loader = LOADER_SPLIT && wave >= 4;
    const int cw = LOADER_SPLIT ? (wave & 3) : wave;                                   // compute-wave id
    constexpr int MI = (VAR == 1) ? 4 : 8, NJ = 4;                                     // 16x16 blocks per wave
    constexpr int NDMA = (VAR == 1) ? 9 : 18;
    // 8 workgroups share one stream (an output tile row of a real GEMM shares its A panel, a column its B panel): L2 reuse
    // (workgroups b and b + 8 share an XCD: the sharing group is 8 workgroups of ONE XCD, like the product's XCD-contiguous tile order)
    const int stream = (blockIdx.x & 7) * 4 + (blockIdx.x >> 3) / 8;
    const char* base = src + ((size_t)stream * 7919 * STAGE) % (src_bytes - (size_t)(ksteps + 4) * STAGE - (1 << 20));
    auto issue = [&](auto v_c, int kt, unsigned stage_lds, int w) {
        constexpr int v = decltype(v_c)::value;
        const int unit = w * NDMA + v;
        glds16_s(base + (size_t)kt * 192 + (size_t)(unit % 24) * 16 * 12288 + (unit / 24) * 64, voff,
                 __builtin_amdgcn_readfirstlane(stage_lds + (unsigned)(unit * UNIT)));
    };
    // prologue: tile 0
    if (!LOADER_SPLIT || is_loader) {
        static_for<NDMA>([&](auto v_c) { issue(v_c, 0, lds0, LOADER_SPLIT ? cw : wave); });
        static_for<NDMA>([&](auto v_c) { issue(v_c, 1, lds0 + STAGE, LOADER_SPLIT ? cw : wave); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (is_loader) {
        for (int ks = 0; ks < ksteps; ++ks) {
            // tile ks + 1 has landed (waited below last round); barrier ks: everyone leaves stage (ks & 1) ... issue tile ks + 2 there
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            static_for<NDMA>([&](auto v_c) { issue(v_c, ks + 2, lds0 + (ks & 1) * STAGE, cw); });
        }
    } else {
        const int wm = (VAR == 1) ? (cw >> 1) : (cw >> 1), wn = cw & 1;
        const unsigned char* Aoff = lds + (VAR == 1 ? wm * 12 : wm * 24) * UNIT;
        const unsigned char* Boff = lds + A_BYTES + wn * 12 * UNIT;
        const unsigned rd = (unsigned)((lane & 15) * 64 + ((lane >> 4) ^ ((0x78 >> (2 * (((lane & 15) >> 2) & 3))) & 3)) * 16);
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 fb[NJ][3], fa[2][3];
        for (int ks = 0; ks < ksteps; ++ks) {
            const unsigned char* st = lds + (ks & 1) * STAGE;
            static_for<NJ * 3>([&](auto x_c) { constexpr int x = decltype(x_c)::value; fb[x / 3][x % 3] = *reinterpret_cast<const u32x4*>(st + (Boff - lds) + x * UNIT + rd); });
            static_for<3>([&](auto p_c) { constexpr int p = decltype(p_c)::value; fa[0][p] = *reinterpret_cast<const u32x4*>(st + (Aoff - lds) + p * UNIT + rd); });
            static_for<MI * NJ * 6>([&](auto s_c) {
                constexpr int sl = decltype(s_c)::value;
                constexpr int i = sl / 24, j = (sl % 24) / 6, pr = sl % 6;
                constexpr int pa = (pr < 3) ? 0 : (pr < 5) ? 1 : 2;
                constexpr int pb = (pr == 0 || pr == 3 || pr == 5) ? 0 : (pr == 1 || pr == 4) ? 1 : 2;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j][pb]), __builtin_bit_cast(bf16x8, fa[i & 1][pa]), acc[i][j], 0, 0, 0);
                if constexpr (i < MI - 1 && (sl % 24) < 9 && (sl % 24) % 3 == 1)
                    fa[(i + 1) & 1][(sl % 24) / 3] = *reinterpret_cast<const u32x4*>(st + (Aoff - lds) + ((i + 1) * 3 + (sl % 24) / 3) * UNIT + rd);
                if constexpr (!LOADER_SPLIT) {
                    // own DMAs: tile ks + 1 ... issued after the barrier of the previous step, spread over the first rows
                    constexpr int every = (VAR == 1) ? 5 : 5;
                    if constexpr (sl % every == 2 && sl / every < NDMA) issue(std::integral_constant<int, sl / every>{}, ks + 2, lds0 + (ks & 1) * STAGE + 0 * sl, wave);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VAR>
static void run(const char* name, const char* src, size_t src_bytes, float* out, unsigned long long* stamps, int ksteps, int nblocks) {
    auto k = lab_kernel<VAR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int threads = VAR == 0 ? 256 : 512;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k, dim3(nblocks), dim3(threads), LDS_BYTES, 0, src, src_bytes, out, ksteps, stamps);
    CK(hipDeviceSynchronize());
    const int reps = 60;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(nblocks), dim3(threads), LDS_BYTES, 0, src, src_bytes, out, ksteps, stamps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * nblocks);
    CK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * 2 * nblocks, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int b = 0; b < nblocks; ++b) { cyc.push_back((double)st[2 * b]); clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double us = ms / reps * 1e3;
    const double flop = 2.0 * 256 * 128 * 32 * 6 * (double)ksteps * nblocks;
    printf("%-44s %8.1f us/launch  %7.1f TF/s bf16 executed (%.3f of 2.5 PF; %5.1f fp32-equiv)  cycles/K-step %7.1f  clock %.3f GHz\n", name, us,
           flop / us / 1e6, flop / us / 1e6 / 2500.0, flop / 6 / us / 1e6, cyc[nblocks / 2] / ksteps, clk[nblocks / 2]);
}

int main(int argc, char** argv) {
    const int ksteps = argc > 1 ? atoi(argv[1]) : 256, nblocks = argc > 2 ? atoi(argv[2]) : 256;
    const size_t src_bytes = (size_t)1 << 30;
    char* src; float* out; unsigned long long* stamps;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&out, (size_t)nblocks * 512 * 4)); CK(hipMalloc(&stamps, (size_t)nblocks * 16));
    {   // random bf16 in [-1, 1): power depends on the data
        std::vector<unsigned short> h(1 << 24);
        srand(7);
        for (auto& v : h) v = (unsigned short)(0x3f00 | (rand() & 0x7f) | ((rand() & 1) << 15));
        for (size_t o = 0; o < src_bytes; o += h.size() * 2) CK(hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    for (int round = 0; round < 2; ++round) {
        run<0>("V4 : 4 waves, own DMAs (18 / wave)", src, src_bytes, out, stamps, ksteps, nblocks);
        run<1>("V8 : 8 waves x (64 x 64), own DMAs (9 / wave)", src, src_bytes, out, stamps, ksteps, nblocks);
        run<2>("V4L: 4 compute + 4 loader waves", src, src_bytes, out, stamps, ksteps, nblocks);
    }
    return 0;
}
