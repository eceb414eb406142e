// shape lab (diagnostic, not product): does the bf16 MFMA SHAPE change the wall time of the x3 K loop at equal work?
// MI355X_MICROARCH.md 'DVFS give-back' item 7: in power-limited bf16 loops v_mfma_f32_16x16x32_bf16 held a higher clock than
// v_mfma_f32_32x32x16_bf16 at about equal cycles per flop.  This lab runs the x3 tile's inner loop (256 x 128 macro tile, 4 waves
// x (128 x 64), three bf16 planes per operand resident in LDS, 36 ds_read_b128 fragment reads and 6 products per block and
// K-step of 32) in both shapes, with and without a VALU load that stands in for the in-loop operand split, on random data:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/shape_lab.hip -o tools/_build/shape_lab && tools/_build/shape_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TBM = 256, TBN = 128, TBK = 32;
constexpr int A_PLANE = TBM * TBK * 2, B_PLANE = TBN * TBK * 2;     // bytes
constexpr int LDS_BYTES = 3 * (A_PLANE + B_PLANE);                  // 73728

__device__ __forceinline__ bf16x8 lds_read16(const char* base, int off) {
    return *reinterpret_cast<const bf16x8*>(base + off);
}

// SHAPE 32: v_mfma_f32_32x32x16_bf16, SHAPE 16: v_mfma_f32_16x16x32_bf16.  VALU: fused multiply-adds per K-step and lane beside
// the MFMAs (the split costs ~264 VALU per wave and K-step in the product kernel).
template <int SHAPE, int VALU>
__global__ __launch_bounds__(256) void loop_kernel(const unsigned* __restrict__ seed, float* __restrict__ out, int ksteps,
                                                   unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill the planes with pseudo-random bf16 in [-1, 1) (same content for every workgroup: power depends on the data)
    for (int i = tid; i < LDS_BYTES / 4; i += 256) {
        unsigned x = seed[i & 4095] * 2654435761u + (unsigned)i * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned lo = 0x3f000000u | ((x & 0x7fu) << 16) | ((x & 0x80u) << 24);       // as fp32 bits -> take high halves
        const unsigned hi = 0x3f000000u | (((x >> 8) & 0x7fu) << 16) | (((x >> 8) & 0x80u) << 24);
        reinterpret_cast<unsigned*>(lds)[i] = (lo >> 16) | (hi & 0xffff0000u);
    }
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;                         // 2 x 2 waves: rows 128*wm, cols 64*wn
    const char* Ab = lds + wm * 128 * 64;                            // [rows][32 k] bf16 images, 64-B rows
    const char* Bb = lds + 3 * A_PLANE + wn * 64 * 64;
    float vacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vacc[i] = (float)(lane + i);
    unsigned long long t0 = 0, r0 = 0;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        const int r = lane & 31, h = lane >> 5;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 a[4][3], b[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 32 * i + r, ch = (2 * kh + h) ^ ((row >> 2) & 3);
                        a[i][p] = lds_read16(Ab + p * A_PLANE, row * 64 + ch * 16);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = 32 * j + r, ch = (2 * kh + h) ^ ((row >> 2) & 3);
                        b[j][p] = lds_read16(Bb + p * B_PLANE, row * 64 + ch * 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        if constexpr (VALU > 0) {
#pragma unroll
                            for (int v = 0; v < VALU / 16; ++v) vacc[v & 7] = __builtin_fmaf(vacc[v & 7], 1.0000001f, vacc[(v + 1) & 7]);
                        }
                    }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) s += acc[i][j][q];
        vacc[0] += s;
    } else {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        const int r = lane & 15, g = lane >> 4;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {                    // two halves of the wave tile's rows: keeps the live fragments at 36 registers
                bf16x8 a[4][3], b[4][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 64 * half + 16 * i + r, ch = g ^ ((row >> 1) & 3);
                        a[i][p] = lds_read16(Ab + p * A_PLANE, row * 64 + ch * 16);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 16 * j + r, ch = g ^ ((row >> 1) & 3);
                        b[j][p] = lds_read16(Bb + p * B_PLANE, row * 64 + ch * 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4& c = acc[4 * half + i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                        if constexpr (VALU > 0) {
#pragma unroll
                            for (int v = 0; v < VALU / 32; ++v) vacc[v & 7] = __builtin_fmaf(vacc[v & 7], 1.0000001f, vacc[(v + 1) & 7]);
                        }
                    }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) s += acc[i][j][q];
        vacc[0] += s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += vacc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int VALU>
static void run(const char* name, const unsigned* seed, float* out, unsigned long long* stamps, int ksteps, int nblocks) {
    auto k = loop_kernel<SHAPE, VALU>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 60; ++w) hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), LDS_BYTES, 0, seed, out, ksteps, stamps);
    CK(hipDeviceSynchronize());
    const int reps = 100;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), LDS_BYTES, 0, seed, out, ksteps, stamps);
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
    const double flop = 2.0 * 256 * 128 * 32 * 6 * (double)ksteps * nblocks;      // executed bf16 flops
    printf("%-34s %8.1f us/launch  %7.1f TF/s bf16 executed (%5.1f fp32-equiv)  cycles/K-step %7.1f  clock %.3f GHz\n", name, us,
           flop / us / 1e6, flop / 6 / us / 1e6, cyc[nblocks / 2] / ksteps, clk[nblocks / 2]);
}

int main(int argc, char** argv) {
    const int ksteps = argc > 1 ? atoi(argv[1]) : 512, nblocks = argc > 2 ? atoi(argv[2]) : 256;
    unsigned* seed; float* out; unsigned long long* stamps;
    CK(hipMalloc(&seed, 4096 * 4)); CK(hipMalloc(&out, (size_t)nblocks * 256 * 4)); CK(hipMalloc(&stamps, (size_t)nblocks * 16));
    std::vector<unsigned> hs(4096);
    srand(1);
    for (auto& v : hs) v = (unsigned)rand() * 2654435761u;
    CK(hipMemcpy(seed, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {
        run<32, 0>("32x32x16, reads + MFMA", seed, out, stamps, ksteps, nblocks);
        run<16, 0>("16x16x32, reads + MFMA", seed, out, stamps, ksteps, nblocks);
        run<32, 256>("32x32x16, + 256 VALU per K-step", seed, out, stamps, ksteps, nblocks);
        run<16, 256>("16x16x32, + 256 VALU per K-step", seed, out, stamps, ksteps, nblocks);
    }
    return 0;
}
