// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the VQ-VAE training step.
// wave = 64 lanes; everything here is written for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VQH_OK 0
#define VQH_ERR_ARG (-1)
#define VQH_ERR_LAUNCH (-2)
#define VQH_ERR_WORKSPACE (-3)

extern "C" void vqh_set_error(const char* msg);

#define VQH_CHECK_ARG(cond, msg)            \
    do {                                    \
        if (!(cond)) {                      \
            vqh_set_error(msg);             \
            return VQH_ERR_ARG;             \
        }                                   \
    } while (0)

#define VQH_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            vqh_set_error(hipGetErrorString(e__));           \
            return VQH_ERR_LAUNCH;                           \
        }                                                    \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: the dropout mask of element e at dropout site s in training step t is
// a pure function of (seed, t, s, e), so backward regenerates it instead of storing a mask.
// rng_state (device memory): [0] = seed, [1] = step counter (advanced once per step by a kernel,
// so a captured hipGraph replays with fresh masks).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

struct DropCfg {
    const unsigned long long* rng_state;  // device: {seed, step}; may be null when p == 0
    uint32_t site;                        // unique id of the dropout site in the network
    float p;                              // drop probability
    float scale;                          // 1/(1-p)
};

// keep-factor (0 or 1/(1-p)) for 4 consecutive elements starting at element index 4*quad.
__device__ __forceinline__ void drop4(const DropCfg& d, unsigned long long seed, unsigned long long step,
                                      unsigned long long quad, float out[4]) {
    uint4 c = make_uint4((uint32_t)quad, (uint32_t)(quad >> 32), d.site, (uint32_t)step);
    uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
    uint4 r = philox4x32_10(c, k);
    const float inv = 2.3283064365386963e-10f;  // 2^-32
    out[0] = (r.x * inv >= d.p) ? d.scale : 0.f;
    out[1] = (r.y * inv >= d.p) ? d.scale : 0.f;
    out[2] = (r.z * inv >= d.p) ? d.scale : 0.f;
    out[3] = (r.w * inv >= d.p) ? d.scale : 0.f;
}

// keep-factor of one element (elem = linear element index inside the dropout site's tensor)
__device__ __forceinline__ float drop1(const DropCfg& d, unsigned long long seed, unsigned long long step,
                                       unsigned long long elem) {
    float f[4];
    drop4(d, seed, step, elem >> 2, f);
    return f[elem & 3];
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float phi = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * phi;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
