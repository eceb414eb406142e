// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the VQ-VAE training step.
// wave = 64 lanes; everything here is written for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
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
// Dropout stream: the keep/drop decision of element e at dropout site s in training step t is a pure function of
// (seed, t, s, e), so backward regenerates it instead of storing a mask.  rng_state (device memory): [0] = seed,
// [1] = step counter (advanced once per step by a kernel, so a captured hipGraph replays with fresh masks).
// The function is a counter hash: key = mix32 chain over (seed, step, site); one mix32 (the 2-multiply "lowbias32"
// avalanche finaliser) of key ^ pair*GOLDEN yields 32 bits = the 16-bit draws of elements 2*pair and 2*pair+1, and an
// element is kept iff its draw >= thr = round(p * 65536).  The drop probability is therefore quantised to 2^-16
// (p = 0.1 -> 0.100006) and the keep scale is 1/(1 - thr/65536), so E[keep * scale] = 1 exactly.  About 6 VALU ops per
// decision; Philox4x32-10, used at first, cost ~25 and was visible in every fused epilogue (64 decisions per lane)
// and in attention.  The attention kernels use the same construction keyed per (row, key pair) -- attention.hip.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

struct DropCfg {
    const unsigned long long* rng_state;  // device: {seed, step}; may be null when p == 0
    uint32_t site;                        // unique id of the dropout site in the network
    float p;                              // effective drop probability thr / 65536 (0 = dropout off)
    float scale;                          // 1/(1-p)
    uint32_t thr;                         // keep iff 16-bit draw >= thr
};

// host: quantise the requested probability (see above)
inline DropCfg make_drop(const unsigned long long* rng_state, unsigned site, float p) {
    DropCfg d;
    d.rng_state = rng_state;
    d.site = site;
    long t = lrintf(p * 65536.f);
    if (t < 0) t = 0;
    if (t > 65535) t = 65535;
    d.thr = (uint32_t)t;
    d.p = (float)t / 65536.f;
    d.scale = 1.f / (1.f - d.p);
    return d;
}

// per-(seed, step, site) key: compute once per thread, then drop_keep / drop4 per element
__device__ __forceinline__ unsigned drop_key(const DropCfg& d, unsigned long long seed, unsigned long long step) {
    unsigned k = mix32((unsigned)seed ^ (d.site * 0x9E3779B9U));
    k = mix32(k ^ (unsigned)(seed >> 32) ^ ((unsigned)step * 0x85EBCA6BU));
    return mix32(k ^ (unsigned)(step >> 32) ^ 0x5bd1e995U);
}

// the two 16-bit draws of elements 2*pair (low half) and 2*pair+1 (high half)
__device__ __forceinline__ unsigned drop_pair_bits(unsigned key, unsigned long long pair) {
    const unsigned hi = (unsigned)(pair >> 32);
    if (hi) key = mix32(key ^ hi);
    return mix32(key ^ ((unsigned)pair * 0x9E3779B9U));
}

// keep-factor (0 or 1/(1-p)) of one element (elem = linear element index inside the dropout site's tensor)
__device__ __forceinline__ float drop_keep(const DropCfg& d, unsigned key, unsigned long long elem) {
    const unsigned bits = drop_pair_bits(key, elem >> 1);
    const unsigned draw = (elem & 1) ? (bits >> 16) : (bits & 0xffffu);
    return (draw >= d.thr) ? d.scale : 0.f;
}

// keep-factors of the 4 consecutive elements 4*quad .. 4*quad+3 (two hashes)
__device__ __forceinline__ void drop4(const DropCfg& d, unsigned key, unsigned long long quad, float out[4]) {
    const unsigned b0 = drop_pair_bits(key, quad * 2), b1 = drop_pair_bits(key, quad * 2 + 1);
    out[0] = ((b0 & 0xffffu) >= d.thr) ? d.scale : 0.f;
    out[1] = ((b0 >> 16) >= d.thr) ? d.scale : 0.f;
    out[2] = ((b1 & 0xffffu) >= d.thr) ? d.scale : 0.f;
    out[3] = ((b1 >> 16) >= d.thr) ? d.scale : 0.f;
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
