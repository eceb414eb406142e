// Fused optimizer kernels over the flat parameter / gradient buffers: global gradient norm
// (Lightning's gradient_clip_val -> torch.nn.utils.clip_grad_norm_, /root/reference/run.py:191-197)
// and AdamW (/root/reference/experiment.py:170; torch.optim.AdamW defaults betas=(0.9,0.999),
// eps=1e-8, decoupled weight decay applied to EVERY parameter).  All step-varying scalars (lr, step
// count, clip) are read from device memory so that a captured hipGraph can be replayed while the
// host updates them between replays.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    const long long n4 = n / 4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = g4[i];
        s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[n4 * 4 + threadIdx.x];
        s += (double)v * v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm_out[0] = sqrt(sum partials) ; norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ void gradnorm_finish_kernel(const double* __restrict__ part, int nparts, float* __restrict__ norm_out,
                                       const float* __restrict__ hyper) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
        // hyper[8] = gradient scale (1/world_size after a SUM all-reduce; 1 otherwise)
        const float gscale = hyper[8];
        const float nrm = (float)sqrt(s) * gscale;
        norm_out[0] = nrm;
        const float max_norm = hyper[5];
        float coef = gscale;
        if (max_norm > 0.f) coef = gscale * fminf(1.f, max_norm / (nrm + 1e-6f));
        norm_out[1] = coef;
    }
}

// hyper (device, floats): [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay [5]=max_norm
//                         [6]=bias_correction1 = 1-beta1^t  [7]=bias_correction2 = 1-beta2^t  [8]=grad scale
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    const float* __restrict__ hyper, const float* __restrict__ norm) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const float bc1 = hyper[6], bc2s = sqrtf(hyper[7]);
    const float coef = norm ? norm[1] : 1.f;
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = gi;                                   // clip_grad_norm_ scales .grad in place
        float pi = p[i];
        pi = pi * (1.f - lr * wd);                   // decoupled decay first (torch order)
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);         // lerp form used by torch
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

}  // namespace

// norm_out: 2 floats {total norm, clip coefficient}; workspace: >= 1024 doubles
extern "C" int vqh_grad_norm(const float* g, long long n, const float* hyper, float* norm_out, double* workspace,
                             hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && g && hyper && norm_out && workspace, "vqh_grad_norm: bad argument");
    VQH_CHECK_ARG((reinterpret_cast<uintptr_t>(g) & 15) == 0, "vqh_grad_norm: gradient buffer must be 16-byte aligned");
    const int nb = 1024;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, g, n, workspace);
    hipLaunchKernelGGL(gradnorm_finish_kernel, dim3(1), dim3(64), 0, stream, workspace, nb, norm_out, hyper);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_adamw_step(float* p, float* g, float* m, float* v, long long n, const float* hyper,
                              const float* norm, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && p && g && m && v && hyper, "vqh_adamw_step: bad argument");
    if (n == 0) return VQH_OK;
    long long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)nb), dim3(256), 0, stream, p, g, m, v, n, hyper, norm);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
