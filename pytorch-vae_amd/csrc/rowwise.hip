// HBM-bound row-wise kernels of the training step: LayerNorm forward/backward, the 3->H input
// embeddings (+dropout +positional table), broadcast parameters, column sums (bias gradients),
// dropout backward.  One wave (64 lanes) per row, float4 accesses when the row pitch allows it.
// Reference call sites: nn.LayerNorm instances /root/reference/models/vq_vae.py:462-465,501,524 and the
// per-layer norms; input_proj/ss_input_proj + pos_enc :642-650; query_embed + pos_enc :750-751.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WAVES_PER_BLOCK = 4;

// ------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * w + b ; saves mean / rstd per row for backward.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ y, int ldy, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int i = lane; i < H; i += 64) s += xr[i];
    const float mu = wave_sum(s) / (float)H;
    float v = 0.f;
    for (int i = lane; i < H; i += 64) {
        const float d = xr[i] - mu;
        v += d * d;
    }
    const float rs = rsqrtf(wave_sum(v) / (float)H + eps);
    float* yr = y + (size_t)row * ldy;
    for (int i = lane; i < H; i += 64) yr[i] = (xr[i] - mu) * rs * w[i] + b[i];
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// 16-byte form: the row lives in registers (NV quads per lane), one read and one write of x/y per element -- the scalar
// kernel above walks the row three times with 4-byte accesses and reached 3.4 TB/s (43 % of HBM) at H = 512.
// Two-pass statistics (mean, then centred squares) on the registers: same formula as the scalar kernel.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ y, int ldy, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int rows, int H, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = H >> 2;
    f32x4 wv[NV], bv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = lane + 64 * j;
        wv[j] = (q < nq) ? *reinterpret_cast<const f32x4*>(w + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j] = (q < nq) ? *reinterpret_cast<const f32x4*>(b + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invH = 1.f / (float)H;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wave; row < rows; row += gridDim.x * WAVES_PER_BLOCK) {
        const float* xr = x + (size_t)row * ldx;
        f32x4 xv[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            xv[j] = (q < nq) ? *reinterpret_cast<const f32x4*>(xr + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
        }
        const float mu = wave_sum(s) * invH;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            if (q < nq) {
                const f32x4 d = xv[j] - mu;
                v += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
        const float rs = rsqrtf(wave_sum(v) * invH + eps);
        float* yr = y + (size_t)row * ldy;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            if (q < nq) *reinterpret_cast<f32x4*>(yr + 4 * q) = (xv[j] - mu) * rs * wv[j] + bv[j];
        }
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * w.
// Each wave walks rows (grid-stride) and keeps dw/db partial sums for its columns in registers;
// the 4 waves of a block are combined through LDS and written as one slab per block:
//   part[block][0][H] = sum dy*xhat ,  part[block][1][H] = sum dy
// (summed by vqh_reduce_slabs: deterministic, no atomics).
// ------------------------------------------------------------------------------------------
template <int VPT>  // values per lane: H <= 64*VPT
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                            const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            int lddx, int accumulate, float* __restrict__ part,
                                                            int rows, int H) {
    __shared__ float red[WAVES_PER_BLOCK][2][64 * VPT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aw[VPT], ab[VPT], wv[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        aw[j] = 0.f;
        ab[j] = 0.f;
        const int i = lane + 64 * j;
        wv[j] = (i < H) ? w[i] : 0.f;
    }
    const float invH = 1.f / (float)H;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wave; row < rows; row += gridDim.x * WAVES_PER_BLOCK) {
        const float* dyr = dy + (size_t)row * lddy;
        const float* xr = x + (size_t)row * ldx;
        const float mu = mean[row], rs = rstd[row];
        float g[VPT], xh[VPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int i = lane + 64 * j;
            float d = 0.f, xv = 0.f;
            if (i < H) {
                d = dyr[i];
                xv = (xr[i] - mu) * rs;
            }
            xh[j] = xv;
            g[j] = d * wv[j];
            aw[j] += d * xv;
            ab[j] += d;
            s1 += g[j];
            s2 += g[j] * xv;
        }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
        float* dxr = dx + (size_t)row * lddx;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int i = lane + 64 * j;
            if (i < H) {
                const float v = rs * (g[j] - s1 - xh[j] * s2);
                dxr[i] = accumulate ? dxr[i] + v : v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        red[wave][0][lane + 64 * j] = aw[j];
        red[wave][1][lane + 64 * j] = ab[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H; i += 256) {
        const int which = i / H, c = i % H;
        float s = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < WAVES_PER_BLOCK; ++wv_) s += red[wv_][which][c];
        part[((size_t)blockIdx.x * 2 + which) * H + c] = s;
    }
}

// float4 variant: lane owns float4 groups q = lane + 64*j (H % 4 == 0, 16-byte aligned rows); 1 KiB per wave
// instruction instead of 256 B.  Same slab layout as the scalar kernel.
template <int NV>  // float4 groups per lane: H <= 256*NV
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, int lddy,
                                                                const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ dx,
                                                                int lddx, int accumulate, float* __restrict__ part,
                                                                int rows, int H, float* __restrict__ dx_drop, DropCfg drop) {
    // dx_drop (optional, [rows, H] contiguous): the final dx times the keep mask of the dropout site that consumes it
    // next in backward -- the residual-stream gradient is written once more here instead of being re-read by a
    // separate dropout-backward pass
    const unsigned dkey = dx_drop ? drop_key(drop, drop.rng_state[0], drop.rng_state[1]) : 0u;
    __shared__ float red[WAVES_PER_BLOCK][2][256 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = H >> 2;
    f32x4 aw[NV], ab[NV], wv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = lane + 64 * j;
        wv[j] = (q < nq) ? *reinterpret_cast<const f32x4*>(w + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invH = 1.f / (float)H;
    for (int row = blockIdx.x * WAVES_PER_BLOCK + wave; row < rows; row += gridDim.x * WAVES_PER_BLOCK) {
        const float* dyr = dy + (size_t)row * lddy;
        const float* xr = x + (size_t)row * ldx;
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            f32x4 d = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
            if (q < nq) {
                d = *reinterpret_cast<const f32x4*>(dyr + 4 * q);
                xv = *reinterpret_cast<const f32x4*>(xr + 4 * q);
                xv = (xv - mu) * rs;
            }
            xh[j] = xv;
            g[j] = d * wv[j];
            aw[j] += d * xv;
            ab[j] += d;
            const f32x4 gx = g[j] * xv;
            s1 += (g[j][0] + g[j][1]) + (g[j][2] + g[j][3]);
            s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
        float* dxr = dx + (size_t)row * lddx;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = lane + 64 * j;
            if (q < nq) {
                f32x4 v = (g[j] - s1 - xh[j] * s2) * rs;
                if (accumulate) v += *reinterpret_cast<const f32x4*>(dxr + 4 * q);
                *reinterpret_cast<f32x4*>(dxr + 4 * q) = v;
                if (dx_drop) {
                    float f[4];
                    drop4(drop, dkey, ((unsigned long long)row * H + 4 * q) >> 2, f);
                    v[0] *= f[0]; v[1] *= f[1]; v[2] *= f[2]; v[3] *= f[3];
                    *reinterpret_cast<f32x4*>(dx_drop + (size_t)row * H + 4 * q) = v;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][0][4 * (lane + 64 * j) + e] = aw[j][e];
            red[wave][1][4 * (lane + 64 * j) + e] = ab[j][e];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H; i += 256) {
        const int which = i / H, c = i % H;
        float s = 0.f;
#pragma unroll
        for (int wv_ = 0; wv_ < WAVES_PER_BLOCK; ++wv_) s += red[wv_][which][c];
        part[((size_t)blockIdx.x * 2 + which) * H + c] = s;
    }
}

// LayerNorm weight / bias gradients from the per-block slabs [S][2][H]: column i < H goes to dw, i >= H to db.
// Same fixed summation order as reduce_slabs_kernel.
__global__ __launch_bounds__(1024) void reduce_slabs2_kernel(const float* __restrict__ slabs, int S, int H,
                                                             float* __restrict__ dw, float* __restrict__ db, float beta) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx, n = 2 * H;
    float s = 0.f;
    if (i < n) {
        int k = ty;
        for (; k + 48 < S; k += 64) {
            const float a = slabs[(size_t)k * n + i], b = slabs[(size_t)(k + 16) * n + i];
            const float c = slabs[(size_t)(k + 32) * n + i], d = slabs[(size_t)(k + 48) * n + i];
            s += (a + b) + (c + d);
        }
        for (; k < S; k += 16) s += slabs[(size_t)k * n + i];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][tx];
        float* o = (i < H) ? dw + i : db + (i - H);
        *o = (beta != 0.f) ? beta * (*o) + t : t;
    }
}

// out[i] = beta*out[i] + sum_s slabs[s*stride + i]
// block = 64 columns x 16 slab lanes: every thread sums S/16 slabs with independent loads, the 16 partials of
// a column are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, int S, long long stride,
                                                            long long n, float* __restrict__ out, float beta) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < n) {
        int k = ty;
        for (; k + 48 < S; k += 64) {
            const float a = slabs[(size_t)k * stride + i], b = slabs[(size_t)(k + 16) * stride + i];
            const float c = slabs[(size_t)(k + 32) * stride + i], d = slabs[(size_t)(k + 48) * stride + i];
            s += (a + b) + (c + d);
        }
        for (; k < S; k += 16) s += slabs[(size_t)k * stride + i];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][tx];
        out[i] = (beta != 0.f) ? beta * out[i] + t : t;
    }
}

// ------------------------------------------------------------------------------------------
// column sums: part[blk][n] = sum over the block's rows of X[m][n]   (bias / broadcast-param grads)
// block (64 columns x 4 row lanes); grid (ceil(N/64), row_blocks)
// ------------------------------------------------------------------------------------------
// float4 variant: block = 64 float4 column groups (256 columns) x 4 row lanes; grid (ceil(N/256), row_blocks)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ X, int ld, int M, int N,
                                                         float* __restrict__ part, int rows_per_block) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c < N) {
        int r = r0 + ry;
        for (; r + 4 < r1; r += 8) {
            s0 += *reinterpret_cast<const f32x4*>(X + (size_t)r * ld + c);
            s1 += *reinterpret_cast<const f32x4*>(X + (size_t)(r + 4) * ld + c);
        }
        for (; r < r1; r += 4) s0 += *reinterpret_cast<const f32x4*>(X + (size_t)r * ld + c);
    }
    red[ry][lane] = s0 + s1;
    __syncthreads();
    if (ry == 0 && c < N) {
        const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * N + c) = t;
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, int M, int N,
                                                     float* __restrict__ part, int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + ry; r < r1; r += 4) s += X[(size_t)r * ld + c];
    red[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < N)
        part[(size_t)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// input embedding: out[r,:] = dropout(W[:, 0:3] . x[r, c0:c0+3] + b) + pe[r % L, :]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, int ldx, int c0,
                                                        const float* __restrict__ W, const float* __restrict__ b,
                                                        const float* __restrict__ pe, float* __restrict__ out,
                                                        int rows, int L, int H, DropCfg drop) {
    unsigned dkey = 0;
    if (drop.p > 0.f) dkey = drop_key(drop, drop.rng_state[0], drop.rng_state[1]);
    const long long total = (long long)rows * H;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / H), hcol = (int)(e % H);
        const float* xr = x + (size_t)r * ldx + c0;
        float v = b[hcol] + W[hcol * 3 + 0] * xr[0] + W[hcol * 3 + 1] * xr[1] + W[hcol * 3 + 2] * xr[2];
        if (drop.p > 0.f) v *= drop_keep(drop, dkey, (unsigned long long)e);
        out[e] = v + pe[(size_t)(r % L) * H + hcol];
    }
}

// embedding backward: part[blk][c][h] (c = 0..2 weight columns, c = 3 bias) over the block's rows
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        int ldx, int c0, float* __restrict__ part, int rows, int H,
                                                        int rows_per_block, DropCfg drop) {
    __shared__ float red[4][4][64];
    unsigned dkey = 0;
    if (drop.p > 0.f) dkey = drop_key(drop, drop.rng_state[0], drop.rng_state[1]);
    const int hcol = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (hcol < H)
        for (int r = r0 + ry; r < r1; r += 4) {
            float d = dy[(size_t)r * H + hcol];
            if (drop.p > 0.f) d *= drop_keep(drop, dkey, (unsigned long long)r * H + hcol);
            const float* xr = x + (size_t)r * ldx + c0;
            a0 += d * xr[0];
            a1 += d * xr[1];
            a2 += d * xr[2];
            a3 += d;
        }
    const int l = threadIdx.x & 63;
    red[ry][0][l] = a0; red[ry][1][l] = a1; red[ry][2][l] = a2; red[ry][3][l] = a3;
    __syncthreads();
    if (ry == 0 && hcol < H) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            part[((size_t)blockIdx.y * 4 + c) * H + hcol] = red[0][c][l] + red[1][c][l] + red[2][c][l] + red[3][c][l];
    }
}

// out[b, i] = p0[i] (+ p1[i])   for b in [0,B): broadcast of a parameter block over the batch
__global__ void bcast_rows_kernel(const float* __restrict__ p0, const float* __restrict__ p1, float* __restrict__ out,
                                  int B, long long n) {
    const long long total = (long long)B * n;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e % n;
        out[e] = p1 ? p0[i] + p1[i] : p0[i];
    }
}

// out = dy * keep(site, elem)   (dropout backward; the same stream as the forward site)
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ out, long long n, DropCfg drop) {
    const unsigned dkey = drop_key(drop, drop.rng_state[0], drop.rng_state[1]);
    const long long nq = (n + 3) / 4;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        float f[4];
        drop4(drop, dkey, (unsigned long long)q, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = q * 4 + e;
            if (i < n) out[i] = dy[i] * f[e];
        }
    }
}

// out = a + b (elementwise, float4 when possible)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// out = dy * y * (1 - y)   (sigmoid backward through its saved output)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = dy[i] * y[i] * (1.f - y[i]);
}

// y = sigmoid(x) in place variant helpers are in the GEMM epilogue; strided copy for concat layouts:
__global__ void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c];
    }
}

// dW[h][c] = beta*dW + s[c][h] ; db[h] = beta*db + s[3][h]
__global__ void embed_scatter_kernel(const float* __restrict__ s, float* __restrict__ dW, float* __restrict__ db,
                                     float beta, int H) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    for (int c = 0; c < 3; ++c) {
        const float v = s[c * H + h];
        dW[h * 3 + c] = (beta != 0.f) ? beta * dW[h * 3 + c] + v : v;
    }
    const float v = s[3 * H + h];
    db[h] = (beta != 0.f) ? beta * db[h] + v : v;
}

// rigid augmentation + coordinate noise of the INPUT only (reference models/vq_vae.py:775-792, _random_rotation :331-345)
// u[b,0:3] uniforms -> unit quaternion -> R_b ; out.xyz = R_b . xyz + t_b (+ noise) ; the SS channels are copied.
__global__ void augment_kernel(const float* __restrict__ x, const float* __restrict__ u, const float* __restrict__ t,
                               const float* __restrict__ noise, float* __restrict__ out, int B, int L) {
    const long long total = (long long)B * L;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(e / L);
        const float* xi = x + e * 6;
        float px = xi[0], py = xi[1], pz = xi[2];
        if (u) {
            const float u1 = u[b * 3], u2 = u[b * 3 + 1], u3 = u[b * 3 + 2];
            const float two_pi = 6.28318530717958647692f;
            const float qx = sqrtf(1.f - u1) * sinf(two_pi * u2), qy = sqrtf(1.f - u1) * cosf(two_pi * u2);
            const float qz = sqrtf(u1) * sinf(two_pi * u3), qw = sqrtf(u1) * cosf(two_pi * u3);
            const float r00 = 1.f - 2.f * (qy * qy + qz * qz), r01 = 2.f * (qx * qy - qz * qw), r02 = 2.f * (qx * qz + qy * qw);
            const float r10 = 2.f * (qx * qy + qz * qw), r11 = 1.f - 2.f * (qx * qx + qz * qz), r12 = 2.f * (qy * qz - qx * qw);
            const float r20 = 2.f * (qx * qz - qy * qw), r21 = 2.f * (qy * qz + qx * qw), r22 = 1.f - 2.f * (qx * qx + qy * qy);
            const float ax = r00 * px + r01 * py + r02 * pz + t[b * 3];
            const float ay = r10 * px + r11 * py + r12 * pz + t[b * 3 + 1];
            const float az = r20 * px + r21 * py + r22 * pz + t[b * 3 + 2];
            px = ax; py = ay; pz = az;
        }
        if (noise) { px += noise[e * 3]; py += noise[e * 3 + 1]; pz += noise[e * 3 + 2]; }
        float* o = out + e * 6;
        o[0] = px; o[1] = py; o[2] = pz; o[3] = xi[3]; o[4] = xi[4]; o[5] = xi[5];
    }
}

// P[r,:] = softmax(scale * S[r,:] + colbias) in place; one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, int ld, const float* __restrict__ colbias,
                                                           float scale, int R, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float* sr = S + (size_t)row * ld;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, scale * sr[k] + (colbias ? colbias[k] : 0.f));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float e = expf(scale * sr[k] + (colbias ? colbias[k] : 0.f) - mx);
        sr[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < K; k += 64) sr[k] *= inv;
}

// dlogits = P * (g - sum_k P_k g_k) in place (g is a per-column vector)
__global__ __launch_bounds__(256) void softmax_bwd_colgrad_kernel(float* __restrict__ P, int ld, const float* __restrict__ g,
                                                                  int R, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float* pr = P + (size_t)row * ld;
    float dotv = 0.f;
    for (int k = lane; k < K; k += 64) dotv += pr[k] * g[k];
    dotv = wave_sum(dotv);
    for (int k = lane; k < K; k += 64) pr[k] = pr[k] * (g[k] - dotv);
}

// usage-entropy regulariser (reference :1299-1309): pc = colsum/R ; reg = lambda * sum pc*log(max(pc,1e-12));
// g[k] = d reg / d P[r,k] = lambda/R * d(pc log clamp(pc))/dpc ; metrics[loss] += reg ; metrics[usage_reg] = reg
__global__ __launch_bounds__(256) void usage_entropy_finish_kernel(const float* __restrict__ colsum, int K, float invR,
                                                                   float lambda, float* __restrict__ g,
                                                                   float* __restrict__ metrics, int i_loss, int i_reg) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float pc = colsum[k] * invR;
        const float lg = logf(fmaxf(pc, 1e-12f));
        acc += pc * lg;
        g[k] = lambda * invR * (lg + (pc > 1e-12f ? 1.f : 0.f));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float reg = lambda * (red[0] + red[1] + red[2] + red[3]);
        metrics[i_loss] += reg;
        metrics[i_reg] = reg;
    }
}

// soft-VQ mix (reference :843-853): z_dec = ze + (((1-alpha)*z_soft + alpha*z_hard) - ze)
__global__ void vq_mix_kernel(const float* __restrict__ ze, const float* __restrict__ zsoft, const float* __restrict__ zhard,
                              float alpha, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float mix = __fadd_rn(__fmul_rn(1.f - alpha, zsoft[i]), __fmul_rn(alpha, zhard[i]));
        out[i] = __fadd_rn(ze[i], __fsub_rn(mix, ze[i]));
    }
}

inline int blocks_for(long long n, int per_block = 256, int cap = 4096) {
    long long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

extern "C" int vqh_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                                 float* mean, float* rstd, int rows, int H, float eps, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && H > 0 && ldx >= H && ldy >= H, "vqh_layernorm_fwd: bad shape");
    if (rows == 0) return VQH_OK;
    VQH_CHECK_ARG(x && w && b && y, "vqh_layernorm_fwd: null pointer");
    const int grid = (rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const bool vec = (H % 4 == 0) && H <= 1024 && ((ldx | ldy) % 4 == 0) &&
                     (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) |
                        reinterpret_cast<uintptr_t>(b)) & 15) == 0);
    if (vec) {
        const int gv = grid > 2048 ? 2048 : grid;            // grid-stride rows: weights stay in registers
        if (H <= 256) hipLaunchKernelGGL((layernorm_fwd_vec_kernel<1>), dim3(gv), dim3(256), 0, stream, x, ldx, w, b, y, ldy, mean, rstd, rows, H, eps);
        else if (H <= 512) hipLaunchKernelGGL((layernorm_fwd_vec_kernel<2>), dim3(gv), dim3(256), 0, stream, x, ldx, w, b, y, ldy, mean, rstd, rows, H, eps);
        else hipLaunchKernelGGL((layernorm_fwd_vec_kernel<4>), dim3(gv), dim3(256), 0, stream, x, ldx, w, b, y, ldy, mean, rstd, rows, H, eps);
    } else {
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, w, b, y, ldy, mean, rstd, rows, H, eps);
    }
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_reduce_slabs(const float* slabs, int S, long long stride, long long n, float* out, float beta,
                                hipStream_t stream) {
    VQH_CHECK_ARG(S >= 0 && n >= 0, "vqh_reduce_slabs: bad shape");
    if (n == 0) return VQH_OK;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, stream, slabs, S, stride, n, out, beta);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

// dw/db are written as beta*old + sum (beta=0 overwrite).  workspace >= 2*H*nblocks floats.
extern "C" int vqh_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* w,
                                 const float* mean, const float* rstd, float* dx, int lddx, int accumulate_dx,
                                 float* dw, float* db, float beta, int rows, int H, float* dx_drop,
                                 const unsigned long long* rng_state, unsigned drop_site, float drop_p, float* workspace,
                                 long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && H > 0 && H <= 2048, "vqh_layernorm_bwd: H must be in [1,2048]");
    if (rows == 0) return VQH_OK;
    VQH_CHECK_ARG(dy && x && w && mean && rstd && dx && dw && db && workspace, "vqh_layernorm_bwd: null pointer");
    VQH_CHECK_ARG(!dx_drop || (rng_state && drop_p > 0.f && drop_p < 1.f), "vqh_layernorm_bwd: dx_drop needs rng_state and 0 < p < 1");
    int nblk = (rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    {   // workgroups = partial-sum slabs the second kernel has to read: 1024 (measured: 512 the same, 256 +0.6 ms per step) for this
        // memory-bound kernel; VQH_LN_BWD_BLOCKS overrides (A/B runs)
        static int cap = 0;
        if (cap == 0) {
            const char* e = getenv("VQH_LN_BWD_BLOCKS");
            cap = e ? atoi(e) : 1024;
            if (cap < 1) cap = 1024;
        }
        if (nblk > cap) nblk = cap;
    }
    VQH_CHECK_ARG((long long)nblk * 2 * H <= workspace_floats, "vqh_layernorm_bwd: workspace too small");
    const bool vec = (H % 4 == 0) && H <= 1024 && ((lddy | ldx | lddx) % 4 == 0) &&
                     (((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                        reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx_drop)) & 15) == 0);
    const DropCfg dcfg = dx_drop ? make_drop(rng_state, drop_site, drop_p) : DropCfg{};
#define LN_BWD_V(V)                                                                                                    \
    hipLaunchKernelGGL((layernorm_bwd_vec_kernel<V>), dim3(nblk), dim3(256), 0, stream, dy, lddy, x, ldx, w, mean, rstd, \
                       dx, lddx, accumulate_dx, workspace, rows, H, dx_drop, dcfg)
    if (vec) {
        if (H <= 256) LN_BWD_V(1);
        else if (H <= 512) LN_BWD_V(2);
        else LN_BWD_V(4);
    } else
#define LN_BWD(V)                                                                                                  \
    hipLaunchKernelGGL((layernorm_bwd_kernel<V>), dim3(nblk), dim3(256), 0, stream, dy, lddy, x, ldx, w, mean, rstd, \
                       dx, lddx, accumulate_dx, workspace, rows, H)
    if (H <= 64) LN_BWD(1);
    else if (H <= 128) LN_BWD(2);
    else if (H <= 256) LN_BWD(4);
    else if (H <= 512) LN_BWD(8);
    else if (H <= 1024) LN_BWD(16);
    else LN_BWD(32);
#undef LN_BWD
#undef LN_BWD_V
    VQH_LAUNCH_CHECK();
    if (dx_drop && !vec) {        // scalar kernel: the mask is applied by the stand-alone pass (needs a dense dx)
        VQH_CHECK_ARG(lddx == H, "vqh_layernorm_bwd: dx_drop on the unaligned path needs a dense dx");
        hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks_for(((long long)rows * H + 3) / 4)), dim3(256), 0, stream, dx, dx_drop,
                           (long long)rows * H, dcfg);
    }
    // slab layout [blk][2][H]: dw = sum_blk slab[blk][0], db = sum_blk slab[blk][1]  (one launch for both)
    hipLaunchKernelGGL(reduce_slabs2_kernel, dim3((2 * H + 63) / 64), dim3(1024), 0, stream, workspace, nblk, H, dw, db, beta);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

// out[n] = beta*out[n] + sum_m X[m][n]
extern "C" int vqh_colsum(const float* X, int ld, int M, int N, float* out, float beta, float* workspace,
                          long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(M >= 0 && N >= 0 && ld >= N, "vqh_colsum: bad shape");
    if (N == 0) return VQH_OK;
    VQH_CHECK_ARG(out && workspace, "vqh_colsum: null pointer");
    int rb = (M + 255) / 256;
    if (rb < 1) rb = 1;
    if (rb > 128) rb = 128;
    const int rows_per_block = (M + rb - 1) / rb > 0 ? (M + rb - 1) / rb : 1;
    VQH_CHECK_ARG((long long)rb * N <= workspace_floats, "vqh_colsum: workspace too small");
    if ((N & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0)
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((N + 255) / 256, rb), dim3(256), 0, stream, X, ld, M, N, workspace, rows_per_block);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, rb), dim3(256), 0, stream, X, ld, M, N, workspace, rows_per_block);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((N + 63) / 64), dim3(1024), 0, stream, workspace, rb, (long long)N,
                       (long long)N, out, beta);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_embed_fwd(const float* x, int ldx, int col0, const float* W, const float* b, const float* pe,
                             float* out, int rows, int L, int H, const unsigned long long* rng_state,
                             unsigned drop_site, float drop_p, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && L > 0 && H > 0 && ldx >= col0 + 3, "vqh_embed_fwd: bad shape");
    if (rows == 0) return VQH_OK;
    VQH_CHECK_ARG(x && W && b && pe && out, "vqh_embed_fwd: null pointer");
    VQH_CHECK_ARG(drop_p == 0.f || rng_state, "vqh_embed_fwd: dropout needs rng_state");
    DropCfg d = make_drop(rng_state, drop_site, drop_p);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blocks_for((long long)rows * H)), dim3(256), 0, stream, x, ldx, col0, W, b,
                       pe, out, rows, L, H, d);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

// dW[H,3] and db[H] (beta*old + sum).  workspace >= 4*H*row_blocks floats
extern "C" int vqh_embed_bwd(const float* dy, const float* x, int ldx, int col0, float* dW, float* db, float beta,
                             int rows, int H, const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                             float* workspace, long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && H > 0, "vqh_embed_bwd: bad shape");
    VQH_CHECK_ARG(dy && x && dW && db && workspace, "vqh_embed_bwd: null pointer");
    VQH_CHECK_ARG(drop_p == 0.f || rng_state, "vqh_embed_bwd: dropout needs rng_state");
    int rb = (rows + 255) / 256;
    if (rb < 1) rb = 1;
    if (rb > 128) rb = 128;
    const int rpb = (rows + rb - 1) / rb > 0 ? (rows + rb - 1) / rb : 1;
    VQH_CHECK_ARG((long long)rb * 4 * H <= workspace_floats, "vqh_embed_bwd: workspace too small");
    DropCfg d = make_drop(rng_state, drop_site, drop_p);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((H + 63) / 64, rb), dim3(256), 0, stream, dy, x, ldx, col0, workspace, rows,
                       H, rpb, d);
    // slabs [blk][4][H]: gather weight column c into dW[h*3+c] needs a transposing reduce; reduce into a
    // scratch [4][H] at the end of the workspace region first, then scatter.
    VQH_LAUNCH_CHECK();
    float* scratch = workspace + (size_t)rb * 4 * H;
    VQH_CHECK_ARG((long long)rb * 4 * H + 4 * H <= workspace_floats, "vqh_embed_bwd: workspace too small");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((4 * H + 63) / 64), dim3(1024), 0, stream, workspace, rb,
                       (long long)4 * H, (long long)4 * H, scratch, 0.f);
    // dW[h][c] = beta*dW + scratch[c][h] ; db[h] = beta*db + scratch[3][h]
    hipLaunchKernelGGL(embed_scatter_kernel, dim3((H + 255) / 256), dim3(256), 0, stream, scratch, dW, db, beta, H);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_bcast_rows(const float* p0, const float* p1, float* out, int B, long long n, hipStream_t stream) {
    VQH_CHECK_ARG(B >= 0 && n >= 0, "vqh_bcast_rows: bad shape");
    if (B == 0 || n == 0) return VQH_OK;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(blocks_for((long long)B * n)), dim3(256), 0, stream, p0, p1, out, B, n);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_dropout_bwd(const float* dy, float* out, long long n, const unsigned long long* rng_state,
                               unsigned drop_site, float drop_p, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && drop_p > 0.f && drop_p < 1.f && rng_state, "vqh_dropout_bwd: bad argument");
    if (n == 0) return VQH_OK;
    DropCfg d = make_drop(rng_state, drop_site, drop_p);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, stream, dy, out, n, d);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_add(const float* a, const float* b, float* out, long long n, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0, "vqh_add: bad n");
    if (n == 0) return VQH_OK;
    hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, a, b, out, n);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_copy2d(const float* src, int lds_, float* dst, int ldd, int rows, int cols, hipStream_t stream) {
    VQH_CHECK_ARG(rows >= 0 && cols >= 0 && lds_ >= cols && ldd >= cols, "vqh_copy2d: bad shape");
    if (rows == 0 || cols == 0) return VQH_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for((long long)rows * cols)), dim3(256), 0, stream, src, lds_, dst, ldd,
                       rows, cols);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_sigmoid_bwd(const float* dy, const float* y, float* out, long long n, hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0, "vqh_sigmoid_bwd: bad n");
    if (n == 0) return VQH_OK;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, dy, y, out, n);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_memset(void* ptr, int value, long long bytes, hipStream_t stream) {
    VQH_CHECK_ARG(bytes >= 0, "vqh_memset: bad size");
    if (bytes == 0) return VQH_OK;
    hipError_t e = hipMemsetAsync(ptr, value, (size_t)bytes, stream);
    if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
    return VQH_OK;
}

extern "C" int vqh_augment(const float* x, const float* u, const float* t, const float* noise, float* out, int B, int L,
                           hipStream_t stream) {
    VQH_CHECK_ARG(B >= 0 && L >= 0 && x && out && (!u || t), "vqh_augment: bad argument");
    if (B == 0 || L == 0) return VQH_OK;
    hipLaunchKernelGGL(augment_kernel, dim3(blocks_for((long long)B * L)), dim3(256), 0, stream, x, u, t, noise, out, B, L);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_softmax_rows(float* S, int ld, const float* colbias, float scale, int R, int K, hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && K > 0 && ld >= K && S, "vqh_softmax_rows: bad argument");
    if (R == 0) return VQH_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, S, ld, colbias, scale, R, K);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_softmax_bwd_colgrad(float* P, int ld, const float* g, int R, int K, hipStream_t stream) {
    VQH_CHECK_ARG(R >= 0 && K > 0 && ld >= K && P && g, "vqh_softmax_bwd_colgrad: bad argument");
    if (R == 0) return VQH_OK;
    hipLaunchKernelGGL(softmax_bwd_colgrad_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, P, ld, g, R, K);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_usage_entropy_finish(const float* colsum, int K, int R, float lambda, float* g, float* metrics,
                                        int i_loss, int i_reg, hipStream_t stream) {
    VQH_CHECK_ARG(K > 0 && R > 0 && colsum && g && metrics, "vqh_usage_entropy_finish: bad argument");
    hipLaunchKernelGGL(usage_entropy_finish_kernel, dim3(1), dim3(256), 0, stream, colsum, K, 1.f / (float)R, lambda, g,
                       metrics, i_loss, i_reg);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}

extern "C" int vqh_vq_mix(const float* ze, const float* zsoft, const float* zhard, float alpha, float* out, long long n,
                          hipStream_t stream) {
    VQH_CHECK_ARG(n >= 0 && ze && zsoft && zhard && out, "vqh_vq_mix: bad argument");
    if (n == 0) return VQH_OK;
    hipLaunchKernelGGL(vq_mix_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, ze, zsoft, zhard, alpha, out, n);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
