// Fused multi-term loss, forward AND gradient w.r.t. the reconstruction / latent, one workgroup per
// sample (reference: VQVAE.loss_function, /root/reference/models/vq_vae.py:1097-1388 and helpers
// _mse_per_sample :903, _kabsch_rt_safe :943, _ss_label_smoothing_ce :920, _dihedral_cos_sin :347,
// _pairwise_pdm :971, _window_kabsch_loss :996, _frenet_regularizers :1040, _long_range_pdm :1070).
//
// Every normaliser of the reference is a count over the mask (global over the batch), so it is known
// before the per-sample work starts: vqh_loss_counts builds an int table from the mask alone, then
// each block computes its sample's numerators and -- because weights and normalisers are known -- the
// finished gradient d loss / d recons[b] in the same pass.  No host sync: the reference's
// `.any()` / boolean-index control flow (:1142-1147) becomes per-sample predicates.
// Gradient accumulation of the stencil terms is ordered in barrier-separated phases (one writer per
// address per phase), so results are bitwise reproducible without atomics.
// The 3x3 Kabsch rotation is solved per sample in fp64 with a one-sided Jacobi SVD.
#include "common.h"

namespace {

// ---- int table built from the mask ---------------------------------------------------------
enum { T_VALID = 0, T_PAIR = 1, T_TRI = 2, T_QUAD = 3, T_FIVE = 4, T_ANY3 = 5, T_PDM = 8, T_LR = 40 };

// ---- per-sample partial sums ---------------------------------------------------------------
enum {
    P_RAW = 0, P_ALN, P_BEST, P_RMSD_RAW, P_RMSD_BEST, P_CE, P_ACC, P_SSTV, P_BL, P_BA, P_DIR, P_DIH, P_TV2, P_TAU,
    P_PDM, P_LRPDM, P_WINK, P_COMMIT, P_KAPPA, P_COUNT
};

// ---- output metric slots (order of the reference's result dict) ------------------------------
enum {
    M_LOSS = 0, M_REC_XYZ, M_MSE_RAW, M_MSE_ALN, M_REC_SS, M_SS_ACC, M_VQ_LOSS, M_BL, M_BA, M_DIR, M_DIH, M_GEOM,
    M_SS_TV, M_USAGE_REG, M_TV2, M_PPL, M_DEAD, M_RMSD_RAW, M_RMSD_ALN, M_PDM, M_WINK, M_KAPPA, M_TAU, M_LRPDM, M_COUNT
};

struct LossCfg {
    int B, L, Ntok, D;
    int Ls;                     // rows per sample IN MEMORY (>= L): the step pads L to a length bucket, the loss sees [B, L]
    const float* Ldev;          // device scalar holding L (as a float) or null: one captured graph serves every L_max of a bucket
    int masked;                 // 0: the reference was called with mask=None (different normalisers)
    int use_vq;
    float rmsd_w, ss_w, bl_w, ba_w, dir_w, dih_w, tv_l, pdm_w, wk_w, kap_w, tau_w, lr_w;
    float alpha, ss_tv_l, label_smoothing, beta;
    int pdm_window, wk_size, wk_stride, lr_sep, lr_stride, lr_max;
    int n_lr_pairs, n_windows;  // enumerated on the host exactly like the reference's python loops
    int t_wk;                   // offset of the per-window selected-sample counts in the table
    int has_stats;              // set_data_stats(): geometry terms see x*sd + mu (reference to_real, :1218-1227)
    float sd[3], mu[3];
};

// Enumeration sizes of the reference's python loops for a batch of padded length L: long-range pairs (:1078-1082) and
// Kabsch windows (:1003).  Evaluated on the host for sizing and on the device when L comes from device memory.
__host__ __device__ inline void loss_enumerate(int L, int lr_sep, int lr_stride, int lr_max, int wk_size, int wk_stride,
                                               int* npairs_out, int* nwin_out) {
    int npairs = 0;
    if (L >= lr_sep + 1) {
        const int st = lr_stride > 1 ? lr_stride : 1;
        for (int off = 0; off < (lr_max > 1 ? lr_max : 1); ++off) {
            const int sep = lr_sep + off;
            if (L - sep > 0) npairs += (L - sep + st - 1) / st;
        }
    }
    int nwin = 0;
    if (L >= 3 && wk_size >= 3 && L - wk_size + 1 > 0) {
        const int st = wk_stride > 1 ? wk_stride : 1;
        nwin = (L - wk_size) / st + 1;
    }
    *npairs_out = npairs;
    *nwin_out = nwin;
}

// kernel-side view of the configuration: L (and what is enumerated from it) taken from device memory when given
__device__ __forceinline__ LossCfg loss_resolve(LossCfg c) {
    if (c.Ldev) {
        int L = (int)(*c.Ldev + 0.5f);
        L = L < 1 ? 1 : (L > c.Ls ? c.Ls : L);
        c.L = L;
        loss_enumerate(L, c.lr_sep, c.lr_stride, c.lr_max, c.wk_size, c.wk_stride, &c.n_lr_pairs, &c.n_windows);
    }
    return c;
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r{x, y, z}; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 ld3(const float* p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void add3(float* p, int i, V3 a) { p[3 * i] += a.x; p[3 * i + 1] += a.y; p[3 * i + 2] += a.z; }

constexpr float UEPS = 1e-8f;
// u = v / (|v| + eps)   (reference _unit, models/vq_vae.py:328)
__device__ __forceinline__ V3 unit(V3 v) { return (1.f / (norm(v) + UEPS)) * v; }
// gradient of unit(): dv = du/(n+eps) - v (v.du) / (n (n+eps)^2)   (the |v| subgradient at 0 is 0)
__device__ __forceinline__ V3 unit_bwd(V3 v, V3 du) {
    const float n = norm(v), ne = n + UEPS;
    V3 r = (1.f / ne) * du;
    if (n > 0.f) r = r - (dot(v, du) / (n * ne * ne)) * v;
    return r;
}
// d|v|/dv = v/|v| (0 at 0)
__device__ __forceinline__ V3 norm_grad(V3 v) {
    const float n = norm(v);
    return n > 0.f ? (1.f / n) * v : v3(0.f, 0.f, 0.f);
}

__device__ __forceinline__ void dihedral_fwd(V3 p0, V3 p1, V3 p2, V3 p3, float& co, float& si) {
    const V3 b1 = unit(p1 - p0), b2 = unit(p2 - p1), b3 = unit(p3 - p2);
    const V3 n1 = unit(cross(b1, b2)), n2 = unit(cross(b2, b3));
    const V3 m1 = cross(n1, unit(b2));
    co = fminf(1.f, fmaxf(-1.f, dot(n1, n2)));
    si = fminf(1.f, fmaxf(-1.f, dot(m1, n2)));
}
__device__ __forceinline__ void dihedral_bwd(V3 p0, V3 p1, V3 p2, V3 p3, float dco, float dsi, V3& g0, V3& g1, V3& g2,
                                             V3& g3) {
    const V3 v1 = p1 - p0, v2 = p2 - p1, v3_ = p3 - p2;
    const V3 b1 = unit(v1), b2 = unit(v2), b3 = unit(v3_);
    const V3 c1 = cross(b1, b2), c2 = cross(b2, b3);
    const V3 n1 = unit(c1), n2 = unit(c2);
    const V3 ub2 = unit(b2);
    const V3 m1 = cross(n1, ub2);
    const float cr = dot(n1, n2), sr = dot(m1, n2);
    if (cr < -1.f || cr > 1.f) dco = 0.f;
    if (sr < -1.f || sr > 1.f) dsi = 0.f;
    V3 dn1 = dco * n2;
    const V3 dn2 = dco * n1 + dsi * m1;
    const V3 dm1 = dsi * n2;
    dn1 = dn1 + cross(ub2, dm1);          // m1 = n1 x ub2
    const V3 dub2 = cross(dm1, n1);
    V3 db2 = unit_bwd(b2, dub2);
    const V3 dc1 = unit_bwd(c1, dn1), dc2 = unit_bwd(c2, dn2);
    const V3 db1 = cross(b2, dc1);        // c1 = b1 x b2
    db2 = db2 + cross(dc1, b1);
    db2 = db2 + cross(b3, dc2);           // c2 = b2 x b3
    const V3 db3 = cross(dc2, b2);
    const V3 dv1 = unit_bwd(v1, db1), dv2 = unit_bwd(v2, db2), dv3 = unit_bwd(v3_, db3);
    g0 = v3(-dv1.x, -dv1.y, -dv1.z);
    g1 = dv1 - dv2;
    g2 = dv2 - dv3;
    g3 = dv3;
}

// block-wide sum (256 threads); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- 3x3 SVD-based Kabsch in fp64: R = U diag(1,1,sign det(U Vh)) Vh of H (row-vector convention x@R) ----
__device__ void kabsch_rotation(const double Hin[9], double R[9]) {
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { A[i][j] = Hin[3 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    // one-sided Jacobi: rotate column pairs of A (and V) until the columns are orthogonal: A = U S, H = U S V^T
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
                if (ga == 0.0) continue;
                off += fabs(ga) / (sqrt(al * be) + 1e-300);
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double sg[3], U[3][3];
    for (int k = 0; k < 3; ++k) sg[k] = sqrt(A[0][k] * A[0][k] + A[1][k] * A[1][k] + A[2][k] * A[2][k]);
    // order: index of the smallest singular value gets the reflection (LAPACK sorts descending, flips last)
    int kmin = 0;
    for (int k = 1; k < 3; ++k) if (sg[k] < sg[kmin]) kmin = k;
    const double smax = fmax(sg[0], fmax(sg[1], sg[2]));
    const double tiny = smax * 1e-14;
    int nz = 0;
    for (int k = 0; k < 3; ++k) {
        if (sg[k] > tiny && sg[k] > 0.0) { for (int i = 0; i < 3; ++i) U[i][k] = A[i][k] / sg[k]; ++nz; }
        else { for (int i = 0; i < 3; ++i) U[i][k] = 0.0; }
    }
    if (nz == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0; }
    else if (nz < 3) {
        // complete an orthonormal basis for the null columns
        for (int k = 0; k < 3; ++k) {
            if (sg[k] > tiny && sg[k] > 0.0) continue;
            // find a unit vector orthogonal to the already set columns
            double best[3] = {0, 0, 0}; double bn = -1.0;
            for (int e = 0; e < 3; ++e) {
                double w[3] = {e == 0 ? 1.0 : 0.0, e == 1 ? 1.0 : 0.0, e == 2 ? 1.0 : 0.0};
                for (int c = 0; c < 3; ++c) {
                    if (c == k) continue;
                    const double nn = U[0][c] * U[0][c] + U[1][c] * U[1][c] + U[2][c] * U[2][c];
                    if (nn == 0.0) continue;
                    const double pr = w[0] * U[0][c] + w[1] * U[1][c] + w[2] * U[2][c];
                    for (int i = 0; i < 3; ++i) w[i] -= pr * U[i][c];
                }
                const double wn = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
                if (wn > bn) { bn = wn; for (int i = 0; i < 3; ++i) best[i] = w[i]; }
            }
            const double inv = 1.0 / sqrt(bn);
            for (int i = 0; i < 3; ++i) U[i][k] = best[i] * inv;
        }
    }
    // det(U V^T) = det(U) det(V)
    auto det3 = [](double M[3][3]) {
        return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
               M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    };
    const double dsign = (det3(U) * det3(V) >= 0.0) ? 1.0 : -1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += U[i][k] * ((k == kmin) ? dsign : 1.0) * V[j][k];
            R[3 * i + j] = s;
        }
}

// Cooperative Kabsch over points [l0, l0+n) of a (moving) and b (target) with weights m: all threads call.
// Outputs R (row-vector convention) and t in shared memory (fp32), returns ok flag.
__device__ bool block_kabsch(const float* a, const float* b, const float* m, int l0, int n, double* dred, float* Rt_s) {
    double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0}, sm = 0;
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const float w = m[l0 + l];
        sm += w;
        for (int c = 0; c < 3; ++c) { sa[c] += (double)w * a[3 * (l0 + l) + c]; sb[c] += (double)w * b[3 * (l0 + l) + c]; }
    }
    sm = block_sum_d(sm, dred);
    const double den = fmax(sm, 1.0);
    double amu[3], bmu[3];
    for (int c = 0; c < 3; ++c) { amu[c] = block_sum_d(sa[c], dred) / den; bmu[c] = block_sum_d(sb[c], dred) / den; }
    double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const double w = m[l0 + l];
        double ac[3], bc[3];
        for (int c = 0; c < 3; ++c) {
            // the reference centres in fp32 (x - mu); mirror that rounding, accumulate in fp64
            ac[c] = (double)((float)a[3 * (l0 + l) + c] - (float)amu[c]);
            bc[c] = (double)((float)b[3 * (l0 + l) + c] - (float)bmu[c]);
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Hm[3 * i + j] += w * ac[i] * bc[j];
    }
    for (int k = 0; k < 9; ++k) Hm[k] = block_sum_d(Hm[k], dred);
    if (threadIdx.x == 0) {
        double R[9];
        kabsch_rotation(Hm, R);
        bool ok = true;
        for (int k = 0; k < 9; ++k) { Rt_s[k] = (float)R[k]; ok = ok && isfinite(R[k]); }
        for (int j = 0; j < 3; ++j) {
            const double t = bmu[j] - (amu[0] * R[0 * 3 + j] + amu[1] * R[1 * 3 + j] + amu[2] * R[2 * 3 + j]);
            Rt_s[9 + j] = (float)t;
            ok = ok && isfinite(t);
        }
        Rt_s[12] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    return Rt_s[12] != 0.f;
}

// ----------------------------------------------------------------------------------------------
// counts from the mask: one block per sample, integer atomics (deterministic)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_counts_kernel(const unsigned char* __restrict__ mask, LossCfg c_in,
                                                          int* __restrict__ table) {
    const LossCfg c = loss_resolve(c_in);
    extern __shared__ unsigned char sm_[];
    unsigned char* m = sm_;
    const int b = blockIdx.x, L = c.L;
    for (int l = threadIdx.x; l < L; l += 256) m[l] = mask ? mask[(size_t)b * c.Ls + l] : 1;
    __syncthreads();
    int nv = 0, np = 0, nt = 0, nq = 0, n5 = 0;
    for (int l = threadIdx.x; l < L; l += 256) {
        nv += m[l];
        if (l + 1 < L) np += m[l] & m[l + 1];
        if (l + 2 < L) nt += m[l] & m[l + 1] & m[l + 2];
        if (l + 3 < L) nq += m[l] & m[l + 1] & m[l + 2] & m[l + 3];
        if (l + 4 < L) n5 += m[l] & m[l + 1] & m[l + 2] & m[l + 3] & m[l + 4];
    }
    __shared__ int red[5];
    if (threadIdx.x < 5) red[threadIdx.x] = 0;
    __syncthreads();
    atomicAdd(&red[0], nv); atomicAdd(&red[1], np); atomicAdd(&red[2], nt); atomicAdd(&red[3], nq); atomicAdd(&red[4], n5);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&table[T_VALID], red[0]); atomicAdd(&table[T_PAIR], red[1]); atomicAdd(&table[T_TRI], red[2]);
        atomicAdd(&table[T_QUAD], red[3]); atomicAdd(&table[T_FIVE], red[4]);
        if (red[0] >= 3) atomicAdd(&table[T_ANY3], 1);
    }
    // local PDM pair counts, d = 1 .. window-1
    for (int d = 1 + (threadIdx.x >> 6); d < c.pdm_window && d < 32; d += 4) {
        int s = 0;
        for (int l = threadIdx.x & 63; l + d < L; l += 64) s += m[l] & m[l + d];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&table[T_PDM + d - 1], s);
    }
    // long-range pairs, enumerated like the reference's double loop (:1078-1082)
    {
        int p = 0;
        for (int off = 0; off < max(1, c.lr_max); ++off)
            for (int i = 0; i < L; i += max(1, c.lr_stride)) {
                const int j = i + c.lr_sep + off;
                if (j >= L) break;
                if ((p & 255) == (int)threadIdx.x && (m[i] & m[j])) atomicAdd(&table[T_LR + p], 1);
                ++p;
            }
    }
    // window Kabsch: sample selected in window w iff it has >= 3 valid points there (:1009,:1029)
    for (int w = threadIdx.x; w < c.n_windows; w += 256) {
        const int s0 = w * max(1, c.wk_stride);
        int cnt = 0;
        for (int l = 0; l < c.wk_size; ++l) cnt += m[s0 + l];
        if (!c.masked || cnt >= 3) atomicAdd(&table[c.t_wk + w], 1);
    }
}

// ----------------------------------------------------------------------------------------------
// per-sample loss + gradient
// ----------------------------------------------------------------------------------------------
#define FOR_POS(n_) for (int base_ = 0; base_ < (n_); base_ += 256)

__global__ __launch_bounds__(256) void loss_sample_kernel(const float* __restrict__ recons,
                                                          const float* __restrict__ target,
                                                          const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ ze, const float* __restrict__ zq,
                                                          const int* __restrict__ table, LossCfg c_in,
                                                          float* __restrict__ d_recons, float* __restrict__ d_ze,
                                                          float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const LossCfg c = loss_resolve(c_in);
    const int L = c.L, b = blockIdx.x, tid = threadIdx.x;
    double* dred = reinterpret_cast<double*>(sm);   // [4] (kept first: 8-byte aligned for any L)
    float* rx = sm + 8;           // [3L] reconstructed xyz
    float* gx = rx + 3 * L;       // [3L] target xyz
    float* lg = gx + 3 * L;       // [3L] logits
    float* pr = lg + 3 * L;       // [3L] softmax probs
    float* g = pr + 3 * L;        // [3L] d loss / d rx
    float* mk = g + 3 * L;        // [L] mask as float
    float* dcs = mk + L;          // [2L] combined d/d(cos,sin) of the reconstructed dihedral track
    float* dhr = dcs + 2 * L;     // [2L] dihedral (cos,sin) of recon
    float* dhg = dhr + 2 * L;     // [2L] dihedral of target
    int* lab = reinterpret_cast<int*>(dhg + 2 * L);   // [L]
    float* red = reinterpret_cast<float*>(lab + L);   // [4]
    float* Rt = red + 4;          // [13] R(9) t(3) ok
    float* rrb = Rt + 16;         // [3L] real-space recon   } only distinct from rx/gx/g when data stats are set
    float* grb = rrb + 3 * L;     // [3L] real-space target  }
    float* gRb = grb + 3 * L;     // [3L] d loss / d real-space recon
    const float* rr = c.has_stats ? rrb : rx;
    const float* gr = c.has_stats ? grb : gx;
    float* gR = c.has_stats ? gRb : g;

    const float* rb = recons + (size_t)b * c.Ls * 6;
    const float* tb = target + (size_t)b * c.Ls * 6;
    for (int l = tid; l < L; l += 256) {
        for (int k = 0; k < 3; ++k) {
            rx[3 * l + k] = rb[6 * l + k];
            gx[3 * l + k] = tb[6 * l + k];
            lg[3 * l + k] = rb[6 * l + 3 + k];
            g[3 * l + k] = 0.f;
            if (c.has_stats) {
                rrb[3 * l + k] = rb[6 * l + k] * c.sd[k] + c.mu[k];
                grb[3 * l + k] = tb[6 * l + k] * c.sd[k] + c.mu[k];
                gRb[3 * l + k] = 0.f;
            }
        }
        mk[l] = mask ? (float)mask[(size_t)b * c.Ls + l] : 1.f;
        // label = argmax of the one-hot target (first maximum)
        const float t0 = tb[6 * l + 3], t1 = tb[6 * l + 4], t2 = tb[6 * l + 5];
        int la = 0; float tm = t0;
        if (t1 > tm) { tm = t1; la = 1; }
        if (t2 > tm) { la = 2; }
        lab[l] = la;
        dcs[2 * l] = 0.f; dcs[2 * l + 1] = 0.f;
    }
    __syncthreads();

    const float invB = 1.f / (float)c.B;
    float nvalid_b = 0.f;
    for (int l = tid; l < L; l += 256) nvalid_b += mk[l];
    nvalid_b = block_sum(nvalid_b, red);
    const float nb = fmaxf(nvalid_b, 1.f);
    auto tden = [&](int slot) { return 1.f / fmaxf((float)table[slot], 1.f); };

    // ---------------- xyz MSE: raw and Kabsch aligned (:1130-1172) ----------------
    float raw = 0.f;
    for (int l = tid; l < L; l += 256) {
        const V3 d = ld3(rx, l) - ld3(gx, l);
        raw += mk[l] * dot(d, d);
    }
    raw = block_sum(raw, red) / nb;
    bool can = (L >= 3);
    if (can && c.masked) can = table[T_ANY3] > 0;
    float aln = raw, best = raw;
    bool use_aln = false;
    if (can) {
        const bool ok = block_kabsch(rx, gx, mk, 0, L, dred, Rt);
        float s = 0.f;
        for (int l = tid; l < L; l += 256) {
            const V3 x = ld3(rx, l);
            const V3 y = v3(x.x * Rt[0] + x.y * Rt[3] + x.z * Rt[6] + Rt[9], x.x * Rt[1] + x.y * Rt[4] + x.z * Rt[7] + Rt[10],
                            x.x * Rt[2] + x.y * Rt[5] + x.z * Rt[8] + Rt[11]);
            const V3 d = y - ld3(gx, l);
            s += mk[l] * dot(d, d);
        }
        aln = block_sum(s, red) / nb;
        const bool sel = ok && (!c.masked || nvalid_b >= 3.f);
        if (sel) { best = fminf(raw, aln); use_aln = aln < raw; }
    }
    {
        const float w_raw = c.rmsd_w * invB * ((1.f - c.alpha) + (use_aln ? 0.f : c.alpha)) * 2.f / nb;
        const float w_aln = use_aln ? c.rmsd_w * invB * c.alpha * 2.f / nb : 0.f;
        for (int l = tid; l < L; l += 256) {
            const V3 x = ld3(rx, l), y0 = ld3(gx, l);
            V3 gr = (w_raw * mk[l]) * (x - y0);
            if (use_aln) {
                const V3 y = v3(x.x * Rt[0] + x.y * Rt[3] + x.z * Rt[6] + Rt[9], x.x * Rt[1] + x.y * Rt[4] + x.z * Rt[7] + Rt[10],
                                x.x * Rt[2] + x.y * Rt[5] + x.z * Rt[8] + Rt[11]);
                const V3 d = y - y0;
                // (d . R^T): row-vector convention y = x R  =>  dx = d R^T
                const V3 dx = v3(d.x * Rt[0] + d.y * Rt[1] + d.z * Rt[2], d.x * Rt[3] + d.y * Rt[4] + d.z * Rt[5],
                                 d.x * Rt[6] + d.y * Rt[7] + d.z * Rt[8]);
                gr = gr + (w_aln * mk[l]) * dx;
            }
            add3(g, l, gr);
        }
    }
    __syncthreads();

    // ---------------- secondary structure: CE / label smoothing, accuracy, TV (:1184-1215) ----------------
    float ce = 0.f, acc = 0.f, sstv = 0.f;
    float* dlg = d_recons + (size_t)b * c.Ls * 6;   // logits grads are written straight to the output
    const float inv_valid = tden(T_VALID), inv_pair = tden(T_PAIR), inv_tri = tden(T_TRI);
    for (int l = tid; l < L; l += 256) {
        const float a0 = lg[3 * l], a1 = lg[3 * l + 1], a2 = lg[3 * l + 2];
        const float mx = fmaxf(a0, fmaxf(a1, a2));
        const float e0 = expf(a0 - mx), e1 = expf(a1 - mx), e2 = expf(a2 - mx);
        const float se = e0 + e1 + e2, lse = mx + logf(se);
        pr[3 * l] = e0 / se; pr[3 * l + 1] = e1 / se; pr[3 * l + 2] = e2 / se;
        const int la = lab[l];
        float per;
        if (c.label_smoothing > 0.f) {
            const float eps = c.label_smoothing, lo = eps / 2.f, hi = 1.f - eps;
            per = 0.f;
            for (int k = 0; k < 3; ++k) {
                const float t = (k == la) ? hi : lo;
                per += t * (logf(t) - (lg[3 * l + k] - lse));
            }
        } else {
            per = -(lg[3 * l + la] - lse);
        }
        ce += mk[l] * per;
        int am = 0; float bm = a0;
        if (a1 > bm) { bm = a1; am = 1; }
        if (a2 > bm) { am = 2; }
        acc += (am == la) ? mk[l] : 0.f;
    }
    __syncthreads();
    const bool tv_on = (c.ss_tv_l > 0.f) && (L >= 2);
    for (int l = tid; l < L; l += 256) {
        // CE gradient
        const int la = lab[l];
        float dl[3];
        const float wce = c.ss_w * inv_valid * mk[l];
        for (int k = 0; k < 3; ++k) {
            float t;
            if (c.label_smoothing > 0.f) t = (k == la) ? 1.f - c.label_smoothing : c.label_smoothing / 2.f;
            else t = (k == la) ? 1.f : 0.f;
            dl[k] = wce * (pr[3 * l + k] - t);
        }
        if (tv_on) {
            float dp[3] = {0.f, 0.f, 0.f};
            const float wtv = c.ss_tv_l * inv_pair;
            if (l + 1 < L) {
                const float pm = mk[l] * mk[l + 1];
                for (int k = 0; k < 3; ++k) {
                    const float d = pr[3 * (l + 1) + k] - pr[3 * l + k];
                    sstv += pm * fabsf(d);
                    dp[k] -= wtv * pm * ((d > 0.f) - (d < 0.f));
                }
            }
            if (l >= 1) {
                const float pm = mk[l] * mk[l - 1];
                for (int k = 0; k < 3; ++k) {
                    const float d = pr[3 * l + k] - pr[3 * (l - 1) + k];
                    dp[k] += wtv * pm * ((d > 0.f) - (d < 0.f));
                }
            }
            const float s = dp[0] * pr[3 * l] + dp[1] * pr[3 * l + 1] + dp[2] * pr[3 * l + 2];
            for (int k = 0; k < 3; ++k) dl[k] += pr[3 * l + k] * (dp[k] - s);
        }
        for (int k = 0; k < 3; ++k) dlg[6 * l + 3 + k] = dl[k];
    }
    ce = block_sum(ce, red); acc = block_sum(acc, red); sstv = block_sum(sstv, red);

    // ---------------- pair terms: bond length (:1230) and direction (:1264) ----------------
    float bl = 0.f, dr = 0.f;
    FOR_POS(L - 1) {
        const int l = base_ + tid;
        const bool on = l < L - 1;
        V3 ga = v3(0, 0, 0);
        if (on) {
            const float pm = mk[l] * mk[l + 1];
            const V3 vr = ld3(rr, l + 1) - ld3(rr, l), vg = ld3(gr, l + 1) - ld3(gr, l);
            const float e = norm(vr) - norm(vg);
            bl += pm * e * e;
            const V3 ur = unit(vr), ug = unit(vg);
            dr += pm * (1.f - dot(ur, ug));
            ga = (2.f * c.bl_w * inv_pair * pm * e) * norm_grad(vr);
            if (c.dir_w != 0.f) ga = ga + unit_bwd(vr, (-c.dir_w * inv_pair * pm) * ug);
        }
        if (on) add3(gR, l + 1, ga);
        __syncthreads();
        if (on) add3(gR, l, v3(-ga.x, -ga.y, -ga.z));
        __syncthreads();
    }
    bl = block_sum(bl, red); dr = block_sum(dr, red);

    // ---------------- triple terms: bond angle (:1244), xyz TV2 (:1312) / Frenet kappa (:1044-1052) ----------------
    float ba = 0.f, tv2 = 0.f, kap = 0.f;
    FOR_POS(L - 2) {
        const int l = base_ + tid;
        const bool on = l < L - 2;
        V3 g0 = v3(0, 0, 0), g1 = g0, g2 = g0, t0 = g0, t1 = g0, t2 = g0;
        if (on) {
            const float tm = mk[l] * mk[l + 1] * mk[l + 2];
            const V3 p0 = ld3(rr, l), p1 = ld3(rr, l + 1), p2 = ld3(rr, l + 2);
            const V3 v1 = p1 - p0, v2 = p2 - p1;
            const V3 u1 = unit(v1), u2 = unit(v2);
            const V3 q0 = ld3(gr, l), q1 = ld3(gr, l + 1), q2 = ld3(gr, l + 2);
            const float cg = dot(unit(q1 - q0), unit(q2 - q1));
            const float e = dot(u1, u2) - cg;
            ba += tm * e * e;
            const float dc = 2.f * c.ba_w * inv_tri * tm * e;
            const V3 dv1 = unit_bwd(v1, dc * u2), dv2 = unit_bwd(v2, dc * u1);
            const V3 d2r = v2 - v1;                                   // Frenet kappa lives in real space
            kap += tm * dot(d2r, d2r);
            const float wk = 2.f * c.kap_w * inv_tri * tm;
            g0 = v3(-dv1.x, -dv1.y, -dv1.z) + wk * d2r;
            g1 = (dv1 - dv2) - (2.f * wk) * d2r;
            g2 = dv2 + wk * d2r;
            const V3 d2n = (ld3(rx, l + 2) - ld3(rx, l + 1)) - (ld3(rx, l + 1) - ld3(rx, l));   // xyz TV2: normalised space
            tv2 += tm * dot(d2n, d2n);
            const float wt = 2.f * c.tv_l * inv_tri * tm;
            t0 = wt * d2n; t1 = (-2.f * wt) * d2n; t2 = wt * d2n;
        }
        if (on) { add3(gR, l, g0); add3(g, l, t0); }
        __syncthreads();
        if (on) { add3(gR, l + 1, g1); add3(g, l + 1, t1); }
        __syncthreads();
        if (on) { add3(gR, l + 2, g2); add3(g, l + 2, t2); }
        __syncthreads();
    }
    kap = block_sum(kap, red);
    ba = block_sum(ba, red); tv2 = block_sum(tv2, red);

    // ---------------- dihedral (:1278) and Frenet tau (:1056-1065) ----------------
    float dih = 0.f, tau = 0.f;
    const int nd = L - 3;
    if (nd > 0) {
        for (int i = tid; i < nd; i += 256) {
            float co, si;
            dihedral_fwd(ld3(rr, i), ld3(rr, i + 1), ld3(rr, i + 2), ld3(rr, i + 3), co, si);
            dhr[2 * i] = co; dhr[2 * i + 1] = si;
            dihedral_fwd(ld3(gr, i), ld3(gr, i + 1), ld3(gr, i + 2), ld3(gr, i + 3), co, si);
            dhg[2 * i] = co; dhg[2 * i + 1] = si;
        }
        __syncthreads();
        const float w_dih = c.masked ? c.dih_w * tden(T_QUAD) : c.dih_w / (2.f * (float)c.B * (float)nd);
        const float w_tau = c.tau_w * tden(T_FIVE);
        for (int i = tid; i < nd; i += 256) {
            const float qm = mk[i] * mk[i + 1] * mk[i + 2] * mk[i + 3];
            const float ec = dhr[2 * i] - dhg[2 * i], es = dhr[2 * i + 1] - dhg[2 * i + 1];
            dih += qm * (ec * ec + es * es);
            float dco = 2.f * w_dih * qm * ec, dsi = 2.f * w_dih * qm * es;
            if (i + 1 < nd) {   // tau term i: (dih[i+1]-dih[i])^2, five-point mask
                const float m5 = qm * mk[i + 4];
                const float a = dhr[2 * (i + 1)] - dhr[2 * i], s = dhr[2 * (i + 1) + 1] - dhr[2 * i + 1];
                tau += m5 * (a * a + s * s);
                dco -= 2.f * w_tau * m5 * a; dsi -= 2.f * w_tau * m5 * s;
            }
            if (i >= 1) {       // tau term i-1
                const float m5 = mk[i - 1] * mk[i] * mk[i + 1] * mk[i + 2] * mk[i + 3];
                const float a = dhr[2 * i] - dhr[2 * (i - 1)], s = dhr[2 * i + 1] - dhr[2 * (i - 1) + 1];
                dco += 2.f * w_tau * m5 * a; dsi += 2.f * w_tau * m5 * s;
            }
            dcs[2 * i] = dco; dcs[2 * i + 1] = dsi;
        }
        __syncthreads();
        if (c.dih_w != 0.f || c.tau_w != 0.f) {
            FOR_POS(nd) {
                const int i = base_ + tid;
                const bool on = i < nd;
                V3 g0 = v3(0, 0, 0), g1 = g0, g2 = g0, g3 = g0;
                if (on) dihedral_bwd(ld3(rr, i), ld3(rr, i + 1), ld3(rr, i + 2), ld3(rr, i + 3), dcs[2 * i], dcs[2 * i + 1], g0, g1, g2, g3);
                if (on) add3(gR, i, g0);
                __syncthreads();
                if (on) add3(gR, i + 1, g1);
                __syncthreads();
                if (on) add3(gR, i + 2, g2);
                __syncthreads();
                if (on) add3(gR, i + 3, g3);
                __syncthreads();
            }
        }
    }
    dih = block_sum(dih, red); tau = block_sum(tau, red);

    // ---------------- local pairwise-distance matrix (:971-994) ----------------
    float pdm = 0.f;
    if (c.pdm_w > 0.f && L >= 2 && c.pdm_window > 1) {
        const float wn = 1.f / fmaxf(1.f, (float)(c.pdm_window - 1));
        for (int d = 1; d < c.pdm_window; ++d) {
            const int np = L - d;
            const float iden = (d - 1 < 32) ? tden(T_PDM + d - 1) : 1.f;
            float s = 0.f;
            FOR_POS(max(np, 0)) {
                const int i = base_ + tid;
                const bool on = i < np;
                V3 ga = v3(0, 0, 0);
                if (on) {
                    const float pm = mk[i] * mk[i + d];
                    const V3 va = ld3(rr, i) - ld3(rr, i + d), vb = ld3(gr, i) - ld3(gr, i + d);
                    const float e = norm(va) - norm(vb);
                    s += pm * e * e;
                    ga = (2.f * c.pdm_w * wn * iden * pm * e) * norm_grad(va);
                }
                if (on) add3(gR, i, ga);
                __syncthreads();
                if (on) add3(gR, i + d, v3(-ga.x, -ga.y, -ga.z));
                __syncthreads();
            }
            pdm += wn * iden * block_sum(s, red);
        }
    }

    // ---------------- long-range PDM (:1070-1095) ----------------
    float lrp = 0.f;
    if (c.lr_w > 0.f && L >= c.lr_sep + 1 && c.n_lr_pairs > 0) {
        const float wn = 1.f / (float)c.n_lr_pairs;
        int p0 = 0;
        const int st = max(1, c.lr_stride);
        for (int off = 0; off < max(1, c.lr_max); ++off) {
            // pairs of this offset: i = 0, st, 2st, ... while i + sep + off < L
            const int sep = c.lr_sep + off;
            const int cntp = (L - sep > 0) ? (L - sep + st - 1) / st : 0;
            float s = 0.f;
            FOR_POS(cntp) {
                const int k = base_ + tid;
                const bool on = k < cntp;
                V3 ga = v3(0, 0, 0);
                int i = 0, j = 0;
                if (on) {
                    i = k * st; j = i + sep;
                    const float pm = mk[i] * mk[j];
                    const float iden = tden(T_LR + p0 + k);
                    const V3 va = ld3(rr, j) - ld3(rr, i), vb = ld3(gr, j) - ld3(gr, i);
                    const float e = norm(va) - norm(vb);
                    s += iden * pm * e * e;
                    ga = (2.f * c.lr_w * wn * iden * pm * e) * norm_grad(va);
                }
                if (on) add3(gR, j, ga);
                __syncthreads();
                if (on) add3(gR, i, v3(-ga.x, -ga.y, -ga.z));
                __syncthreads();
            }
            lrp += wn * block_sum(s, red);
            p0 += cntp;
        }
    }

    // ---------------- window Kabsch (:996-1038) ----------------
    float wink = 0.f;
    if (c.wk_w > 0.f && L >= 3 && c.wk_size >= 3 && c.n_windows > 0) {
        int nwin = 0;
        for (int w = 0; w < c.n_windows; ++w) nwin += table[c.t_wk + w] > 0;
        if (nwin > 0) {
            const float wn = 1.f / (float)nwin;
            for (int w = 0; w < c.n_windows; ++w) {
                const int nsel = table[c.t_wk + w];
                if (nsel == 0) continue;
                const int s0 = w * max(1, c.wk_stride);
                float cntv = 0.f;
                for (int l = tid; l < c.wk_size; l += 256) cntv += mk[s0 + l];
                cntv = block_sum(cntv, red);
                const bool selected = !c.masked || cntv >= 3.f;
                if (!selected) continue;          // uniform across the block
                const bool ok = block_kabsch(rr, gr, mk, s0, c.wk_size, dred, Rt);
                if (!ok) continue;
                const float den = c.masked ? fmaxf(cntv, 1.f) : (float)(c.wk_size * 3);
                const float coef = c.wk_w * wn / ((float)nsel * den);
                float s = 0.f;
                for (int l = tid; l < c.wk_size; l += 256) {
                    const int p = s0 + l;
                    const V3 x = ld3(rr, p);
                    const V3 y = v3(x.x * Rt[0] + x.y * Rt[3] + x.z * Rt[6] + Rt[9], x.x * Rt[1] + x.y * Rt[4] + x.z * Rt[7] + Rt[10],
                                    x.x * Rt[2] + x.y * Rt[5] + x.z * Rt[8] + Rt[11]);
                    const V3 d = y - ld3(gr, p);
                    s += mk[p] * dot(d, d);
                    const V3 dx = v3(d.x * Rt[0] + d.y * Rt[1] + d.z * Rt[2], d.x * Rt[3] + d.y * Rt[4] + d.z * Rt[5],
                                     d.x * Rt[6] + d.y * Rt[7] + d.z * Rt[8]);
                    add3(gR, p, (2.f * coef * mk[p]) * dx);
                }
                wink += wn / ((float)nsel * den) * block_sum(s, red);
            }
        }
    }
    __syncthreads();

    // ---------------- commitment term (:1292-1296) ----------------
    float commit = 0.f;
    if (c.use_vq && ze) {
        const int n = c.Ntok * c.D;
        const float wq = 2.f * c.beta / ((float)c.B * (float)n);
        const float* zeb = ze + (size_t)b * n;
        const float* zqb = zq + (size_t)b * n;
        float* dzb = d_ze + (size_t)b * n;
        for (int i = tid; i < n; i += 256) {
            const float d = zeb[i] - zqb[i];
            commit += d * d;
            dzb[i] = wq * d;
        }
        commit = block_sum(commit, red);
    }

    // ---------------- outputs ----------------
    float* dxb = d_recons + (size_t)b * c.Ls * 6;
    for (int l = tid; l < L; l += 256)
        for (int k = 0; k < 3; ++k) dxb[6 * l + k] = c.has_stats ? g[3 * l + k] + gRb[3 * l + k] * c.sd[k] : g[3 * l + k];
    for (int i = 6 * L + tid; i < 6 * c.Ls; i += 256) dxb[i] = 0.f;      // bucket padding: no gradient
    if (tid == 0) {
        float* p = part + (size_t)b * P_COUNT;
        p[P_RAW] = raw; p[P_ALN] = aln; p[P_BEST] = best;
        p[P_RMSD_RAW] = sqrtf(fmaxf(raw, 1e-12f)); p[P_RMSD_BEST] = sqrtf(fmaxf(best, 1e-12f));
        p[P_CE] = ce; p[P_ACC] = acc; p[P_SSTV] = sstv; p[P_BL] = bl; p[P_BA] = ba; p[P_DIR] = dr; p[P_DIH] = dih;
        p[P_TV2] = tv2; p[P_TAU] = tau; p[P_KAPPA] = kap; p[P_PDM] = pdm; p[P_LRPDM] = lrp; p[P_WINK] = wink; p[P_COMMIT] = commit;
    }
}

// sum the per-sample partials in sample order and assemble the metric vector
__global__ void loss_finish_kernel(const float* __restrict__ part, const int* __restrict__ table, LossCfg c_in,
                                   const float* __restrict__ vq_stats, float* __restrict__ metrics) {
    const LossCfg c = loss_resolve(c_in);
    __shared__ double acc[P_COUNT];
    if (threadIdx.x < P_COUNT) {
        double s = 0.0;
        for (int b = 0; b < c.B; ++b) s += (double)part[(size_t)b * P_COUNT + threadIdx.x];
        acc[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    auto cnt = [&](int slot) { return fmax((double)table[slot], 1.0); };
    const double B = c.B;
    const double l_raw = acc[P_RAW] / B, l_aln = acc[P_BEST] / B;
    const double l_xyz = c.alpha * l_aln + (1.0 - c.alpha) * l_raw;
    const double l_ss = acc[P_CE] / cnt(T_VALID);
    const double ss_tv = (c.ss_tv_l > 0.f && c.L >= 2) ? acc[P_SSTV] / cnt(T_PAIR) : 0.0;
    const double bl = c.L >= 2 ? acc[P_BL] / cnt(T_PAIR) : 0.0;
    const double ba = c.L >= 3 ? acc[P_BA] / cnt(T_TRI) : 0.0;
    const double dr = c.L >= 2 ? acc[P_DIR] / cnt(T_PAIR) : 0.0;
    double dih = 0.0;
    if (c.L >= 4) dih = c.masked ? acc[P_DIH] / cnt(T_QUAD) : acc[P_DIH] / (2.0 * B * (c.L - 3));
    const double geom = c.bl_w * bl + c.ba_w * ba + c.dir_w * dr + c.dih_w * dih;
    const double xyz_tv = (c.tv_l > 0.f && c.L >= 3) ? acc[P_TV2] / cnt(T_TRI) : 0.0;
    const double kappa = (c.kap_w > 0.f && c.L >= 3) ? acc[P_KAPPA] / cnt(T_TRI) : 0.0;
    const double tau = (c.tau_w > 0.f && c.L >= 5) ? acc[P_TAU] / cnt(T_FIVE) : 0.0;
    const double vq = c.use_vq ? c.beta * acc[P_COMMIT] / (B * c.Ntok * c.D) : 0.0;
    const double pdm = acc[P_PDM], lrp = acc[P_LRPDM], wink = acc[P_WINK];
    const double total = c.rmsd_w * l_xyz + c.ss_w * l_ss + vq + geom + c.ss_tv_l * ss_tv + c.tv_l * xyz_tv + c.pdm_w * pdm +
                         c.wk_w * wink + c.kap_w * kappa + c.tau_w * tau + c.lr_w * lrp;
    metrics[M_LOSS] = (float)total;
    metrics[M_REC_XYZ] = (float)l_xyz;
    metrics[M_MSE_RAW] = (float)l_raw;
    metrics[M_MSE_ALN] = (float)(acc[P_ALN] / B);
    metrics[M_REC_SS] = (float)l_ss;
    metrics[M_SS_ACC] = (float)(acc[P_ACC] / cnt(T_VALID));
    metrics[M_VQ_LOSS] = (float)vq;
    metrics[M_BL] = (float)bl; metrics[M_BA] = (float)ba; metrics[M_DIR] = (float)dr; metrics[M_DIH] = (float)dih;
    metrics[M_GEOM] = (float)geom;
    metrics[M_SS_TV] = (float)ss_tv;
    metrics[M_USAGE_REG] = 0.f;
    metrics[M_TV2] = (float)xyz_tv;
    metrics[M_PPL] = vq_stats ? vq_stats[0] : 0.f;
    metrics[M_DEAD] = vq_stats ? vq_stats[1] : 0.f;
    metrics[M_RMSD_RAW] = (float)(acc[P_RMSD_RAW] / B);
    metrics[M_RMSD_ALN] = (float)(acc[P_RMSD_BEST] / B);
    metrics[M_PDM] = (float)pdm; metrics[M_WINK] = (float)wink; metrics[M_KAPPA] = (float)kappa;
    metrics[M_TAU] = (float)tau; metrics[M_LRPDM] = (float)lrp;
}

}  // namespace

// weights: 16 floats {rmsd_w, ss_w, bl_w, ba_w, dir_w, dih_w, xyz_tv_lambda, pdm_w, win_kabsch_w, kappa_w, tau_w,
//                     lr_pdm_w, xyz_align_alpha, ss_tv_lambda, label_smoothing, beta}
// data_stats (HOST, may be NULL): {std_x, std_y, std_z, mean_x, mean_y, mean_z} of set_data_stats()
// iparams: 6 ints {pdm_window, win_kabsch_size, win_kabsch_stride, lr_min_sep, lr_stride, lr_max_offsets}
// metrics: 24 floats (order = reference dict + the 5 optional keys);  workspace: ints table + per-sample partials.
extern "C" int vqh_loss_fwd_bwd(const float* recons, const float* target, const unsigned char* mask, int masked,
                                const float* ze, const float* zq, const float* vq_stats, int B, int L, int L_stride, const float* L_dev,
                                int Ntok, int D, int use_vq, const float* weights, const int* iparams, const float* data_stats,
                                float* d_recons, float* d_ze, float* metrics, float* workspace,
                                long long workspace_floats, hipStream_t stream) {
    VQH_CHECK_ARG(B > 0 && L > 0 && L_stride >= L, "vqh_loss_fwd_bwd: bad shape");
    VQH_CHECK_ARG(recons && target && weights && iparams && d_recons && metrics && workspace, "vqh_loss_fwd_bwd: null pointer");
    VQH_CHECK_ARG(!use_vq || (ze && zq && d_ze), "vqh_loss_fwd_bwd: VQ tensors missing");
    LossCfg c{};
    c.B = B; c.L = L_dev ? L_stride : L; c.Ls = L_stride; c.Ldev = L_dev; c.Ntok = Ntok; c.D = D; c.masked = masked ? 1 : 0; c.use_vq = use_vq ? 1 : 0;
    c.rmsd_w = weights[0]; c.ss_w = weights[1]; c.bl_w = weights[2]; c.ba_w = weights[3]; c.dir_w = weights[4];
    c.dih_w = weights[5]; c.tv_l = weights[6]; c.pdm_w = weights[7]; c.wk_w = weights[8]; c.kap_w = weights[9];
    c.tau_w = weights[10]; c.lr_w = weights[11]; c.alpha = weights[12]; c.ss_tv_l = weights[13];
    c.label_smoothing = weights[14]; c.beta = weights[15];
    c.pdm_window = iparams[0]; c.wk_size = iparams[1]; c.wk_stride = iparams[2]; c.lr_sep = iparams[3];
    c.lr_stride = iparams[4]; c.lr_max = iparams[5];
    c.has_stats = data_stats ? 1 : 0;
    for (int k = 0; k < 3; ++k) { c.sd[k] = data_stats ? data_stats[k] : 1.f; c.mu[k] = data_stats ? data_stats[3 + k] : 0.f; }
    VQH_CHECK_ARG(c.pdm_window <= 33, "vqh_loss_fwd_bwd: pdm_window > 33 unsupported");
    // enumerate long-range pairs and windows exactly like the reference loops; with L on the device the tables are sized
    // for the longest batch the buffers can hold (L_stride) and the kernels enumerate for the actual L
    int npairs = 0, nwin = 0;
    loss_enumerate(L_dev ? L_stride : L, c.lr_sep, c.lr_stride, c.lr_max, c.wk_size, c.wk_stride, &npairs, &nwin);
    c.n_lr_pairs = npairs;
    c.n_windows = nwin;
    c.t_wk = T_LR + npairs;
    const long long table_ints = c.t_wk + nwin + 8;
    const long long need = table_ints + (long long)B * P_COUNT;
    VQH_CHECK_ARG(need <= workspace_floats, "vqh_loss_fwd_bwd: workspace too small");
    int* table = reinterpret_cast<int*>(workspace);
    float* part = workspace + table_ints;
    hipError_t e = hipMemsetAsync(table, 0, sizeof(int) * table_ints, stream);
    if (e != hipSuccess) { vqh_set_error(hipGetErrorString(e)); return VQH_ERR_LAUNCH; }
    const int Lmax = c.L;        // == L, or L_stride when L comes from the device
    hipLaunchKernelGGL(loss_counts_kernel, dim3(B), dim3(256), (size_t)Lmax, stream, mask, c, table);
    const size_t smem = sizeof(float) * (size_t)(8 + 23 * Lmax + 4 + 16 + 9 * Lmax + 4);
    VQH_CHECK_ARG(smem <= 160 * 1024, "vqh_loss_fwd_bwd: sequence too long for the LDS-resident loss kernel");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_sample_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) { vqh_set_error(hipGetErrorString(ea)); return VQH_ERR_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(loss_sample_kernel, dim3(B), dim3(256), smem, stream, recons, target, mask, ze, zq, table, c,
                       d_recons, d_ze, part);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, stream, part, table, c, vq_stats, metrics);
    VQH_LAUNCH_CHECK();
    return VQH_OK;
}
