/*
 * vqvae_hip.h -- C ABI of libvqvae_hip.so: the MI355X (gfx950) kernels of the VQ-VAE training step.
 *
 * The reference (jluuser/PyTorch-VAE) has no FFI of its own: its hot path is Python calling torch ops.
 * Each entry point below replaces the torch call(s) cited next to it (paths relative to the reference
 * root); INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted
 *   - row-major fp32 tensors with explicit leading dimensions (ld*, in elements)
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no hidden synchronisation, no
 *     allocation, no ownership transfer: the caller owns every buffer including `workspace`
 *   - returns 0 on success, <0 on error; vqh_last_error() returns a thread-local message
 *   - safe to capture into a hipGraph (all step-varying scalars are read from device memory)
 *   - dropout masks are counter-hash functions of (seed, step, site, element): rng_state points to
 *     two device uint64 {seed, step}; backward regenerates the forward mask from the same site id;
 *     drop probabilities are quantised to multiples of 2^-16 (keep scale 1/(1 - quantised p))
 */
#ifndef VQVAE_HIP_H
#define VQVAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vqh_stream_t; /* hipStream_t */

const char* vqh_last_error(void);
int vqh_abi_version(void);

/* rng_state[1] += 1  (once per training step) */
int vqh_rng_advance(unsigned long long* rng_state, vqh_stream_t stream);
int vqh_memset(void* ptr, int value, long long bytes, vqh_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM:  C[M,N] = epilogue( opA(A)[M,K] . opB(B)[K,N] ) (+ beta*C)
 *   a_kcontig=1: A stored [M,K] (lda>=K)   a_kcontig=0: A stored [K,M] (lda>=M)
 *   b_kcontig=1: B stored [N,K] (ldb>=K)   b_kcontig=0: B stored [K,N] (ldb>=N)
 * replaces: every nn.Linear / MultiheadAttention projection forward (models/vq_vae.py:455-533),
 *           and their autograd dgrad / wgrad products.
 * epilogue modes (bias may be NULL):
 *   0 LINEAR       v = acc + bias
 *   1 RELU_DROP    v = dropout(relu(acc + bias))                  TransformerEncoderLayer._ff_block
 *   2 GELU         aux_out = acc + bias ; v = gelu_erf(aux_out)   tokenizer ffn / fuse_mlp (:302-306, :497-502)
 *   3 DROP_RESID   v = aux_in + dropout(acc + bias)               residual adds (x = x + dropout(...))
 *   4 SIGMOID      v = sigmoid(acc + bias)                        latent_sigmoid (:740-742)
 *   5 MUL_POSMASK  v = acc * (aux_in > 0 ? 1/(1-p) : 0)           backward of mode 1 through its saved output
 *   6 MUL_GELUGRAD v = acc * gelu'(aux_in)                        backward of mode 2
 *   7 MUL_SIGGRAD  v = acc * aux_in * (1 - aux_in)
 * workspace (may be NULL): split-K slabs; used automatically when the output has < 256 tiles.
 * ------------------------------------------------------------------------------------------- */
int vqh_gemm(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             float* C, int ldc, const float* bias, int mode, const float* aux_in, float* aux_out, int ldaux,
             float beta, const unsigned long long* rng_state, unsigned drop_site, float drop_p, float* workspace,
             long long workspace_floats, vqh_stream_t stream);

/* dW[n_out,k_in] = dY[rows,n_out]^T . X[rows,k_in]  and  db[n_out] = column sums of dY, one launch (+ split-K reduce):
 * the autograd weight/bias gradient of every nn.Linear; db may be NULL */
int vqh_gemm_wgrad(int rows, int n_out, int k_in, const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw,
                   float* db, float beta, float* workspace, long long workspace_floats, vqh_stream_t stream);

/* All weight-gradient products of ONE transformer layer in one launch (each: dW = dY^T X, db = column sums of dY, beta 0):
 * autograd's per-Linear weight-gradient GEMMs (models/vq_vae.py:455-533, one per nn.Linear / MHA projection) batched with a
 * common K-chunk so that every workgroup runs a long K range and a tile needs few split-K slabs.  Products that do not tile
 * evenly (n_out % 256, k_in % 128, rows % 32) are executed one by one exactly as vqh_gemm_wgrad would.  The operands must
 * stay untouched until the call; workspace_floats >= sum over products of splits * (n_out * k_in + n_out). */
typedef struct vqh_wgrad_t {
    int rows, n_out, k_in;
    const float* dY; int lddy;
    const float* X; int ldx;
    float* dW; int lddw;
    float* db;                  /* may be null */
} vqh_wgrad_t;
int vqh_gemm_wgrad_group(int n, const vqh_wgrad_t* problems, float* workspace, long long workspace_floats, vqh_stream_t stream);

/* tuning knobs of vqh_gemm (returns the previous value): bit0 = XCD-aware tile order (default on);
 * bits 1,2 are timing-only diagnostics that produce WRONG results (skip stores / skip loads): honoured only by a library
 * built with -DVQH_DIAG (labs), masked off by the product build;
 * bit 4 = no epilogue-operand prefetch; bit 5 = no fragment pipelining across the K-step barrier; bit 6 = no skinny-shape
 * streaming kernels (everything on the MFMA tile kernel); bit 7 = no 256x128 LDS-DMA kernel (everything that tiles
 * evenly stays on the 128x128 register-staged kernel); bit 8 = vqh_gemm_wgrad_group runs its products one by one;
 * bit 10 (1024) = no XCD-contiguous order in the grouped kernels, bit 11 (2048) = x3 tiles stage linear outputs through LDS (A/B);
 * bit 9 (512) = the 256x128 tiles run on the native fp32 MFMA (v_mfma_f32_32x32x2_f32) instead of the default, which feeds
 * the bf16 matrix pipes with an EXACT three-way split of every fp32 operand (a = h + m + l, six cross products, fp32
 * accumulation): same accuracy as the fp32 MFMA against fp64 (csrc/gemm_dma.inc, tools/x3_lab.hip), 1.6-1.7x its speed */
int vqh_gemm_set_flags(int flags);

/* ---------------------------------------------------------------------------------------------
 * Plane tensors ("P3") and the GEMM that consumes them (csrc/gemm_p3.inc) -- the default path of every Linear that tiles evenly.
 * A plane tensor holds an fp32 matrix X[R][C] (C % 32 == 0) as its EXACT three-way bf16 split x = h + m + l (h = bf16(x),
 * m = bf16(x - h), l = x - h - m; round to nearest even; 3 x 8 significant bits = fp32's 24):
 *     bf16 P[R][C / 32][3 planes: h, m, l][32]      element (r, c) of plane p at byte r * pitch + (c >> 5) * 192 + p * 64 + (c & 31) * 2
 * i.e. 1.5x the fp32 bytes.  Guaranteed domain: finite x with 2^-100 <= |x| <= 3.38e38 (exact; below 2^-100 the low planes reach
 * the bf16 denormal range and the product keeps at least the high plane's 8 bits; above 3.38e38 h rounds to Inf and the
 * product is NaN where an fp32 MFMA returns the value; Inf and NaN operands give NaN).
 * The producers of GEMM operands (LayerNorm, GEMM epilogues, attention, the per-step weight split) write planes themselves;
 * vqh_p3_split is the generic converter.  vqh_gemm_p3 computes C = epilogue(opA(A) . opB(B)) like vqh_gemm with both operands
 * given as plane tensors (six bf16 MFMA products per fp32 product, fp32 accumulation: as accurate against fp64 as the fp32
 * MFMA), writing fp32 (C), a plane tensor (Cp) or both; shapes must satisfy vqh_gemm_p3_eligible (M % 256, N % 128, K % 32).
 * sign_bits: EPI_RELU_DROP writes, EPI_MUL_POSMASK reads 1 bit per output (v > 0) in the kernel's own order (M * N / 8 bytes),
 * instead of an fp32 copy of the activation.  Layouts: (a_kcontig, b_kcontig) = (1, 1) forward, (1, 0) dgrad, (0, 0) wgrad.
 * STAGE IMAGES (pitch 0 in vqh_p3_split and for the A / B operands of vqh_gemm_p3): the same planes stored as the 1 KB units the
 * kernel's LDS-DMA moves -- [R / 256][C / 32][16 row blocks][3 planes][64 lanes x 16 B], lane L of a unit = row L >> 2 of the
 * block, 16-byte chunk (L & 3) ^ F((L >> 4) & 3), F = (0, 2, 3, 1) -- so that every DMA instruction reads consecutive memory
 * (R % 256 == 0; same size, same values, bit-identical results; one image serves the k-contiguous and the row-contiguous use).
 * ------------------------------------------------------------------------------------------- */
int vqh_p3_split(const float* X, int ldx, void* P, long long pitch_bytes, int rows, int cols, vqh_stream_t stream);
typedef struct vqh_p3_item_t { const float* X; void* P; int rows, cols; long long ldx, pitch_bytes; } vqh_p3_item_t;
/* many matrices in one launch: the per-step split of the weights (they change once per step, in AdamW) */
int vqh_p3_split_multi(int n, const vqh_p3_item_t* items, vqh_stream_t stream);
int vqh_gemm_p3_eligible(int M, int N, int K);
int vqh_gemm_p3(int a_kcontig, int b_kcontig, int M, int N, int K, const void* Ap, long long pitch_a, const void* Bp,
                long long pitch_b, float* C, int ldc, void* Cp, long long pitch_c, const float* bias, int mode,
                const float* aux_in, float* aux_out, int ldaux, unsigned* sign_bits, float beta,
                const unsigned long long* rng_state, unsigned drop_site, float drop_p, float* workspace,
                long long workspace_floats, vqh_stream_t stream);
/* all weight-gradient products of one layer on plane operands (see vqh_gemm_wgrad_group); every product must tile evenly
 * (n_out % 256, k_in % 128, rows % 32); db (may be NULL) = column sums of dY, computed by MFMAs against a ones fragment */
typedef struct vqh_wgrad_p3_t {
    int rows, n_out, k_in;
    const void* dYp; long long pitch_dy;
    const void* Xp; long long pitch_x;
    float* dW; int lddw;
    float* db;
} vqh_wgrad_p3_t;
int vqh_gemm_p3_wgrad_group(int n, const vqh_wgrad_p3_t* problems, float* workspace, long long workspace_floats, vqh_stream_t stream);

/* Live timing of the GEMM main kernels with HIP events on their launch stream (bench.py's roofline figure):
 * begin(), run eager (non-captured) steps, end(out) with out = double[7][4][9][3]: for kernel family (0 = gemm_f32_mfma,
 * the 128x128 register-staged tile; 1 = gemm_f32_dma, the 256x128 LDS-DMA tile; 2 = gemm_f32_dma_group, one entry per launch;
 * 3 = gemm_f32_x3 and 4 = gemm_f32_x3_group, the same tiles on the bf16 pipes with the operand split inside the K loop;
 * 5 = gemm_p3 and 6 = gemm_p3_group, plane-tensor operands), operand layout (a_kcontig*2 +
 * b_kcontig) and kernel template MODE+1 (0 = generic kernel, 1.. = epilogue-specialised): launches, kernel seconds,
 * sum of 2*M*N*K.  The split-K reduce launch is not inside the bracket. */
int vqh_gemm_profile_begin(void);
int vqh_gemm_profile_end(double* out);

/* nn.LayerNorm forward/backward (eps 1e-5, biased variance); 35 instances on the path */
int vqh_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, float* mean,
                      float* rstd, int rows, int H, float eps, vqh_stream_t stream);
/* dx_drop (optional, [rows,H] dense): also writes dx * keep-mask(drop_site) -- the dropout backward of the residual
 * branch that consumes dx next (dropout1/dropout2 of the Transformer layers), folded into the producing kernel */
int vqh_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* w, const float* mean,
                      const float* rstd, float* dx, int lddx, int accumulate_dx, float* dw, float* db, float beta,
                      int rows, int H, float* dx_drop, const unsigned long long* rng_state, unsigned drop_site,
                      float drop_p, float* workspace, long long workspace_floats, vqh_stream_t stream);

/* out[i] = beta*out[i] + sum_s slabs[s*stride + i] ;  out[n] = beta*out[n] + sum_m X[m][n] */
int vqh_reduce_slabs(const float* slabs, int S, long long stride, long long n, float* out, float beta,
                     vqh_stream_t stream);
int vqh_colsum(const float* X, int ld, int M, int N, float* out, float beta, float* workspace,
               long long workspace_floats, vqh_stream_t stream);

/* input_proj / ss_input_proj + inp_dropout + pos_enc (models/vq_vae.py:642-643, 649-650) and its weight grads */
int vqh_embed_fwd(const float* x, int ldx, int col0, const float* W, const float* b, const float* pe, float* out,
                  int rows, int L, int H, const unsigned long long* rng_state, unsigned drop_site, float drop_p,
                  vqh_stream_t stream);
int vqh_embed_bwd(const float* dy, const float* x, int ldx, int col0, float* dW, float* db, float beta, int rows,
                  int H, const unsigned long long* rng_state, unsigned drop_site, float drop_p, float* workspace,
                  long long workspace_floats, vqh_stream_t stream);

/* out[b,:] = p0 (+ p1): tokenizer.queries expand (:313), query_embed[:L] + pos_enc[:L] (:750-751) */
int vqh_bcast_rows(const float* p0, const float* p1, float* out, int B, long long n, vqh_stream_t stream);
int vqh_dropout_bwd(const float* dy, float* out, long long n, const unsigned long long* rng_state,
                    unsigned drop_site, float drop_p, vqh_stream_t stream);
int vqh_add(const float* a, const float* b, float* out, long long n, vqh_stream_t stream);
int vqh_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols, vqh_stream_t stream);
int vqh_sigmoid_bwd(const float* dy, const float* y, float* out, long long n, vqh_stream_t stream);

/* input-only rigid augmentation + coordinate noise (models/vq_vae.py:775-792; u,t,noise device tensors or NULL) */
int vqh_augment(const float* x, const float* u, const float* t, const float* noise, float* out, int B, int L,
                vqh_stream_t stream);
/* P = softmax(scale*S + colbias) per row, in place: soft-VQ probabilities (:838-841), usage-entropy logits (:1305-1306) */
int vqh_softmax_rows(float* S, int ld, const float* colbias, float scale, int R, int K, vqh_stream_t stream);
int vqh_softmax_bwd_colgrad(float* P, int ld, const float* g, int R, int K, vqh_stream_t stream);
int vqh_usage_entropy_finish(const float* colsum, int K, int R, float lambda, float* g, float* metrics, int i_loss,
                             int i_reg, vqh_stream_t stream);
int vqh_vq_mix(const float* ze, const float* zsoft, const float* zhard, float alpha, float* out, long long n,
               vqh_stream_t stream);

/* nn.MultiheadAttention core (scaled scores, key padding mask, softmax, dropout, P.V), flash style.
 * Q/K/V/O element (b, t, head, d) at ptr[(b*T + t)*ld + head*dh + d]; kvalid [B,S] bytes (1 = attend) or NULL.
 * qkv_shared bit 0: Q holds ONE sample ([T, ldq]) shared by all B batch entries; bit 1: the same for K and V ([S, ld]).
 * The first decoder layer sees the same query_embed+pos_enc rows for every sample (models/vq_vae.py:750-751) and the
 * first tokenizer layer the same learned queries (:313), so their projections are computed once; outputs and the
 * gradients dQ/dK/dV stay per sample. */
int vqh_attn_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                 float* LSE, const unsigned char* kvalid, int B, int nh, int T, int S, int dh, int qkv_shared,
                 const unsigned long long* rng_state, unsigned drop_site, float drop_p, vqh_stream_t stream);
int vqh_attn_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                 const float* LSE, const float* dO, int lddo, float* Dsum, float* dQ, int lddq, float* dK, int lddk,
                 float* dV, int lddv, const unsigned char* kvalid, int B, int nh, int T, int S, int dh, int qkv_shared,
                 const unsigned long long* rng_state, unsigned drop_site, float drop_p, vqh_stream_t stream);
/* diagnostic: bit 0 = do not use the fused short-sequence (T,S <= 64) kernels; bit 1 = short-sequence backward in its
 * round-1 two-wave-group form instead of the quartered one; bit 2 = the kernels of head dim 64 (general and short-sequence
 * ones) on the native fp32 MFMA instead of the bf16 pipes fed by exact 3-way operand splits (A/B runs: same results to fp32
 * round-off); returns the previous flags */
int vqh_attn_set_flags(int flags);

/* ids of positions outside `valid` (bytes, 1 = valid) become -1, which the statistics entry points below ignore:
 * VectorQuantizerEMA.forward(mask=...) restricts the EMA statistics / usage histogram to valid positions (:192-205, :251-256) */
int vqh_vq_mask_ids(const long long* idx, const unsigned char* valid, long long* out, int R, vqh_stream_t stream);
/* VectorQuantizerEMA (models/vq_vae.py:19-283).
 * vqh_vq_nearest = distances + argmin (:183-188 / :238-244) without the R x K matrix: fp32 MFMA scores with top-2 tracking,
 * rows inside the fp32 noise band re-evaluated in fp64 (first-minimum tie rule of torch.argmin).  For D = 8..256 (power
 * of two) the codebook streams through LDS in 64-code tiles shared by 4 waves that keep their Z rows in registers.
 * workspace_floats >= 3T + 4K + 2R + 3*nsplit*R + R/4 + 16 + K*(6D+16)/4 + R + 3*min(R,16384)*ceil(K/256) with T = smallest power
 * of two >= 2K (>= 64), nsplit <= K/256 + 1.  Rows flagged ambiguous are compacted into a list and re-evaluated by (row, 256-code
 * chunk) work items spread over the whole chip, merged with the first-minimum rule */
int vqh_vq_nearest(const float* Z, int ldz, const float* E, int lde, long long* idx_out, int idx_offset, int R, int K,
                   int D, float rel_tol, float* workspace, long long workspace_floats, vqh_stream_t stream);
/* Which score kernel vqh_vq_nearest takes for a shape and workspace: 3 = plane-tensor form (D = 128 / 256 / 512, R % 256 == 0,
 * K % 128 == 0: Z and the codebook are split once into bf16 plane tensors and scored by the LDS-DMA fed 256 x 128 tile loop of
 * vqh_gemm_p3 with a running top-2 in registers instead of an output; taken only when the workspace holds
 * 1.5 (R + K) D + K + 6 ns R + 16 floats MORE than the formula above, ns <= 8 the code split), 2 = rows of Z as
 * register-resident planes against code tiles in LDS (D = 64 / 128 / 256), 1 = fp32 MFMA with code tiles in LDS, 0 = per-wave
 * gather.  All forms return the same indices (the refinement settles every row inside the noise band exactly) */
int vqh_vq_nearest_form(int R, int K, int D, long long workspace_floats);
/* workspace floats with which vqh_vq_nearest takes its fastest form for the shape (the minimum above + the plane-tensor extra) */
int vqh_vq_nearest_workspace(int R, int K, int D, long long* floats_out);
/* diagnostic: bit 0 = always use the per-wave global-gather nearest kernel (round-1 form); bit 1 = score with the fp32 MFMA
 * instead of the bf16 pipes on exactly split operands (D = 64 / 128 / 256; same indices); segment sums of tables beyond the LDS
 * run as a stable radix sort by code + a segmented sum (skew-proof) unless bit 2 (one workgroup per code) or bit 4
 * ((row chunk, code range) workgroups with LDS tables) is set; bit 3 = one-wave-per-row refinement instead of the chip-wide one;
 * bit 5 = never take the plane-tensor score form (A/B runs: same results); returns the previous flags */
int vqh_vq_set_flags(int flags);
/* live timing of the nearest-neighbour main kernel with HIP events on its launch stream (bench.py --vq-only):
 * begin(), eager calls, end(out) with out = double[3] = {launches, kernel seconds, sum of 2*R*K*D} */
int vqh_vq_profile_begin(void);
int vqh_vq_profile_end(double* out);
int vqh_vq_gather(const float* E, int lde, const long long* idx, int idx_offset, const float* rows_in, int ldr,
                  float* zq_level, float* res_out, int R, int D, vqh_stream_t stream);
int vqh_vq_finish(const float* zq_levels, int Q, const float* ze, int ldz, float* zq, float* zst, int R, int D,
                  vqh_stream_t stream);
int vqh_vq_segment_sum(const float* rows, int ldr, const long long* idx, int R, int D, int k0, int Kn, float* cnt,
                       float* sum, float* workspace, long long workspace_floats, vqh_stream_t stream);
int vqh_vq_ema_apply(const float* cnt, const float* sum, float* ema_cnt, float* ema_emb, float* emb, int K, int D,
                     float decay, float one_minus_decay, float eps, vqh_stream_t stream);
int vqh_row_sqnorm(const float* X, int ld, int rows, int D, float* out, float scale, vqh_stream_t stream);
/* _maybe_reinit_dead_codes (models/vq_vae.py:91-107): codes with usage <= threshold take row pick[k] of `rows` */
int vqh_vq_reinit(const float* usage, float threshold, const long long* pick, const float* rows, int ldr, float* emb,
                  float* ema_emb, float* ema_cnt, int K, int D, vqh_stream_t stream);
int vqh_vq_usage_stats(const float* usage, int K, float n_positions, float* ep_usage, float* ep_cnt, float* stats,
                       vqh_stream_t stream);

/* VQVAE.loss_function forward + gradient w.r.t. recons / z_e (models/vq_vae.py:1097-1388).
 * weights[16] = {rmsd_w, ss_w, bond_length_w, bond_angle_w, dir_w, dih_w, xyz_tv_lambda, pdm_w, win_kabsch_w,
 *                kappa_w, tau_w, lr_pdm_w, xyz_align_alpha, ss_tv_lambda, label_smoothing, beta}   (HOST pointer)
 * data_stats  = NULL or 6 HOST floats {std xyz, mean xyz} of set_data_stats() (:568-574, to_real :1218-1227)
 * iparams[6]  = {pdm_window, win_kabsch_size, win_kabsch_stride, lr_min_sep, lr_stride, lr_max_offsets} (HOST)
 * metrics[24] (device): loss, Reconstruction_Loss_XYZ, XYZ_MSE_Raw, XYZ_MSE_Aligned, Reconstruction_Loss_SS,
 *   SS_Accuracy, VQ_Loss, Geom_BondLength_Loss, Geom_BondAngle_Loss, Geom_Direction_Loss, Geom_Dihedral_Loss,
 *   Geom_Loss, SS_TV, Usage_Reg, XYZ_TV2, VQ_Perplexity, VQ_DeadRatio, RMSD_Raw, RMSD_Aligned, Geom_LocalPDM,
 *   Geom_WinKabsch, Frenet_Kappa, Frenet_Tau, Geom_LongRangePDM
 * L = the batch's own padded length (pad_collate's L_max: the window / long-range-pair enumerations of :996-1095 depend on it);
 * L_stride >= L = rows per sample in memory (recons, target, mask, d_recons): the fused step pads L up to a length bucket
 * and the loss still sees the reference's [B, L] view; d_recons rows L..L_stride-1 are written as zeros.
 * L_dev (device, may be NULL): L as ONE float in device memory, read by the kernels instead of the argument, so that a captured
 * hipGraph serves every L_max that falls into the same bucket (tables are then sized for L_stride) */
int vqh_loss_fwd_bwd(const float* recons, const float* target, const unsigned char* mask, int masked, const float* ze,
                     const float* zq, const float* vq_stats, int B, int L, int L_stride, const float* L_dev, int Ntok, int D,
                     int use_vq,
                     const float* weights, const int* iparams, const float* data_stats, float* d_recons, float* d_ze,
                     float* metrics, float* workspace, long long workspace_floats, vqh_stream_t stream);

/* clip_grad_norm_ + torch.optim.AdamW over flat buffers (experiment.py:170, run.py:191-197).
 * hyper (device, 9 floats): lr, beta1, beta2, eps, weight_decay, max_norm, 1-beta1^t, 1-beta2^t,
 *                            grad_scale (1/world_size when the gradient buffer holds a SUM over ranks)
 * norm_out (device, 2 floats): total grad norm, clip coefficient */
int vqh_grad_norm(const float* g, long long n, const float* hyper, float* norm_out, double* workspace,
                  vqh_stream_t stream);
int vqh_adamw_step(float* p, float* g, float* m, float* v, long long n, const float* hyper, const float* norm,
                   vqh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQVAE_HIP_H */
