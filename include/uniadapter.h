/*
 * uniadapter.h - C ABI of libuniadapter_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the Uni-Adapter per-sample test-time hot path.  The reference has no
 * FFI of its own (it is pure Python on PyTorch); the seams it does have are Python duck-typed
 * call sites, cited per entry point below as `file:line` under the reference root.  Every entry
 * point here is what a binding for that call site would bind (see INTEGRATION.md for the ctypes
 * stubs).
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch / C++ types in any signature
 *   - return 0 (UA_OK) or a negative UA_E* code; no exception crosses the ABI; ua_strerror(code)
 *   - every buffer (inputs, outputs, state, workspace) is owned by the caller and must stay alive
 *     until the stream reaches the enqueued work; the library never allocates, frees or keeps
 *     pointers, has no global mutable state and never synchronises the device
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *   - tensors are row-major, contiguous unless a leading dimension (ld*) is given, fp32 unless
 *     stated; pointers must be 16-byte aligned, feature dims multiples of 4
 *   - GMM state layout: mu,var (K,M,D) with D fastest; pi,c (K,M); class_counts (K)
 */
#ifndef UNIADAPTER_H
#define UNIADAPTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UA_OK             0
#define UA_EINVAL        -1   /* null pointer / non-positive or inconsistent shape            */
#define UA_EALIGN        -2   /* pointer or leading dimension not 16-byte aligned             */
#define UA_EUNSUPPORTED  -3   /* shape outside the compiled envelope (see each function)      */
#define UA_EWORKSPACE    -4   /* workspace too small                                          */
#define UA_ELAUNCH       -5   /* hipGetLastError() != hipSuccess after the launch             */

#define UA_VERSION 103        /* 0.1.2: round 3 (ua_tta_state.replay_lean, ua_gemm_split_set_persistent) */

int         ua_version(void);
const char* ua_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * MODE-DOTA cache (reference class DOTA_mix, dota_mixture.py:7-274)
 * Envelope of the register-resident kernels: D % 256 == 0, D <= 2048, 1 <= M <= 16.
 * ------------------------------------------------------------------------------------------ */

/* DOTA_mix.predict (dota_mixture.py:236-267) + _log_likelihood (:121-156) for B queries:
 *   v = max(var+eps,1e-8); ll[b,k,m] = -1/2 (sum_d ln v + sum_d (x-mu)^2/v)
 *   joint = ln(pi+1e-10) + ll ; scores[b,k] = logsumexp_m joint
 * Any of scores (B,K), joint (B,K,M), loglik (B,K,M) may be NULL.  Also the batched forward of
 * compute_text_alignment_loss (Uni_Adapter.py:219-244) with x = class embeddings (B = K).      */
int ua_gmm_loglik(const float* mu, const float* var, const float* pi, const float* x,
                  int32_t B, int32_t K, int32_t M, int32_t D, float eps,
                  float* scores, float* joint, float* loglik, void* stream);

/* Gradient of sum(grad_ll * loglik) w.r.t. x: the backward that autograd takes through
 * DOTA_mix._log_likelihood in Uni_Adapter.py:228,467-469.  grad_ll (B,K,M) -> grad_x (B,D).    */
int ua_gmm_loglik_bwd(const float* mu, const float* var, const float* x, const float* grad_ll,
                      int32_t B, int32_t K, int32_t M, int32_t D, float eps,
                      float* grad_x, void* stream);

/* DOTA_mix.fit (dota_mixture.py:162-234): one streaming-EM step, state updated in place.
 * x (B,D), gamma_class (B,K); B <= 64.                                                         */
int ua_gmm_fit(float* mu, float* var, float* pi, float* c, float* class_counts,
               const float* x, const float* gamma_class,
               int32_t B, int32_t K, int32_t M, int32_t D, float eps, void* stream);

/* The cache step of one test sample in ONE pass over the state (Uni_Adapter.py:416-430):
 *   scores = predict(fp16-rounded xq) ; fit(x1, gamma) ; fit(x2, gamma)
 * xq,x1,x2 (1,D); gamma (1,K); scores (1,K).  xq_round_fp16 != 0 applies the `.half()` of
 * Uni_Adapter.py:416 to xq inside the kernel.                                                  */
int ua_gmm_step_fused(float* mu, float* var, float* pi, float* c, float* class_counts,
                      const float* xq, int32_t xq_round_fp16, const float* x1, const float* x2,
                      const float* gamma_class, float* scores,
                      int32_t K, int32_t M, int32_t D, float eps, void* stream);

/* ua_gmm_step_fused followed by ua_gmm_prepare_alignment on the updated state, in the same launch (the rows are still in
 * registers): inv_var (K,M,D) and row_const (K,M) as ua_gmm_prepare_alignment writes them, bit for bit.  Used by the
 * composite replay when residual learning follows the cache step (Uni_Adapter.py:430 -> :444).                          */
int ua_gmm_step_fused_prepare(float* mu, float* var, float* pi, float* c, float* class_counts,
                              const float* xq, int32_t xq_round_fp16, const float* x1, const float* x2,
                              const float* gamma_class, float* scores,
                              int32_t K, int32_t M, int32_t D, float eps, float* inv_var, float* row_const, void* stream);

/* ------------------------------------------------------------------------------------------
 * Zero-shot logits and fusion (Uni_Adapter.py)
 * ------------------------------------------------------------------------------------------ */

/* get_logits_wrapper (Uni_Adapter.py:53-75), uni3d branch, batch 1, given the encoder output:
 *   fhat = feat/|feat| (applied `n_normalize` times: 1 for :56, 2 for :56 followed by :429)
 *   logits = (scale*fhat) @ text^T ; prob = softmax ; entropy = -sum p ln(p+1e-10) ; pred=argmax
 * text is the CURRENT (K,D) text matrix (the reference holds its transpose).
 * Outputs (any may be NULL): fhat (D), logits (K), prob (K), entropy (1), pred (1, int32).     */
int ua_clip_logits(const float* feat, const float* text, int32_t K, int32_t D, float scale,
                   int32_t n_normalize, float* fhat, float* logits, float* prob, float* entropy,
                   int32_t* pred, void* stream);

/* The sample and its augmented copy in ONE launch (two workgroups): identical to
 *   ua_clip_logits(feat, text, K, D, scale, 1, fhat, logits, prob, NULL, NULL) and
 *   ua_clip_logits(feat_aug, text, K, D, scale, 2, faug, NULL, NULL, NULL, NULL)   (Uni_Adapter.py:406,424,429). */
int ua_clip_logits_pair(const float* feat, const float* feat_aug, const float* text, int32_t K, int32_t D,
                        float scale, float* fhat, float* faug, float* logits, float* prob, void* stream);

/* Entropy-weighted fusion (Uni_Adapter.py:491-521 incl. the weight-reuse quirk at :512-513):
 *   w = min(rho*mean(c)/batch, eta); d = w*dota; wc=1/(H(clip)+1e-3); wd=1/(H(d)+1e-3);
 *   wc = wc/(wc+wd); wd = wd/(wc+wd); final = wc*clip + wd*d
 * Optionally (target != NULL) writes rank = #classes with final > final[target] (top-k
 * accuracy, utils/utils.py:154-167) and the arg-max.                                           */
int ua_fuse_logits(const float* clip_logits, const float* dota_logits, const float* c,
                   int32_t K, int32_t M, float rho, float eta, int32_t batch,
                   const int32_t* target, float* final_logits, int32_t* rank, int32_t* pred,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Residual learning of the text anchors (Uni_Adapter.py:191-270, 444-476)
 * ------------------------------------------------------------------------------------------ */

/* e = (text0+res)/|text0+res| per row (Uni_Adapter.py:388-393, 451-456); norm_out (K) optional.
 * clamp_eps: F.normalize's max(|t|,eps) (1e-12 at :392) or 0 for the plain division at :456.   */
int ua_text_normalize(const float* text0, const float* res, int32_t K, int32_t D, float clamp_eps,
                      float* e_out, float* norm_out, void* stream);

/* Loss of compute_text_alignment_loss (Uni_Adapter.py:248-253) from L (K,K) and its closed-form
 * gradient pushed through logsumexp_m: grad_ll[i,k,m] = dLoss/dL[i,k] * exp(joint[i,k,m]-L[i,k]).
 * K <= 160.  loss_out (1), grad_ll (K,K,M).                                                    */
int ua_align_loss(const float* L, const float* joint, int32_t K, int32_t M,
                  float* loss_out, float* grad_ll, void* stream);

/* Fast path of the alignment forward/backward.  The GMM state is constant during the res_steps Adam
 * steps of one sample, so 1/max(var+eps,1e-8) (K,M,D) and row_const[k,m] = ln(pi+1e-10) - 1/2 sum_d
 * ln v (K,M) are prepared once per sample; forward then is joint = row_const - 1/2 sum (e-mu)^2*inv_var
 * and L = logsumexp_m joint (same quantities as ua_gmm_loglik with B = K queries, Uni_Adapter.py:219-244),
 * backward is grad_e = sum_{k,m} grad_ll (mu - e)*inv_var (same as ua_gmm_loglik_bwd).            */
int ua_gmm_prepare_alignment(const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                             float* inv_var, float* row_const, void* stream);
int ua_align_forward(const float* mu, const float* inv_var, const float* row_const, const float* e,
                     int32_t B, int32_t K, int32_t M, int32_t D, float* L, float* joint, void* stream);
int ua_align_backward(const float* mu, const float* inv_var, const float* e, const float* grad_ll,
                      int32_t B, int32_t K, int32_t M, int32_t D, float* grad_e, void* stream);

/* Chain grad_e through e = t/|t| (t = text0+res), apply one torch.optim.Adam step
 * (lr, betas, eps as Uni_Adapter.py:350-352; `step` is the 1-based step count) to res and its
 * moments, then refresh e and norm for the next evaluation.  All (K,D) except norm (K).        */
int ua_residual_adam(const float* text0, float* res, float* adam_m, float* adam_v,
                     const float* grad_e, float* e, float* norm,
                     int32_t K, int32_t D, int32_t step, double lr, double beta1, double beta2,
                     double adam_eps, void* stream);

/* One residual-learning step after ua_align_forward in ONE launch: ua_align_loss + ua_align_backward
 * + ua_residual_adam with bit-identical results (Uni_Adapter.py:191-270,444-476).  L (K,K), joint
 * (K,K,M) from ua_align_forward on e; updates res / adam_m / adam_v / e / norm in place; grad_e (K,D)
 * is workspace and output; loss (1) and grad_ll (K,K,M) are optional outputs; tickets: K int32 that
 * are zero at the call (left zero on completion).  K <= 128, D % 256 == 0.                       */
int ua_align_step(const float* mu, const float* inv_var, const float* L, const float* joint,
                  const float* text0, float* res, float* adam_m, float* adam_v, float* e, float* norm,
                  int32_t K, int32_t M, int32_t D, int32_t step, double lr, double beta1, double beta2,
                  double adam_eps, float* loss, float* grad_ll, float* grad_e, int32_t* tickets,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Uni3D point-cloud encoder (models/point_encoder.py, models/uni3d.py, timm EVA02 blocks)
 * ------------------------------------------------------------------------------------------ */

/* fps (models/point_encoder.py:7-14 -> pointnet2_ops furthest_point_sample + gather).
 * xyz (B,N,3); N <= 4096.  start: first index (0 = pointnet2 convention); skip_eps >= 0 skips
 * points with |p|^2 <= skip_eps as the pointnet2 CUDA kernel does (1e-3), < 0 disables.
 * Ties go to the lowest index.  center_idx (B,G) int32 (optional), centers (B,G,3).            */
int ua_fps(const float* xyz, int32_t B, int32_t N, int32_t G, int32_t start, float skip_eps,
           int32_t* center_idx, float* centers, void* stream);

/* knn_point + Group.forward (models/point_encoder.py:17-49, 99-127): the S nearest points of
 * every centre under the expanded-form distance -2c.p+|c|^2+|p|^2, gathered as
 * [xyz - centre, rgb] (rgb == NULL: xyz-only groups of models/ulip/pointbert/dvae.py:152-181, the
 * colour columns are zero).  feat (B*G*S, ldf) with ldf >= 6 (columns 6..ldf-1 are zero-filled);
 * idx (B,G,S) int32 optional (ascending index order inside a group).  N <= 4096, S <= 256.     */
int ua_knn_group(const float* xyz, const float* rgb, const float* centers,
                 int32_t B, int32_t N, int32_t G, int32_t S,
                 float* feat, int32_t ldf, int32_t* idx, void* stream);

/* Ball query + gather of the OpenShape PPAT set-abstraction layer (models/openshape/pointnet_util.py:
 * 89-146): per centre the first S points in index order with square_distance <= radius^2, padded with
 * the first hit; feat row = [p - centre (3) | point_feat[p] (F) | zeros up to ldf].  xyz (B,N,3),
 * point_feat (B,N,F), centers (B,G,3); feat (B*G*S, ldf), ldf >= 3+F; idx (B,G,S) optional.  N <= 8192. */
int ua_ball_group(const float* xyz, const float* point_feat, const float* centers,
                  int32_t B, int32_t N, int32_t G, int32_t S, int32_t F, float radius,
                  float* feat, int32_t ldf, int32_t* idx, void* stream);

/* Epilogue of ua_gemm; all pointers optional (NULL = skipped).  Order of operations:
 *   v = acc + (bias[n] + rowgroup_add[m / group_rows, n])     (group_rows: a multiple of 32)
 *   v = v*scale[n] + shift[n]                (eval BatchNorm, models/point_encoder.py:135,141)
 *   act: 0 none | 1 ReLU | 2 GELU(erf) | 3 SwiGLU (columns come in blocks of 32 gate / 32 up:
 *        out[:, j] = silu(gate)*up, N_out = N/2; timm SwiGLU of EvaBlock)
 *   v += residual[m, n]
 *   C[m,n] = v (if C) ; pool_out[m / pool_rows, n] = max over the row group (pool_rows = 64 or 32)
 *   split_out (optional): the same output as an H32L32 split-precision matrix (see ua_split_f16)
 * LayerNorm folded around the GEMM (ua_gemm_split only; the EvaBlock norm1 / norm2 / SwiGLU ffn_ln, timm eva.py, never
 * exist as a pass over HBM): LN(a).W0^T = rstd*(a.(gamma (.) W0)^T - mean * rowsum(gamma (.) W0)) + W0.beta, so
 *   producer: row_stats_out[t*ld_stats + m] = (sum, sum of squares) of the OUTPUT values v[m, :] this launch wrote from
 *             its t-th tile of 256 input columns (float2; t < ceil(N/256)); ua_row_stats_finalize reduces the tiles to
 *             (mean, rstd) per row;
 *   consumer: A holds the UN-normalised rows, W = gamma (.) W0 and bias = b0 + W0.beta (prepared by the caller);
 *             ln_mean_rstd (M float2) and ln_colsum[n] = sum_k W[n,k] change the first line above to
 *             v = rstd[m]*(acc - mean[m]*ln_colsum[n]) + (bias[n] + rowgroup_add[..]).
 *             Instead of ln_mean_rstd the consumer can take the producer's partials directly (ln_stats_in: (ln_parts,
 *             ln_stats_ld) float2 with ln_parts <= 4, ln_n values per row, ln_eps) and reduce them itself - the same
 *             arithmetic as ua_row_stats_finalize, bit-identical results, one launch less.                        */
typedef struct {
    const float* bias;
    const float* scale;
    const float* shift;
    const float* rowgroup_add; int64_t ldg; int32_t group_rows;
    int32_t      act;
    const float* residual;     int64_t ldr;
    float*       pool_out;     int64_t ldp; int32_t pool_rows;
    void*        split_out;    int64_t ld_split; float split_scale;  /* optional H32L32 copy of the output */
    float*       row_stats_out; int64_t ld_stats;                    /* ua_gemm_split only */
    const float* ln_mean_rstd;  const float* ln_colsum;              /* ua_gemm_split only */
    const float* ln_stats_in;   int64_t ln_stats_ld; int32_t ln_parts; int32_t ln_n; float ln_eps;   /* ua_gemm_split only */
} ua_gemm_epilogue;

/* C (M,N) = A (M,K) @ W (N,K)^T with fused epilogue: every Linear / Conv1d(k=1) of the encoder
 * (models/point_encoder.py:133-144,175-186; timm EvaBlock q/k/v/proj/fc1/fc2).  fp32 inputs,
 * exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  K % 4 == 0; lda, ldw, ldc % 4 == 0.               */
int ua_gemm(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
            int32_t M, int32_t N, int32_t K, const ua_gemm_epilogue* ep, void* stream);

/* Split-precision ("H32L32") operands for ua_gemm_split.  A (rows, cols) fp32 matrix is stored
 * with the row pitch of fp32 (ld*4 bytes, ld % 32 == 0); each group of 32 consecutive columns is 32
 * fp16 "hi" values (64 B) followed by 32 fp16 "lo" values (64 B) with x*scale = hi + lo, hi =
 * rn16(x*scale), lo = rn16(x*scale - hi); scale is a power of two.  Columns cols..cols_pad-1 are
 * zero.  ua_split_f16 converts an fp32 matrix (weights at load time, activations that no producer
 * kernel emits in split form).                                                                  */
int ua_split_f16(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols,
                 int32_t cols_pad, float scale, void* stream);

/* C = A @ W^T with the accuracy of an fp32 GEMM on the f16 matrix cores: both operands are
 * H32L32 pairs and the product is evaluated as hi*hi + hi*lo + lo*hi (three
 * v_mfma_f32_16x16x32_f16 per block by default - UA_SPLIT_SCHEDULE=8 selects the 32x32x16 form -, fp32 accumulate; the dropped
 * lo*lo term is 2^-22 relative).
 * gfx950 has no TF32/xf32 path and its f32 MFMA runs at 1/16 of the f16 rate, so this is 16/3 x
 * the f32-MFMA roofline at the same numerical quality (measured: tests/test_gpu_kernels.py).
 * A: (M, K) split, scale a_scale; W: (N, K) split, scale w_scale; accumulators are multiplied by
 * 1/(a_scale*w_scale) before the epilogue.  K % 32 == 0, lda/ldw % 32 == 0.  Same epilogue as
 * ua_gemm.  Replaces the same reference Linear/Conv1d call sites as ua_gemm.                     */
int ua_gemm_split(const void* A, int64_t lda, float a_scale, const void* W, int64_t ldw, float w_scale,
                  float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                  const ua_gemm_epilogue* ep, void* stream);

/* Tuning knob of ua_gemm_split (process-wide, not a data-path entry): mode 1 = launches of more than one round of tiles run as
 * <= one resident workgroup per CU walking the tile list (no workgroup dispatch between a CU's tiles), 2 = only launches of
 * >= 8 rounds, 0 = one workgroup per tile always (the default: measured 0..+4 % slower to 3 % faster by shape, nothing on
 * the whole encoder - DESIGN.md section 7), -1 = take the default from the environment (UA_SPLIT_PERSIST) at the next
 * call.  Results are bit-identical in every mode.                                                                      */
int ua_gemm_split_set_persistent(int32_t mode);

/* Row statistics for the folded LayerNorm: partials (parts, ld_stats) float2 as written through
 * ua_gemm_epilogue.row_stats_out -> mean_rstd[m] = (mean, 1/sqrt(var + eps)) over n values per row (biased variance,
 * F.layer_norm); the tiles are summed in index order in fp64, so the result does not depend on scheduling.          */
int ua_row_stats_finalize(const float* partials, int32_t parts, int64_t ld_stats, int32_t rows, int32_t n, float eps,
                          float* mean_rstd, void* stream);

/* ------------------------------------------------------------------------------------------
 * MFMA (GEMM-form) similarity kernel: DOTA_mix.predict / _log_likelihood for a BATCH of B queries
 * (dota_mixture.py:121-156,236-267 with B > 1; Uni_Adapter.py:219-244, the B = K class embeddings of the
 * alignment matrix).  joint[b,n] = rc[n] - 1/2 ([x_b^2 | x_b] . [1/v_n | -2 mu_n/v_n]), n = k*M+m,
 * rc[n] = ln(pi_n+1e-10) - 1/2 sum ln v_n - 1/2 sum mu_n^2/v_n, scores[b,k] = logsumexp_m joint[b,k*M+m]:
 * a (B x 2D).(2D x KM) product on the matrix cores with H32L32 split operands (terms = 3: fp32-class) or
 * plain fp16 operands (terms = 1: the hi halves only - the BASELINE configs[4] "fp16 feature x prototype"
 * roofline variant; raw-score error ~0.1, reported, never used for the parity claim).
 *   ua_gmm_mfma_operand : builds the state-side operand, row constants and per-row scales in `workspace` from
 *                         (mu, var, pi); valid until the state changes.
 *   ua_gmm_loglik_mfma  : scores (B,K) and/or joint (B,K*M) for queries x (B,D) against that operand.
 * workspace: >= ua_gmm_loglik_mfma_workspace_bytes(B,K,M,D) bytes, 256-byte aligned, the same buffer for both calls
 * (the operand call may pass B = 1).  D % 256 == 0, D <= 2048.                                                       */
size_t ua_gmm_loglik_mfma_workspace_bytes(int32_t B, int32_t K, int32_t M, int32_t D);
int ua_gmm_mfma_operand(const float* mu, const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                        void* workspace, size_t workspace_bytes, void* stream);
int ua_gmm_loglik_mfma(const float* x, int32_t B, int32_t K, int32_t M, int32_t D, int32_t terms,
                       float* scores, float* joint, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Residual learning on the matrix cores (Uni_Adapter.py:191-270,444-476; same step as ua_align_forward + ua_align_step,
 * both contractions with the state as split-f16 MFMA products; csrc/ua_align_mfma.hip).  Envelope: K <= 64, M in
 * {1,2,4,8,16}, D % 256 == 0 (ua_align_mfma_supported); outside it the VALU entry points above are used.
 *   ua_align_mfma_prepare : once per sample, after the cache step: forward operand [1/v | -2 mu/v] + row constants
 *                           (= ua_gmm_mfma_operand) and the transposed operand [mu/v | 1/v] of every 32-feature block.
 *   ua_align_forward_mfma : from the UNNORMALISED embeddings t = text0 + res (K,D): joint (K,K*M), L (K,K) =
 *                           likelihood_matrix (:244), U (K,K*M) = sum_d e (mu - e)/v, nrm (K) = |t|.
 *   ua_align_step_mfma    : loss (:248-253), its gradient through L, grad_e for all rows, and the Adam step
 *                           (torch.optim.Adam, lr/beta/eps as given, `step` = 1-based step count) on res / adam_m / adam_v,
 *                           t <- text0 + res, in ONE launch without any cross-workgroup hand-off (the row dot product
 *                           grad_e . e comes from U).  grad_e (K,D) optional output.
 * After the last step of a sample the caller refreshes e = t/|t| with ua_text_normalize.                              */
int ua_align_mfma_supported(int32_t K, int32_t M, int32_t D);
size_t ua_align_mfma_workspace_bytes(int32_t K, int32_t M, int32_t D);
int ua_align_mfma_prepare(const float* mu, const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                          void* workspace, size_t workspace_bytes, void* stream);
int ua_align_forward_mfma(const float* t, int32_t K, int32_t M, int32_t D, const void* workspace, size_t workspace_bytes,
                          float* L, float* joint, float* U, float* nrm, void* stream);
int ua_align_step_mfma(const float* L, const float* joint, const float* U, const float* nrm, const float* text0,
                       float* res, float* adam_m, float* adam_v, float* t, int32_t K, int32_t M, int32_t D, int32_t step,
                       double lr, double beta1, double beta2, double adam_eps, const void* workspace, size_t workspace_bytes,
                       float* loss, float* grad_e, void* stream);

/* LayerNorm over the last dim (timm LayerNorm eps 1e-6): rows x cols, cols % 4 == 0, cols<=8192.
 * Columns cols..cols_pad-1 of y are zero-filled (GEMM K padding).                              */
int ua_layernorm(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                 float* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad, void* stream);
/* Same, writing the result as an H32L32 split matrix (cols_pad % 32 == 0) for ua_gemm_split.     */
int ua_layernorm_split(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                       void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad,
                       float scale, void* stream);

/* Multi-head self-attention softmax(q k^T / sqrt(hd)) v of timm EvaAttention without rope
 * (models/point_encoder.py:217-218) for Bc clouds of G patch tokens + 1 CLS token.
 * Token rows: cloud b patch g -> b*G+g ; CLS of cloud b -> Bc*G+b.  qkv (rows, 3*H*hd) with
 * q|k|v column blocks; out (rows, H*hd).  hd == 64.                                            */
int ua_attention(const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                 int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream);
/* The CLS query of every (cloud, head) only (output rows Bc*G .. Bc*G+Bc-1); see ua_attention_split_cls. */
int ua_attention_cls(const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                     int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream);

/* Same attention with fp32-accurate products on the f16 matrix cores: qkv is an H32L32 split matrix
 * (rows, 3*H*hd) with scale qkv_scale (as emitted by ua_gemm_split's split_out), the result is
 * written as fp32 (out, optional) and/or as an H32L32 split matrix (out_split, optional) for the
 * proj GEMM.  Softmax statistics and all accumulation are fp32.  G % 128 == 0, G <= 1024 (the CLS
 * query's one-query path keeps the G + 1 scores of a (cloud, head) pair in LDS).                  */
int ua_attention_split(const void* qkv, int64_t ldqkv, float qkv_scale, float* out, int64_t ldo,
                       void* out_split, int64_t ldos, float out_scale,
                       int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream);
/* The same for the CLS query of every (cloud, head) only - rows Bc*G .. Bc*G+Bc-1 of the output; q is read for those rows, k and v
 * for all rows.  For a last transformer block whose patch-token outputs nobody reads (models/point_encoder.py:219 keeps x[:, 0]). */
int ua_attention_split_cls(const void* qkv, int64_t ldqkv, float qkv_scale, float* out, int64_t ldo,
                       void* out_split, int64_t ldos, float out_scale,
                       int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream);

/* Token assembly (models/point_encoder.py:198-208): x[b*G+g] = tok[b*G+g] + pos[b*G+g];
 * x[Bc*G+b] = cls_token + cls_pos.                                                             */
int ua_assemble_tokens(const float* tok, const float* pos, const float* cls_token,
                       const float* cls_pos, float* x, int32_t Bc, int32_t G, int32_t C,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Composite: the in-order cache replay of a chunk of samples in ONE host call
 * (Uni_Adapter.py:388-521 per sample: clip logits -> predict+fit+fit -> [sample index > 0:
 * res_steps x (alignment forward, loss+grad, backward, Adam)] -> fusion).  Launch orchestration
 * only: identical to calling the entry points above in sequence (with align_mfma: ua_gmm_step_fused,
 * ua_align_mfma_prepare, res_steps x (ua_align_forward_mfma, ua_align_step_mfma), ua_text_normalize).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t K, M, D;
    float   eps, rho, eta, logit_scale;           /* dota_cfg (utils/params.py:103-106), 100 (:57)       */
    int32_t res_learning, res_steps;              /* --res-learning, 10 (:467)                            */
    double  lr, beta1, beta2, adam_eps;           /* torch.optim.Adam defaults, lr 1e-3 (:350-352)        */
    float  *mu, *var, *pi, *c, *class_counts;     /* GMM state                                            */
    const float* text0;                           /* initial text features (K,D)                          */
    float  *res, *adam_m, *adam_v, *e, *norm;     /* residuals, Adam moments, current text (K,D), |t| (K) */
    float  *fhat, *faug, *clip_logits, *prob, *dota_scores;          /* scratch: (D),(D),(K),(K),(K)      */
    float  *L, *joint, *grad_ll, *grad_e, *loss;                     /* scratch: (K,K),(K,K,M)x2,(K,D),(1)*/
    float  *inv_var, *row_const;                                     /* scratch: (K,M,D),(K,M)            */
    int32_t *tickets;                                                /* scratch: (K) int32 (ua_align_step) */
    /* residual learning on the matrix cores (ua_align_*_mfma; used when align_mfma != 0 and the shape is supported): */
    int32_t align_mfma;                                              /* 0: VALU kernels, 1: MFMA kernels  */
    float  *t, *U;                                                   /* text0 + res (K,D); scratch (K,K,M) */
    void   *align_ws; size_t align_ws_bytes;                         /* ua_align_mfma_workspace_bytes      */
    /* VALU residual step as ua_align_loss + ua_align_backward + ua_residual_adam (bit-identical to ua_align_step) instead of the
     * one-launch ua_align_step: two launches more per Adam step, but ~45 % of the CU-time (the loss statistics are computed once,
     * not by each of the 4 K workgroups) - for a replay that shares the GPU with the encoder (throughput), not for a replay
     * rank that other GPUs wait on (latency). */
    int32_t replay_lean;
} ua_tta_state;

/* feats (n,2,D): [:,0] encoder output of the sample, [:,1] of its augmented copy.
 * targets (n) int32 or NULL; final_logits (n,K); ranks (n) (required with targets); preds (n) or NULL.
 * sample_index0: stream index of the first sample (residual learning starts at index 1, :444);
 * adam_step0: Adam steps taken so far (the caller adds res_steps per sample with index > 0).     */
int ua_tta_chunk(const ua_tta_state* s, const float* feats, const int32_t* targets, int32_t n,
                 int64_t sample_index0, int64_t adam_step0, float* final_logits,
                 int32_t* ranks, int32_t* preds, void* stream);

/* ------------------------------------------------------------------------------------------
 * Full-covariance DOTA (dota.py:19-87).
 * ua_dota_fit (:42-66): Sigma (K,D,D), mu (K,D), c (K) updated in place from x (B,D) and soft labels
 *   y (B,K), B <= 64; also writes overall_Sigma = mean_k Sigma_k (D,D) and A = (1-eps) overall + eps I
 *   (the matrix ua-side update() inverts, :70-71).  D % 4 == 0.
 * ua_spd_inverse_small: out = inverse of the n x n block at A (n <= 64; principal blocks / Schur
 *   complements of an SPD matrix): base case of the blocked inversion driven from the host with ua_gemm.
 * ua_to_half: fp32 -> fp16 (round to nearest even), e.g. Lambda.half() (:72).
 * ua_dota_predict (:75-86): Lambda_half (D,D) fp16, mu (K,D) fp32, X_half (B,D) fp16 ->
 *   W_half (D,K) fp16 workspace and scores (B,K) (fp16 values stored as fp32): every tensor the
 *   reference materialises is rounded to fp16 at the same point, sums accumulate in fp32.
 * ------------------------------------------------------------------------------------------ */
int ua_dota_fit(float* Sigma, float* mu, float* c, const float* x, const float* y,
                int32_t B, int32_t K, int32_t D, float eps, float* overall_Sigma, float* A, void* stream);
int ua_spd_inverse_small(const float* A, int64_t lda, int32_t n, float* out, int64_t ldo, void* stream);
int ua_to_half(const float* x, void* y_half, int64_t n, void* stream);
int ua_dota_predict(const void* Lambda_half, const float* mu, const void* X_half, int32_t B, int32_t K, int32_t D,
                    void* W_half, float* scores, void* stream);

/* ------------------------------------------------------------------------------------------
 * Original Uni-Adapter cluster cache (Uni_Adapter.py:78-156,541-574; utils/math_utils.py:21-95).
 * ua_cluster_merge (:93-116): the n_slots prototypes of ONE (full) class - keys (n,D), conf (n), count (n) int32,
 *   prob (n,K) - absorb the unit feature f with confidence *conf_new into the most similar prototype; the prototype's
 *   prob / confidence are recomputed from 100 * key @ text^T (text (K,D)).
 * ua_cluster_laplacian (math_utils.py:50-58): W (P,P) = normalised keys @ keys^T in, L_reg = I - S W' S + 2 lambda I out
 *   (W' = W with entries < threshold zeroed, S = diag(1/(sqrt(rowsum W') + 1e-8))); scratch_s: P floats.
 * ua_cluster_cg_step (math_utils.py:31-43): one conjugate-gradient iteration after ApT = pT @ A^T (ua_gemm); all
 *   vectors transposed, (K right-hand sides, P) with row pitch ld (>= P, zero padding; ld % 4 == 0 for the GEMM); sets *not_done if any residual norm^2 >= tol.
 * ua_cluster_logits (:148-156, :552-574): prototype classes from the refined values xT (K,P), affinity f . keys,
 *   cache_logits (K, optional) and final = softmax(clip/100)/H1 + softmax(cache_logits)/H2 (K).
 * ------------------------------------------------------------------------------------------ */
int ua_cluster_merge(float* keys, float* conf, int32_t* count, float* prob, const float* f, const float* conf_new,
                     const float* text, int32_t n_slots, int32_t K, int32_t D, float beta, void* stream);
int ua_cluster_laplacian(float* W, int32_t P, int32_t ld, float threshold, float lambda_reg, float* scratch_s, void* stream);
int ua_cluster_cg_step(float* xT, float* rT, float* pT, const float* ApT, float* rz_old, int32_t K, int32_t P, int32_t ld, float tol,
                       int32_t* not_done, void* stream);
int ua_cluster_logits(const float* xT, int32_t ld, const float* keys, const float* f, const float* clip_logits, int32_t P, int32_t K, int32_t D,
                      int32_t* cls_scratch, float* aff_scratch, float* cache_logits, float* final_logits, void* stream);

/* PointBERT / ULIP-2 variant (models/ulip/pointbert/point_encoder.py:95-98,173-192):
 * ua_add_pos: x += pos before every block (patch rows get pos[b*G+g], CLS rows cls_pos).
 * ua_cls_max_concat: out (Bc, 2C) = [x_cls | max over the G patch tokens] (:190).                */
int ua_add_pos(float* x, const float* pos, const float* cls_pos, int32_t Bc, int32_t G, int32_t C, void* stream);
int ua_cls_max_concat(const float* x, float* out, int32_t Bc, int32_t G, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * "Prototype deltas" multi-GPU mode (BASELINE.json north_star: "RCCL all-gather over xGMI of per-rank
 * prototype deltas after each chunk").  No reference counterpart - its loop is single-process (Uni_Adapter.py:368-579);
 * the statistics are the ones its M-step carries (dota_mixture.py:196-230).  Every rank replays its own shard of a step
 * from the common state S0 = (mu0, var0, c0), then
 * ua_gmm_delta_pack: out = [dS1 (K*M*D) | dS2 (K*M*D) | dC (K*M)],  dS1 = c*mu - c0*mu0,  dS2 = c*var - c0*var0,  dC = c - c0
 *   (fp64 products, stored fp32); D % 4 == 0.
 * ua_gmm_delta_merge: deltas = W packed buffers, buffer r at deltas + r*stride (stride % 4 == 0), summed in rank order:
 *   c = c0 + sum dC, mu = (c0*mu0 + sum dS1)/(c+1e-10), var = max((c0*var0 + sum dS2)/(c+1e-10), 1e-8), pi = c/(sum_m c + 1e-10).
 * ua_vec_delta: out = cur - base.   ua_vec_merge: out = base + scale * sum_r deltas[r*stride + i]  (class counts: scale 1;
 *   text residuals and their Adam moments: scale 1/W, "average of the per-rank Adam deltas").
 * ------------------------------------------------------------------------------------------ */
int ua_gmm_delta_pack(const float* mu, const float* var, const float* c, const float* mu0, const float* var0, const float* c0,
                      int32_t K, int32_t M, int32_t D, float* out, void* stream);
int ua_gmm_delta_merge(float* mu, float* var, float* pi, float* c, const float* mu0, const float* var0, const float* c0,
                       const float* deltas, int32_t W, int64_t stride, int32_t K, int32_t M, int32_t D, void* stream);
int ua_vec_delta(const float* cur, const float* base, int64_t n, float* out, void* stream);
int ua_vec_merge(const float* base, const float* deltas, int32_t W, int64_t stride, int64_t n, float scale, float* out, void* stream);

/* Class-sharded exact replay for large K ("rank r owns classes [rK/W,(r+1)K/W) ... exchange = all_gather of
 * K/W scores per sample + the sum of c").  Every formula of dota_mixture.py:162-258 is independent per class, so a rank runs
 * ua_gmm_step_fused on its slice of the state through offset pointers; the cross-class couplings are the softmax over K
 * (clip logits: computed on every rank) and mean(c) at Uni_Adapter.py:491.  per = ceil(K/W) classes per rank.
 * ua_class_shard_pack: out (per+2 floats) = [this rank's K_local <= per scores, zero padded | fp64 sum of its c slice as a
 *   (hi, lo) float pair].
 * ua_fuse_logits_sharded: ua_fuse_logits for n samples from the W gathered buffers - gathered is (W, n, per+2), rank order,
 *   i.e. the all_gather of every rank's (n, per+2) pack buffers of a chunk; clip_logits / final_logits / dota_out (optional)
 *   are (n,K), target / rank / pred (n).  With residual learning off nothing the fusion produces feeds back into the state
 *   (Uni_Adapter.py:417,430,491), so ONE collective per chunk carries all n score slices.                                */
int ua_class_shard_pack(const float* scores, const float* c, int32_t K_local, int32_t M, int32_t per, float* out, void* stream);
int ua_fuse_logits_sharded(const float* clip_logits, const float* gathered, int32_t W, int32_t n, int32_t per, int32_t K, int32_t M,
                           float rho, float eta, int32_t batch, const int32_t* target, float* final_logits, int32_t* rank,
                           int32_t* pred, float* dota_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIADAPTER_H */
