// Zero-shot logits, entropy-weighted fusion and the text-residual learning step for gfx950.
//
// Reference semantics: Uni_Adapter.py:21-26 (softmax_entropy), :53-75 (get_logits_wrapper),
// :191-270 (compute_text_alignment_loss), :388-393 / :444-476 (text residuals + Adam),
// :491-521 (fusion).  All of these are K- or K^2-sized (K = classes): latency-bound, so each is
// one launch of one (or K) workgroups with wavefront reductions and no intermediate HBM traffic.
#include "ua_common.h"
#include "ua_align_common.h"

namespace {

using ua::block_sum;
using ua::wave_sum;

__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = ua::wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
    return t;
}

// softmax over z[0..K) in LDS -> p (LDS), returns entropy -sum p ln(p+1e-10) (Uni_Adapter.py:21-26)
__device__ float softmax_entropy_lds(const float* z, float* p, int K, double* sd, float* sf) {
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < K; i += blockDim.x) mx = fmaxf(mx, z[i]);
    mx = block_max(mx, sf);
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) { const float e = expf(z[i] - mx); p[i] = e; s += (double)e; }
    const float sum = (float)block_sum(s, sd);
    double h = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const float pi = p[i] / sum;
        p[i] = pi;
        h += (double)(pi * logf(pi + 1e-10f));
    }
    return -(float)block_sum(h, sd);
}

// arg-max with the first index winning ties (the serial `if (z[i] > z[b]) b = i` scan), block-wide; result valid on thread 0
__device__ int block_argmax_first(const float* z, int K, float* sf, int* si) {
    float best = -INFINITY; int idx = 0x7fffffff;
    if ((int)threadIdx.x < K) { best = z[threadIdx.x]; idx = threadIdx.x; }
    for (int i = threadIdx.x + blockDim.x; i < K; i += blockDim.x) if (z[i] > best) { best = z[i]; idx = i; }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(idx, o);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sf[w] = best; si[w] = idx; }
    __syncthreads();
    for (int i = 0; i < nw; ++i) if (sf[i] > best || (sf[i] == best && si[i] < idx)) { best = sf[i]; idx = si[i]; }
    return idx;
}

// ------------------------------------------------------------------------------------------
// get_logits_wrapper, one workgroup of 256 threads.  dynamic LDS: D + 2K floats.
// ------------------------------------------------------------------------------------------
struct ClipJob {            // one feature vector through :56-57 (and :429 for the augmented copy)
    const float* feat; int n_norm; float* fhat; float* logits; float* prob; float* entropy; int* pred;
};
struct ClipArgs { ClipJob job[2]; const float* text; int K, D; float scale; };

// grid = number of jobs (1 or 2: the sample and its augmented copy in one launch), block 1024.
__global__ void __launch_bounds__(1024)
clip_logits_kernel(ClipArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    const ClipJob j = a.job[blockIdx.x];
    const int K = a.K, D = a.D, nt = blockDim.x;
    const float scale = a.scale;
    float* f = smem;          // D
    float* z = smem + D;      // K
    float* p = z + K;         // K
    for (int i = threadIdx.x; i < D; i += nt) f[i] = j.feat[i];
    __syncthreads();
    for (int r = 0; r < j.n_norm; ++r) {                   // x / x.norm(dim=-1)   :56 (and :429)
        double s = 0.0;
        for (int i = threadIdx.x; i < D; i += nt) s += (double)(f[i] * f[i]);
        const float nrm = (float)sqrt(block_sum(s, sd));
        for (int i = threadIdx.x; i < D; i += nt) f[i] = f[i] / nrm;
        __syncthreads();
    }
    if (j.fhat) for (int i = threadIdx.x; i < D; i += nt) j.fhat[i] = f[i];
    if (!j.logits && !j.prob && !j.entropy && !j.pred) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = nt >> 6;
    for (int k = w; k < K; k += nw) {                      // (100*f) @ text^T          :57
        const float4* t4 = reinterpret_cast<const float4*>(a.text + (size_t)k * D);
        double acc = 0.0;
        for (int i = lane; i < D / 4; i += 64) {
            const float4 t = t4[i];
            acc += (double)((scale * f[4 * i + 0]) * t.x) + (double)((scale * f[4 * i + 1]) * t.y)
                 + (double)((scale * f[4 * i + 2]) * t.z) + (double)((scale * f[4 * i + 3]) * t.w);
        }
        acc = wave_sum(acc);
        if (lane == 0) z[k] = (float)acc;
    }
    __syncthreads();
    const float h = softmax_entropy_lds(z, p, K, sd, sf);
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += nt) {
        if (j.logits) j.logits[i] = z[i];
        if (j.prob) j.prob[i] = p[i];
    }
    if (j.pred) {                                          // uniform branch
        __shared__ int si[16];
        const int b = block_argmax_first(z, K, sf, si);
        if (threadIdx.x == 0) *j.pred = b;
    }
    if (threadIdx.x == 0 && j.entropy) *j.entropy = h;
}

// Large K (hundreds of classes and more): the (K,D) text bank no longer fits one workgroup's time budget (16 MB at
// K = 4096), so the rows are spread over the chip - grid (ceil(K/16), jobs), wave = one class row, same per-row
// arithmetic as above - and a second one-workgroup launch does the softmax / entropy / arg-max on the K logits.
// Results are bit-identical to clip_logits_kernel.
__global__ void __launch_bounds__(1024)
clip_rows_kernel(ClipArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    const ClipJob j = a.job[blockIdx.y];
    const int K = a.K, D = a.D, nt = blockDim.x;
    const float scale = a.scale;
    if (!j.logits && blockIdx.x != 0) return;              // this job only wants fhat
    float* f = smem;
    for (int i = threadIdx.x; i < D; i += nt) f[i] = j.feat[i];
    __syncthreads();
    for (int r = 0; r < j.n_norm; ++r) {
        double s = 0.0;
        for (int i = threadIdx.x; i < D; i += nt) s += (double)(f[i] * f[i]);
        const float nrm = (float)sqrt(block_sum(s, sd));
        for (int i = threadIdx.x; i < D; i += nt) f[i] = f[i] / nrm;
        __syncthreads();
    }
    if (j.fhat && blockIdx.x == 0) for (int i = threadIdx.x; i < D; i += nt) j.fhat[i] = f[i];
    if (!j.logits) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = nt >> 6;
    const int k = blockIdx.x * nw + w;
    if (k >= K) return;
    const float4* t4 = reinterpret_cast<const float4*>(a.text + (size_t)k * D);
    double acc = 0.0;
    for (int i = lane; i < D / 4; i += 64) {
        const float4 t = t4[i];
        acc += (double)((scale * f[4 * i + 0]) * t.x) + (double)((scale * f[4 * i + 1]) * t.y)
             + (double)((scale * f[4 * i + 2]) * t.z) + (double)((scale * f[4 * i + 3]) * t.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) j.logits[k] = (float)acc;
}

__global__ void __launch_bounds__(1024)
clip_softmax_kernel(ClipArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ int si[16];
    const ClipJob j = a.job[blockIdx.x];
    if (!j.logits || (!j.prob && !j.entropy && !j.pred)) return;
    const int K = a.K, nt = blockDim.x;
    float* z = smem; float* p = smem + K;
    for (int i = threadIdx.x; i < K; i += nt) z[i] = j.logits[i];
    __syncthreads();
    const float h = softmax_entropy_lds(z, p, K, sd, sf);
    __syncthreads();
    if (j.prob) for (int i = threadIdx.x; i < K; i += nt) j.prob[i] = p[i];
    if (j.pred) {
        const int b = block_argmax_first(z, K, sf, si);
        if (threadIdx.x == 0) *j.pred = b;
    }
    if (threadIdx.x == 0 && j.entropy) *j.entropy = h;
}

// ------------------------------------------------------------------------------------------
// Fusion, one workgroup.  dynamic LDS: 3K floats.
// ------------------------------------------------------------------------------------------
// everything after the fusion weight: zc / zd hold clip and w*dota (K each), p is scratch (K)
__device__ __forceinline__ void fuse_core(float* zc, float* zd, float* p, int K, double* sd, float* sf,
                                          const int* __restrict__ target, float* __restrict__ final_logits,
                                          int* __restrict__ rank, int* __restrict__ pred) {
    const float hc = softmax_entropy_lds(zc, p, K, sd, sf);                      // :508
    __syncthreads();
    const float hd = softmax_entropy_lds(zd, p, K, sd, sf);                      // :509
    float wc = 1.f / (hc + 1e-3f);                                               // :510
    float wd = 1.f / (hd + 1e-3f);                                               // :511
    wc = wc / (wc + wd);                                                         // :512
    wd = wd / (wc + wd);                                                         // :513 (reuses new wc)
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const float v = wc * zc[i] + wd * zd[i];                                 // :521
        p[i] = v;
        final_logits[i] = v;
    }
    __syncthreads();
    if (pred) {                                                                  // uniform branches
        __shared__ int si[16];
        const int b = block_argmax_first(p, K, sf, si);
        if (threadIdx.x == 0) *pred = b;
    }
    if (rank && target) {                                                        // utils/utils.py:154-167
        const int t = *target;
        double rk = 0.0;
        if (t >= 0 && t < K) { for (int i = threadIdx.x; i < K; i += blockDim.x) rk += (p[i] > p[t]) || (p[i] == p[t] && i < t); }
        __syncthreads();
        rk = block_sum(rk, sd);
        if (threadIdx.x == 0) *rank = (t >= 0 && t < K) ? (int)rk : K;
    }
}

__global__ void __launch_bounds__(1024)
fuse_logits_kernel(const float* __restrict__ clip, const float* __restrict__ dota, const float* __restrict__ c,
                   int K, int M, float rho, float eta, int batch, const int* __restrict__ target,
                   float* __restrict__ final_logits, int* __restrict__ rank, int* __restrict__ pred) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    float* zc = smem; float* zd = smem + K; float* p = smem + 2 * K;
    double s = 0.0;
    for (int i = threadIdx.x; i < K * M; i += blockDim.x) s += (double)c[i];
    const float cmean = (float)(block_sum(s, sd) / (double)(K * M));
    const float w = fminf(rho * cmean / (float)batch, eta);                      // :491
    for (int i = threadIdx.x; i < K; i += blockDim.x) { zc[i] = clip[i]; zd[i] = w * dota[i]; }   // :498
    __syncthreads();
    fuse_core(zc, zd, p, K, sd, sf, target, final_logits, rank, pred);
}

// Class-sharded replay (large K): rank r owns classes [r*per, min((r+1)*per, K)) of the state.
// pack: this rank's score slice + the fp64 sum of its slice of c as a (hi, lo) float pair -> out (per + 2 floats).
__global__ void __launch_bounds__(256)
class_shard_pack_kernel(const float* __restrict__ scores, const float* __restrict__ c, int Kl, int M, int per,
                        float* __restrict__ out) {
    __shared__ double sd[16];
    for (int i = threadIdx.x; i < per; i += 256) out[i] = i < Kl ? scores[i] : 0.f;
    double s = 0.0;
    for (int i = threadIdx.x; i < Kl * M; i += 256) s += (double)c[i];
    s = block_sum(s, sd);
    if (threadIdx.x == 0) { const float hi = (float)s; out[per] = hi; out[per + 1] = (float)(s - (double)hi); }
}

// fuse n samples from the W gathered buffers (rank order; buffer of rank r, sample j at gathered[(r*n + j)*(per+2)]):
// dota scores re-assembled, mean(c) from the partial sums.  One workgroup per sample: nothing the fusion produces feeds back
// into the state when residual learning is off (fit uses the zero-shot prob_map, Uni_Adapter.py:417,430; mean(c) is only
// read by the fusion weight :491), so a whole chunk's slices are exchanged with ONE collective and fused afterwards.
__global__ void __launch_bounds__(1024)
fuse_logits_sharded_kernel(const float* __restrict__ clip, const float* __restrict__ gathered, int W, int n, int per,
                           int K, int M, float rho, float eta, int batch, const int* __restrict__ target,
                           float* __restrict__ final_logits, int* __restrict__ rank, int* __restrict__ pred,
                           float* __restrict__ dota_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    const int j = blockIdx.x;
    float* zc = smem; float* zd = smem + K; float* p = smem + 2 * K;
    const size_t rs = (size_t)n * (per + 2);                           // rank stride
    const float* gj = gathered + (size_t)j * (per + 2);
    double s = 0.0;
    for (int r = 0; r < W; ++r) s += (double)gj[(size_t)r * rs + per] + (double)gj[(size_t)r * rs + per + 1];
    const float cmean = (float)(s / (double)(K * M));
    const float w = fminf(rho * cmean / (float)batch, eta);
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const int r = i / per;
        const float d = gj[(size_t)r * rs + (i - r * per)];
        if (dota_out) dota_out[(size_t)j * K + i] = d;
        zc[i] = clip[(size_t)j * K + i]; zd[i] = w * d;
    }
    __syncthreads();
    fuse_core(zc, zd, p, K, sd, sf, target ? target + j : nullptr, final_logits + (size_t)j * K, rank ? rank + j : nullptr,
              pred ? pred + j : nullptr);
}

// ------------------------------------------------------------------------------------------
// e = (text0+res)/|text0+res|, one workgroup per class row.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
text_normalize_kernel(const float* __restrict__ text0, const float* __restrict__ res, int D, float clamp_eps,
                      float* __restrict__ e, float* __restrict__ nrm_out) {
    __shared__ double sd[16];
    const size_t o = (size_t)blockIdx.x * D;
    double s = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) { const float t = text0[o + i] + (res ? res[o + i] : 0.f); s += (double)(t * t); }
    float n = (float)sqrt(block_sum(s, sd));
    if (nrm_out && threadIdx.x == 0) nrm_out[blockIdx.x] = n;
    n = fmaxf(n, clamp_eps);
    for (int i = threadIdx.x; i < D; i += 256) e[o + i] = (text0[o + i] + (res ? res[o + i] : 0.f)) / n;
}

// ------------------------------------------------------------------------------------------
// Alignment loss + gradient w.r.t. the per-mode log-likelihoods, one workgroup of 1024.
//   Ln = L/max(L); P = exp(exp(Ln)); loss = -mean(diag P / rowsum) - mean(diag P / colsum)
// dynamic LDS: K*K (P) + 4K floats.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
align_loss_kernel(const float* __restrict__ L, const float* __restrict__ joint, int K, int M,
                  float* __restrict__ loss_out, float* __restrict__ grad_ll) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ int s_arg;
    float* P = smem;                 // K*K
    float* s1 = smem + K * K;        // row sums
    float* s2 = s1 + K;              // col sums
    float* dg = s2 + K;              // diagonal
    const int n = K * K, tid = threadIdx.x, nt = blockDim.x;
    __shared__ double sc[16 * 64];
    const ua::AlignStats st = ua::align_stats(L, nullptr, K, P, s1, s2, dg, sd, sf, &s_arg, sc);
    if (tid == 0 && loss_out) *loss_out = (float)(-st.tot / (double)K);             // :253
    if (!grad_ll) return;
    const float mx = st.mx, corr = st.corr;
    const int amax = st.amax;
    for (int ik = tid; ik < n; ik += nt) {
        float G = P[ik] / mx;
        if (ik == amax) G += corr;
        const float l = L[ik];
        for (int m = 0; m < M; ++m)
            grad_ll[(size_t)ik * M + m] = G * expf(joint[(size_t)ik * M + m] - l);     // d logsumexp / d joint = softmax
    }
}

// ------------------------------------------------------------------------------------------
// grad_e -> grad_t through e = t/|t|, one Adam step on res, refresh e and |t|.
// One workgroup per class row.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
residual_adam_kernel(const float* __restrict__ text0, float* __restrict__ res, float* __restrict__ am,
                     float* __restrict__ av, const float* __restrict__ ge, float* __restrict__ e,
                     float* __restrict__ nrm, int D, float step_size, float w1, float beta2, float omb2,
                     float bc2_sqrt, float adam_eps) {
    __shared__ double sd[16];
    const int k = blockIdx.x;
    const size_t o = (size_t)k * D;
    const float n = nrm[k];
    double dot = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) dot += (double)(ge[o + i] * e[o + i]);
    const float gdote = (float)block_sum(dot, sd);
    double s = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float g = (ge[o + i] - e[o + i] * gdote) / n;          // d(t/|t|)/dt applied to grad_e
        float m = am[o + i], v = av[o + i];
        m = m + w1 * (g - m);                                        // exp_avg.lerp_(grad, 1-beta1)
        v = v * beta2 + omb2 * (g * g);                              // mul_(beta2).addcmul_(g,g,1-beta2)
        const float den = sqrtf(v) / bc2_sqrt + adam_eps;
        const float r = res[o + i] + ((-step_size) * m) / den;       // addcdiv_(m, den, value=-step_size)
        am[o + i] = m; av[o + i] = v; res[o + i] = r;
        const float t = text0[o + i] + r;
        s += (double)(t * t);
    }
    const float nn = (float)sqrt(block_sum(s, sd));
    if (threadIdx.x == 0) nrm[k] = nn;
    for (int i = threadIdx.x; i < D; i += 256) e[o + i] = (text0[o + i] + res[o + i]) / nn;
}

}  // namespace

static int launch_clip(const ClipArgs& a, int jobs, void* stream) {
    if (a.K <= 0 || a.D <= 0) return UA_EINVAL;
    if (a.D % 4 != 0) return UA_EUNSUPPORTED;
    const size_t lds = (size_t)(a.D + 2 * a.K) * sizeof(float);
    if (lds > 150 * 1024) return UA_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(clip_logits_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return UA_ELAUNCH;
    bool wide = a.K >= 256;
    for (int i = 0; i < jobs; ++i) {
        const ClipJob& j = a.job[i];
        if ((j.prob || j.entropy || j.pred) && !j.logits) wide = false;     // the wide path keeps the logits in the caller's buffer
    }
    if (wide) {
        hipLaunchKernelGGL(clip_rows_kernel, dim3((a.K + 15) / 16, jobs), dim3(1024), (size_t)a.D * sizeof(float), ua::as_stream(stream), a);
        UA_LAUNCH_CHECK();
        const size_t lds2 = (size_t)(2 * a.K) * sizeof(float);
        if (lds2 > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(clip_softmax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds2) != hipSuccess) return UA_ELAUNCH;
        hipLaunchKernelGGL(clip_softmax_kernel, dim3(jobs), dim3(1024), lds2, ua::as_stream(stream), a);
        UA_LAUNCH_CHECK();
        return UA_OK;
    }
    hipLaunchKernelGGL(clip_logits_kernel, dim3(jobs), dim3(1024), lds, ua::as_stream(stream), a);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_clip_logits(const float* feat, const float* text, int32_t K, int32_t D, float scale,
                              int32_t n_normalize, float* fhat, float* logits, float* prob, float* entropy,
                              int32_t* pred, void* stream) {
    UA_CHECK_PTR(feat); UA_CHECK_PTR(text); UA_CHECK_ALIGN16(text);
    if (n_normalize < 0) return UA_EINVAL;
    ClipArgs a{};
    a.job[0] = ClipJob{feat, n_normalize, fhat, logits, prob, entropy, pred};
    a.text = text; a.K = K; a.D = D; a.scale = scale;
    return launch_clip(a, 1, stream);
}

extern "C" int ua_clip_logits_pair(const float* feat, const float* feat_aug, const float* text, int32_t K, int32_t D,
                                   float scale, float* fhat, float* faug, float* logits, float* prob, void* stream) {
    UA_CHECK_PTR(feat); UA_CHECK_PTR(feat_aug); UA_CHECK_PTR(text); UA_CHECK_PTR(fhat); UA_CHECK_PTR(faug);
    UA_CHECK_ALIGN16(text);
    ClipArgs a{};
    a.job[0] = ClipJob{feat, 1, fhat, logits, prob, nullptr, nullptr};          // Uni_Adapter.py:406
    a.job[1] = ClipJob{feat_aug, 2, faug, nullptr, nullptr, nullptr, nullptr};  // :424, :429 (normalised twice)
    a.text = text; a.K = K; a.D = D; a.scale = scale;
    return launch_clip(a, 2, stream);
}

extern "C" int ua_fuse_logits(const float* clip_logits, const float* dota_logits, const float* c,
                              int32_t K, int32_t M, float rho, float eta, int32_t batch,
                              const int32_t* target, float* final_logits, int32_t* rank, int32_t* pred,
                              void* stream) {
    UA_CHECK_PTR(clip_logits); UA_CHECK_PTR(dota_logits); UA_CHECK_PTR(c); UA_CHECK_PTR(final_logits);
    if (K <= 0 || M <= 0 || batch <= 0) return UA_EINVAL;
    const size_t lds = (size_t)(3 * K) * sizeof(float);
    if (lds > 150 * 1024) return UA_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fuse_logits_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return UA_ELAUNCH;
    hipLaunchKernelGGL(fuse_logits_kernel, dim3(1), dim3(K > 1024 ? 1024 : 256), lds, ua::as_stream(stream),
                       clip_logits, dota_logits, c, K, M, rho, eta, batch, target, final_logits, rank, pred);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_class_shard_pack(const float* scores, const float* c, int32_t K_local, int32_t M, int32_t per, float* out,
                                   void* stream) {
    UA_CHECK_PTR(out);
    if (K_local < 0 || K_local > per || per <= 0 || M <= 0) return UA_EINVAL;
    if (K_local > 0) { UA_CHECK_PTR(scores); UA_CHECK_PTR(c); }
    hipLaunchKernelGGL(class_shard_pack_kernel, dim3(1), dim3(256), 0, ua::as_stream(stream), scores, c, K_local, M, per, out);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_fuse_logits_sharded(const float* clip_logits, const float* gathered, int32_t W, int32_t n, int32_t per, int32_t K,
                                      int32_t M, float rho, float eta, int32_t batch, const int32_t* target, float* final_logits,
                                      int32_t* rank, int32_t* pred, float* dota_out, void* stream) {
    UA_CHECK_PTR(clip_logits); UA_CHECK_PTR(gathered); UA_CHECK_PTR(final_logits);
    if (K <= 0 || M <= 0 || batch <= 0 || W <= 0 || n <= 0 || per <= 0 || (int64_t)W * per < K) return UA_EINVAL;
    const size_t lds = (size_t)(3 * K) * sizeof(float);
    if (lds > 150 * 1024) return UA_EUNSUPPORTED;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fuse_logits_sharded_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return UA_ELAUNCH;
    hipLaunchKernelGGL(fuse_logits_sharded_kernel, dim3(n), dim3(K > 1024 ? 1024 : 256), lds, ua::as_stream(stream),
                       clip_logits, gathered, W, n, per, K, M, rho, eta, batch, target, final_logits, rank, pred, dota_out);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_text_normalize(const float* text0, const float* res, int32_t K, int32_t D, float clamp_eps,
                                 float* e_out, float* norm_out, void* stream) {
    UA_CHECK_PTR(text0); UA_CHECK_PTR(e_out);
    if (K <= 0 || D <= 0) return UA_EINVAL;
    hipLaunchKernelGGL(text_normalize_kernel, dim3(K), dim3(256), 0, ua::as_stream(stream),
                       text0, res, D, clamp_eps, e_out, norm_out);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_align_loss(const float* L, const float* joint, int32_t K, int32_t M,
                             float* loss_out, float* grad_ll, void* stream) {
    UA_CHECK_PTR(L);
    if (K <= 0 || M <= 0) return UA_EINVAL;
    if (grad_ll && !joint) return UA_EINVAL;
    if (K > 160) return UA_EUNSUPPORTED;
    const size_t lds = (size_t)(K * K + 4 * K) * sizeof(float);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(align_loss_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return UA_ELAUNCH;
    hipLaunchKernelGGL(align_loss_kernel, dim3(1), dim3(1024), lds, ua::as_stream(stream),
                       L, joint, K, M, loss_out, grad_ll);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_residual_adam(const float* text0, float* res, float* adam_m, float* adam_v,
                                const float* grad_e, float* e, float* norm,
                                int32_t K, int32_t D, int32_t step, double lr, double beta1, double beta2,
                                double adam_eps, void* stream) {
    UA_CHECK_PTR(text0); UA_CHECK_PTR(res); UA_CHECK_PTR(adam_m); UA_CHECK_PTR(adam_v);
    UA_CHECK_PTR(grad_e); UA_CHECK_PTR(e); UA_CHECK_PTR(norm);
    if (K <= 0 || D <= 0 || step <= 0) return UA_EINVAL;
    // scalar bookkeeping exactly as torch.optim.adam._single_tensor_adam (python floats = double)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - beta1);
    const float omb2 = (float)(1.0 - beta2);
    hipLaunchKernelGGL(residual_adam_kernel, dim3(K), dim3(256), 0, ua::as_stream(stream),
                       text0, res, adam_m, adam_v, grad_e, e, norm, D, step_size, w1, (float)beta2, omb2,
                       bc2_sqrt, (float)adam_eps);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_version(void) { return UA_VERSION; }

extern "C" const char* ua_strerror(int code) {
    switch (code) {
        case UA_OK: return "ok";
        case UA_EINVAL: return "invalid argument (null pointer or bad shape)";
        case UA_EALIGN: return "pointer or leading dimension not 16-byte aligned";
        case UA_EUNSUPPORTED: return "shape outside the compiled envelope";
        case UA_EWORKSPACE: return "workspace too small";
        case UA_ELAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
