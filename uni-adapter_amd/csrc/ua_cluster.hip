// Original Uni-Adapter cluster cache: prototype merge, graph-Laplacian label
// refinement by conjugate gradients, one-hot cache values and the entropy-weighted combination.
//
// Reference semantics: Uni_Adapter.py:78-118 (update_cache), :120-156 (compute_cache_logits), :541-574 (loop
// body); utils/math_utils.py:21-45 (conjugate_gradient), :48-95 (online_value_refinement_new).
// The cache holds at most K * shot_capacity (40 * 30 = 1200) prototypes, so every kernel here is
// latency-bound; the two dense products (keys @ keys^T, A @ p inside CG) go through ua_gemm.
// Bookkeeping that depends on the predicted class (which slot, how many prototypes) stays on the
// host exactly as in the reference, which reads `pred` to the host every sample.
#include "ua_common.h"

namespace {

using ua::wave_sum;
using ua::block_sum;

__device__ __forceinline__ float block_max_f32(float v, float* scratch) {
    v = ua::wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, scratch[i]);
    return t;
}

// One workgroup (256).  A full class (n_slots prototypes of one class, contiguous): merge the sample into the most similar
// prototype (:93-116).  keys (n,D), conf (n), count (n) int, prob (n,K) of that class; f (D) unit feature; conf_new scalar.
__global__ void __launch_bounds__(256)
cache_merge_kernel(float* __restrict__ keys, float* __restrict__ conf, int* __restrict__ count, float* __restrict__ prob,
                   const float* __restrict__ f, const float* __restrict__ conf_new_p, const float* __restrict__ text,
                   int n, int K, int D, float beta, float log2K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];    // nf (D) | z (K) | sims (n)
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ int s_best;
    float* nf = smem; float* z = smem + D; float* sims = z + K;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    for (int s = wv; s < n; s += 4) {                                // sim = f @ feat_cluster^T (:97)
        double acc = 0.0;
        for (int d = lane; d < D; d += 64) acc += (double)(f[d] * keys[(size_t)s * D + d]);
        acc = wave_sum(acc);
        if (lane == 0) sims[s] = (float)acc;
    }
    __syncthreads();
    if (tid == 0) { int b = 0; for (int s = 1; s < n; ++s) if (sims[s] > sims[b]) b = s; s_best = b; }   // torch.argmax: first maximum
    __syncthreads();
    const int bi = s_best;
    const float cc = conf[bi], cn = *conf_new_p;
    const float nc = (float)count[bi];
    const float wc = cc * nc, den = nc * cc + cn;                    // :104
    double ss = 0.0;
    for (int d = tid; d < D; d += 256) {
        const float v = (wc * keys[(size_t)bi * D + d] + cn * f[d]) / den;
        nf[d] = v; ss += (double)(v * v);
    }
    const float nrm = (float)sqrt(block_sum(ss, sd));
    for (int d = tid; d < D; d += 256) { const float v = nf[d] / nrm; nf[d] = v; keys[(size_t)bi * D + d] = v; }   // :105
    __syncthreads();
    for (int k = wv; k < K; k += 4) {                                // 100 * new_feat @ clip_weights (:108)
        double acc = 0.0;
        for (int d = lane; d < D; d += 64) acc += (double)((100.f * nf[d]) * text[(size_t)k * D + d]);
        acc = wave_sum(acc);
        if (lane == 0) z[k] = (float)acc;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = tid; k < K; k += 256) mx = fmaxf(mx, z[k]);
    mx = block_max_f32(mx, sf);
    double se = 0.0;
    for (int k = tid; k < K; k += 256) se += (double)expf(z[k] - mx);
    const float sum = (float)block_sum(se, sd);
    double h = 0.0;
    for (int k = tid; k < K; k += 256) {
        const float p = expf(z[k] - mx) / sum;
        prob[(size_t)bi * K + k] = p;                                // new_prob (:108)
        h += (double)(p * logf(p + 1e-10f));
    }
    const float ent = -(float)block_sum(h, sd);                      // :109
    if (tid == 0) { conf[bi] = expf(-beta * (ent / log2K)); count[bi] = count[bi] + 1; }   // :110-114
}

// W[W < thr] = 0 (in place); s[i] = 1 / (sqrt(sum_j W[i][j]) + 1e-8).  grid P, block 256.
__global__ void __launch_bounds__(256)
lap_rows_kernel(float* __restrict__ W, int P, int ld, float thr, float* __restrict__ s) {
    __shared__ double sd[16];
    const int i = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < P; j += 256) {
        float w = W[(size_t)i * ld + j];
        if (w < thr) { w = 0.f; W[(size_t)i * ld + j] = 0.f; }
        acc += (double)w;
    }
    const float deg = (float)block_sum(acc, sd);
    if (threadIdx.x == 0) s[i] = 1.0f / (sqrtf(deg) + 1e-8f);
}

// L_reg = (I - S W S) + 2 lambda I  (math_utils.py:55-58), in place over W.
__global__ void __launch_bounds__(256)
lap_build_kernel(float* __restrict__ W, const float* __restrict__ s, int P, int ld, float two_lambda) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)P * P) return;
    const int i = (int)(e / P), j = (int)(e % P);
    const size_t o = (size_t)i * ld + j;
    const float x = (s[i] * W[o]) * s[j];
    const float eye = i == j ? 1.f : 0.f;
    W[o] = (eye - x) + two_lambda * eye;
}

// One CG iteration after Ap = A @ p (all vectors stored transposed: row k = right-hand side k, P entries).  grid K.
__global__ void __launch_bounds__(256)
cg_step_kernel(float* __restrict__ x, float* __restrict__ r, float* __restrict__ p, const float* __restrict__ Ap,
               float* __restrict__ rz_old, int P, int ld, float tol, int* __restrict__ not_done) {
    __shared__ double sd[16];
    const int k = blockIdx.x;
    const size_t o = (size_t)k * ld;
    double acc = 0.0;
    for (int j = threadIdx.x; j < P; j += 256) acc += (double)(p[o + j] * Ap[o + j]);
    const float pAp = (float)block_sum(acc, sd);
    const float rzo = rz_old[k];
    const float alpha = rzo / (pAp + 1e-8f);                          // :33
    double rr = 0.0;
    for (int j = threadIdx.x; j < P; j += 256) {
        x[o + j] = x[o + j] + alpha * p[o + j];                      // :35
        const float rn = r[o + j] - alpha * Ap[o + j];               // :36
        r[o + j] = rn;
        rr += (double)(rn * rn);
    }
    const float rzn = (float)block_sum(rr, sd);                      // :38
    if (threadIdx.x == 0 && !(rzn < tol)) atomicOr(not_done, 1);     // :39
    const float beta = rzn / (rzo + 1e-8f);                          // :41
    for (int j = threadIdx.x; j < P; j += 256) p[o + j] = r[o + j] + beta * p[o + j];   // :43
    __syncthreads();
    if (threadIdx.x == 0) rz_old[k] = rzn;
}

// Per prototype: class = argmax_k (x[k][p] / sum_k x[k][p]) (:148) and affinity = f . key_p (:154).  grid P, block 64.
__global__ void __launch_bounds__(64)
key_assign_kernel(const float* __restrict__ xT, int ld, const float* __restrict__ keys, const float* __restrict__ f, int P, int K, int D,
                  int* __restrict__ cls, float* __restrict__ aff) {
    const int pidx = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) acc += (double)(f[d] * keys[(size_t)pidx * D + d]);
    acc = wave_sum(acc);
    if (lane == 0) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += xT[(size_t)k * ld + pidx];
        int b = 0; float bv = xT[pidx] / s;
        for (int k = 1; k < K; ++k) { const float v = xT[(size_t)k * ld + pidx] / s; if (v > bv) { bv = v; b = k; } }
        cls[pidx] = b; aff[pidx] = (float)acc;
    }
}

// cache_logits[k] = sum_{p: cls_p = k} aff_p / (count_k + 1e-6); final = prob1 / ent1 + prob2 / ent2 (:552-574).  One workgroup (256).
__global__ void __launch_bounds__(256)
cluster_fuse_kernel(const int* __restrict__ cls, const float* __restrict__ aff, int P, const float* __restrict__ clip_logits, int K,
                    float* __restrict__ cache_logits, float* __restrict__ final_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];    // z2 (K) | cnt (K) | p1 (K) | p2 (K)
    __shared__ double sd[16];
    __shared__ float sf[16];
    float* z2 = smem; float* cnt = smem + K; float* p1 = cnt + K; float* p2 = p1 + K;
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) {                             // few prototypes: serial per class, prototype order
        float c = 0.f, a = 0.f;
        for (int q = 0; q < P; ++q) if (cls[q] == k) c += 1.f;
        const float val = 1.f / (c + 1e-6f);                         // one_hot / (counts + 1e-6)  :149-151
        for (int q = 0; q < P; ++q) if (cls[q] == k) a += aff[q] * val;
        z2[k] = a; cnt[k] = c;
        if (cache_logits) cache_logits[k] = a;
    }
    __syncthreads();
    auto softmax_ent = [&](const float* z, float scale, float* p) {  // softmax(z*scale), entropy of it
        float mx = -INFINITY;
        for (int k = tid; k < K; k += 256) mx = fmaxf(mx, z[k] * scale);
        mx = block_max_f32(mx, sf);
        double se = 0.0;
        for (int k = tid; k < K; k += 256) se += (double)expf(z[k] * scale - mx);
        const float sum = (float)block_sum(se, sd);
        double h = 0.0;
        for (int k = tid; k < K; k += 256) { const float v = expf(z[k] * scale - mx) / sum; p[k] = v; h += (double)(v * logf(v + 1e-10f)); }
        return -(float)block_sum(h, sd);
    };
    const float e1 = softmax_ent(clip_logits, 0.01f, p1);            // clip_logits / 100 (:552-554)
    __syncthreads();
    const float e2 = softmax_ent(z2, 1.0f, p2);                      // :570-572
    __syncthreads();
    for (int k = tid; k < K; k += 256) final_out[k] = (1.f / e1) * p1[k] + (1.f / e2) * p2[k];   // :572
}

}  // namespace

extern "C" int ua_cluster_merge(float* keys, float* conf, int32_t* count, float* prob, const float* f, const float* conf_new,
                                const float* text, int32_t n_slots, int32_t K, int32_t D, float beta, void* stream) {
    UA_CHECK_PTR(keys); UA_CHECK_PTR(conf); UA_CHECK_PTR(count); UA_CHECK_PTR(prob); UA_CHECK_PTR(f); UA_CHECK_PTR(conf_new); UA_CHECK_PTR(text);
    if (n_slots <= 0 || K <= 1 || D <= 0) return UA_EINVAL;
    const size_t lds = (size_t)(D + K + n_slots) * sizeof(float);
    if (lds > 60 * 1024) return UA_EUNSUPPORTED;
    hipLaunchKernelGGL(cache_merge_kernel, dim3(1), dim3(256), lds, ua::as_stream(stream), keys, conf, count, prob, f, conf_new, text,
                       n_slots, K, D, beta, log2f((float)K));
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_cluster_laplacian(float* W, int32_t P, int32_t ld, float threshold, float lambda_reg, float* scratch_s, void* stream) {
    UA_CHECK_PTR(W); UA_CHECK_PTR(scratch_s);
    if (P <= 0 || ld < P) return UA_EINVAL;
    hipStream_t st = ua::as_stream(stream);
    hipLaunchKernelGGL(lap_rows_kernel, dim3(P), dim3(256), 0, st, W, P, ld, threshold, scratch_s);
    UA_LAUNCH_CHECK();
    hipLaunchKernelGGL(lap_build_kernel, dim3((unsigned)(((size_t)P * P + 255) / 256)), dim3(256), 0, st, W, scratch_s, P, ld, 2.0f * lambda_reg);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_cluster_cg_step(float* xT, float* rT, float* pT, const float* ApT, float* rz_old, int32_t K, int32_t P, int32_t ld, float tol,
                                  int32_t* not_done, void* stream) {
    UA_CHECK_PTR(xT); UA_CHECK_PTR(rT); UA_CHECK_PTR(pT); UA_CHECK_PTR(ApT); UA_CHECK_PTR(rz_old); UA_CHECK_PTR(not_done);
    if (K <= 0 || P <= 0 || ld < P) return UA_EINVAL;
    hipLaunchKernelGGL(cg_step_kernel, dim3(K), dim3(256), 0, ua::as_stream(stream), xT, rT, pT, ApT, rz_old, P, ld, tol, not_done);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_cluster_logits(const float* xT, int32_t ld, const float* keys, const float* f, const float* clip_logits, int32_t P, int32_t K, int32_t D,
                                 int32_t* cls_scratch, float* aff_scratch, float* cache_logits, float* final_logits, void* stream) {
    UA_CHECK_PTR(xT); UA_CHECK_PTR(keys); UA_CHECK_PTR(f); UA_CHECK_PTR(clip_logits); UA_CHECK_PTR(cls_scratch); UA_CHECK_PTR(aff_scratch);
    UA_CHECK_PTR(final_logits);
    if (P <= 0 || K <= 1 || D <= 0 || ld < P) return UA_EINVAL;
    if ((size_t)4 * K * sizeof(float) > 60 * 1024) return UA_EUNSUPPORTED;
    hipStream_t st = ua::as_stream(stream);
    hipLaunchKernelGGL(key_assign_kernel, dim3(P), dim3(64), 0, st, xT, ld, keys, f, P, K, D, cls_scratch, aff_scratch);
    UA_LAUNCH_CHECK();
    hipLaunchKernelGGL(cluster_fuse_kernel, dim3(1), dim3(256), (size_t)4 * K * sizeof(float), st, cls_scratch, aff_scratch, P, clip_logits, K,
                       cache_logits, final_logits);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
