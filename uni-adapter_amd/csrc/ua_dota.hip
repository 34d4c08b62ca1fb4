// Full-covariance DOTA: the reference's `dota.py` cache - per-class (D,D)
// covariance updated by weighted rank-1 terms, one shared precision matrix Lambda = inv((1-eps) mean_k Sigma_k
// + eps I) per step, scores  X Lambda M - 1/2 diag(M^T Lambda M)  in fp16 arithmetic.
//
// Reference semantics: dota.py:42-66 (fit), :69-72 (update), :75-86 (predict).
//   * dota_fit_kernel      - HBM streaming: every Sigma_k element is read once and written once, the mean over
//                            classes and the regularised matrix A are produced in the same pass
//                            (2*K*D*D*4 bytes per sample; K=40, D=512: 84 MB).
//   * dota_mean_kernel     - class means / counts (uses the OLD mean for the covariance, so it runs second).
//   * spd_inverse_small    - <= 64 x 64 block inverse (Gauss-Jordan in LDS, fp64): the base case of the blocked
//                            Schur-complement inversion that the host drives with ua_gemm (uni-adapter_amd/dota.py).
//   * dota_w_kernel / dota_score_kernel - W = Lambda_h M_h, c = 1/2 sum M_h o W, scores = X_h W - c with the
//                            reference's rounding points (every materialised tensor is fp16, sums accumulate in fp32).
#include "ua_common.h"
#include <hip/hip_fp16.h>

namespace {

using ua::wave_sum;

__device__ __forceinline__ float h16(float x) { return __half2float(__float2half_rn(x)); }

// grid (D/4/64 , D): thread = row i, 4 consecutive columns j.  B <= 64.
__global__ void __launch_bounds__(64)
dota_fit_kernel(float* __restrict__ Sigma, const float* __restrict__ mu, const float* __restrict__ c,
                const float* __restrict__ x, const float* __restrict__ y, int B, int K, int D,
                float one_minus_eps, float eps, float* __restrict__ overall, float* __restrict__ A) {
    const int i = blockIdx.y, j = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (j >= D) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
        float sw = 0.f;
        float4 delta = make_float4(0.f, 0.f, 0.f, 0.f);
        const float mi = mu[(size_t)k * D + i];
        const float4 mj = *reinterpret_cast<const float4*>(mu + (size_t)k * D + j);
        for (int b = 0; b < B; ++b) {                                        // einsum('bji,bjk->jik') :57-59
            const float yk = y[(size_t)b * K + k];
            sw += yk;
            const float wi = yk * (x[(size_t)b * D + i] - mi);               // weighted_x_minus_mu[b,k,i]
            const float4 xj = *reinterpret_cast<const float4*>(x + (size_t)b * D + j);
            delta.x += wi * (xj.x - mj.x); delta.y += wi * (xj.y - mj.y);
            delta.z += wi * (xj.z - mj.z); delta.w += wi * (xj.w - mj.w);
        }
        const float ck = c[k], den = ck + sw;
        float4* ps = reinterpret_cast<float4*>(Sigma + ((size_t)k * D + i) * D + j);
        float4 s = *ps;
        s.x = (ck * s.x + delta.x) / den; s.y = (ck * s.y + delta.y) / den;   // :60
        s.z = (ck * s.z + delta.z) / den; s.w = (ck * s.w + delta.w) / den;
        *ps = s;
        acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;             // torch.mean(dim=0) :63
    }
    const float fk = (float)K;
    acc.x /= fk; acc.y /= fk; acc.z /= fk; acc.w /= fk;
    *reinterpret_cast<float4*>(overall + (size_t)i * D + j) = acc;
    float4 a;                                                                // (1-eps) * overall + eps * eye  :70-71
    a.x = one_minus_eps * acc.x + (i == j ? eps : 0.f);     a.y = one_minus_eps * acc.y + (i == j + 1 ? eps : 0.f);
    a.z = one_minus_eps * acc.z + (i == j + 2 ? eps : 0.f); a.w = one_minus_eps * acc.w + (i == j + 3 ? eps : 0.f);
    *reinterpret_cast<float4*>(A + (size_t)i * D + j) = a;
}

// grid K, block 256: new_mu = (y^T x + c mu) / (sum_w + c), c += sum_w   (:51-54, :64-65)
__global__ void __launch_bounds__(256)
dota_mean_kernel(float* __restrict__ mu, float* __restrict__ c, const float* __restrict__ x, const float* __restrict__ y,
                 int B, int K, int D) {
    const int k = blockIdx.x;
    float sw = 0.f;
    for (int b = 0; b < B; ++b) sw += y[(size_t)b * K + k];
    const float ck = c[k];
    for (int d = threadIdx.x; d < D; d += 256) {
        float wx = 0.f;
        for (int b = 0; b < B; ++b) wx += y[(size_t)b * K + k] * x[(size_t)b * D + d];
        mu[(size_t)k * D + d] = (wx + ck * mu[(size_t)k * D + d]) / (sw + ck);
    }
    __syncthreads();
    if (threadIdx.x == 0) c[k] = ck + sw;
}

// One workgroup: out = inverse of the n x n (n <= 64) block at A (leading dimension lda), in-place Gauss-Jordan
// without pivoting in fp64 (the blocks are principal blocks / Schur complements of an SPD matrix).
__global__ void __launch_bounds__(256)
spd_inverse_small_kernel(const float* __restrict__ A, long long lda, int n, float* __restrict__ out, long long ldo) {
    __shared__ double a[64][65];
    __shared__ double colp[64];
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 256) a[e / n][e % n] = (double)A[(size_t)(e / n) * lda + (e % n)];
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        if (tid < n) colp[tid] = a[tid][p];
        __syncthreads();
        const double d = 1.0 / colp[p];
        if (tid < n) a[p][tid] = (tid == p ? 1.0 : a[p][tid]) * d;          // scaled pivot row (the pivot itself becomes 1/pivot)
        __syncthreads();
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, j = e % n;
            if (i != p) a[i][j] = (j == p ? 0.0 : a[i][j]) - colp[i] * a[p][j];
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) out[(size_t)(e / n) * ldo + (e % n)] = (float)a[e / n][e % n];
}

// W[d][k] = half( sum_e Lambda_h[d][e] * half(mu[k][e]) ), grid D, block 256 (4 waves, one class per wave at a time)
__global__ void __launch_bounds__(256)
dota_w_kernel(const __half* __restrict__ Lh, const float* __restrict__ mu, int K, int D, __half* __restrict__ W) {
    extern __shared__ __attribute__((aligned(16))) float s_l[];               // Lambda row d
    const int d = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < D; e += 256) s_l[e] = __half2float(Lh[(size_t)d * D + e]);
    __syncthreads();
    for (int k = wv; k < K; k += 4) {
        float acc = 0.f;
        for (int e = lane; e < D; e += 64) acc += s_l[e] * h16(mu[(size_t)k * D + e]);
        acc = wave_sum(acc);
        if (lane == 0) W[(size_t)d * K + k] = __float2half_rn(acc);
    }
}

// c[k] = half(0.5 * half(sum_d half(M[d][k] * W[d][k])));  scores[b][k] = half( half(sum_d X_h[b][d] W[d][k]) - c[k] )
// grid K, block 256.
__global__ void __launch_bounds__(256)
dota_score_kernel(const __half* __restrict__ W, const float* __restrict__ mu, const __half* __restrict__ Xh, int B, int K, int D,
                  float* __restrict__ scores) {
    __shared__ float sf[4];
    const int k = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) acc += h16(h16(mu[(size_t)k * D + d]) * __half2float(W[(size_t)d * K + k]));
    acc = wave_sum(acc);
    if (lane == 0) sf[wv] = acc;
    __syncthreads();
    const float ck = h16(0.5f * h16((sf[0] + sf[1]) + (sf[2] + sf[3])));
    for (int b = wv; b < B; b += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += __half2float(Xh[(size_t)b * D + d]) * __half2float(W[(size_t)d * K + k]);
        s = wave_sum(s);
        if (lane == 0) scores[(size_t)b * K + k] = h16(h16(s) - ck);
    }
}

__global__ void __launch_bounds__(256)
to_half_kernel(const float* __restrict__ x, __half* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = __float2half_rn(x[i]);
}

}  // namespace

extern "C" int ua_dota_fit(float* Sigma, float* mu, float* c, const float* x, const float* y,
                           int32_t B, int32_t K, int32_t D, float eps, float* overall_Sigma, float* A, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(c); UA_CHECK_PTR(x); UA_CHECK_PTR(y);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(x);
    if (B <= 0 || K <= 0 || D <= 0) return UA_EINVAL;
    if (D % 4 != 0 || B > 64) return UA_EUNSUPPORTED;
    hipStream_t st = ua::as_stream(stream);
    if (Sigma) {      // Sigma == NULL: streaming_update_Sigma = False (dota.py:54) - covariances, their mean and A stay as they are
        UA_CHECK_PTR(overall_Sigma); UA_CHECK_PTR(A);
        UA_CHECK_ALIGN16(Sigma); UA_CHECK_ALIGN16(overall_Sigma); UA_CHECK_ALIGN16(A);
        hipLaunchKernelGGL(dota_fit_kernel, dim3((D / 4 + 63) / 64, D), dim3(64), 0, st, Sigma, mu, c, x, y, B, K, D,
                           (float)(1.0 - (double)eps), eps, overall_Sigma, A);
        UA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(dota_mean_kernel, dim3(K), dim3(256), 0, st, mu, c, x, y, B, K, D);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_spd_inverse_small(const float* A, int64_t lda, int32_t n, float* out, int64_t ldo, void* stream) {
    UA_CHECK_PTR(A); UA_CHECK_PTR(out);
    if (n <= 0 || lda < n || ldo < n) return UA_EINVAL;
    if (n > 64) return UA_EUNSUPPORTED;
    hipLaunchKernelGGL(spd_inverse_small_kernel, dim3(1), dim3(256), 0, ua::as_stream(stream), A, (long long)lda, n, out, (long long)ldo);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_to_half(const float* x, void* y_half, int64_t n, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(y_half);
    if (n <= 0) return UA_EINVAL;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ua::as_stream(stream), x,
                       reinterpret_cast<__half*>(y_half), (size_t)n);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_dota_predict(const void* Lambda_half, const float* mu, const void* X_half, int32_t B, int32_t K, int32_t D,
                               void* W_half, float* scores, void* stream) {
    UA_CHECK_PTR(Lambda_half); UA_CHECK_PTR(mu); UA_CHECK_PTR(X_half); UA_CHECK_PTR(W_half); UA_CHECK_PTR(scores);
    if (B <= 0 || K <= 0 || D <= 0) return UA_EINVAL;
    if ((size_t)D * sizeof(float) > 60 * 1024) return UA_EUNSUPPORTED;
    hipStream_t st = ua::as_stream(stream);
    hipLaunchKernelGGL(dota_w_kernel, dim3(D), dim3(256), (size_t)D * sizeof(float), st, reinterpret_cast<const __half*>(Lambda_half), mu, K, D,
                       reinterpret_cast<__half*>(W_half));
    UA_LAUNCH_CHECK();
    hipLaunchKernelGGL(dota_score_kernel, dim3(K), dim3(256), 0, st, reinterpret_cast<const __half*>(W_half), mu,
                       reinterpret_cast<const __half*>(X_half), B, K, D, scores);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
