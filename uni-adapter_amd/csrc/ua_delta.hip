// "Prototype deltas" multi-GPU mode (BASELINE.json north_star): every rank replays its own
// shard of a step from the common state S0, packs the change of the sufficient statistics
//     dC = c - c0,   dS1 = c*mu - c0*mu0,   dS2 = c*var - c0*var0          (per class k, mode m[, feature d])
// ONE all_gather moves the packed buffers, and every rank merges all W of them in rank order:
//     c = c0 + sum dC,  mu = (c0*mu0 + sum dS1)/c,  var = max((c0*var0 + sum dS2)/c, 1e-8),  pi = c / sum_m c.
// The reference has no counterpart (its loop is single-process, Uni_Adapter.py:368-579); the statistics are those
// its M-step keeps (dota_mixture.py:196-230).  Products and sums are formed in fp64 so the merge of W = 1 reproduces the
// sequential state to one fp32 rounding; the exchanged values are fp32.  HBM-bound streaming kernels: float4 per lane,
// pack reads 4 and writes 2 state-sized arrays, merge reads 2 + 2W and writes 2.
#include "ua_common.h"

namespace {

using ua::kWave;

__global__ void __launch_bounds__(256)
delta_pack_kernel(const float4* __restrict__ mu, const float4* __restrict__ var, const float* __restrict__ c,
                  const float4* __restrict__ mu0, const float4* __restrict__ var0, const float* __restrict__ c0,
                  int KM, int D4, float4* __restrict__ dS1, float4* __restrict__ dS2, float* __restrict__ dC) {
    const size_t total = (size_t)KM * D4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / D4);
        const double cn = c[row], co = c0[row];
        const float4 a = mu[i], b = mu0[i], p = var[i], q = var0[i];
        float4 o1, o2;
        o1.x = (float)(cn * a.x - co * b.x); o1.y = (float)(cn * a.y - co * b.y);
        o1.z = (float)(cn * a.z - co * b.z); o1.w = (float)(cn * a.w - co * b.w);
        o2.x = (float)(cn * p.x - co * q.x); o2.y = (float)(cn * p.y - co * q.y);
        o2.z = (float)(cn * p.z - co * q.z); o2.w = (float)(cn * p.w - co * q.w);
        dS1[i] = o1; dS2[i] = o2;
        if (i % D4 == 0) dC[row] = (float)(cn - co);
    }
}

// grid K, block M*64: wave m owns row (k,m)
__global__ void __launch_bounds__(1024)
delta_merge_kernel(float4* __restrict__ mu, float4* __restrict__ var, float* __restrict__ pi, float* __restrict__ c,
                   const float4* __restrict__ mu0, const float4* __restrict__ var0, const float* __restrict__ c0,
                   const float* __restrict__ deltas, int W, size_t stride, int K, int M, int D4) {
    __shared__ float s_c[16];
    const int k = blockIdx.x, m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = k * M + m;
    const size_t nS = (size_t)K * M * D4 * 4;
    const double co = c0[row];
    double cs = co;
    for (int r = 0; r < W; ++r) cs += (double)deltas[r * stride + 2 * nS + row];
    const float cn = (float)cs;
    const double den = (double)cn + 1e-10;
    for (int v = lane; v < D4; v += kWave) {
        const size_t i = (size_t)row * D4 + v;
        const float4 b = mu0[i], q = var0[i];
        double s1x = co * b.x, s1y = co * b.y, s1z = co * b.z, s1w = co * b.w;
        double s2x = co * q.x, s2y = co * q.y, s2z = co * q.z, s2w = co * q.w;
        for (int r = 0; r < W; ++r) {
            const float4 d1 = reinterpret_cast<const float4*>(deltas + r * stride)[i];
            const float4 d2 = reinterpret_cast<const float4*>(deltas + r * stride + nS)[i];
            s1x += d1.x; s1y += d1.y; s1z += d1.z; s1w += d1.w;
            s2x += d2.x; s2y += d2.y; s2z += d2.z; s2w += d2.w;
        }
        float4 o1, o2;
        o1.x = (float)(s1x / den); o1.y = (float)(s1y / den); o1.z = (float)(s1z / den); o1.w = (float)(s1w / den);
        o2.x = fmaxf((float)(s2x / den), 1e-8f); o2.y = fmaxf((float)(s2y / den), 1e-8f);
        o2.z = fmaxf((float)(s2z / den), 1e-8f); o2.w = fmaxf((float)(s2w / den), 1e-8f);
        mu[i] = o1; var[i] = o2;
    }
    if (lane == 0) s_c[m] = cn;
    __syncthreads();
    if (lane == 0) {
        float Ck = 0.f;
        for (int i = 0; i < M; ++i) Ck += s_c[i];
        c[row] = cn;
        pi[row] = cn / (Ck + 1e-10f);                       // dota_mixture.py:229-230
    }
}

__global__ void __launch_bounds__(256)
vec_delta_kernel(const float* __restrict__ cur, const float* __restrict__ base, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = cur[i] - base[i];
}

__global__ void __launch_bounds__(256)
vec_merge_kernel(const float* __restrict__ base, const float* __restrict__ deltas, int W, size_t stride, size_t n, double scale,
                 float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < W; ++r) s += (double)deltas[r * stride + i];
        out[i] = (float)((double)base[i] + scale * s);
    }
}

inline int grid_for(size_t items) {
    const size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int ua_gmm_delta_pack(const float* mu, const float* var, const float* c, const float* mu0, const float* var0,
                                 const float* c0, int32_t K, int32_t M, int32_t D, float* out, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(c); UA_CHECK_PTR(mu0); UA_CHECK_PTR(var0); UA_CHECK_PTR(c0); UA_CHECK_PTR(out);
    if (K <= 0 || M <= 0 || D <= 0 || (D & 3)) return UA_EINVAL;
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(mu0); UA_CHECK_ALIGN16(var0); UA_CHECK_ALIGN16(out);
    const size_t nS = (size_t)K * M * D;
    delta_pack_kernel<<<grid_for(nS / 4), 256, 0, ua::as_stream(stream)>>>(
        reinterpret_cast<const float4*>(mu), reinterpret_cast<const float4*>(var), c, reinterpret_cast<const float4*>(mu0),
        reinterpret_cast<const float4*>(var0), c0, K * M, D / 4, reinterpret_cast<float4*>(out), reinterpret_cast<float4*>(out + nS),
        out + 2 * nS);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_gmm_delta_merge(float* mu, float* var, float* pi, float* c, const float* mu0, const float* var0, const float* c0,
                                  const float* deltas, int32_t W, int64_t stride, int32_t K, int32_t M, int32_t D, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(c); UA_CHECK_PTR(mu0); UA_CHECK_PTR(var0); UA_CHECK_PTR(c0);
    UA_CHECK_PTR(deltas);
    if (K <= 0 || M <= 0 || M > 16 || D <= 0 || (D & 3) || W <= 0) return UA_EINVAL;
    const int64_t nS = (int64_t)K * M * D;
    if (stride < 2 * nS + (int64_t)K * M || (stride & 3)) return UA_EINVAL;
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(mu0); UA_CHECK_ALIGN16(var0); UA_CHECK_ALIGN16(deltas);
    delta_merge_kernel<<<K, M * kWave, 0, ua::as_stream(stream)>>>(
        reinterpret_cast<float4*>(mu), reinterpret_cast<float4*>(var), pi, c, reinterpret_cast<const float4*>(mu0),
        reinterpret_cast<const float4*>(var0), c0, deltas, W, (size_t)stride, K, M, D / 4);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_vec_delta(const float* cur, const float* base, int64_t n, float* out, void* stream) {
    UA_CHECK_PTR(cur); UA_CHECK_PTR(base); UA_CHECK_PTR(out);
    if (n <= 0) return UA_EINVAL;
    vec_delta_kernel<<<grid_for((size_t)n), 256, 0, ua::as_stream(stream)>>>(cur, base, (size_t)n, out);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_vec_merge(const float* base, const float* deltas, int32_t W, int64_t stride, int64_t n, float scale, float* out,
                            void* stream) {
    UA_CHECK_PTR(base); UA_CHECK_PTR(deltas); UA_CHECK_PTR(out);
    if (n <= 0 || W <= 0 || stride < n) return UA_EINVAL;
    vec_merge_kernel<<<grid_for((size_t)n), 256, 0, ua::as_stream(stream)>>>(base, deltas, W, (size_t)stride, (size_t)n, (double)scale, out);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
