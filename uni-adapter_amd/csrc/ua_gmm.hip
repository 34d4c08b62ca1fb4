// MODE-DOTA cache kernels for gfx950 (MI355X): per-class M-mode diagonal Gaussian mixture.
//
// Reference semantics: dota_mixture.py:117-267 (DOTA_mix._get_var/_log_likelihood/fit/predict).
//
// Layout / mapping.  State mu,var are (K,M,D) fp32 with D fastest.  One workgroup owns one class
// k; wave m of the workgroup owns mode row (k,m) and keeps the WHOLE row of mu and var in
// registers (D = 256*VEC floats -> VEC float4 per lane per array; lane l holds float4 #(v*64+l),
// so every wave-instruction moves 1 KiB of contiguous HBM).  Everything the reference does per
// sample to that row (predict, fit, fit) is then done from registers, and the row is written
// back once: 4*K*M*D*4 bytes of HBM traffic per sample instead of the reference's 10 passes.
// Cross-lane reductions (sum over D) are wavefront butterflies; the sum over modes (logsumexp,
// softmax responsibilities, sum_m c) goes through a few floats of LDS per class.
//
// Numerics.  Per-element arithmetic is fp32 in the reference's operation order (this file is
// compiled with -ffp-contract=off so a*b+c is not fused, matching the separate torch ops).
// The D-long sums are accumulated in fp64 and rounded to fp32 once, which is at least as
// accurate as torch's fp32 pairwise sums (|sum| ~ 9e3, fp32 ulp there ~ 1e-3).
#include "ua_common.h"
#include "ua_align_common.h"
#include <cmath>
#include <cstdlib>
#include <hip/hip_fp16.h>

namespace {

using ua::wave_sum;

constexpr int kMaxModes = 16;   // waves per workgroup
constexpr int kMaxFitB  = 64;   // samples per fit call
constexpr int kQChunk   = 32;   // queries per workgroup in the batched log-lik kernel

template <int VEC>
struct Row {
    float4 mu[VEC];
    float4 var[VEC];
};

template <int VEC>
__device__ __forceinline__ void load_row(Row<VEC>& r, const float* mu, const float* var, size_t off, int lane) {
    const float4* pm = reinterpret_cast<const float4*>(mu + off);
    const float4* pv = reinterpret_cast<const float4*>(var + off);
#pragma unroll
    for (int v = 0; v < VEC; ++v) { r.mu[v] = pm[v * 64 + lane]; r.var[v] = pv[v * 64 + lane]; }
}

template <int VEC>
__device__ __forceinline__ void store_row(const Row<VEC>& r, float* mu, float* var, size_t off, int lane) {
    float4* pm = reinterpret_cast<float4*>(mu + off);
    float4* pv = reinterpret_cast<float4*>(var + off);
#pragma unroll
    for (int v = 0; v < VEC; ++v) { pm[v * 64 + lane] = r.mu[v]; pv[v * 64 + lane] = r.var[v]; }
}

__device__ __forceinline__ float reg_var(float v, float eps) { return fmaxf(v + eps, 1e-8f); }   // :117-119

// a / b for the streaming kernels: the full IEEE sequence (v_div_scale x2, v_rcp, 5 x v_fma, v_div_fmas, v_div_fixup) was
// 55 % of gmm_step_kernel's VALU instructions and the kernel is VALU-bound.  Here: hardware reciprocal (1 ulp) and one
// residual correction, q = a*r;  q += (a - q*b)*r  - error 0.5 ulp + O(2^-46), i.e. the correctly rounded quotient except
// in rare double-rounding cases.  Divisors on this path are normal numbers in [1e-8, 1e6] (regularised variances, counts).
__device__ __forceinline__ float div_r(float a, float b, float r) {
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);
}
__device__ __forceinline__ float div_nr(float a, float b) { return div_r(a, b, __builtin_amdgcn_rcpf(b)); }

// logf for a NORMAL, finite, positive argument (regularised variances are >= 1e-8): the compiler's own expansion of logf -
// v_log_f32, then ln2 applied as a (hi, lo) pair, r = y*c; r += fma(y, cc, fma(y, c, -r)) - without its denormal pre-scaling
// (compare, select, ldexp, select, subtract) and infinity select, 5 instead of 12 instructions, same bits for such arguments.
__device__ __forceinline__ float log_normal(float x) {
    const float y = __builtin_amdgcn_logf(x);
    const float c = __uint_as_float(0x3f317217u), cc = __uint_as_float(0x3377d1cfu);
    const float r = y * c;
    return r + __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
}

// sum_d ln(max(var+eps,1e-8)) over the wave's row (all lanes get the total)
// UA_GMM_SUM4: the four terms of a float4 are added in fp32, (x + z) + (y + w), before they join the fp64 sum: 2 instead of 8
// instructions per float4 and accumulator.  The partial sums are ~40 in magnitude (error <= 4e-6 each, unbiased), 256 of them per
// row: ~5e-5 on sums that are rounded to fp32 at an ulp of 2.4e-4 .. 1e-3 afterwards.  All helpers use the same association, so the
// kernels stay bit-identical to one another.
#ifndef UA_GMM_SUM4
#define UA_GMM_SUM4 1          // -DUA_GMM_SUM4=0: every term joins the fp64 sum on its own (the round-1/2 arithmetic; A/B aid)
#endif
#if UA_GMM_SUM4
#define UA_ACC4(acc, a, b, c, d) acc += (double)(((a) + (c)) + ((b) + (d)))
#else
#define UA_ACC4(acc, a, b, c, d) do { acc += (double)(a); acc += (double)(b); acc += (double)(c); acc += (double)(d); } while (0)
#endif

template <int VEC>
__device__ __forceinline__ double row_logdet(const Row<VEC>& r, float eps) {
    double acc = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        UA_ACC4(acc, log_normal(reg_var(r.var[v].x, eps)), log_normal(reg_var(r.var[v].y, eps)),
                log_normal(reg_var(r.var[v].z, eps)), log_normal(reg_var(r.var[v].w, eps)));
    return wave_sum(acc);
}

__device__ __forceinline__ float round_fp16(float x) { return __half2float(__float2half_rn(x)); }

// sum_d (x-mu)^2 / v over the wave's row
template <int VEC, bool ROUND16>
__device__ __forceinline__ double row_maha(const Row<VEC>& r, const float* x, float eps, int lane) {
    const float4* px = reinterpret_cast<const float4*>(x);
    double acc = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float4 xv = px[v * 64 + lane];
        if (ROUND16) { xv.x = round_fp16(xv.x); xv.y = round_fp16(xv.y); xv.z = round_fp16(xv.z); xv.w = round_fp16(xv.w); }
        const float dx = xv.x - r.mu[v].x, dy = xv.y - r.mu[v].y, dz = xv.z - r.mu[v].z, dw = xv.w - r.mu[v].w;
        UA_ACC4(acc, div_nr(dx * dx, reg_var(r.var[v].x, eps)), div_nr(dy * dy, reg_var(r.var[v].y, eps)),
                div_nr(dz * dz, reg_var(r.var[v].z, eps)), div_nr(dw * dw, reg_var(r.var[v].w, eps)));
    }
    return wave_sum(acc);
}

// torch.logsumexp over M values held in LDS
__device__ __forceinline__ float lse_modes(const float* j, int M) {
    float mx = j[0];
    for (int i = 1; i < M; ++i) mx = fmaxf(mx, j[i]);
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += expf(j[i] - mx);
    return mx + logf(s);
}

// row_logdet and one or two row_maha in ONE pass over the row: the regularised variance and its reciprocal are formed
// once per element.  Each accumulator sees the same terms in the same order as the separate passes: same bits.
template <int VEC, bool TWO>
__device__ __forceinline__ void row_logdet_maha(const Row<VEC>& r, const float* xa, const float* xb, float eps, int lane,
                                                double& ld, double& ma, double& mb) {
    const float4* pa = reinterpret_cast<const float4*>(xa);
    const float4* pb = reinterpret_cast<const float4*>(xb);
    double al = 0.0, aa = 0.0, ab = 0.0;
#define UA_LM(F)                                                                        \
        const float rv_##F = reg_var(r.var[v].F, eps);                                  \
        const float rc_##F = __builtin_amdgcn_rcpf(rv_##F);                             \
        const float ln_##F = log_normal(rv_##F);                                        \
        const float da_##F = xav.F - r.mu[v].F, qa_##F = div_r(da_##F * da_##F, rv_##F, rc_##F);   \
        const float db_##F = xbv.F - r.mu[v].F, qb_##F = TWO ? div_r(db_##F * db_##F, rv_##F, rc_##F) : 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float4 xav = pa[v * 64 + lane];
        float4 xbv = xav;
        if (TWO) xbv = pb[v * 64 + lane];
        UA_LM(x) UA_LM(y) UA_LM(z) UA_LM(w)
        UA_ACC4(al, ln_x, ln_y, ln_z, ln_w);
        UA_ACC4(aa, qa_x, qa_y, qa_z, qa_w);
        if (TWO) UA_ACC4(ab, qb_x, qb_y, qb_z, qb_w);
    }
#undef UA_LM
    ld = wave_sum(al); ma = wave_sum(aa);
    if (TWO) mb = wave_sum(ab);
}

// lse_modes with the work spread over the wave: lane i < M forms expf(j[i] - max), the sum is then taken in the same
// left-to-right order through v_readlane, so the result has the same bits as lse_modes (the maximum is exact in any
// order).  ~95 instead of ~200 instructions for M = 16 - a quarter of gmm_step1_kernel's instruction stream otherwise,
// as every wave needs the value.  All 64 lanes must be active; M <= 64.
__device__ __forceinline__ float lse_modes_wave(const float* j, int M, int lane) {
    const float v = lane < M ? j[lane] : -INFINITY;
    const float mx = ua::wave_max(v);
    const float e = lane < M ? expf(v - mx) : 0.f;
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), i));
    return mx + logf(s);
}

// ---- two columns per instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) for the stream kernel --------------------
// gmm_step1_kernel saturates the VALU issue port (profiles/r03_gmm_step_phase_trace.txt: the youngest wave of every SIMD is never
// parked); on gfx950 a packed fp32 FMA costs 3.5 cycles per SIMD for two columns against 3.2 for one (tools/_probe/valu_rate.hip).
// Every operation below is the same IEEE operation on the same operands as in the scalar helpers above - only two at a time -
// so sums, state and scores keep their bits (test_stream_step_kernel_is_bit_identical_to_the_generic_one).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 reg_var2(f2 v, float eps) { const f2 t = v + splat2(eps); return f2{fmaxf(t.x, 1e-8f), fmaxf(t.y, 1e-8f)}; }
__device__ __forceinline__ f2 rcp2(f2 v) { return f2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ f2 div_r2(f2 a, f2 b, f2 r) { const f2 q = a * r; return fma2(fma2(-q, b, a), r, q); }
__device__ __forceinline__ f2 log_normal2(f2 x) {
    const f2 y = {__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
    const f2 c = splat2(__uint_as_float(0x3f317217u)), cc = splat2(__uint_as_float(0x3377d1cfu));
    const f2 r = y * c;
    return r + fma2(y, cc, fma2(y, c, -r));
}
#define UA_LO2(v4) (f2{(v4).x, (v4).y})
#define UA_HI2(v4) (f2{(v4).z, (v4).w})

// row_logdet_maha on column pairs: accumulators see the same terms in the same order (x, y, z, w of every float4)
template <int VEC, bool TWO>
__device__ __forceinline__ void row_logdet_maha_pk(const Row<VEC>& r, const float* xa, const float* xb, float eps, int lane,
                                                   double& ld, double& ma, double& mb) {
    const float4* pa = reinterpret_cast<const float4*>(xa);
    const float4* pb = reinterpret_cast<const float4*>(xb);
    double al = 0.0, aa = 0.0, ab = 0.0;
#define UA_LM2(SEL, T)                                                                  \
        const f2 rv_##T = reg_var2(SEL(r.var[v]), eps);                                 \
        const f2 rc_##T = rcp2(rv_##T);                                                 \
        const f2 ln_##T = log_normal2(rv_##T);                                          \
        const f2 mu_##T = SEL(r.mu[v]);                                                 \
        const f2 da_##T = SEL(xav) - mu_##T, qa_##T = div_r2(da_##T * da_##T, rv_##T, rc_##T);   \
        const f2 db_##T = SEL(xbv) - mu_##T, qb_##T = TWO ? div_r2(db_##T * db_##T, rv_##T, rc_##T) : f2{0.f, 0.f};
#if UA_GMM_SUM4
#define UA_ACC4P(acc, lo, hi) do { const f2 t_ = (lo) + (hi); acc += (double)(t_.x + t_.y); } while (0)        /* (x + z) + (y + w) */
#else
#define UA_ACC4P(acc, lo, hi) do { acc += (double)(lo).x; acc += (double)(lo).y; acc += (double)(hi).x; acc += (double)(hi).y; } while (0)
#endif
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float4 xav = pa[v * 64 + lane];
        float4 xbv = xav;
        if (TWO) xbv = pb[v * 64 + lane];
        UA_LM2(UA_LO2, l) UA_LM2(UA_HI2, h)
        UA_ACC4P(al, ln_l, ln_h);
        UA_ACC4P(aa, qa_l, qa_h);
        if (TWO) UA_ACC4P(ab, qb_l, qb_h);
    }
#undef UA_LM2
#undef UA_ACC4P
    ld = wave_sum(al); ma = wave_sum(aa);
    if (TWO) mb = wave_sum(ab);
}

// ------------------------------------------------------------------------------------------
// Batched log-likelihood / predict: grid (K, ceil(B/kQChunk)), block M*64.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(1024)
gmm_loglik_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ pi,
                  const float* __restrict__ x, int B, int K, int M, float eps,
                  float* __restrict__ scores, float* __restrict__ joint, float* __restrict__ loglik) {
    constexpr int D = 256 * VEC;
    __shared__ float s_j[kQChunk][kMaxModes];
    const int k = blockIdx.x, m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = blockIdx.y * kQChunk;
    const int nq = min(kQChunk, B - q0);
    Row<VEC> r;
    load_row<VEC>(r, mu, var, ((size_t)k * M + m) * D, lane);
    const double ld = row_logdet<VEC>(r, eps);
    const float logpi = logf(pi[k * M + m] + 1e-10f);
    for (int qi = 0; qi < nq; ++qi) {
        const int q = q0 + qi;
        const double mh = row_maha<VEC, false>(r, x + (size_t)q * D, eps, lane);
        const float ll = (float)(-0.5 * (ld + mh));
        if (lane == 0) {
            const float jv = logpi + ll;
            s_j[qi][m] = jv;
            const size_t o = ((size_t)q * K + k) * M + m;
            if (joint) joint[o] = jv;
            if (loglik) loglik[o] = ll;
        }
    }
    if (scores) {
        __syncthreads();
        for (int qi = threadIdx.x; qi < nq; qi += blockDim.x)
            scores[(size_t)(q0 + qi) * K + k] = lse_modes(s_j[qi], M);
    }
}

// ------------------------------------------------------------------------------------------
// Class-resident step kernel: optional predict, then up to two fits, one pass over the state.
// ------------------------------------------------------------------------------------------
struct StepArgs {
    float* mu; float* var; float* pi; float* c; float* class_counts;
    const float* xq; int nq; int round_q; float* scores;
    const float* xf[2]; const float* gf[2]; int bf[2]; int nfit;
    int K, M; float eps;
    float* inv_var; float* row_const;       // gmm_step1_kernel only: also emit what gmm_prepare_kernel would compute from the new state
};

template <int VEC>
__global__ void __launch_bounds__(1024) gmm_step_kernel(StepArgs a) {
    constexpr int D = 256 * VEC;
    __shared__ float s_j[kMaxFitB][kMaxModes];   // joint log-probabilities [sample][mode]
    __shared__ float s_c[kMaxModes];
    const int k = blockIdx.x, m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int M = a.M, K = a.K;
    const float eps = a.eps;
    const size_t off = ((size_t)k * M + m) * D;
    Row<VEC> r;
    load_row<VEC>(r, a.mu, a.var, off, lane);
    float c_km = a.c[k * M + m];
    float pi_km = a.pi[k * M + m];
    double ld = row_logdet<VEC>(r, eps);

    // ---- predict (dota_mixture.py:250-258) on the state BEFORE any fit ----
    for (int q = 0; q < a.nq; ++q) {
        const float* xq = a.xq + (size_t)q * D;
        const double mh = a.round_q ? row_maha<VEC, true>(r, xq, eps, lane) : row_maha<VEC, false>(r, xq, eps, lane);
        const float ll = (float)(-0.5 * (ld + mh));
        __syncthreads();
        if (lane == 0) s_j[0][m] = logf(pi_km + 1e-10f) + ll;
        __syncthreads();
        if (threadIdx.x == 0) a.scores[(size_t)q * K + k] = lse_modes(s_j[0], M);
    }

    // ---- fits (dota_mixture.py:162-234) ----
    for (int f = 0; f < a.nfit; ++f) {
        const float* X = a.xf[f];
        const float* G = a.gf[f];
        const int B = a.bf[f];
        if (f > 0) ld = row_logdet<VEC>(r, eps);          // var changed
        const float logpi = logf(pi_km + 1e-10f);
        __syncthreads();
        for (int b = 0; b < B; ++b) {                       // E-step :176-180
            const double mh = row_maha<VEC, false>(r, X + (size_t)b * D, eps, lane);
            if (lane == 0) s_j[b][m] = logpi + (float)(-0.5 * (ld + mh));
        }
        __syncthreads();
        // responsibilities of this wave's mode, gamma_{b,k,m} = gamma_class[b,k] * r   :183-187
        float sum_gamma = 0.f;
        float4 wx[VEC], wxx[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { wx[v] = make_float4(0.f, 0.f, 0.f, 0.f); wxx[v] = wx[v]; }
        for (int b = 0; b < B; ++b) {
            const float lse = lse_modes(s_j[b], M);
            const float g = G[(size_t)b * K + k] * expf(s_j[b][m] - lse);
            sum_gamma += g;                                                            // :190
            const float4* px = reinterpret_cast<const float4*>(X + (size_t)b * D);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {                                            // :202,:213-214
                const float4 xv = px[v * 64 + lane];
                wx[v].x += g * xv.x; wx[v].y += g * xv.y; wx[v].z += g * xv.z; wx[v].w += g * xv.w;
                wxx[v].x += g * (xv.x * xv.x); wxx[v].y += g * (xv.y * xv.y);
                wxx[v].z += g * (xv.z * xv.z); wxx[v].w += g * (xv.w * xv.w);
            }
        }
        const float c_new = c_km + sum_gamma;                                          // :196
        const float den = c_new + 1e-10f;
        const float rden = __builtin_amdgcn_rcpf(den);
#define UA_UPD(F)                                                                                     \
        {                                                                                             \
            const float mo = r.mu[v].F, vo = r.var[v].F, w1 = wx[v].F, w2 = wxx[v].F;                 \
            const float mn = div_r(c_km * mo + w1, den, rden);                         /* :205-206 */ \
            const float sq = (w2 + (-2.0f * mo) * w1) + sum_gamma * (mo * mo);         /* :215-217 */ \
            const float vn = div_r(c_km * vo + sq, den, rden);                         /* :219-220 */ \
            r.mu[v].F = mn; r.var[v].F = fmaxf(vn, 1e-8f);                             /* :223     */ \
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) { UA_UPD(x) UA_UPD(y) UA_UPD(z) UA_UPD(w) }
#undef UA_UPD
        c_km = c_new;
        if (lane == 0) s_c[m] = c_new;
        __syncthreads();
        float Ck = 0.f;
        for (int i = 0; i < M; ++i) Ck += s_c[i];
        pi_km = c_km / (Ck + 1e-10f);                                                  // :229-230
        if (threadIdx.x == 0) {                                                        // :233
            float gs = 0.f;
            for (int b = 0; b < B; ++b) gs += G[(size_t)b * K + k];
            a.class_counts[k] += gs;
        }
    }
    if (a.nfit > 0) {
        store_row<VEC>(r, a.mu, a.var, off, lane);
        if (lane == 0) { a.c[k * M + m] = c_km; a.pi[k * M + m] = pi_km; }
    }
}

// ------------------------------------------------------------------------------------------
// The stream case of the step kernel - one query, two single-sample fits (Uni_Adapter.py:416,417,430) - as a
// grid-stride loop over classes with the NEXT class's rows prefetched into registers while the current class is
// computed, and the updated rows stored asynchronously: with one 1024-thread workgroup per CU the generic kernel
// runs load -> compute -> store back to back (VALU idle while loading, HBM idle while computing).  Without the
// per-fit accumulators of the batched path (for B = 1, sum_b g*x is g*x) the rows of two classes fit the 128-VGPR
// budget of a 1024-thread workgroup.  Same expressions in the same order as gmm_step_kernel: bit-identical state
// and scores (tests/test_gpu_kernels.py::test_stream_step_kernel_is_bit_identical_to_the_generic_one).
// ------------------------------------------------------------------------------------------
// logsumexp over the first M lanes' values (lane i holds value i; lanes >= M are ignored): same association as lse_modes
__device__ __forceinline__ float lse_lanes(float v, int M, int lane) {
    v = lane < M ? v : -INFINITY;
    const float mx = ua::wave_max(v);
    const float e = lane < M ? expf(v - mx) : 0.f;
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), i));
    return mx + logf(s);
}

#ifdef UA_GMM_TRACE
// Instrumented build only (tools/gmm_trace.py): s_memtime stamps of one steady-state class of workgroup 0, per wave.
__device__ unsigned long long g_gmm_trace[16 * 16];
#define UA_GSTAMP(i) do { if (blockIdx.x == 0 && lane == 0 && it_ == 3) g_gmm_trace[m * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UA_GSTAMP(i) do { } while (0)
#endif

// Exchanges per class (one LDS slot each, one barrier between a slot's write and its reads):
//   A: query joint + first-fit joint of every mode (both use the old pi)   -> barrier A -> scores, responsibilities of fit 1
//   B: count after fit 1 + second-fit log-likelihood (does not depend on pi) -> barrier B -> every wave forms sum_m c, the
//      new pi_i and joint_i of ALL modes on its lanes (one division and one logf per lane, in parallel), its logsumexp,
//      and fit 2;  the count after fit 2 is exchanged at barrier A of the NEXT class (the final pi of a class is only
//      stored, nobody waits for it) - unless the per-sample constants of the residual loop are emitted, which need it at once.
// Two barriers per class instead of five; expressions and summation orders are those of gmm_step_kernel (bit-identical).
template <int VEC>
__global__ void __launch_bounds__(1024) gmm_step1_kernel(StepArgs a) {
    constexpr int D = 256 * VEC;
    __shared__ __attribute__((aligned(16))) float s_x[3][D];     // query (fp16-rounded if asked), sample, augmented sample
    __shared__ __attribute__((aligned(16))) float s_xx[2][D];    // squares of the two samples (:213)
    __shared__ float s_jq[kMaxModes], s_j0[kMaxModes], s_c0[kMaxModes], s_l1[kMaxModes], s_c1[kMaxModes];
    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int M = a.M, K = a.K;
    const float eps = a.eps;
    // Everything a class needs from global memory is requested one class ahead: the vector-memory counter retires in
    // order, so ANY later load inside the class body would also wait for the prefetched rows and undo the overlap.
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float q = a.xq[i];
        s_x[0][i] = a.round_q ? round_fp16(q) : q;
        const float x1 = a.xf[0][i], x2 = a.xf[1][i];
        s_x[1][i] = x1; s_x[2][i] = x2;
        s_xx[0][i] = x1 * x1; s_xx[1][i] = x2 * x2;
    }
    Row<VEC> r, nx;
    int k = blockIdx.x;
    float c_km = 0.f, pi_km = 0.f, g0 = 0.f, g1 = 0.f, cc = 0.f;
    float c_nx = 0.f, pi_nx = 0.f, g0_nx = 0.f, g1_nx = 0.f, cc_nx = 0.f;
    if (k < K) {
        c_km = a.c[k * M + m]; pi_km = a.pi[k * M + m]; g0 = a.gf[0][k]; g1 = a.gf[1][k]; cc = a.class_counts[k];
        load_row<VEC>(r, a.mu, a.var, ((size_t)k * M + m) * D, lane);
    }
    // first class resident before the loop: otherwise the compiler's wait for these loads sits at the first use inside the
    // loop body and - the counter being in-order, and the prefetch below conditional - also waits for the prefetch there
    __builtin_amdgcn_s_waitcnt(0x0F70);                                                  // vmcnt(0)
    __syncthreads();
    const bool now = a.inv_var != nullptr;                                               // final pi needed inside the class body
    bool have_prev = false;                                                              // (wave-uniform)
#ifdef UA_GMM_TRACE
    int it_ = -1;
#endif
    for (; k < K; k += gridDim.x) {
#ifdef UA_GMM_TRACE
        ++it_;
#endif
        UA_GSTAMP(0);
        const int kn = k + gridDim.x;
        if (kn < K) {
            c_nx = a.c[kn * M + m]; pi_nx = a.pi[kn * M + m]; g0_nx = a.gf[0][kn]; g1_nx = a.gf[1][kn]; cc_nx = a.class_counts[kn];
            load_row<VEC>(nx, a.mu, a.var, ((size_t)kn * M + m) * D, lane);             // in flight during this class
        }
        const size_t off = ((size_t)k * M + m) * D;
        double ld, mh, mh1 = 0.0;
        UA_GSTAMP(1);
        row_logdet_maha_pk<VEC, true>(r, s_x[0], s_x[1], eps, lane, ld, mh, mh1);     // log-det, query and first-fit distances
        UA_GSTAMP(2);
        {
            const float logpi = logf(pi_km + 1e-10f);
            if (lane == 0) {
                s_jq[m] = logpi + (float)(-0.5 * (ld + mh));                            // predict (dota_mixture.py:250-258)
                s_j0[m] = logpi + (float)(-0.5 * (ld + mh1));                           // E-step of fit 1 (:176-180)
            }
        }
        UA_GSTAMP(3);
        __syncthreads();                                                                // A
        UA_GSTAMP(4);
        if (have_prev) {                                                                // final pi of the previous class (:229-230)
            float Ck = 0.f;
            for (int i = 0; i < M; ++i) Ck += s_c1[i];
            if (lane == 0) a.pi[(k - (int)gridDim.x) * M + m] = s_c1[m] / (Ck + 1e-10f);
        }
        if (m == 0) {                                                                   // wave-uniform
            const float sc = lse_modes_wave(s_jq, M, lane);
            if (lane == 0) a.scores[k] = sc;
        }
#pragma unroll 1
        for (int f = 0; f < 2; ++f) {                                                   // fits :162-234, B = 1
            const float* X = s_x[1 + f];
            float g;
            if (f == 0) {
                const float lse = lse_modes_wave(s_j0, M, lane);
                g = g0 * expf(s_j0[m] - lse);                                          // :183-187
            } else {
                // what this wave contributes to fit 2's E-step: its count and its log-likelihood under the updated row (var changed)
                double unused;
                UA_GSTAMP(7);
                row_logdet_maha_pk<VEC, false>(r, X, X, eps, lane, ld, mh1, unused);
                if (lane == 0) { s_c0[m] = c_km; s_l1[m] = (float)(-0.5 * (ld + mh1)); }
                UA_GSTAMP(8);
                __syncthreads();                                                        // B
                UA_GSTAMP(9);
                // pi of every mode after fit 1 (:229-230), joint of every mode on the lanes, responsibilities
                float Ck = 0.f;
                for (int i = 0; i < M; ++i) Ck += s_c0[i];
                const float ci = lane < M ? s_c0[lane] : 0.f;
                const float pii = ci / (Ck + 1e-10f);
                const float ji = lane < M ? logf(pii + 1e-10f) + s_l1[lane] : -INFINITY;
                const float lse = lse_lanes(ji, M, lane);
                const float jm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ji), m));
                g = g1 * expf(jm - lse);
            }
            const float sum_gamma = g;                                                  // :190
            const float c_new = c_km + sum_gamma;                                       // :196
            const float den = c_new + 1e-10f;
            const float rden = __builtin_amdgcn_rcpf(den);
            const float4* px = reinterpret_cast<const float4*>(X);
            const float4* pxx = reinterpret_cast<const float4*>(s_xx[f]);
            if (f == 0) UA_GSTAMP(5); else UA_GSTAMP(10);
            // two columns per instruction; x*x comes from LDS (formed once per launch: the same product)
            const f2 g2 = splat2(g), c2 = splat2(c_km), den2 = splat2(den), rden2 = splat2(rden), sg2 = splat2(sum_gamma), m2c = splat2(-2.0f);
#define UA_UPD2(SEL, LO)                                                                              \
            {                                                                                         \
                const f2 mo = SEL(r.mu[v]), vo = SEL(r.var[v]);                                       \
                const f2 w1 = g2 * SEL(xv), w2 = g2 * SEL(xxv);     /* :202,:213-214; 0 + g*x of the batched sum dropped (differs for -0 only) */ \
                const f2 mn = div_r2(c2 * mo + w1, den2, rden2);                       /* :205-206 */ \
                const f2 sq = (w2 + (m2c * mo) * w1) + sg2 * (mo * mo);                /* :215-217 */ \
                const f2 vn = div_r2(c2 * vo + sq, den2, rden2);                       /* :219-220 */ \
                if (LO) { r.mu[v].x = mn.x; r.mu[v].y = mn.y; r.var[v].x = fmaxf(vn.x, 1e-8f); r.var[v].y = fmaxf(vn.y, 1e-8f); }   /* :223 */ \
                else    { r.mu[v].z = mn.x; r.mu[v].w = mn.y; r.var[v].z = fmaxf(vn.x, 1e-8f); r.var[v].w = fmaxf(vn.y, 1e-8f); }              \
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float4 xv = px[v * 64 + lane];
                const float4 xxv = pxx[v * 64 + lane];
                UA_UPD2(UA_LO2, true) UA_UPD2(UA_HI2, false)
            }
#undef UA_UPD2
            c_km = c_new;
            cc += f ? g1 : g0;                                                          // :233
            if (f == 0) UA_GSTAMP(6); else UA_GSTAMP(11);
        }
        store_row<VEC>(r, a.mu, a.var, off, lane);
        UA_GSTAMP(12);
        if (lane == 0) { a.c[k * M + m] = c_km; s_c1[m] = c_km; }
        if (threadIdx.x == 0) a.class_counts[k] = cc;
        if (now) {                                         // workgroup-uniform: pi at once, then gmm_prepare_kernel on the rows still in registers
            __syncthreads();
            float Ck = 0.f;
            for (int i = 0; i < M; ++i) Ck += s_c1[i];
            pi_km = c_km / (Ck + 1e-10f);
            if (lane == 0) a.pi[k * M + m] = pi_km;
            float4* po = reinterpret_cast<float4*>(a.inv_var + off);
            double acc = 0.0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float x = reg_var(r.var[v].x, eps), y = reg_var(r.var[v].y, eps), z = reg_var(r.var[v].z, eps), w = reg_var(r.var[v].w, eps);
                acc += (double)log_normal(x) + (double)log_normal(y) + (double)log_normal(z) + (double)log_normal(w);
                po[v * 64 + lane] = make_float4(1.0f / x, 1.0f / y, 1.0f / z, 1.0f / w);
            }
            acc = wave_sum(acc);
            if (lane == 0) a.row_const[k * M + m] = (float)((double)logf(pi_km + 1e-10f) - 0.5 * acc);
            __syncthreads();                               // s_c1 is rewritten by the next class
        } else {
            have_prev = true;
        }
        if (kn < K) { r = nx; c_km = c_nx; pi_km = pi_nx; g0 = g0_nx; g1 = g1_nx; cc = cc_nx; }
        UA_GSTAMP(13);
    }
    if (have_prev) {                                       // final pi of this workgroup's last class (k has run past it by one stride)
        __syncthreads();
        float Ck = 0.f;
        for (int i = 0; i < M; ++i) Ck += s_c1[i];
        if (lane == 0) a.pi[(k - (int)gridDim.x) * M + m] = s_c1[m] / (Ck + 1e-10f);
    }
}

// ------------------------------------------------------------------------------------------
// Backward of the log-likelihood w.r.t. x: grad_x[b,d] = sum_{k,m} w[b,k,m] * (mu-x)/v.
// grid (B, D/256), block 256: thread = one feature d, loop over the K*M rows (coalesced 1 KiB
// row segments, state served from L2 / Infinity Cache).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gmm_loglik_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ x,
                      const float* __restrict__ w, int KM, int D, float eps, float* __restrict__ gx) {
    __shared__ float s_w[1024];
    const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
    const float xv = x[(size_t)b * D + d];
    double acc = 0.0;
    for (int r0 = 0; r0 < KM; r0 += 1024) {
        const int n = min(1024, KM - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) s_w[i] = w[(size_t)b * KM + r0 + i];
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)(r0 + i) * D + d;
            const float v = reg_var(var[o], eps);
            acc += (double)(s_w[i] * ((mu[o] - xv) / v));
        }
    }
    gx[(size_t)b * D + d] = (float)acc;
}

// ------------------------------------------------------------------------------------------
// Residual-learning fast path (Uni_Adapter.py:444-476).  The GMM state does not change during the
// 10 Adam steps of a sample, so 1/v and the per-row constant ln(pi+1e-10) - 1/2 sum ln v are
// computed ONCE per sample (gmm_prepare_kernel) and the 11 forward / 10 backward evaluations are
// pure multiply-add streams over L2-resident rows.
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(1024)
gmm_prepare_kernel(const float* __restrict__ var, const float* __restrict__ pi, int M, float eps,
                   float* __restrict__ inv_var, float* __restrict__ row_const) {
    constexpr int D = 256 * VEC;
    const int k = blockIdx.x, m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t off = ((size_t)k * M + m) * D;
    const float4* pv = reinterpret_cast<const float4*>(var + off);
    float4* po = reinterpret_cast<float4*>(inv_var + off);
    double acc = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float4 a = pv[v * 64 + lane];
        const float x = reg_var(a.x, eps), y = reg_var(a.y, eps), z = reg_var(a.z, eps), w = reg_var(a.w, eps);
        acc += (double)log_normal(x) + (double)log_normal(y) + (double)log_normal(z) + (double)log_normal(w);
        po[v * 64 + lane] = make_float4(1.0f / x, 1.0f / y, 1.0f / z, 1.0f / w);
    }
    acc = wave_sum(acc);
    if (lane == 0) row_const[k * M + m] = (float)((double)logf(pi[k * M + m] + 1e-10f) - 0.5 * acc);
}

// queries per workgroup in align_fwd_kernel (template parameter Q): the workgroup's rows stay in registers for Q queries, so the launch
// has ceil(B/Q) * K workgroups - fewer, longer workgroups cost the same latency (measured at Q = 2, 4, 8 in round 1) and less
// CU-time, which is what the replay is paid in when it shares the GPU with the encoder.

// grid (K classes, ceil(B/kAlignQ)), block M*64.  joint[b,k,m] = const[k,m] - 1/2 sum_d (e-mu)^2 * iv
template <int VEC, int kAlignQ>
__global__ void __launch_bounds__(1024)
align_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ inv_var, const float* __restrict__ row_const,
                 const float* __restrict__ e, int B, int K, int M, float* __restrict__ L, float* __restrict__ joint) {
    constexpr int D = 256 * VEC;
    __shared__ float s_j[kAlignQ][kMaxModes];
    const int k = blockIdx.x, m = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = blockIdx.y * kAlignQ, nq = min(kAlignQ, B - q0);
    const size_t off = ((size_t)k * M + m) * D;
    const float4* pm = reinterpret_cast<const float4*>(mu + off);
    const float4* pi4 = reinterpret_cast<const float4*>(inv_var + off);
    float4 rm[VEC], ri[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { rm[v] = pm[v * 64 + lane]; ri[v] = pi4[v * 64 + lane]; }
    const float cst = row_const[k * M + m];
    for (int qi = 0; qi < nq; ++qi) {
        const float4* pe = reinterpret_cast<const float4*>(e + (size_t)(q0 + qi) * D);
        double acc = 0.0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float4 x = pe[v * 64 + lane];
            float d;
            d = x.x - rm[v].x; acc += (double)((d * d) * ri[v].x);
            d = x.y - rm[v].y; acc += (double)((d * d) * ri[v].y);
            d = x.z - rm[v].z; acc += (double)((d * d) * ri[v].z);
            d = x.w - rm[v].w; acc += (double)((d * d) * ri[v].w);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const float jv = (float)((double)cst - 0.5 * acc);
            s_j[qi][m] = jv;
            joint[((size_t)(q0 + qi) * K + k) * M + m] = jv;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nq) L[(size_t)(q0 + threadIdx.x) * K + k] = lse_modes(s_j[threadIdx.x], M);
}

// grad_e[b, d] = sum_{k,m} w[b,k,m] (mu - e) * iv.  grid (B, D/256), block 1024 = 16 waves; every
// wave owns rows km = wave, wave+16, ... and 256 features (one float4 per lane); LDS tree at the end.
__global__ void __launch_bounds__(1024)
align_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ inv_var, const float* __restrict__ e,
                 const float* __restrict__ w, int KM, int D, float* __restrict__ ge) {
    __shared__ double s_acc[16][256];
    __shared__ float s_w[2048];
    const int b = blockIdx.x, c = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 x = reinterpret_cast<const float4*>(e + (size_t)b * D + c * 256)[lane];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int r0 = 0; r0 < KM; r0 += 2048) {
        const int n = min(2048, KM - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) s_w[i] = w[(size_t)b * KM + r0 + i];
        __syncthreads();
#pragma unroll 4
        for (int i = wv; i < n; i += 16) {
            const size_t o = (size_t)(r0 + i) * D + c * 256;
            const float4 m4 = reinterpret_cast<const float4*>(mu + o)[lane];
            const float4 i4 = reinterpret_cast<const float4*>(inv_var + o)[lane];
            const float ww = s_w[i];
            a0 += (double)(ww * ((m4.x - x.x) * i4.x));
            a1 += (double)(ww * ((m4.y - x.y) * i4.y));
            a2 += (double)(ww * ((m4.z - x.z) * i4.z));
            a3 += (double)(ww * ((m4.w - x.w) * i4.w));
        }
    }
    s_acc[wv][lane * 4 + 0] = a0; s_acc[wv][lane * 4 + 1] = a1; s_acc[wv][lane * 4 + 2] = a2; s_acc[wv][lane * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 256) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s_acc[i][threadIdx.x];
        ge[(size_t)b * D + c * 256 + threadIdx.x] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------
// One residual-learning step in ONE launch: alignment loss + its gradient (align_loss_kernel of
// ua_logits.hip), the backward through the log-likelihoods (align_bwd_kernel above) and the Adam
// update with renormalisation (residual_adam_kernel) - Uni_Adapter.py:191-270,444-476 after the
// forward.  grid (B = K rows, D/256 column chunks), block 1024.
//   * every workgroup recomputes the K x K loss statistics from L (K <= 160: <= 25,600 values;
//     same code and reduction order as align_loss_kernel) and forms ONLY its own row b of the
//     weights w[b,c,m] - no separate loss launch, no grad_ll round trip;
//   * its 256 columns of grad_e[b] are accumulated exactly as in align_bwd_kernel;
//   * the D/256 workgroups of a row hand over through an agent-scope release/acquire ticket;
//     the last arriver applies the Adam step to row b with its first 256 threads (the block
//     shape of residual_adam_kernel, so the sums associate identically).
// Results are bit-identical to the three separate launches.  `tickets` (B int32) must be zero
// at launch; the last arriver re-zeroes its entry.
// dynamic LDS: 2*K*K + 3K floats (P, row sums, column sums, diagonal, L).
// ------------------------------------------------------------------------------------------
struct AlignStepArgs {
    const float* mu; const float* inv_var; const float* L; const float* joint;
    const float* text0; float* res; float* am; float* av; float* e; float* nrm;
    float* ge; float* loss_out; float* grad_ll; int* tickets;
    int K, M, D;
    float step_size, w1, beta2, omb2, bc2_sqrt, adam_eps;
};

// block_sum over the first `nw` waves only (all threads of the block must call it)
__device__ __forceinline__ double block_sum_first(double v, double* scratch, int nw) {
    v = ua::wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && w < nw) scratch[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

#ifdef UA_STEP_TRACE
#define ST_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ST_MARK(i)
#endif
constexpr int kStepRows = 12;       // (mu - e) * inv_var rows a wave prefetches into registers (rows wave, wave+16, ...)

__global__ void __launch_bounds__(1024)
align_step_kernel(AlignStepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double s_acc[16][256];
    __shared__ float s_w[2048];
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ int s_arg, s_last;
    const int K = a.K, M = a.M, D = a.D, KM = K * M;
    const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, nt = 1024;
    const int wv = tid >> 6, lane = tid & 63;
    float* P = smem;                 // K*K
    float* s1 = smem + K * K;        // row sums
    float* s2 = s1 + K;              // col sums
    float* dg = s2 + K;              // diagonal
    float* Ls = dg + K;              // K*K copy of L (read once from L2)
#ifdef UA_STEP_TRACE
    unsigned long long st_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ST_MARK(0);
    // ---- issue the long-latency reads first: the wave's (mu - e) * inv_var rows do not depend on the
    //      loss, only their weights do.  d[j] is formed with the same operations as align_bwd_kernel.
    const float4 x = reinterpret_cast<const float4*>(a.e + (size_t)b * D + c * 256)[lane];
    const bool pre = KM <= 2048;                                // block-uniform (one pass over the weights)
    // the weight this thread forms after the loss phase needs joint[b, km]; the Adam tail (if this
    // workgroup turns out to be the row's last arriver) needs the row's state: all requested now.
    float jpre = 0.f;
    if (pre && tid < KM) jpre = a.joint[(size_t)b * KM + tid];
    const size_t orow = (size_t)b * D;
    float4 p_am = make_float4(0.f, 0.f, 0.f, 0.f), p_av = p_am, p_res = p_am, p_t0 = p_am, p_e = p_am;
    if (tid < 256 && D == 1024) {
        p_am = make_float4(a.am[orow + tid], a.am[orow + tid + 256], a.am[orow + tid + 512], a.am[orow + tid + 768]);
        p_av = make_float4(a.av[orow + tid], a.av[orow + tid + 256], a.av[orow + tid + 512], a.av[orow + tid + 768]);
        p_res = make_float4(a.res[orow + tid], a.res[orow + tid + 256], a.res[orow + tid + 512], a.res[orow + tid + 768]);
        p_t0 = make_float4(a.text0[orow + tid], a.text0[orow + tid + 256], a.text0[orow + tid + 512], a.text0[orow + tid + 768]);
        p_e = make_float4(a.e[orow + tid], a.e[orow + tid + 256], a.e[orow + tid + 512], a.e[orow + tid + 768]);
    }
    float4 d[kStepRows];
    if (pre) {
#pragma unroll
        for (int j = 0; j < kStepRows; ++j) {
            const int i = wv + 16 * j;
            if (i < KM) {
                const size_t o = (size_t)i * D + c * 256;
                const float4 m4 = reinterpret_cast<const float4*>(a.mu + o)[lane];
                const float4 i4 = reinterpret_cast<const float4*>(a.inv_var + o)[lane];
                d[j] = make_float4((m4.x - x.x) * i4.x, (m4.y - x.y) * i4.y, (m4.z - x.z) * i4.z, (m4.w - x.w) * i4.w);
            }
        }
    }
    ST_MARK(1);
    // ---- loss statistics: the code of align_loss_kernel (ua_align_common.h), L staged in LDS ----
    double* sc = &s_acc[0][0];                              // (s_acc is first written after the statistics)
    const ua::AlignStats stt = ua::align_stats(a.L, Ls, K, P, s1, s2, dg, sd, sf, &s_arg, sc);
    if (b == 0 && c == 0 && tid == 0 && a.loss_out) *a.loss_out = (float)(-stt.tot / (double)K);
    const float mx = stt.mx, corr = stt.corr;
    const int amax = stt.amax;
    ST_MARK(6);
    // ---- backward for row b, columns c*256 .. +255 (align_bwd_kernel with w formed in place) ----
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int r0 = 0; r0 < KM; r0 += 2048) {
        const int nn = min(2048, KM - r0);
        __syncthreads();
        for (int i = tid; i < nn; i += nt) {
            const int km = r0 + i, k = km / M, ik = b * K + k;
            float G = P[ik] / mx;
            if (ik == amax) G += corr;
            const float jv = pre ? jpre : a.joint[(size_t)ik * M + (km - k * M)];       // pre: i == tid == km
            const float wgt = G * expf(jv - Ls[ik]);
            s_w[i] = wgt;
            if (a.grad_ll && c == 0) a.grad_ll[(size_t)b * KM + km] = wgt;
        }
        __syncthreads();
        int i_first = wv;
        if (pre) {
#pragma unroll
            for (int j = 0; j < kStepRows; ++j) {
                const int i = wv + 16 * j;
                if (i < nn) {
                    const float ww = s_w[i];
                    a0 += (double)(ww * d[j].x); a1 += (double)(ww * d[j].y); a2 += (double)(ww * d[j].z); a3 += (double)(ww * d[j].w);
                }
            }
            i_first = wv + 16 * kStepRows;                     // the rest (same row order) straight from L2
        }
        {
#pragma unroll 4
            for (int i = i_first; i < nn; i += 16) {
                const size_t o = (size_t)(r0 + i) * D + c * 256;
                const float4 m4 = reinterpret_cast<const float4*>(a.mu + o)[lane];
                const float4 i4 = reinterpret_cast<const float4*>(a.inv_var + o)[lane];
                const float ww = s_w[i];
                a0 += (double)(ww * ((m4.x - x.x) * i4.x));
                a1 += (double)(ww * ((m4.y - x.y) * i4.y));
                a2 += (double)(ww * ((m4.z - x.z) * i4.z));
                a3 += (double)(ww * ((m4.w - x.w) * i4.w));
            }
        }
    }
    s_acc[wv][lane * 4 + 0] = a0; s_acc[wv][lane * 4 + 1] = a1; s_acc[wv][lane * 4 + 2] = a2; s_acc[wv][lane * 4 + 3] = a3;
    __syncthreads();
    if (tid < 256) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s_acc[i][tid];
        a.ge[(size_t)b * D + c * 256 + tid] = (float)t;
    }
    ST_MARK(7);
    // ---- hand-off: the last of the D/256 workgroups of row b applies Adam to the row ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(&a.tickets[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (int)gridDim.y - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a.tickets[b] = 0;                                  // the next launch is stream-ordered behind this one
        }
        s_last = last;
    }
    __syncthreads();
    ST_MARK(8);
#ifdef UA_STEP_TRACE
    if (!s_last) { if (b == 1 && tid == 0 && a.grad_ll) for (int i = 0; i < 9; ++i) a.grad_ll[i] = (float)(st_t[i] - st_t[0]); return; }
#endif
    if (!s_last) return;                                   // workgroup-uniform
    const size_t o = (size_t)b * D;
    const bool act = tid < 256;                            // residual_adam_kernel's block shape
    const float nrm = a.nrm[b];
    if (D == 1024) {
        // same arithmetic as below with this thread's 4 columns (tid, tid+256, ...) held in registers
        const float pe[4] = {p_e.x, p_e.y, p_e.z, p_e.w}, pt[4] = {p_t0.x, p_t0.y, p_t0.z, p_t0.w};
        float pm[4] = {p_am.x, p_am.y, p_am.z, p_am.w}, pv[4] = {p_av.x, p_av.y, p_av.z, p_av.w}, pr[4] = {p_res.x, p_res.y, p_res.z, p_res.w};
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        double dot = 0.0;
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { gv[q] = a.ge[o + tid + 256 * q]; dot += (double)(gv[q] * pe[q]); }
        }
        const float gdote = (float)block_sum_first(dot, sd, 4);
        double s = 0.0;
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = (gv[q] - pe[q] * gdote) / nrm;
                float m = pm[q], v = pv[q];
                m = m + a.w1 * (g - m);
                v = v * a.beta2 + a.omb2 * (g * g);
                const float den = sqrtf(v) / a.bc2_sqrt + a.adam_eps;
                const float r = pr[q] + ((-a.step_size) * m) / den;
                a.am[o + tid + 256 * q] = m; a.av[o + tid + 256 * q] = v; a.res[o + tid + 256 * q] = r;
                pr[q] = r;
                const float t = pt[q] + r;
                s += (double)(t * t);
            }
        }
        const float nn2 = (float)sqrt(block_sum_first(s, sd, 4));
        if (tid == 0) a.nrm[b] = nn2;
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a.e[o + tid + 256 * q] = (pt[q] + pr[q]) / nn2;
        }
    } else {
        double dot = 0.0;
        if (act) for (int i = tid; i < D; i += 256) dot += (double)(a.ge[o + i] * a.e[o + i]);
        const float gdote = (float)block_sum_first(dot, sd, 4);
        double s = 0.0;
        if (act) for (int i = tid; i < D; i += 256) {
            const float g = (a.ge[o + i] - a.e[o + i] * gdote) / nrm;
            float m = a.am[o + i], v = a.av[o + i];
            m = m + a.w1 * (g - m);
            v = v * a.beta2 + a.omb2 * (g * g);
            const float den = sqrtf(v) / a.bc2_sqrt + a.adam_eps;
            const float r = a.res[o + i] + ((-a.step_size) * m) / den;
            a.am[o + i] = m; a.av[o + i] = v; a.res[o + i] = r;
            const float t = a.text0[o + i] + r;
            s += (double)(t * t);
        }
        const float nn2 = (float)sqrt(block_sum_first(s, sd, 4));
        if (tid == 0) a.nrm[b] = nn2;
        if (act) for (int i = tid; i < D; i += 256) a.e[o + i] = (a.text0[o + i] + a.res[o + i]) / nn2;
    }
    ST_MARK(9);
#ifdef UA_STEP_TRACE
    if (b == 0 && tid == 0 && a.grad_ll) for (int i = 0; i < 10; ++i) a.grad_ll[16 + i] = (float)(st_t[i] - st_t[0]);
#endif
}

template <typename F>
int dispatch_vec(int D, F&& f) {
    switch (D / 256) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 5: return f(std::integral_constant<int, 5>{});
        case 6: return f(std::integral_constant<int, 6>{});
        case 7: return f(std::integral_constant<int, 7>{});
        case 8: return f(std::integral_constant<int, 8>{});
        default: return UA_EUNSUPPORTED;
    }
}

int check_envelope(int K, int M, int D) {
    if (K <= 0 || M <= 0 || D <= 0) return UA_EINVAL;
    if (D % 256 != 0 || D > 2048 || M > kMaxModes) return UA_EUNSUPPORTED;
    return UA_OK;
}

int launch_step(const StepArgs& a, int D, hipStream_t st) {
    return dispatch_vec(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        hipLaunchKernelGGL(gmm_step_kernel<VEC>, dim3(a.K), dim3(a.M * 64), 0, st, a);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
}

}  // namespace

extern "C" int ua_gmm_loglik(const float* mu, const float* var, const float* pi, const float* x,
                             int32_t B, int32_t K, int32_t M, int32_t D, float eps,
                             float* scores, float* joint, float* loglik, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(x);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(x);
    if (B <= 0) return UA_EINVAL;
    if (int e = check_envelope(K, M, D)) return e;
    return dispatch_vec(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        dim3 grid(K, (B + kQChunk - 1) / kQChunk);
        hipLaunchKernelGGL(gmm_loglik_kernel<VEC>, grid, dim3(M * 64), 0, ua::as_stream(stream),
                           mu, var, pi, x, B, K, M, eps, scores, joint, loglik);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
}

extern "C" int ua_gmm_loglik_bwd(const float* mu, const float* var, const float* x, const float* grad_ll,
                                 int32_t B, int32_t K, int32_t M, int32_t D, float eps,
                                 float* grad_x, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(x); UA_CHECK_PTR(grad_ll); UA_CHECK_PTR(grad_x);
    if (B <= 0 || K <= 0 || M <= 0 || D <= 0) return UA_EINVAL;
    if (D % 256 != 0) return UA_EUNSUPPORTED;
    hipLaunchKernelGGL(gmm_loglik_bwd_kernel, dim3(B, D / 256), dim3(256), 0, ua::as_stream(stream),
                       mu, var, x, grad_ll, K * M, D, eps, grad_x);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_gmm_fit(float* mu, float* var, float* pi, float* c, float* class_counts,
                          const float* x, const float* gamma_class,
                          int32_t B, int32_t K, int32_t M, int32_t D, float eps, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(c); UA_CHECK_PTR(class_counts);
    UA_CHECK_PTR(x); UA_CHECK_PTR(gamma_class);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(x);
    if (B <= 0) return UA_EINVAL;
    if (B > kMaxFitB) return UA_EUNSUPPORTED;
    if (int e = check_envelope(K, M, D)) return e;
    StepArgs a{};
    a.mu = mu; a.var = var; a.pi = pi; a.c = c; a.class_counts = class_counts;
    a.nq = 0; a.nfit = 1; a.xf[0] = x; a.gf[0] = gamma_class; a.bf[0] = B;
    a.K = K; a.M = M; a.eps = eps;
    return launch_step(a, D, ua::as_stream(stream));
}

static int step_fused_impl(float* mu, float* var, float* pi, float* c, float* class_counts,
                           const float* xq, int32_t xq_round_fp16, const float* x1, const float* x2,
                           const float* gamma_class, float* scores,
                           int32_t K, int32_t M, int32_t D, float eps, float* inv_var, float* row_const, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(c); UA_CHECK_PTR(class_counts);
    UA_CHECK_PTR(xq); UA_CHECK_PTR(x1); UA_CHECK_PTR(x2); UA_CHECK_PTR(gamma_class); UA_CHECK_PTR(scores);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(xq); UA_CHECK_ALIGN16(x1); UA_CHECK_ALIGN16(x2);
    if (int e = check_envelope(K, M, D)) return e;
    StepArgs a{};
    a.mu = mu; a.var = var; a.pi = pi; a.c = c; a.class_counts = class_counts;
    a.xq = xq; a.nq = 1; a.round_q = xq_round_fp16; a.scores = scores;
    a.nfit = 2; a.xf[0] = x1; a.xf[1] = x2; a.gf[0] = gamma_class; a.gf[1] = gamma_class; a.bf[0] = 1; a.bf[1] = 1;
    a.K = K; a.M = M; a.eps = eps;
    if (D > 1024) {                                                      // two classes' rows do not fit the register budget
        if (int e = launch_step(a, D, ua::as_stream(stream))) return e;
        return inv_var ? ua_gmm_prepare_alignment(var, pi, K, M, D, eps, inv_var, row_const, stream) : UA_OK;
    }
    a.inv_var = inv_var; a.row_const = row_const;
    // persistent form: at most one workgroup per CU and round-robin classes, so every CU streams its next class
    // while it computes the current one
    return dispatch_vec(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        if constexpr (VEC <= 4) {
            // 119 VGPRs: 4 waves per SIMD = one 16-wave workgroup per CU, or two of <= 8 waves (M <= 8): fill both slots
            // (K = 1156, M = 8: 54.0 -> 46.0 us; a 64-VGPR build that fits two 16-wave workgroups spills and is slower: 0.32 vs 0.40)
            const int cap = M <= 8 ? 512 : 256;
            hipLaunchKernelGGL(gmm_step1_kernel<VEC>, dim3(K < cap ? K : cap), dim3(M * 64), 0, ua::as_stream(stream), a);
            UA_LAUNCH_CHECK();
        }
        return UA_OK;
    });
}

#ifdef UA_GMM_TRACE
extern "C" int ua_debug_gmm_trace(unsigned long long* host256) {
    return hipMemcpyFromSymbol(host256, HIP_SYMBOL(g_gmm_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ua_gmm_step_fused(float* mu, float* var, float* pi, float* c, float* class_counts,
                                 const float* xq, int32_t xq_round_fp16, const float* x1, const float* x2,
                                 const float* gamma_class, float* scores,
                                 int32_t K, int32_t M, int32_t D, float eps, void* stream) {
    return step_fused_impl(mu, var, pi, c, class_counts, xq, xq_round_fp16, x1, x2, gamma_class, scores, K, M, D, eps, nullptr, nullptr, stream);
}

extern "C" int ua_gmm_step_fused_prepare(float* mu, float* var, float* pi, float* c, float* class_counts,
                                         const float* xq, int32_t xq_round_fp16, const float* x1, const float* x2,
                                         const float* gamma_class, float* scores,
                                         int32_t K, int32_t M, int32_t D, float eps, float* inv_var, float* row_const, void* stream) {
    UA_CHECK_PTR(inv_var); UA_CHECK_PTR(row_const); UA_CHECK_ALIGN16(inv_var);
    return step_fused_impl(mu, var, pi, c, class_counts, xq, xq_round_fp16, x1, x2, gamma_class, scores, K, M, D, eps, inv_var, row_const, stream);
}

extern "C" int ua_gmm_prepare_alignment(const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                                        float* inv_var, float* row_const, void* stream) {
    UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(inv_var); UA_CHECK_PTR(row_const);
    UA_CHECK_ALIGN16(var); UA_CHECK_ALIGN16(inv_var);
    if (int e = check_envelope(K, M, D)) return e;
    return dispatch_vec(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        hipLaunchKernelGGL(gmm_prepare_kernel<VEC>, dim3(K), dim3(M * 64), 0, ua::as_stream(stream),
                           var, pi, M, eps, inv_var, row_const);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
}

extern "C" int ua_align_forward(const float* mu, const float* inv_var, const float* row_const, const float* e,
                                int32_t B, int32_t K, int32_t M, int32_t D, float* L, float* joint, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(inv_var); UA_CHECK_PTR(row_const); UA_CHECK_PTR(e); UA_CHECK_PTR(L); UA_CHECK_PTR(joint);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(inv_var); UA_CHECK_ALIGN16(e);
    if (B <= 0) return UA_EINVAL;
    if (int er = check_envelope(K, M, D)) return er;
    return dispatch_vec(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        static const int q = [] { const char* e = getenv("UA_ALIGN_FWD_Q"); return e ? atoi(e) : 4; }();
        if (q >= 20)      hipLaunchKernelGGL((align_fwd_kernel<VEC, 20>), dim3(K, (B + 19) / 20), dim3(M * 64), 0, ua::as_stream(stream), mu, inv_var, row_const, e, B, K, M, L, joint);
        else if (q >= 10) hipLaunchKernelGGL((align_fwd_kernel<VEC, 10>), dim3(K, (B + 9) / 10), dim3(M * 64), 0, ua::as_stream(stream), mu, inv_var, row_const, e, B, K, M, L, joint);
        else              hipLaunchKernelGGL((align_fwd_kernel<VEC, 4>), dim3(K, (B + 3) / 4), dim3(M * 64), 0, ua::as_stream(stream), mu, inv_var, row_const, e, B, K, M, L, joint);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
}

extern "C" int ua_align_backward(const float* mu, const float* inv_var, const float* e, const float* grad_ll,
                                 int32_t B, int32_t K, int32_t M, int32_t D, float* grad_e, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(inv_var); UA_CHECK_PTR(e); UA_CHECK_PTR(grad_ll); UA_CHECK_PTR(grad_e);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(inv_var); UA_CHECK_ALIGN16(e);
    if (B <= 0 || K <= 0 || M <= 0 || D <= 0) return UA_EINVAL;
    if (D % 256 != 0) return UA_EUNSUPPORTED;
    hipLaunchKernelGGL(align_bwd_kernel, dim3(B, D / 256), dim3(1024), 0, ua::as_stream(stream),
                       mu, inv_var, e, grad_ll, K * M, D, grad_e);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_align_step(const float* mu, const float* inv_var, const float* L, const float* joint,
                             const float* text0, float* res, float* adam_m, float* adam_v, float* e, float* norm,
                             int32_t K, int32_t M, int32_t D, int32_t step, double lr, double beta1, double beta2,
                             double adam_eps, float* loss, float* grad_ll, float* grad_e, int32_t* tickets,
                             void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(inv_var); UA_CHECK_PTR(L); UA_CHECK_PTR(joint); UA_CHECK_PTR(text0);
    UA_CHECK_PTR(res); UA_CHECK_PTR(adam_m); UA_CHECK_PTR(adam_v); UA_CHECK_PTR(e); UA_CHECK_PTR(norm);
    UA_CHECK_PTR(grad_e); UA_CHECK_PTR(tickets);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(inv_var); UA_CHECK_ALIGN16(e);
    if (K <= 0 || M <= 0 || D <= 0 || step <= 0) return UA_EINVAL;
    if (D % 256 != 0 || K > 128) return UA_EUNSUPPORTED;          // 2 K^2 floats of LDS next to 41 KB static
    AlignStepArgs a{};
    a.mu = mu; a.inv_var = inv_var; a.L = L; a.joint = joint; a.text0 = text0; a.res = res; a.am = adam_m; a.av = adam_v;
    a.e = e; a.nrm = norm; a.ge = grad_e; a.loss_out = loss; a.grad_ll = grad_ll; a.tickets = tickets;
    a.K = K; a.M = M; a.D = D;
    // scalar bookkeeping exactly as ua_residual_adam (torch.optim.adam._single_tensor_adam, python floats = double)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.adam_eps = (float)adam_eps;
    const size_t lds = (size_t)(2 * K * K + 4 * K) * sizeof(float);
    if (lds > 16 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(align_step_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return UA_ELAUNCH;
    hipLaunchKernelGGL(align_step_kernel, dim3(K, D / 256), dim3(1024), lds, ua::as_stream(stream), a);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
