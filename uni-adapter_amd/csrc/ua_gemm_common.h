// Shared pieces of the two GEMM kernels (exact-f32 MFMA and 3xf16-split MFMA): epilogue arguments,
// the LDS-staged float4 epilogue and the split-precision ("H32L32") storage format.
#pragma once
#include "ua_common.h"

namespace ua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// H32L32 split-precision storage.  A (rows, cols) fp32 matrix x is stored with the SAME row pitch
// as fp32 (ld*4 bytes); every group of 32 consecutive columns is 32 fp16 "hi" values (64 B)
// followed by 32 fp16 "lo" values (64 B) with  x*s = hi + lo  (s a power of two chosen so that hi
// stays in fp16 range and lo is a normal number).  hi = rn16(x*s), lo = rn16(x*s - hi): the pair
// carries 22 significant bits, and a product of two pairs dropped to hi*hi + hi*lo + lo*hi has a
// relative error of 2^-21 - below the accumulated fp32 rounding of a K >= 64 dot product.
// One 32-column group of a row is exactly one 128-byte line = one GEMM K-slab row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split4(float4 v, float s, f16x4& hi, f16x4& lo) {
    const float x = v.x * s, y = v.y * s, z = v.z * s, w = v.w * s;
    hi[0] = (_Float16)x; hi[1] = (_Float16)y; hi[2] = (_Float16)z; hi[3] = (_Float16)w;
    lo[0] = (_Float16)(x - (float)hi[0]); lo[1] = (_Float16)(y - (float)hi[1]);
    lo[2] = (_Float16)(z - (float)hi[2]); lo[3] = (_Float16)(w - (float)hi[3]);
}
// byte offset of column c (multiple of 4) inside a row of an H32L32 matrix; lo part is +64
__device__ __forceinline__ size_t split_col_off(int c) { return (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2; }

// ---- helpers of the MFMA similarity / alignment kernels (ua_gemm_split.hip, ua_align_mfma.hip) ----
__device__ __forceinline__ float pow2_scale(float amax) {
    // largest power of two s with amax * s <= 30000 (fp16 max 65504), clamped to [2^-14, 2^14]
    if (!(amax > 0.f)) return 1.f;
    int e;
    frexpf(30000.0f / amax, &e);                       // 30000/amax = f * 2^e, f in [0.5, 1)
    e = max(-14, min(14, e - 1));
    return ldexpf(1.0f, e);
}

// Contraction layout: K-tile 2i holds features d = 32i..32i+31 of the first part (x^2 | 1/v), K-tile 2i+1 the same
// features of the second part (x | -2 mu/v), so the running sum of a GEMM accumulator follows
// sum_{d' < d} ((x - mu)^2 - mu^2)/v instead of first climbing to sum x^2/v (~5e3) and then coming back down: the fp32
// roundings of the accumulate chain happen at about half the magnitude.
__device__ __forceinline__ int sim_col(int part, int d) { return (d >> 5) * 64 + part * 32 + (d & 31); }

__device__ __forceinline__ void store_split4(unsigned char* row, int c, float4 v, float s) {
    f16x4 hi, lo;
    split4(v, s, hi, lo);
    unsigned char* q = row + split_col_off(c);
    *reinterpret_cast<f16x4*>(q) = hi;
    *reinterpret_cast<f16x4*>(q + 64) = lo;
}


inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct MfmaWs { unsigned char* Wop; float* rc; float* colscale; unsigned char* Aop; float* row_scale; float* joint; size_t total; };

inline MfmaWs mfma_ws_layout(unsigned char* base, int B, int K, int M, int D, bool need_joint) {
    MfmaWs w{};
    size_t o = 0;
    const size_t KM = (size_t)K * M;
    w.Wop = base + o; o += align256(KM * 2 * D * 4);
    w.rc = reinterpret_cast<float*>(base + o); o += align256(KM * 4);
    w.colscale = reinterpret_cast<float*>(base + o); o += align256(KM * 4);
    w.Aop = base + o; o += align256((size_t)B * 2 * D * 4);
    w.row_scale = reinterpret_cast<float*>(base + o); o += align256((size_t)B * 4);
    if (need_joint) { w.joint = reinterpret_cast<float*>(base + o); o += align256((size_t)B * KM * 4); }
    w.total = o;
    return w;
}


// State-side operand row n of the MFMA similarity: [1/v | -2 mu/v] as split fp16 pairs with one power-of-two scale per row,
// rc[n] = ln(pi_n + 1e-10) - 1/2 sum ln v - 1/2 sum mu^2/v, colscale[n] = -1/(2 s_n).  One wave (all 64 lanes) per row.
template <int VEC>
__device__ __forceinline__ void gmm_operand_row(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ pi,
                                                int n, float eps, int lane, unsigned char* __restrict__ Wop,
                                                float* __restrict__ rc, float* __restrict__ colscale) {
    constexpr int D = 256 * VEC;
    const float4* pm = reinterpret_cast<const float4*>(mu + (size_t)n * D);
    const float4* pv = reinterpret_cast<const float4*>(var + (size_t)n * D);
    float4 a[VEC], b[VEC];
    double ld = 0.0, mq = 0.0;
    float amax = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float4 m4 = pm[v * 64 + lane], v4 = pv[v * 64 + lane];
#define UA_OP(F)                                                                           \
        {                                                                                  \
            const float rv = fmaxf(v4.F + eps, 1e-8f);                 /* dota_mixture.py:117-119 */ \
            const float iv = 1.0f / rv;                                                    \
            ld += (double)logf(rv);                                                        \
            mq += (double)((m4.F * m4.F) * iv);                                            \
            a[v].F = iv; b[v].F = (-2.0f * m4.F) * iv;                                     \
            amax = fmaxf(amax, fmaxf(iv, fabsf(b[v].F)));                                  \
        }
        UA_OP(x) UA_OP(y) UA_OP(z) UA_OP(w)
#undef UA_OP
    }
    ld = wave_sum(ld); mq = wave_sum(mq);
    amax = wave_max(amax);
    const float s = pow2_scale(amax);
    unsigned char* row = Wop + (size_t)n * (2 * D) * 4;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = (v * 64 + lane) * 4;
        store_split4(row, sim_col(0, c), a[v], s);
        store_split4(row, sim_col(1, c), b[v], s);
    }
    if (lane == 0) {
        rc[n] = (float)((double)logf(pi[n] + 1e-10f) - 0.5 * ld - 0.5 * mq);
        colscale[n] = -0.5f / s;
    }
}

struct EpiArgs {
    int M, N;                      // logical output rows / INPUT columns of the accumulator tile
    float acc_scale;               // accumulators are multiplied by this first (split kernel: 1/(sA*sW))
    const float* bias; const float* scale; const float* shift;
    const float* rga; long long ldg; int group_rows;
    int act;                       // 0 none, 1 relu, 2 gelu(erf), 3 swiglu
    const float* residual; long long ldr;
    float* C; long long ldc;
    float* pool; long long ldp; int pool_rows;   // 64 (Uni3D groups) or 32 (PointBERT groups)
    unsigned char* Cs; long long ldcs; float cs_scale;     // optional H32L32 copy of the output
    // internal extensions (not in the public ua_gemm_epilogue; used by the MFMA similarity kernel, ua_gmm_loglik_mfma):
    const float* row_scale;        // per-output-row factor applied to the accumulator (operand rows carry their own power-of-two scale)
    float* lse; long long ld_lse; int lse_group;   // act 4: logsumexp over groups of lse_group (4, 8 or 16) consecutive columns -> lse[m, n / group]
    // LayerNorm folded around the GEMM (split kernel only; see ua_gemm_epilogue in include/uniadapter.h)
    float2* stats; long long ld_stats;             // out: per (column tile, row) partial (sum, sum of squares) of the output values
    const float2* ln_mr; const float* ln_c;        // in: per-row (mean, rstd) of A's rows and the row sums of W: v = rstd*(acc - mean*ln_c[n]) + bias[n]
    int lean_ok;                                   // the compile-time-specialised row loops may be used (UA_EPI_LEAN=0: A/B aid)
    const float2* ln_in; long long ln_in_ld; int ln_parts; double ln_inv_n, ln_eps;   // in (instead of ln_mr): the producer's partials, reduced by the consumer
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// x * sigmoid(x) with the hardware 2^x and reciprocal (about 3 ulp; the libm expf + IEEE division form costs
// ~25 VALU instructions per element, which made the SwiGLU epilogue 1.6x as long as a plain one)
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

constexpr int EP_LD = 68;          // floats per staged row (64 + 4: conflict-free ds_read_b128)

// Write the four 32x32 accumulator blocks of a 64x64 sub-tile into the wave's staging area.
// acc[i][j][e]: row i*32 + (e&3) + 8*(e>>2) + 4*lh, column j*32 + li.
__device__ __forceinline__ void stage_64x64(float* stage, const f32x16& a00, const f32x16& a01, const f32x16& a10,
                                            const f32x16& a11, int li, int lh) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
        stage[r * EP_LD + li] = a00[e];
        stage[r * EP_LD + 32 + li] = a01[e];
        stage[(32 + r) * EP_LD + li] = a10[e];
        stage[(32 + r) * EP_LD + 32 + li] = a11[e];
    }
}

// Same for 16x16 accumulator blocks (v_mfma_f32_16x16x32_f16): acc[ib0 + ib][jb][j] is row
// ib*16 + 4*(lane>>4) + j, column jb*16 + (lane&15) of the 64x64 sub-tile (ib, jb = 0..3).
__device__ __forceinline__ void stage_64x64_b16(float* stage, const f32x4 (&acc)[8][4], int ib0, int lane) {
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                stage[(ib * 16 + 4 * lq + j) * EP_LD + jb * 16 + l15] = acc[ib0 + ib][jb][j];
}

// Staging with the folded LayerNorm applied: staged = acc * (as * rstd[row]) - mean[row] * rstd[row] * colsum[col].
// mr: (mean, rstd) of the 64 rows of this half, lc: column sums of the wave's 64 columns (both in LDS).
__device__ __forceinline__ void stage_64x64_b16_fold(float* stage, const f32x4 (&acc)[8][4], int ib0, int lane, float as,
                                                     const float2* mr, const float* lc) {
    const int l15 = lane & 15, lq = lane >> 4;
    float c[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) c[jb] = lc[jb * 16 + l15];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 m = mr[ib * 16 + 4 * lq + j];
            const float a = as * m.y, b = -m.x * m.y;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                stage[(ib * 16 + 4 * lq + j) * EP_LD + jb * 16 + l15] = __builtin_fmaf(acc[ib0 + ib][jb][j], a, b * c[jb]);
        }
}

// float4 epilogue of one staged 64x64 sub-tile owned by one wave.  mbase: first output row;
// ncol_base: first INPUT column of the sub-tile (multiple of 64).  Every lane handles 4 consecutive
// columns (c4) of rows lr, lr+4, ...; C is written in 256-byte row segments with 16-byte stores.
// FULL: the sub-tile lies entirely inside the matrix - no per-lane row / column guards.  ACT is a
// template parameter so the activation switch is outside the row loop; the optional operands are
// wave-uniform (scalar) branches and all row addresses advance by pointer increments.
// LEAN != 0: a launch whose optional operands are known at compile time (bit 0 fp32 output, bit 1 split output, bit 2 residual,
// bit 3 row statistics, bit 4 64-row max-pool, bit 5 BN affine + row-group addend; never a per-row scale or a logsumexp): the row loop is
// VALU-issue-bound (~80-110 instructions per iteration in the generic form, 32 iterations per wave), so the wave-uniform
// flag tests and the pooling maxima it never stores are compiled out.  Same arithmetic, same bytes.
template <bool FULL, int ACT, int LEAN = 0>
__device__ __forceinline__ void epilogue_64x64_impl(const EpiArgs& g, const float* stage, int lane, int mbase, int ncol_base, float2* part, bool prescaled) {
    // the library is built with -ffp-contract=off for the element-wise cache kernels; here acc*scale + bias and the BN
    // affine may fuse (one rounding instead of two, half the VALU instructions of the row loop)
#pragma clang fp contract(fast)
    constexpr bool lean = LEAN != 0;
    const bool have_res = lean ? (LEAN & 4) != 0 : g.residual != nullptr;
    const bool have_rga = lean ? (LEAN & 32) != 0 : g.rga != nullptr;
    const bool have_rs = lean ? false : g.row_scale != nullptr;
    const bool have_pool = lean ? (LEAN & 16) != 0 : g.pool != nullptr;
    constexpr bool swiglu = ACT == 3;
    constexpr bool lse = ACT == 4;
    const int lr = lane >> 4, c4 = (lane & 15) * 4;
    const int ncol0 = ncol_base + c4;                                // input column of .x
    const int nout0 = swiglu ? ncol_base / 2 + c4 : ncol0;           // output column of .x
    const int n_out = swiglu ? g.N / 2 : g.N;
    const bool active = swiglu ? (c4 < 32) : true;                   // SwiGLU: gate lanes produce output
    auto ld4 = [&](const float* p, int n, int lim, float fill) {     // guarded float4 of a per-column vector
        float4 r = make_float4(fill, fill, fill, fill);
        if (p) {
            if (FULL || n + 3 < lim) r = *reinterpret_cast<const float4*>(p + n);
            else { if (n < lim) r.x = p[n]; if (n + 1 < lim) r.y = p[n + 1]; if (n + 2 < lim) r.z = p[n + 2]; }
        }
        return r;
    };
    auto st4 = [&](float* q, const float4& v) {
        if (FULL || nout0 + 3 < n_out) *reinterpret_cast<float4*>(q) = v;
        else { if (nout0 < n_out) q[0] = v.x; if (nout0 + 1 < n_out) q[1] = v.y; if (nout0 + 2 < n_out) q[2] = v.z; }
    };
    const float as = prescaled ? 1.0f : g.acc_scale;               // (folded LayerNorm: the staging step applied acc_scale * rstd already)
    const float4 b4 = ld4(g.bias, ncol0, g.N, 0.f);
    const float4 s4 = ld4(lean && !(LEAN & 32) ? nullptr : g.scale, ncol0, g.N, 1.f);   // v = (acc*as + bias + rga) * scale + shift
    const float4 t4 = ld4(lean && !(LEAN & 32) ? nullptr : g.shift, ncol0, g.N, 0.f);
    // Every global LOAD of the sub-tile is issued here, before the first store: vmcnt counts loads and
    // stores in one in-order counter, so a load inside the store loop would make each iteration wait
    // for the previous iteration's store to be acknowledged.
    // row-group addend (group_rows is a multiple of 32): one vector for rows 0..31, one for rows 32..63
    float4 ba = b4, bb = b4;
    if (have_rga) {
        if (FULL || mbase < g.M) {
            const float4 a = ld4(g.rga + (size_t)(mbase / g.group_rows) * g.ldg, ncol0, g.N, 0.f);
            ba.x += a.x; ba.y += a.y; ba.z += a.z; ba.w += a.w;
        }
        if (FULL || mbase + 32 < g.M) {
            const float4 a = ld4(g.rga + (size_t)((mbase + 32) / g.group_rows) * g.ldg, ncol0, g.N, 0.f);
            bb.x += a.x; bb.y += a.y; bb.z += a.z; bb.w += a.w;
        }
    }
    const int m_first = mbase + lr;
    float4 res[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) res[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have_res && active) {
        const float* rp = g.residual + (size_t)m_first * g.ldr;            // ld4 adds the column
        const size_t rstep = (size_t)4 * g.ldr;
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (FULL || m_first + p * 4 < g.M) res[p] = ld4(rp + p * rstep, nout0, n_out, 0.f);
    }
    float rs[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) rs[p] = 1.f;
    if (have_rs) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (FULL || m_first + p * 4 < g.M) rs[p] = g.row_scale[m_first + p * 4];
    }
    const bool have_stats = lean ? (LEAN & 8) != 0 : part != nullptr;    // wave-uniform
    const bool affine = lean ? (LEAN & 32) != 0 : (g.scale || g.shift);
    float4 pm = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);     // rows 0..63 (pool 64) / 0..31 (pool 32)
    float4 pm2 = pm;                                                          // rows 32..63 when pooling 32-row groups
    const bool pool32 = lean ? false : g.pool_rows == 32;
    const float* sp = stage + lr * EP_LD + c4;
    // wave-uniform flags stay scalar: testing the per-lane pointers instead makes every `if` an exec-mask branch
    const bool have_c = lean ? (LEAN & 1) != 0 : g.C != nullptr, have_s = lean ? (LEAN & 2) != 0 : g.Cs != nullptr;
    float* cp = g.C + (size_t)m_first * g.ldc + nout0;
    unsigned char* qp = g.Cs + (size_t)m_first * g.ldcs * 4 + split_col_off(nout0);
    const size_t cstep = (size_t)4 * g.ldc, qstep = (size_t)16 * g.ldcs;
    // all 16 staged rows are requested from LDS before the first one is used (one latency instead of sixteen)
    float4 vin[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) vin[p] = *reinterpret_cast<const float4*>(sp + p * 4 * EP_LD);
    // Specialised loops: the 16 lanes of a row leave their (sum, sumsq) partials in the wave's staging area - free once the rows
    // above are in registers; LDS operations of a wave execute in order - and one lane per row adds them after the loop in the
    // association of the DPP tree of the generic loop (bit-identical statistics, ~1/4 of the instructions).  17 slots per row.
    float2* const scr = reinterpret_cast<float2*>(const_cast<float*>(stage));
    constexpr bool lds_stats = lean && FULL;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        float4 v = vin[p];
        const float4 bp = p < 8 ? ba : bb;
        const float asr = as * rs[p];
        v.x = v.x * asr + bp.x; v.y = v.y * asr + bp.y; v.z = v.z * asr + bp.z; v.w = v.w * asr + bp.w;
        if (affine) { v.x = v.x * s4.x + t4.x; v.y = v.y * s4.y + t4.y; v.z = v.z * s4.z + t4.z; v.w = v.w * s4.w + t4.w; }
        if constexpr (swiglu && lean && FULL) {
            // SwiGLU on BOTH lanes of a gate / up pair (lanes L and L+8 of a 16-lane row).  The generic form below produces the four
            // outputs of columns 4L..4L+3 on the gate lane and leaves the up lane idle through silu, split and statistics; here the
            // pair swaps two values (one row_ror:8 each way) and each lane finishes two columns: gate lane 4L, 4L+1, up lane
            // 4L+2, 4L+3.  Same values, same bytes, and the statistics associate exactly as in the generic loop
            // ((x + y) + (z + w) per pair first, then the 8-lane tree), so they are bit-identical too.
            const bool gl = (lane & 8) == 0;                                    // gate lane
            const float sa = gl ? v.z : v.x, sb = gl ? v.w : v.y;              // what the partner needs
            const float ra = dpp_f<0x128, 0xF>(sa), rb = dpp_f<0x128, 0xF>(sb);   // row_ror:8 = the partner's value
            const float o0 = silu(gl ? v.x : ra) * (gl ? ra : v.z);
            const float o1 = silu(gl ? v.y : rb) * (gl ? rb : v.w);
            const int oc = ncol_base / 2 + 4 * ((lane & 15) & 7) + (gl ? 0 : 2);
            if (have_stats) scr[(lr + 4 * p) * 17 + (lane & 15)] = make_float2(o0 + o1, o0 * o0 + o1 * o1);
            const size_t row = (size_t)(m_first + p * 4);
            if (have_c) *reinterpret_cast<float2*>(g.C + row * g.ldc + oc) = make_float2(o0, o1);
            if (have_s) {
                const float x0 = o0 * g.cs_scale, x1 = o1 * g.cs_scale;
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                f16x2 hi, lo;
                hi[0] = (_Float16)x0; hi[1] = (_Float16)x1;
                lo[0] = (_Float16)(x0 - (float)hi[0]); lo[1] = (_Float16)(x1 - (float)hi[1]);
                unsigned char* q = g.Cs + row * g.ldcs * 4 + split_col_off(oc);
                *reinterpret_cast<f16x2*>(q) = hi; *reinterpret_cast<f16x2*>(q + 64) = lo;
            }
            continue;
        }
        if (ACT == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (ACT == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (swiglu) {
            // the up-projection of column c sits 32 columns to the right = 8 lanes up in this row
            float4 u;
            u.x = __shfl_down(v.x, 8); u.y = __shfl_down(v.y, 8); u.z = __shfl_down(v.z, 8); u.w = __shfl_down(v.w, 8);
            v.x = silu(v.x) * u.x; v.y = silu(v.y) * u.y; v.z = silu(v.z) * u.z; v.w = silu(v.w) * u.w;
        }
        v.x += res[p].x; v.y += res[p].y; v.z += res[p].z; v.w += res[p].w;
        if (lse) {
            // logsumexp over lse_group consecutive columns = 1, 2 or 4 adjacent lanes of this row (torch.logsumexp: max, exp, sum, log)
            const int G = g.lse_group;                                 // wave-uniform: 4, 8 or 16
            float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            if (G >= 8) mx = fmaxf(mx, dpp_f<0xB1, 0xF>(mx));          // quad_perm [1,0,3,2]
            if (G == 16) mx = fmaxf(mx, dpp_f<0x4E, 0xF>(mx));         // quad_perm [2,3,0,1]
            float sm = (expf(v.x - mx) + expf(v.y - mx)) + (expf(v.z - mx) + expf(v.w - mx));
            if (G >= 8) sm += dpp_f<0xB1, 0xF>(sm);
            if (G == 16) sm += dpp_f<0x4E, 0xF>(sm);
            const float out = mx + logf(sm);
            const bool writer = (lane & ((G >> 2) - 1)) == 0;
            if (writer && (FULL || (m_first + p * 4 < g.M && ncol0 < g.N)))
                g.lse[(size_t)(m_first + p * 4) * g.ld_lse + ncol0 / G] = out;
        }
        const bool lane_ok = (FULL || m_first + p * 4 < g.M) && active;         // FULL and not SwiGLU: constant true
        if (have_stats) {
            // (sum, sum of squares) of this row's 64 (SwiGLU: 32) output columns: 4 columns per lane, 16 lanes per row
            float4 t = v;
            if (!lane_ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (!FULL) { if (nout0 >= n_out) t.x = 0.f; if (nout0 + 1 >= n_out) t.y = 0.f; if (nout0 + 2 >= n_out) t.z = 0.f; if (nout0 + 3 >= n_out) t.w = 0.f; }
            float s1 = (t.x + t.y) + (t.z + t.w);
            float s2 = (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
            if constexpr (lds_stats) { scr[(lr + 4 * p) * 17 + (lane & 15)] = make_float2(s1, s2); }
            else {
            s1 += dpp_f<0xB1, 0xF>(s1);  s2 += dpp_f<0xB1, 0xF>(s2);           // quad_perm [1,0,3,2]
            s1 += dpp_f<0x4E, 0xF>(s1);  s2 += dpp_f<0x4E, 0xF>(s2);           // quad_perm [2,3,0,1]
            s1 += dpp_f<0x141, 0xF>(s1); s2 += dpp_f<0x141, 0xF>(s2);          // row_half_mirror
            s1 += dpp_f<0x140, 0xF>(s1); s2 += dpp_f<0x140, 0xF>(s2);          // row_mirror
            if ((lane & 15) == 0) part[lr + 4 * p] = make_float2(s1, s2);
            }
        }
        if (FULL && !swiglu) {
            if (have_c) *reinterpret_cast<float4*>(cp + p * cstep) = v;
            if (have_s) {
                f16x4 hi, lo;
                split4(v, g.cs_scale, hi, lo);
                *reinterpret_cast<f16x4*>(qp + p * qstep) = hi;
                *reinterpret_cast<f16x4*>(qp + p * qstep + 64) = lo;
            }
        } else if (lane_ok) {
            if (have_c) st4(cp + p * cstep, v);
            if (have_s && (FULL || nout0 + 3 < n_out)) {                  // n_out % 4 == 0 is required with Cs
                f16x4 hi, lo;
                split4(v, g.cs_scale, hi, lo);
                *reinterpret_cast<f16x4*>(qp + p * qstep) = hi;
                *reinterpret_cast<f16x4*>(qp + p * qstep + 64) = lo;
            }
        }
        if ((lean ? (LEAN & 16) != 0 : true) && lane_ok) {
            if (pool32 && p >= 8) { pm2.x = fmaxf(pm2.x, v.x); pm2.y = fmaxf(pm2.y, v.y); pm2.z = fmaxf(pm2.z, v.z); pm2.w = fmaxf(pm2.w, v.w); }
            else { pm.x = fmaxf(pm.x, v.x); pm.y = fmaxf(pm.y, v.y); pm.z = fmaxf(pm.z, v.z); pm.w = fmaxf(pm.w, v.w); }
        }
    }
    if constexpr (lds_stats) {
        if (have_stats) {
            // lane r adds the 16 partials of row r: xor-1 pairs, xor-2 pairs, the two quads of an 8-lane half (lane 7's order
            // for the upper quad), the two halves - what lane 0 of the row holds after the generic loop's four DPP steps
            const float2* rp = scr + lane * 17;
            float2 l[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) l[j] = rp[j];
            auto add = [](float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); };
            float2 tot;
            if constexpr (swiglu) {
                float2 P[8];                                                    // the generic loop's per-lane values: gate + up lane
#pragma unroll
                for (int j = 0; j < 8; ++j) P[j] = add(l[j], l[j + 8]);
                tot = add(add(add(P[0], P[1]), add(P[2], P[3])), add(add(P[7], P[6]), add(P[5], P[4])));
            } else {
                const float2 c0 = add(add(add(l[0], l[1]), add(l[2], l[3])), add(add(l[7], l[6]), add(l[5], l[4])));
                const float2 c15 = add(add(add(l[15], l[14]), add(l[13], l[12])), add(add(l[8], l[9]), add(l[10], l[11])));
                tot = add(c0, c15);
            }
            part[lane] = tot;
        }
    }
    if (have_pool) {                                                 // max over this wave's 64 rows
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            pm.x = fmaxf(pm.x, __shfl_xor(pm.x, o)); pm.y = fmaxf(pm.y, __shfl_xor(pm.y, o));
            pm.z = fmaxf(pm.z, __shfl_xor(pm.z, o)); pm.w = fmaxf(pm.w, __shfl_xor(pm.w, o));
        }
        if (pool32) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                pm2.x = fmaxf(pm2.x, __shfl_xor(pm2.x, o)); pm2.y = fmaxf(pm2.y, __shfl_xor(pm2.y, o));
                pm2.z = fmaxf(pm2.z, __shfl_xor(pm2.z, o)); pm2.w = fmaxf(pm2.w, __shfl_xor(pm2.w, o));
            }
        }
        if (lr == 0 && active && (FULL || mbase < g.M)) {
            if (!pool32) st4(g.pool + (size_t)(mbase / 64) * g.ldp + nout0, pm);
            else {
                st4(g.pool + (size_t)(mbase / 32) * g.ldp + nout0, pm);
                if (FULL || mbase + 32 < g.M) st4(g.pool + (size_t)(mbase / 32 + 1) * g.ldp + nout0, pm2);
            }
        }
    }
}

template <bool FULL>
__device__ __forceinline__ void epilogue_64x64_act(const EpiArgs& g, const float* stage, int lane, int mbase, int ncol_base, float2* part, bool prescaled) {
    switch (g.act) {                                                 // wave-uniform
        case 1:  epilogue_64x64_impl<FULL, 1>(g, stage, lane, mbase, ncol_base, part, prescaled); break;
        case 2:  epilogue_64x64_impl<FULL, 2>(g, stage, lane, mbase, ncol_base, part, prescaled); break;
        case 3:  epilogue_64x64_impl<FULL, 3>(g, stage, lane, mbase, ncol_base, part, prescaled); break;
        case 4:  epilogue_64x64_impl<FULL, 4>(g, stage, lane, mbase, ncol_base, part, prescaled); break;
        default: epilogue_64x64_impl<FULL, 0>(g, stage, lane, mbase, ncol_base, part, prescaled); break;
    }
}

// part: LDS slots of this wave's 64 rows for the row statistics (stats output requested), else nullptr
__device__ __forceinline__ void epilogue_64x64(const EpiArgs& g, const float* stage, int lane, int mbase, int ncol_base, float2* part = nullptr,
                                               bool prescaled = false) {
    if (mbase + 64 <= g.M && ncol_base + 64 <= g.N) {                                                    // wave-uniform
        // the transformer blocks' launches (point_encoder.py: qkv | fc1 | proj, fc2 with and without the folded LayerNorm's outputs)
        const int code = (g.C ? 1 : 0) | (g.Cs ? 2 : 0) | (g.residual ? 4 : 0) | (part ? 8 : 0);
        if (g.lean_ok && !g.scale && !g.shift && !g.rga && !g.row_scale && !g.pool && (g.act == 0 || g.act == 3)) {
            if (g.act == 0) {
                if (code == 2)  { epilogue_64x64_impl<true, 0, 2>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
                if (code == 15) { epilogue_64x64_impl<true, 0, 15>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
                if (code == 5)  { epilogue_64x64_impl<true, 0, 5>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
            } else {
                if (code == 10) { epilogue_64x64_impl<true, 3, 10>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
                if (code == 2)  { epilogue_64x64_impl<true, 3, 2>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
                if (code == 1)  { epilogue_64x64_impl<true, 3, 1>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
            }
        }
        // the mini-PointNet's per-point convolutions (K = 128 .. 512: the epilogue is up to three quarters of a tile's time):
        // conv2 = 64-row max-pool + split copy, conv3 = row-group addend + BN affine + ReLU -> split copy, conv4 = max-pool only
        if (g.lean_ok == 1 && !g.row_scale && !g.residual && !part && !g.C) {
            const bool aff = g.scale && g.shift && g.rga, none = !g.scale && !g.shift && !g.rga;
            const bool pool64 = g.pool && g.pool_rows == 64;
            if (g.act == 0 && none && pool64 && g.Cs)  { epilogue_64x64_impl<true, 0, 2 | 16>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
            if (g.act == 0 && none && pool64 && !g.Cs) { epilogue_64x64_impl<true, 0, 16>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
            if (g.act == 1 && aff && !g.pool && g.Cs)  { epilogue_64x64_impl<true, 1, 2 | 32>(g, stage, lane, mbase, ncol_base, part, prescaled); return; }
        }
        epilogue_64x64_act<true>(g, stage, lane, mbase, ncol_base, part, prescaled);
    } else epilogue_64x64_act<false>(g, stage, lane, mbase, ncol_base, part, prescaled);
}

// Host-side validation + translation of the public epilogue descriptor (shared by both entry points).
static inline int fill_epi(EpiArgs& e, const ua_gemm_epilogue* ep, int M, int N, float* C, long long ldc) {
    e = EpiArgs{};
    e.M = M; e.N = N; e.acc_scale = 1.0f; e.C = C; e.ldc = ldc;
    static const int lean_ok = [] { const char* v = getenv("UA_EPI_LEAN"); return v ? atoi(v) : 1; }();
    e.lean_ok = lean_ok;
    if (ep) {
        e.bias = ep->bias; e.scale = ep->scale; e.shift = ep->shift;
        e.rga = ep->rowgroup_add; e.ldg = ep->ldg; e.group_rows = ep->group_rows;
        e.act = ep->act; e.residual = ep->residual; e.ldr = ep->ldr;
        e.pool = ep->pool_out; e.ldp = ep->ldp; e.pool_rows = ep->pool_rows;
        e.Cs = reinterpret_cast<unsigned char*>(ep->split_out); e.ldcs = ep->ld_split; e.cs_scale = ep->split_scale;
        e.stats = reinterpret_cast<float2*>(ep->row_stats_out); e.ld_stats = ep->ld_stats;
        e.ln_mr = reinterpret_cast<const float2*>(ep->ln_mean_rstd); e.ln_c = ep->ln_colsum;
        e.ln_in = reinterpret_cast<const float2*>(ep->ln_stats_in); e.ln_in_ld = ep->ln_stats_ld; e.ln_parts = ep->ln_parts;
        if (e.ln_in && e.ln_mr) return UA_EINVAL;
        if ((e.ln_mr != nullptr || e.ln_in != nullptr) != (e.ln_c != nullptr)) return UA_EINVAL;
        if (e.ln_in) {
            if (ep->ln_parts < 1 || ep->ln_n < 1 || ep->ln_stats_ld < M || !(ep->ln_eps >= 0.f)) return UA_EINVAL;
            if (ep->ln_parts > 4) return UA_EUNSUPPORTED;              // (LDS budget; more tiles go through ua_row_stats_finalize)
            e.ln_inv_n = 1.0 / ep->ln_n; e.ln_eps = (double)ep->ln_eps;
        }
        if (e.stats && e.ld_stats < M) return UA_EINVAL;
        if (e.act < 0 || e.act > 3) return UA_EINVAL;
        if (e.rga && e.group_rows <= 0) return UA_EINVAL;
        if (e.rga && e.group_rows % 32 != 0) return UA_EUNSUPPORTED;    // the epilogue holds one addend per 32 rows
        if (e.pool && ((ep->pool_rows != 64 && ep->pool_rows != 32) || M % ep->pool_rows != 0)) return UA_EUNSUPPORTED;
        if (e.act == 3 && (N % 64 != 0 || e.rga || e.scale || e.shift)) return UA_EUNSUPPORTED;
    }
    if (!C && !e.pool && !e.Cs) return UA_EINVAL;
    const int n_out = e.act == 3 ? N / 2 : N;
    if (C && ldc < n_out) return UA_EINVAL;
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    // float4 epilogue: row starts of C / residual / pool / rowgroup_add must be 16-byte aligned
    if (C && (ldc % 4 || mis(C))) return UA_EALIGN;
    if (e.residual && (e.ldr % 4 || mis(e.residual))) return UA_EALIGN;
    if (e.pool && (e.ldp % 4 || mis(e.pool))) return UA_EALIGN;
    if (e.rga && (e.ldg % 4 || mis(e.rga))) return UA_EALIGN;
    if ((e.bias && mis(e.bias)) || (e.scale && mis(e.scale)) || (e.shift && mis(e.shift))) return UA_EALIGN;
    if ((e.ln_c && mis(e.ln_c)) || (reinterpret_cast<uintptr_t>(e.ln_mr) & 7) || (reinterpret_cast<uintptr_t>(e.stats) & 7) ||
        (reinterpret_cast<uintptr_t>(e.ln_in) & 7)) return UA_EALIGN;
    if (e.Cs) {
        if (mis(e.Cs) || e.ldcs % 32 || e.ldcs < n_out || n_out % 4) return UA_EALIGN;
        if (!(e.cs_scale > 0.f)) return UA_EINVAL;
    }
    return UA_OK;
}

}  // namespace ua
