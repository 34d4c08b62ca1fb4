// Shared pieces of the two GEMM kernels (exact-f32 MFMA and 3xf16-split MFMA): epilogue arguments,
// the LDS-staged float4 epilogue and the split-precision ("H32L32") storage format.
#pragma once
#include "ua_common.h"

namespace ua {

typedef float f32x16 __attribute__((ext_vector_type(16)));
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
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

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

// float4 epilogue of one staged 64x64 sub-tile owned by one wave.  mbase: first output row;
// ncol_base: first INPUT column of the sub-tile (multiple of 64).  Every lane handles 4 consecutive
// columns (c4) of rows lr, lr+4, ...; C is written in 256-byte row segments with 16-byte stores.
__device__ __forceinline__ void epilogue_64x64(const EpiArgs& g, const float* stage, int lane, int mbase, int ncol_base) {
    const bool swiglu = g.act == 3;
    const int lr = lane >> 4, c4 = (lane & 15) * 4;
    const int ncol0 = ncol_base + c4;                                // input column of .x
    const int nout0 = swiglu ? ncol_base / 2 + c4 : ncol0;           // output column of .x
    const int n_out = swiglu ? g.N / 2 : g.N;
    const bool active = swiglu ? (c4 < 32) : true;                   // SwiGLU: gate lanes produce output
    auto ld4 = [&](const float* p, int n, int lim, float fill) {     // guarded float4 of a per-column vector
        float4 r = make_float4(fill, fill, fill, fill);
        if (p) {
            if (n + 3 < lim) r = *reinterpret_cast<const float4*>(p + n);
            else { if (n < lim) r.x = p[n]; if (n + 1 < lim) r.y = p[n + 1]; if (n + 2 < lim) r.z = p[n + 2]; }
        }
        return r;
    };
    const float4 b4 = ld4(g.bias, ncol0, g.N, 0.f);
    const float4 s4 = ld4(g.scale, ncol0, g.N, 1.f);
    const float4 t4 = ld4(g.shift, ncol0, g.N, 0.f);
    const float as = g.acc_scale;
    float4 pm = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);     // rows 0..63 (pool 64) / 0..31 (pool 32)
    float4 pm2 = pm;                                                          // rows 32..63 when pooling 32-row groups
    const bool pool32 = g.pool_rows == 32;
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int r = p * 4 + lr, m = mbase + r;
        float4 v = *reinterpret_cast<const float4*>(stage + r * EP_LD + c4);
        v.x = v.x * as + b4.x; v.y = v.y * as + b4.y; v.z = v.z * as + b4.z; v.w = v.w * as + b4.w;
        const bool row_ok = m < g.M;
        if (g.rga && row_ok) {
            const float4 a = ld4(g.rga + (size_t)(m / g.group_rows) * g.ldg, ncol0, g.N, 0.f);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (g.scale) { v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w; }
        if (g.shift) { v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w; }
        if (g.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (g.act == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (swiglu) {
            // the up-projection of column c sits 32 columns to the right = 8 lanes up in this row
            float4 u;
            u.x = __shfl_down(v.x, 8); u.y = __shfl_down(v.y, 8); u.z = __shfl_down(v.z, 8); u.w = __shfl_down(v.w, 8);
            v.x = silu(v.x) * u.x; v.y = silu(v.y) * u.y; v.z = silu(v.z) * u.z; v.w = silu(v.w) * u.w;
        }
        if (row_ok && active) {
            if (g.residual) {
                const float4 a = ld4(g.residual + (size_t)m * g.ldr, nout0, n_out, 0.f);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            if (g.C) {
                float* c = g.C + (size_t)m * g.ldc + nout0;
                if (nout0 + 3 < n_out) *reinterpret_cast<float4*>(c) = v;
                else { if (nout0 < n_out) c[0] = v.x; if (nout0 + 1 < n_out) c[1] = v.y; if (nout0 + 2 < n_out) c[2] = v.z; }
            }
            if (g.Cs && nout0 + 3 < n_out) {                          // n_out % 4 == 0 is required with Cs
                f16x4 hi, lo;
                split4(v, g.cs_scale, hi, lo);
                unsigned char* q = g.Cs + (size_t)m * g.ldcs * 4 + split_col_off(nout0);
                *reinterpret_cast<f16x4*>(q) = hi;
                *reinterpret_cast<f16x4*>(q + 64) = lo;
            }
            if (pool32 && p >= 8) { pm2.x = fmaxf(pm2.x, v.x); pm2.y = fmaxf(pm2.y, v.y); pm2.z = fmaxf(pm2.z, v.z); pm2.w = fmaxf(pm2.w, v.w); }
            else { pm.x = fmaxf(pm.x, v.x); pm.y = fmaxf(pm.y, v.y); pm.z = fmaxf(pm.z, v.z); pm.w = fmaxf(pm.w, v.w); }
        }
    }
    if (g.pool) {                                                    // max over this wave's 64 rows
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
        auto put = [&](float* q, const float4& v) {
            if (nout0 + 3 < n_out) *reinterpret_cast<float4*>(q) = v;
            else { if (nout0 < n_out) q[0] = v.x; if (nout0 + 1 < n_out) q[1] = v.y; if (nout0 + 2 < n_out) q[2] = v.z; }
        };
        if (lr == 0 && mbase < g.M) {
            if (!pool32) put(g.pool + (size_t)(mbase / 64) * g.ldp + nout0, pm);
            else {
                put(g.pool + (size_t)(mbase / 32) * g.ldp + nout0, pm);
                if (mbase + 32 < g.M) put(g.pool + (size_t)(mbase / 32 + 1) * g.ldp + nout0, pm2);
            }
        }
    }
}

// Host-side validation + translation of the public epilogue descriptor (shared by both entry points).
static inline int fill_epi(EpiArgs& e, const ua_gemm_epilogue* ep, int M, int N, float* C, long long ldc) {
    e = EpiArgs{};
    e.M = M; e.N = N; e.acc_scale = 1.0f; e.C = C; e.ldc = ldc;
    if (ep) {
        e.bias = ep->bias; e.scale = ep->scale; e.shift = ep->shift;
        e.rga = ep->rowgroup_add; e.ldg = ep->ldg; e.group_rows = ep->group_rows;
        e.act = ep->act; e.residual = ep->residual; e.ldr = ep->ldr;
        e.pool = ep->pool_out; e.ldp = ep->ldp; e.pool_rows = ep->pool_rows;
        e.Cs = reinterpret_cast<unsigned char*>(ep->split_out); e.ldcs = ep->ld_split; e.cs_scale = ep->split_scale;
        if (e.act < 0 || e.act > 3) return UA_EINVAL;
        if (e.rga && e.group_rows <= 0) return UA_EINVAL;
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
    if (e.Cs) {
        if (mis(e.Cs) || e.ldcs % 32 || e.ldcs < n_out || n_out % 4) return UA_EALIGN;
        if (!(e.cs_scale > 0.f)) return UA_EINVAL;
    }
    return UA_OK;
}

}  // namespace ua
