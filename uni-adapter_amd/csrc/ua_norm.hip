// LayerNorm and token assembly for gfx950.
//
// Reference semantics: timm LayerNorm (eps 1e-6) inside EvaBlock / visual.norm / fc_norm
// (models/point_encoder.py:217-220); token assembly models/point_encoder.py:198-208.
// Both are HBM/L2-streaming ops: one wavefront per row, 16-byte accesses, fp64 moments.
#include "ua_gemm_common.h"

namespace {

using ua::wave_sum;

// One wavefront per row; the row lives in registers (NV float4 per lane), so HBM/L2 sees exactly
// one read and one write per element.  Columns >= cols (K padding) are written as zeros.
// SPLIT: the result is written as an H32L32 split-precision matrix (operand of ua_gemm_split).
template <int NV, bool SPLIT>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                 float eps, float* __restrict__ y, int64_t ldy, int rows, int cols, int cols_pad, float split_scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
    float4* yr = reinterpret_cast<float4*>(y + (size_t)row * ldy);
    const int nv = (cols + 3) / 4;                 // float4 chunks holding data (tail chunk masked)
    float4 v[NV];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = j * 64 + lane, c0 = 4 * i;
        v[j] = (i < nv) ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 0 >= cols) v[j].x = 0.f;
        if (c0 + 1 >= cols) v[j].y = 0.f;
        if (c0 + 2 >= cols) v[j].z = 0.f;
        if (c0 + 3 >= cols) v[j].w = 0.f;
        s += (double)v[j].x + (double)v[j].y + (double)v[j].z + (double)v[j].w;
    }
    const float mean = (float)(wave_sum(s) / (double)cols);
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c0 = 4 * (j * 64 + lane);
        float d;
        if (c0 + 0 < cols) { d = v[j].x - mean; q += (double)(d * d); }
        if (c0 + 1 < cols) { d = v[j].y - mean; q += (double)(d * d); }
        if (c0 + 2 < cols) { d = v[j].z - mean; q += (double)(d * d); }
        if (c0 + 3 < cols) { d = v[j].w - mean; q += (double)(d * d); }
    }
    const float var = (float)(wave_sum(q) / (double)cols);
    const float rstd = 1.0f / sqrtf(var + eps);
    const int nvp = cols_pad / 4;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = j * 64 + lane, c0 = 4 * i;
        if (i >= nvp) continue;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 3 < cols) {
            const float4 ww = *reinterpret_cast<const float4*>(w + c0), bb = *reinterpret_cast<const float4*>(b + c0);
            o.x = (v[j].x - mean) * rstd * ww.x + bb.x; o.y = (v[j].y - mean) * rstd * ww.y + bb.y;
            o.z = (v[j].z - mean) * rstd * ww.z + bb.z; o.w = (v[j].w - mean) * rstd * ww.w + bb.w;
        } else {
            if (c0 + 0 < cols) o.x = (v[j].x - mean) * rstd * w[c0] + b[c0];
            if (c0 + 1 < cols) o.y = (v[j].y - mean) * rstd * w[c0 + 1] + b[c0 + 1];
            if (c0 + 2 < cols) o.z = (v[j].z - mean) * rstd * w[c0 + 2] + b[c0 + 2];
        }
        if (SPLIT) {
            ua::f16x4 hi, lo;
            ua::split4(o, split_scale, hi, lo);
            unsigned char* q = reinterpret_cast<unsigned char*>(y) + (size_t)row * ldy * 4 + ua::split_col_off(c0);
            *reinterpret_cast<ua::f16x4*>(q) = hi;
            *reinterpret_cast<ua::f16x4*>(q + 64) = lo;
        } else {
            yr[i] = o;
        }
    }
}

__global__ void __launch_bounds__(256)
assemble_tokens_kernel(const float* __restrict__ tok, const float* __restrict__ pos, const float* __restrict__ cls_token,
                       const float* __restrict__ cls_pos, float* __restrict__ x, int n_patch_rows, int n_rows, int C4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n_rows * C4;
    if (i >= total) return;
    const size_t row = i / C4, c = i % C4;
    float4 a, p;
    if (row < (size_t)n_patch_rows) {
        a = reinterpret_cast<const float4*>(tok)[i];
        p = reinterpret_cast<const float4*>(pos)[i];
    } else {
        a = reinterpret_cast<const float4*>(cls_token)[c];
        p = reinterpret_cast<const float4*>(cls_pos)[c];
    }
    reinterpret_cast<float4*>(x)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);   // :208
}

// x[b*G+g] += pos[b*G+g] ; x[Bc*G+b] += cls_pos   (PointBERT re-adds the positional embedding before
// every block: models/ulip/pointbert/point_encoder.py:95-98)
__global__ void __launch_bounds__(256)
add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, const float* __restrict__ cls_pos,
               int n_patch_rows, int n_rows, int C4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_rows * C4) return;
    const size_t row = i / C4, c = i % C4;
    const float4 p = (row < (size_t)n_patch_rows) ? reinterpret_cast<const float4*>(pos)[i] : reinterpret_cast<const float4*>(cls_pos)[c];
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    reinterpret_cast<float4*>(x)[i] = v;
}

// out[b, 0:C] = x[Bc*G+b] (CLS) ; out[b, C:2C] = max_g x[b*G+g]   (point_encoder.py:190: cat[cls, max])
__global__ void __launch_bounds__(256)
cls_max_concat_kernel(const float* __restrict__ x, float* __restrict__ out, int Bc, int G, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = -INFINITY;
    const float* p = x + (size_t)b * G * C + c;
    for (int g = 0; g < G; ++g) m = fmaxf(m, p[(size_t)g * C]);
    out[(size_t)b * 2 * C + c] = x[((size_t)Bc * G + b) * C + c];
    out[(size_t)b * 2 * C + C + c] = m;
}

}  // namespace

extern "C" int ua_add_pos(float* x, const float* pos, const float* cls_pos, int32_t Bc, int32_t G, int32_t C, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(pos); UA_CHECK_PTR(cls_pos);
    UA_CHECK_ALIGN16(x); UA_CHECK_ALIGN16(pos); UA_CHECK_ALIGN16(cls_pos);
    if (Bc <= 0 || G <= 0 || C <= 0) return UA_EINVAL;
    if (C % 4) return UA_EUNSUPPORTED;
    const int rows = Bc * G + Bc;
    const size_t total = (size_t)rows * (C / 4);
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ua::as_stream(stream),
                       x, pos, cls_pos, Bc * G, rows, C / 4);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_cls_max_concat(const float* x, float* out, int32_t Bc, int32_t G, int32_t C, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(out);
    if (Bc <= 0 || G <= 0 || C <= 0) return UA_EINVAL;
    hipLaunchKernelGGL(cls_max_concat_kernel, dim3((C + 255) / 256, Bc), dim3(256), 0, ua::as_stream(stream), x, out, Bc, G, C);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_layernorm(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                            float* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(w); UA_CHECK_PTR(b); UA_CHECK_PTR(y);
    UA_CHECK_ALIGN16(x); UA_CHECK_ALIGN16(y);
    if (rows <= 0 || cols <= 0 || cols_pad < cols) return UA_EINVAL;
    if (ldx % 4 || ldy % 4 || cols_pad % 4) return UA_EALIGN;
    if (ldx < ((cols + 3) / 4) * 4 || ldy < cols_pad) return UA_EINVAL;
    if (cols_pad > 8192) return UA_EUNSUPPORTED;
    const int nv = (cols_pad / 4 + 63) / 64;       // float4 per lane
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = ua::as_stream(stream);
#define UA_LN(NV) hipLaunchKernelGGL((layernorm_kernel<NV, false>), grid, block, 0, st, x, ldx, w, b, eps, y, ldy, rows, cols, cols_pad, 1.0f)
    if (nv <= 2) UA_LN(2); else if (nv <= 4) UA_LN(4); else if (nv <= 6) UA_LN(6); else if (nv <= 8) UA_LN(8);
    else if (nv <= 12) UA_LN(12); else if (nv <= 16) UA_LN(16); else if (nv <= 24) UA_LN(24); else UA_LN(32);
#undef UA_LN
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_layernorm_split(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                                  void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad,
                                  float scale, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(w); UA_CHECK_PTR(b); UA_CHECK_PTR(y);
    UA_CHECK_ALIGN16(x); UA_CHECK_ALIGN16(y);
    if (rows <= 0 || cols <= 0 || cols_pad < cols || !(scale > 0.f)) return UA_EINVAL;
    if (ldx % 4 || ldy % 32 || cols_pad % 32) return UA_EALIGN;
    if (ldx < ((cols + 3) / 4) * 4 || ldy < cols_pad) return UA_EINVAL;
    if (cols_pad > 8192) return UA_EUNSUPPORTED;
    const int nv = (cols_pad / 4 + 63) / 64;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = ua::as_stream(stream);
    float* yf = reinterpret_cast<float*>(y);
#define UA_LN(NV) hipLaunchKernelGGL((layernorm_kernel<NV, true>), grid, block, 0, st, x, ldx, w, b, eps, yf, ldy, rows, cols, cols_pad, scale)
    if (nv <= 2) UA_LN(2); else if (nv <= 4) UA_LN(4); else if (nv <= 6) UA_LN(6); else if (nv <= 8) UA_LN(8);
    else if (nv <= 12) UA_LN(12); else if (nv <= 16) UA_LN(16); else if (nv <= 24) UA_LN(24); else UA_LN(32);
#undef UA_LN
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_assemble_tokens(const float* tok, const float* pos, const float* cls_token,
                                  const float* cls_pos, float* x, int32_t Bc, int32_t G, int32_t C,
                                  void* stream) {
    UA_CHECK_PTR(tok); UA_CHECK_PTR(pos); UA_CHECK_PTR(cls_token); UA_CHECK_PTR(cls_pos); UA_CHECK_PTR(x);
    UA_CHECK_ALIGN16(tok); UA_CHECK_ALIGN16(pos); UA_CHECK_ALIGN16(cls_token); UA_CHECK_ALIGN16(cls_pos); UA_CHECK_ALIGN16(x);
    if (Bc <= 0 || G <= 0 || C <= 0) return UA_EINVAL;
    if (C % 4) return UA_EUNSUPPORTED;
    const int rows = Bc * G + Bc;
    const size_t total = (size_t)rows * (C / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ua::as_stream(stream), tok, pos, cls_token, cls_pos, x, Bc * G, rows, C / 4);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
