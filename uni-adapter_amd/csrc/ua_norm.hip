// LayerNorm and token assembly for gfx950.
//
// Reference semantics: timm LayerNorm (eps 1e-6) inside EvaBlock / visual.norm / fc_norm
// (models/point_encoder.py:217-220); token assembly models/point_encoder.py:198-208.
// Both are HBM/L2-streaming ops: one wavefront per row, 16-byte accesses, fp64 moments.
#include "ua_common.h"

namespace {

using ua::wave_sum;

__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                 float eps, float* __restrict__ y, int64_t ldy, int rows, int cols, int cols_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float* yr = y + (size_t)row * ldy;
    const int nv = (cols + 3) / 4;                 // float4 chunks (tail chunk masked)
    double s = 0.0;
    for (int i = lane; i < nv; i += 64) {
        const float4 v = reinterpret_cast<const float4*>(xr)[i];
        const int c0 = 4 * i;
        s += (double)v.x;
        if (c0 + 1 < cols) s += (double)v.y;
        if (c0 + 2 < cols) s += (double)v.z;
        if (c0 + 3 < cols) s += (double)v.w;
    }
    const float mean = (float)(wave_sum(s) / (double)cols);
    double q = 0.0;
    for (int i = lane; i < nv; i += 64) {
        const float4 v = reinterpret_cast<const float4*>(xr)[i];
        const int c0 = 4 * i;
        float d = v.x - mean; q += (double)(d * d);
        if (c0 + 1 < cols) { d = v.y - mean; q += (double)(d * d); }
        if (c0 + 2 < cols) { d = v.z - mean; q += (double)(d * d); }
        if (c0 + 3 < cols) { d = v.w - mean; q += (double)(d * d); }
    }
    const float var = (float)(wave_sum(q) / (double)cols);
    const float rstd = 1.0f / sqrtf(var + eps);
    const int nvp = cols_pad / 4;
    for (int i = lane; i < nvp; i += 64) {
        const int c0 = 4 * i;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 < cols) {
            const float4 v = reinterpret_cast<const float4*>(xr)[i];
            o.x = (v.x - mean) * rstd * w[c0] + b[c0];
            if (c0 + 1 < cols) o.y = (v.y - mean) * rstd * w[c0 + 1] + b[c0 + 1];
            if (c0 + 2 < cols) o.z = (v.z - mean) * rstd * w[c0 + 2] + b[c0 + 2];
            if (c0 + 3 < cols) o.w = (v.w - mean) * rstd * w[c0 + 3] + b[c0 + 3];
        }
        reinterpret_cast<float4*>(yr)[i] = o;
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

}  // namespace

extern "C" int ua_layernorm(const float* x, int64_t ldx, const float* w, const float* b, float eps,
                            float* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(w); UA_CHECK_PTR(b); UA_CHECK_PTR(y);
    UA_CHECK_ALIGN16(x); UA_CHECK_ALIGN16(y);
    if (rows <= 0 || cols <= 0 || cols_pad < cols) return UA_EINVAL;
    if (ldx % 4 || ldy % 4 || cols_pad % 4) return UA_EALIGN;
    if (ldx < ((cols + 3) / 4) * 4 || ldy < cols_pad) return UA_EINVAL;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, ua::as_stream(stream),
                       x, ldx, w, b, eps, y, ldy, rows, cols, cols_pad);
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
