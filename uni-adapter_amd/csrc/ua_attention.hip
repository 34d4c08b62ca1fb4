// Multi-head self-attention for the Uni3D token sequence (512 patch tokens + CLS per cloud) on
// gfx950, exact-fp32 MFMA, flash-style (the T x T score matrix never leaves registers).
//
// Reference semantics: timm EvaAttention called without rope (models/point_encoder.py:217-218;
// arithmetic spelled out in extract_attention.py:121-177): softmax(q k^T * hd^-0.5) v per head.
//
// Token rows (ua_attention in include/uniadapter.h): patch g of cloud b at b*G+g, CLS of cloud b
// at Bc*G+b, so the big GEMMs see a multiple of 128 rows plus one thin tile.
//
// Mapping.  Workgroup = (cloud, head, 128 query rows), 4 wavefronts x 32 queries.  Key/value
// tiles of 32 keys are staged through double-buffered LDS and shared by the 4 waves.
//   S^T = K Q^T  ("swapped" product): v_mfma_f32_32x32x2_f32 with A = K tile, B = Q^T puts the
//   QUERY on the lane (col = lane&31) and the 32 keys in the 16 accumulator registers of the two
//   lane halves, so the softmax row-max / row-sum are in-lane reductions plus ONE lane^32 shuffle.
//   O^T += V^T P^T: the P registers are already the B operand of the next MFMA (k index = key),
//   lane (d, h) reads V[key(r,h)][d] from LDS as the A operand - no data movement for P at all.
// Q is pre-scaled by hd^-0.5 = 2^-3 (exact).  Online softmax keeps (m, l) per lane; the two lane
// halves hold disjoint key subsets of the same query and are combined once at the end.
#include "ua_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;                 // head dim
constexpr int KT = 32;                 // keys per tile
constexpr int K_LD = HD + 4;           // LDS row stride of the K tile (floats): conflict-free ds_read_b128
constexpr int V_LD = HD;               // V tile rows are read 32 consecutive floats at a time

__global__ void __launch_bounds__(256, 2)
attention_kernel(const float* __restrict__ qkv, long long ldqkv, float* __restrict__ out, long long ldo,
                 int Bc, int G, int H, int cls_only) {
    __shared__ __attribute__((aligned(16))) float s_k[2][KT * K_LD];
    __shared__ __attribute__((aligned(16))) float s_v[2][KT * V_LD];
    const int C = H * HD;
    const int n_qb_patch = G / 128;                       // query blocks of patch tokens
    // XCD-aware block order (see attention_split_kernel): ids congruent mod 8 walk the query blocks of one (cloud, head)
    // pair, so the pair's K and V are fetched by one XCD's L2 instead of up to five
    int qb, pair;
    {
        const int nq = n_qb_patch + 1, L = blockIdx.x;
        if (cls_only) { pair = L; qb = n_qb_patch; }                       // one workgroup per (cloud, head): its CLS query only
        else if (((H * Bc) & 7) == 0) { const int slot = L >> 3; pair = (slot / nq) * 8 + (L & 7); qb = slot % nq; }
        else { pair = L / nq; qb = L % nq; }
    }
    const int h = pair % H, b = pair / H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- this wave's 32 query rows ----
    const bool cls_block = qb >= n_qb_patch;
    const int q_local = wid * 32 + li;
    long long q_row; bool q_valid;
    if (!cls_block) { q_row = (long long)b * G + qb * 128 + q_local; q_valid = true; }
    else            { q_row = (long long)Bc * G + b; q_valid = (q_local == 0); }
    const bool wave_active = !cls_block || wid == 0;      // wave-uniform

    float4 qf[8];                                          // Q[q][8c+4h .. +3], c = 0..7, pre-scaled
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q_valid) v = *reinterpret_cast<const float4*>(qkv + q_row * ldqkv + h * HD + c * 8 + lh * 4);
        qf[c] = make_float4(v.x * 0.125f, v.y * 0.125f, v.z * 0.125f, v.w * 0.125f);
    }

    f32x16 o0, o1;                                         // O^T rows d (0..31 | 32..63), col = query
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // ---- K/V tile staging: tile t < n_pt holds patch keys 32t..32t+31, the last tile the CLS key ----
    const int n_pt = G / KT, n_tiles = n_pt + 1;
    const int s_key = tid >> 3, s_d = (tid & 7) * 4;       // 256 threads cover 32 keys x 32 floats; 2 passes
    float4 rk[2], rv[2];
    auto fetch = [&](int t) {
        long long row; bool ok;
        if (t < n_pt) { row = (long long)b * G + t * KT + s_key; ok = true; }
        else          { row = (long long)Bc * G + b; ok = (s_key == 0); }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            rk[p] = make_float4(0.f, 0.f, 0.f, 0.f); rv[p] = rk[p];
            if (ok) {
                const float* base = qkv + row * ldqkv + h * HD + s_d + p * 32;
                rk[p] = *reinterpret_cast<const float4*>(base + C);
                rv[p] = *reinterpret_cast<const float4*>(base + 2 * C);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<float4*>(&s_k[buf][s_key * K_LD + s_d + p * 32]) = rk[p];
            *reinterpret_cast<float4*>(&s_v[buf][s_key * V_LD + s_d + p * 32]) = rv[p];
        }
    };

    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) fetch(t + 1);
        if (wave_active) {
            // ---- S^T[key][q] = sum_d K[key][d] Q[q][d] ----
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const float* kp = &s_k[buf][li * K_LD + lh * 4];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + c * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c].w, s, 0, 0, 0);
            }
            // s[e] = score of key (e&3)+8(e>>2)+4lh for query li
            const int n_valid = (t < n_pt) ? KT : 1;
            if (n_valid < KT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (key >= n_valid) s[e] = -INFINITY;
                }
            }
            // ---- online softmax (row = lane's query; the other half of the row is lane^32) ----
            float mt = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = expf(m_run - m_new);       // 0 on the first tile (m_run = -inf)
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = expf(s[e] - m_new); psum += s[e]; }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
            // ---- O^T[d][q] += sum_key V[key][d] P[q][key] ; MFMA step e uses keys key(e,0), key(e,1) ----
            const float* vp = &s_v[buf][(4 * lh) * V_LD + li];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int krow = (e & 3) + 8 * (e >> 2);
                const float v0 = vp[krow * V_LD];
                const float v1 = vp[krow * V_LD + 32];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], o1, 0, 0, 0);
            }
        }
        if (t + 1 < n_tiles) stash(buf ^ 1);
        __syncthreads();
    }

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q_valid) {
        float* orow = out + q_row * ldo + h * HD;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {                   // registers 4e4..4e4+3 = 4 consecutive d
            const int d = 8 * e4 + 4 * lh;
            *reinterpret_cast<float4*>(orow + d) =
                make_float4(o0[4 * e4] * inv, o0[4 * e4 + 1] * inv, o0[4 * e4 + 2] * inv, o0[4 * e4 + 3] * inv);
            *reinterpret_cast<float4*>(orow + 32 + d) =
                make_float4(o1[4 * e4] * inv, o1[4 * e4 + 1] * inv, o1[4 * e4 + 2] * inv, o1[4 * e4 + 3] * inv);
        }
    }
}

}  // namespace

static int attention_impl(int cls_only, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                          int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    UA_CHECK_PTR(qkv); UA_CHECK_PTR(out);
    UA_CHECK_ALIGN16(qkv); UA_CHECK_ALIGN16(out);
    if (Bc <= 0 || G <= 0 || H <= 0) return UA_EINVAL;
    if (hd != HD || G % 128 != 0) return UA_EUNSUPPORTED;
    if (ldqkv % 4 || ldo % 4) return UA_EALIGN;
    if (ldqkv < 3LL * H * HD || ldo < (int64_t)H * HD) return UA_EINVAL;
    if ((long long)(G / 128 + 1) * H * Bc > 0x7fffffffLL) return UA_EUNSUPPORTED;
    dim3 grid(cls_only ? (unsigned)(H * Bc) : (unsigned)(G / 128 + 1) * H * Bc);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, ua::as_stream(stream),
                       qkv, (long long)ldqkv, out, (long long)ldo, Bc, G, H, cls_only);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_attention(const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                            int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    return attention_impl(0, qkv, ldqkv, out, ldo, Bc, G, H, hd, stream);
}

extern "C" int ua_attention_cls(const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                                int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    return attention_impl(1, qkv, ldqkv, out, ldo, Bc, G, H, hd, stream);
}
