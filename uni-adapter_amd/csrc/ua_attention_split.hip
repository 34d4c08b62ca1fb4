// Multi-head self-attention with fp32-accurate products on the f16 matrix cores (3-term split,
// fp32 accumulate, fp32 softmax) - the ua_gemm_split companion of ua_attention.hip.
//
// Reference semantics: timm EvaAttention without rope (models/point_encoder.py:217-218;
// extract_attention.py:121-177): softmax(q k^T * hd^-0.5) v per head.
//
// Operands arrive in the H32L32 split format straight from the qkv GEMM epilogue (no fp32 qkv is
// ever written) and the result is written in split form for the proj GEMM.  Per head the 64
// features of q/k/v are two 32-column groups = 256 contiguous bytes [hi 0-31 | lo 0-31 | hi 32-63 |
// lo 32-63] of a token row.
//
// Mapping (as ua_attention.hip): workgroup = (cloud, head, 128 queries), 4 wavefronts x 32 queries;
// 32-key K/V tiles double-buffered in LDS.  S^T = K Q^T with v_mfma_f32_32x32x16_f16 (query on the
// lane, keys in the 16 accumulator registers -> in-lane softmax row reductions); each product is
// hi*hi + hi*lo + lo*hi.  P is split in registers (p in [0,1]) and is directly the B operand of
// O^T += V^T P^T; the A operand needs V^T: V is staged row-major exactly like K and transposed on
// the way out of LDS by ds_read_b64_tr_b16 (each 16-lane group reads a 4-key x 16-feature block
// and every lane receives the 4 keys of its feature), two transposed reads per fragment.
// Per key tile a wave issues 24 MFMAs (768 matrix-pipe cycles) instead of 64 f32 MFMAs (4096).
#include "ua_gemm_common.h"

namespace {

using ua::f32x16;
using ua::f16x8;
using ua::f16x4;

constexpr int HD = 64, KT = 32;
constexpr int KROW = 272;              // LDS bytes per key row (256 + 16 pad: conflict-free ds_read_b128)
constexpr int VROW = 320;              // LDS bytes per key row of V (256 + 64: rows 16 banks apart -> conflict-free ds_read_b64_tr_b16)
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

__global__ void __launch_bounds__(256, 2)
attention_split_kernel(const unsigned char* __restrict__ qkv, long long ldqkv, float c2, float v_inv_scale,
                       float* __restrict__ out, long long ldo, unsigned char* __restrict__ out_split, long long ldos,
                       float out_scale, int Bc, int G, int H, int cls_only) {
    __shared__ __attribute__((aligned(16))) unsigned char s_k[2][KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char s_v[2][KT * VROW];
    const int C = H * HD;
    // XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by linear id, and the query blocks of one
    // (cloud, head) pair all read that pair's K and V (263 KB at G = 512): with the pair's blocks on adjacent ids every
    // XCD's L2 fetched them separately (measured: 3.0x the algorithmic traffic, L2 hit rate 0.37, 5.2 TB/s - the kernel
    // was fabric-bound).  Here ids congruent mod 8 walk the blocks of one pair, then the next pair of the same XCD.
    const int n_qb_patch = G / 128;
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
    const size_t row_bytes = (size_t)ldqkv * 4;
    const size_t q_off = (size_t)(h * HD) * 4, k_off = (size_t)(C + h * HD) * 4, v_off = (size_t)(2 * C + h * HD) * 4;

    const bool cls_block = qb >= n_qb_patch;
    if (cls_block) {                                                            // workgroup-uniform
        // The CLS token of this (cloud, head) is ONE query against G + 1 keys: a 32-query MFMA wave would idle 31 lanes through 17
        // key tiles while the other three waves only stage (measured: 15 % of the kernel's time at 32 clouds).  Here the 256 threads
        // take one key each per pass for the scores (fp32 dot products of the exact hi + lo values), a block-wide softmax, and
        // wave w accumulates p * V over keys w, w+4, ... with one output feature per lane.
        // (LDS: the key/value tile buffers of the patch path are unused here and are re-used)
        float* s_q = reinterpret_cast<float*>(&s_k[0][0]);                      // [64]
        float* s_red = s_q + HD;                                                // [8]
        float* s_p = s_red + 8;                                                 // scores, then probabilities: [G + 1]
        float* s_o = reinterpret_cast<float*>(&s_v[0][0]);                      // [32 key slots][64]
        static_assert(sizeof(s_k) >= (HD + 8 + 1024 + 1) * 4 && sizeof(s_v) >= 32 * HD * 4, "CLS path scratch");
        const int nkeys = G + 1;
        const long long qrow = (long long)Bc * G + b;
        if (tid < HD) {
            const unsigned char* p = qkv + (size_t)qrow * row_bytes + q_off + (tid >> 5) * 128 + (tid & 31) * 2;
            s_q[tid] = (float)*reinterpret_cast<const _Float16*>(p) + (float)*reinterpret_cast<const _Float16*>(p + 64);
        }
        __syncthreads();
        // 8 threads per key (8 features each: one 16-byte hi and one 16-byte lo load), 32 keys per pass
        const int g8 = tid & 7, slot = tid >> 3;
        const size_t foff = (size_t)(g8 >> 2) * 128 + (size_t)(g8 & 3) * 16;
        float qf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = s_q[8 * g8 + j];
        float mloc = -INFINITY;
        for (int k0 = slot; k0 < nkeys; k0 += 128) {                            // 4 independent keys in flight per thread
            f16x8 kh[4], kl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k0 + 32 * u;
                const long long row = key < G ? (long long)b * G + key : qrow;  // (keys past the end re-read the CLS row; result dropped)
                const unsigned char* kp = qkv + (size_t)row * row_bytes + k_off + foff;
                kh[u] = *reinterpret_cast<const f16x8*>(kp);
                kl[u] = *reinterpret_cast<const f16x8*>(kp + 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k0 + 32 * u;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = __builtin_fmaf((float)kh[u][j] + (float)kl[u][j], qf[j], acc);
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
                if (key < nkeys) {
                    if (g8 == 0) s_p[key] = acc;
                    mloc = fmaxf(mloc, acc);
                }
            }
        }
        mloc = ua::wave_max(mloc);
        if (lane == 0) s_red[wid] = mloc;
        __syncthreads();
        const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        float lsum = 0.f;
        for (int key = tid; key < nkeys; key += 256) {
            const float pv = __builtin_amdgcn_exp2f((s_p[key] - mx) * c2);      // c2 = hd^-0.5 / s^2 * log2(e) > 0
            s_p[key] = pv;
            lsum += pv;
        }
        lsum = ua::wave_sum(lsum);
        if (lane == 0) s_red[4 + wid] = lsum;
        __syncthreads();
        const float ltot = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        for (int k0 = slot; k0 < nkeys; k0 += 128) {
            f16x8 vh[4], vl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k0 + 32 * u;
                const long long row = key < G ? (long long)b * G + key : qrow;
                const unsigned char* vp = qkv + (size_t)row * row_bytes + v_off + foff;
                vh[u] = *reinterpret_cast<const f16x8*>(vp);
                vl[u] = *reinterpret_cast<const f16x8*>(vp + 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k0 + 32 * u;
                const float pk = key < nkeys ? s_p[key] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(pk, (float)vh[u][j] + (float)vl[u][j], o[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s_o[slot * HD + 8 * g8 + j] = o[j];
        __syncthreads();
        if (tid < HD) {
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) r += s_o[q * HD + tid];
            r *= v_inv_scale / ltot;
            if (out) out[qrow * ldo + h * HD + tid] = r;
            if (out_split) {
                const float x = r * out_scale;
                const _Float16 hi = (_Float16)x;
                unsigned char* q0 = out_split + (size_t)qrow * ldos * 4 + ua::split_col_off((h * HD + tid) & ~3) + (tid & 3) * 2;
                *reinterpret_cast<_Float16*>(q0) = hi;
                *reinterpret_cast<_Float16*>(q0 + 64) = (_Float16)(x - (float)hi);
            }
        }
        return;
    }
    const int q_local = wid * 32 + li;
    const long long q_row = (long long)b * G + qb * 128 + q_local;
    constexpr bool q_valid = true, wave_active = true;                          // (every wave of a patch block holds 32 real queries)

    // Q fragments: k16 step s covers features 16s..16s+15; lane half lh holds 8 of them
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned char* p = qkv + (size_t)q_row * row_bytes + q_off + (s >> 1) * 128 + (16 * (s & 1) + 8 * lh) * 2;
        if (q_valid) { qh[s] = *reinterpret_cast<const f16x8*>(p); ql[s] = *reinterpret_cast<const f16x8*>(p + 64); }
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qh[s][j] = (_Float16)0.f; ql[s][j] = (_Float16)0.f; }
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging: 256 threads x 2 chunks of 16 B cover 32 keys x 256 B for K and for V
    const int n_pt = G / KT, n_tiles = n_pt + 1;
    uint4 rk[2], rv[2];
    auto fetch = [&](int t) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p, key = f >> 4, c = f & 15;
            long long row; bool ok;
            if (t < n_pt) { row = (long long)b * G + t * KT + key; ok = true; }
            else          { row = (long long)Bc * G + b; ok = (key == 0); }
            rk[p] = make_uint4(0u, 0u, 0u, 0u); rv[p] = rk[p];
            if (ok) {
                const unsigned char* base = qkv + (size_t)row * row_bytes + c * 16;
                rk[p] = *reinterpret_cast<const uint4*>(base + k_off);
                rv[p] = *reinterpret_cast<const uint4*>(base + v_off);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = tid + 256 * p, key = f >> 4, c = f & 15;
            *reinterpret_cast<uint4*>(&s_k[buf][key * KROW + c * 16]) = rk[p];
            *reinterpret_cast<uint4*>(&s_v[buf][key * VROW + c * 16]) = rv[p];
        }
    };

    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) fetch(t + 1);
        if (wave_active) {
            // ---- S^T[key][q] = sum_d K[key][d] Q[q][d] (3 f16 MFMAs per k16 step) ----
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
            const unsigned char* kp = &s_k[buf][li * KROW];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = (s >> 1) * 128 + (16 * (s & 1) + 8 * lh) * 2;
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kp + o);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(kp + o + 64);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc, 0, 0, 0);
            }
            // ---- online softmax, fp32 (row = lane's query; other half of the row is lane^32).  The raw
            // scores stay unscaled: with c2 = hd^-0.5 / s^2 * log2(e) > 0 the row maximum is taken on the
            // raw values and p = 2^(s*c2 - m*c2) is one v_fma + one v_exp_f32 per element.
            if (t == n_pt) {                                                       // CLS tile: only key 0 exists
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (key >= 1) sacc[e] = -INFINITY;
                }
            }
            float mt = sacc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, sacc[e]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);
            const float mc = m_new * c2;
            float psum = 0.f;
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], c2, -mc));
                psum += p;
                const _Float16 hi = (_Float16)p;
                ph[e >> 3][e & 7] = hi;
                pl[e >> 3][e & 7] = (_Float16)(p - (float)hi);
            }
            if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {                     // wave-uniform: some row's maximum moved
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);  // exactly 1 for the rows that did not move
                l_run *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
                m_run = m_new;
            }
            l_run += psum;
            // ---- O^T[d][q] += sum_key V[key][d] P[q][key]; slot i of half lh <-> key (i&3)+8(i>>2)+4lh+16ks ----
            // transposed-read addressing: 16-lane group Gq = lane>>4 handles features 16*(Gq&1).. of the
            // 32-feature block and keys 4*(Gq>>1)+{0..3} (+16ks, +8); lane 4q+p of the group supplies row q,
            // features 4p..4p+3; it receives feature (lane&15) of the 4 keys = the MFMA A-operand slots.
            const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
            const unsigned char* vp = &s_v[buf][(4 * (gq >> 1) + tq) * VROW + (16 * (gq & 1) + 4 * tp) * 2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned char* r = vp + (16 * ks) * VROW + db * 128;
                    typedef __attribute__((address_space(3))) s16x4* lp_t;
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 8 * VROW));
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 64));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 8 * VROW + 64));
                    // (whole-vector concatenation + one bit-cast: element-wise bit_casts of the v4i16
                    //  results were mis-lowered by hipcc 7.2 into duplicated registers)
                    const f16x8 vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                    const f16x8 vl = __builtin_bit_cast(f16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if (db == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[ks], o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[ks], o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[ks], o0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[ks], o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[ks], o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[ks], o1, 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 < n_tiles) stash(buf ^ 1);
        __syncthreads();
    }

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = v_inv_scale / l_tot;
    if (q_valid) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int d = 8 * e4 + 4 * lh;
            const float4 v0 = make_float4(o0[4 * e4] * inv, o0[4 * e4 + 1] * inv, o0[4 * e4 + 2] * inv, o0[4 * e4 + 3] * inv);
            const float4 v1 = make_float4(o1[4 * e4] * inv, o1[4 * e4 + 1] * inv, o1[4 * e4 + 2] * inv, o1[4 * e4 + 3] * inv);
            if (out) {
                float* orow = out + q_row * ldo + h * HD;
                *reinterpret_cast<float4*>(orow + d) = v0;
                *reinterpret_cast<float4*>(orow + 32 + d) = v1;
            }
            if (out_split) {
                unsigned char* srow = out_split + (size_t)q_row * ldos * 4;
                f16x4 hi, lo;
                ua::split4(v0, out_scale, hi, lo);
                unsigned char* q0 = srow + ua::split_col_off(h * HD + d);
                *reinterpret_cast<f16x4*>(q0) = hi; *reinterpret_cast<f16x4*>(q0 + 64) = lo;
                ua::split4(v1, out_scale, hi, lo);
                unsigned char* q1 = srow + ua::split_col_off(h * HD + 32 + d);
                *reinterpret_cast<f16x4*>(q1) = hi; *reinterpret_cast<f16x4*>(q1 + 64) = lo;
            }
        }
    }
}

}  // namespace

static int attention_split_impl(int cls_only, const void* qkv, int64_t ldqkv, float qkv_scale, float* out, int64_t ldo,
                                  void* out_split, int64_t ldos, float out_scale,
                                  int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    UA_CHECK_PTR(qkv); UA_CHECK_ALIGN16(qkv);
    if (!out && !out_split) return UA_EINVAL;
    if (Bc <= 0 || G <= 0 || H <= 0 || !(qkv_scale > 0.f)) return UA_EINVAL;
    if (hd != HD || G % 128 != 0 || G > 1024) return UA_EUNSUPPORTED;
    if (ldqkv % 32 || ldqkv < 3LL * H * HD) return UA_EALIGN;
    if (out && (ldo % 4 || ldo < (int64_t)H * HD || (reinterpret_cast<uintptr_t>(out) & 15))) return UA_EALIGN;
    if (out_split && (ldos % 32 || ldos < (int64_t)H * HD || (reinterpret_cast<uintptr_t>(out_split) & 15) || !(out_scale > 0.f))) return UA_EALIGN;
    if ((long long)(G / 128 + 1) * H * Bc > 0x7fffffffLL) return UA_EUNSUPPORTED;
    dim3 grid(cls_only ? (unsigned)(H * Bc) : (unsigned)(G / 128 + 1) * H * Bc);
    const float c_scale = (0.125f / (qkv_scale * qkv_scale)) * 1.44269504088896340736f;   // hd^-0.5 / s^2 * log2(e)
    hipLaunchKernelGGL(attention_split_kernel, grid, dim3(256), 0, ua::as_stream(stream),
                       reinterpret_cast<const unsigned char*>(qkv), (long long)ldqkv, c_scale, 1.0f / qkv_scale,
                       out, (long long)ldo, reinterpret_cast<unsigned char*>(out_split), (long long)ldos, out_scale, Bc, G, H, cls_only);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_attention_split(const void* qkv, int64_t ldqkv, float qkv_scale, float* out, int64_t ldo,
                                  void* out_split, int64_t ldos, float out_scale,
                                  int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    return attention_split_impl(0, qkv, ldqkv, qkv_scale, out, ldo, out_split, ldos, out_scale, Bc, G, H, hd, stream);
}

extern "C" int ua_attention_split_cls(const void* qkv, int64_t ldqkv, float qkv_scale, float* out, int64_t ldo,
                                      void* out_split, int64_t ldos, float out_scale,
                                      int32_t Bc, int32_t G, int32_t H, int32_t hd, void* stream) {
    return attention_split_impl(1, qkv, ldqkv, qkv_scale, out, ldo, out_split, ldos, out_scale, Bc, G, H, hd, stream);
}
