// Residual learning of the text anchors (Uni_Adapter.py:191-270, 444-476) on the matrix cores.
//
// Per sample the GMM state is fixed while 10 Adam steps move the K class embeddings e_b = t_b / |t_b|,
// t = text0 + res.  One step is two contractions with the state, and both are GEMMs:
//
//   forward   J[b,r]  = rc[r] - 1/2 (T1 + T2),  T1 = sum_d e_bd^2 / v_rd,  T2 = -2 sum_d e_bd mu_rd / v_rd     (r = k*M + m)
//             L[b,k]  = logsumexp_m J[b, k*M + m]                                                             (Uni_Adapter.py:219-244)
//   backward  ge[b,d] = sum_r w[b,r] (mu_rd - e_bd) / v_rd = G1[b,d] - e_bd G2[b,d],   G1 = w . (mu/v),  G2 = w . (1/v)
//             w[b,r]  = dLoss/dL[b,k] * exp(J[b,r] - L[b,k])                                                  (autograd of :244-253)
//
// The VALU kernels of ua_gmm.hip (align_fwd_kernel, align_step_kernel) re-read the state once per query group: 26 MB from
// L2 per forward and 105 MB per backward at K = 40, M = 8, D = 1024 (655 KB per workgroup), and the D/256 workgroups of a row
// hand over through a release/acquire ticket before Adam.  Here
//   * align_fwd_mfma_kernel: 16 queries x 16 state rows per workgroup, the 2D-long contraction split over its 8 waves
//     (v_mfma_f32_16x16x32_f16 on H32L32 split operands, partial blocks summed in fp64 through LDS).  The query operand
//     [t^2 | t] is formed on the fly from the UNNORMALISED t (1/|t|^2 and 1/|t| are applied to the two partial sums; the
//     workgroup also forms |t|^2 of its 16 queries), so no kernel ever has to renormalise a row;
//   * align_bwd_adam_mfma_kernel: one workgroup per 32 feature columns holds the TRANSPOSED state operand of its columns
//     ([mu/v | 1/v] over all rows r: 64 x KM split values, 82 KB) in LDS, recomputes the K x K loss statistics (the code of
//     align_loss_kernel), writes the weights w as a split operand into LDS, and one 16 x 16 x KM MFMA chain per wave gives
//     G1, G2 for all K queries.  Adam then needs g = (ge - e (ge . e)) / |t|, and the row dot product is available WITHOUT the
//     other column blocks:  ge_b . e_b = sum_r w[b,r] sum_d e (mu - e)/v = sum_r w[b,r] (-T2/2 - T1) - the forward pass stores
//     U = -T2/2 - T1 next to J.  So every workgroup applies Adam to its own 32 columns at once: no ticket, no fence, no second
//     pass, and per backward 82 KB instead of 655 KB per workgroup (2.6 MB instead of 105 MB in all).
// The state-side operands are rebuilt once per sample (ua_align_mfma_prepare: ua_gmm_mfma_operand + the transposed operand).
#include "ua_gemm_common.h"
#include "ua_align_common.h"
#include <cmath>

extern "C" int ua_gmm_mfma_operand(const float* mu, const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                                   void* workspace, size_t workspace_bytes, void* stream);

namespace {

using ua::f16x8;
using ua::f32x4;
using ua::pow2_scale;
using ua::wave_max;
using ua::wave_sum;

struct AlignWs {
    ua::MfmaWs f;                  // forward operand [1/v | -2 mu/v] (KM x 2D split), rc, colscale = -1/(2 s_r)
    unsigned char* ST;             // backward operand: (D/32) blocks x 64 rows [mu/v of 32 features | 1/v of 32 features] x KMp split columns
    float* st_inv;                 // 1 / (row scale) of every ST row
    int KMp;
    size_t total;
};

inline AlignWs align_ws_layout(unsigned char* base, int K, int M, int D) {
    AlignWs w{};
    w.f = ua::mfma_ws_layout(base, 1, K, M, D, false);
    size_t o = w.f.total;
    w.KMp = (K * M + 31) / 32 * 32;
    w.ST = base + o; o += ua::align256((size_t)(D / 32) * 64 * w.KMp * 4);
    w.st_inv = reinterpret_cast<float*>(base + o); o += ua::align256((size_t)(D / 32) * 64 * 4);
    w.total = o;
    return w;
}

inline size_t bwd_lds_bytes(int K, int KMp) {
    const size_t stats = ((size_t)(2 * K * K + 3 * K) * 4 + 15) / 16 * 16;
    return stats + (size_t)(64 + K) * (KMp * 4 + 16);
}

inline bool envelope_ok(int K, int M, int D) {
    if (K <= 0 || M <= 0 || D <= 0 || D % 256 != 0 || D > 2048) return false;
    if (K > 64 || 16 % M != 0) return false;
    const int KMp = (K * M + 31) / 32 * 32;
    if (KMp > 384) return false;                                     // register-resident joint / U rows of the backward kernel
    return bwd_lds_bytes(K, KMp) <= 150 * 1024 && (size_t)64 * (KMp + 1) * 4 <= 150 * 1024;
}

// ---------------------------------------------------------------------------------------------
// Both state-side operands in ONE launch, block 1024, dynamic LDS 64 x (KMp + 1) floats:
//   blocks 0 .. D/32-1      : transposed operand of features 32j..32j+31 ([mu/v | 1/v] over all rows, one scale per row)
//   blocks D/32 .. +KM/16-1 : forward operand rows (ua::gmm_operand_row, one wave per state row)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(1024)
align_prepare_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ pi, int KM, int KMp, float eps,
                     unsigned char* __restrict__ Wop, float* __restrict__ rc, float* __restrict__ colscale,
                     unsigned char* __restrict__ ST, float* __restrict__ st_inv) {
    constexpr int D = 256 * VEC;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if ((int)blockIdx.x >= D / 32) {                                          // workgroup-uniform
        const int n = ((int)blockIdx.x - D / 32) * 16 + w;
        if (n < KM) ua::gmm_operand_row<VEC>(mu, var, pi, n, eps, lane, Wop, rc, colscale);
        return;
    }
    const int j = blockIdx.x, d0 = j * 32, dl = tid & 31, rr = tid >> 5, P = KMp + 1;
    // rows r = rr, rr + 32, ...: all loads of a thread are issued before the first use (KMp <= 1024 -> at most 32 passes; 8 at a time)
    for (int r0 = 0; r0 < KMp; r0 += 256) {
        float m[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + rr + 32 * u;
            m[u] = 0.f; v[u] = 1.f;
            if (r < KM) { m[u] = mu[(size_t)r * D + d0 + dl]; v[u] = var[(size_t)r * D + d0 + dl]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + rr + 32 * u;
            if (r < KMp) {
                float a = 0.f, b = 0.f;
                if (r < KM) { const float iv = 1.0f / fmaxf(v[u] + eps, 1e-8f); a = m[u] * iv; b = iv; }   // dota_mixture.py:117-119
                tile[dl * P + r] = a;
                tile[(32 + dl) * P + r] = b;
            }
        }
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) {                                             // 16 waves x 4 rows
        const int row = w * 4 + i;
        const float* src = tile + row * P;
        float amax = 0.f;
        for (int r = lane; r < KMp; r += 64) amax = fmaxf(amax, fabsf(src[r]));
        amax = wave_max(amax);
        const float s = pow2_scale(amax);
        unsigned char* dst = ST + ((size_t)j * 64 + row) * KMp * 4;
        for (int c = lane * 4; c < KMp; c += 256)
            ua::store_split4(dst, c, make_float4(src[c], src[c + 1], src[c + 2], src[c + 3]), s);
        if (lane == 0) st_inv[j * 64 + row] = 1.0f / s;
    }
}

__device__ __forceinline__ void split8(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = v[i] * s;
        hi[i] = (_Float16)x;
        lo[i] = (_Float16)(x - (float)hi[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward.  grid (ceil(KM/16), ceil(K/16)), block 512; NDB = D/256 feature blocks of 32 per wave.
// ---------------------------------------------------------------------------------------------
template <int NDB>
__global__ void __launch_bounds__(512)
align_fwd_mfma_kernel(const float* __restrict__ t, const unsigned char* __restrict__ Wop, const float* __restrict__ rc,
                      const float* __restrict__ colscale, int K, int KM, int M,
                      float* __restrict__ joint, float* __restrict__ U, float* __restrict__ L, float* __restrict__ nrm) {
    constexpr int D = 256 * NDB;
    __shared__ double s_p1[8][256];
    __shared__ double s_p2[8][256];
    __shared__ double s_n2[8][64];
    __shared__ float s_j[16][17];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int lr = lane & 15, kg = lane >> 4;
    const float* pt = t + (size_t)min(b0 + lr, K - 1) * D + (w * NDB) * 32 + kg * 8;
    const unsigned char* pw = Wop + (size_t)min(n0 + lr, KM - 1) * (2 * D) * 4 + (size_t)(w * NDB) * 256 + kg * 16;
    float4 xa[NDB], xb[NDB];
    f16x8 wh0[NDB], wl0[NDB], wh1[NDB], wl1[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        xa[i] = *reinterpret_cast<const float4*>(pt + i * 32);
        xb[i] = *reinterpret_cast<const float4*>(pt + i * 32 + 4);
        wh0[i] = *reinterpret_cast<const f16x8*>(pw + (2 * i) * 128);          // K-tile 2i: 1/v of features 32i..32i+31
        wl0[i] = *reinterpret_cast<const f16x8*>(pw + (2 * i) * 128 + 64);
        wh1[i] = *reinterpret_cast<const f16x8*>(pw + (2 * i + 1) * 128);      // K-tile 2i+1: -2 mu/v of the same features
        wl1[i] = *reinterpret_cast<const f16x8*>(pw + (2 * i + 1) * 128 + 64);
    }
    // row constants of this thread's output element, requested now (used after the cross-wave sum)
    float p_rc = 0.f, p_cs = 0.f;
    if (tid < 256) { const int n = min(n0 + (tid & 15), KM - 1); p_rc = rc[n]; p_cs = colscale[n]; }
    float amax = 0.f;
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const float v[8] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w, xb[i].x, xb[i].y, xb[i].z, xb[i].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { amax = fmaxf(amax, fabsf(v[e])); n2 += (double)(v[e] * v[e]); }
    }
    amax = wave_max(amax);
    const float s = pow2_scale(fmaxf(amax, amax * amax));            // one scale for this wave's slice of [t^2 | t]
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const float v[8] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w, xb[i].x, xb[i].y, xb[i].z, xb[i].w};
        float v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v2[e] = v[e] * v[e];
        f16x8 qh, ql, xh, xl;
        split8(v2, s, qh, ql);
        split8(v, s, xh, xl);
        // state rows as the MFMA's first operand (output rows), queries as its second (output columns)
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0[i], qh, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0[i], qh, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0[i], ql, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1[i], xh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1[i], xh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1[i], xl, acc2, 0, 0, 0);
    }
    const double inv_s = 1.0 / (double)s;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        s_p1[w][lane * 4 + jj] = (double)acc1[jj] * inv_s;
        s_p2[w][lane * 4 + jj] = (double)acc2[jj] * inv_s;
    }
    s_n2[w][lane] = n2;
    __syncthreads();
    if (tid < 256) {
        // thread = (query q, state row c): accumulator element of lane (c >> 2) * 16 + q, register c & 3
        const int q = tid >> 4, c = tid & 15;
        const int idx = ((c >> 2) * 16 + q) * 4 + (c & 3);
        double T1 = 0.0, T2 = 0.0, nn2 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { T1 += s_p1[i][idx]; T2 += s_p2[i][idx]; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) nn2 += s_n2[i][g * 16 + q];
        const int b = b0 + q, n = n0 + c;
        float jv = 0.f;
        if (b < K && n < KM) {
            const double inv_sr = -2.0 * (double)p_cs;              // colscale = -1/(2 s_r)
            const double rn2 = 1.0 / nn2, rn = sqrt(rn2);
            const double t1 = T1 * inv_sr * rn2, t2 = T2 * inv_sr * rn;
            jv = (float)((double)p_rc - 0.5 * (t1 + t2));
            joint[(size_t)b * KM + n] = jv;
            U[(size_t)b * KM + n] = (float)(-0.5 * t2 - t1);        // sum_d e (mu - e) / v
            if (blockIdx.x == 0 && c == 0) nrm[b] = (float)sqrt(nn2);
        }
        s_j[q][c] = jv;
    }
    __syncthreads();
    if (tid < 256) {
        const int q = tid >> 4, c = tid & 15;
        const int b = b0 + q, n = n0 + c;
        if (c % M == 0 && b < K && n < KM) {                        // M divides 16: the class lies inside the block
            float mx = s_j[q][c];
            for (int m = 1; m < M; ++m) mx = fmaxf(mx, s_j[q][c + m]);
            float sm = 0.f;
            for (int m = 0; m < M; ++m) sm += expf(s_j[q][c + m] - mx);
            L[(size_t)b * K + n / M] = mx + logf(sm);               // torch.logsumexp
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward + Adam.  grid D/32, block 1024.
// dynamic LDS: [P K*K | s1 K | s2 K | dg K | L K*K] floats (16-byte padded) | S operand 64 x PITCH | weight operand K x PITCH,
// PITCH = KMp*4 + 16 bytes (rows 16 bytes apart in the bank pattern: conflict-free ds_read_b128 of 16 rows).
// ---------------------------------------------------------------------------------------------
struct AlignBwdArgs {
    const float* L; const float* joint; const float* U; const float* nrm;
    float* t; const float* text0; float* res; float* am; float* av;
    const unsigned char* ST; const float* st_inv;
    float* loss_out; float* ge_out;
    int K, M, D, KM, KMp;
    float step_size, w1, beta2, omb2, bc2_sqrt, adam_eps;
};

#ifdef UA_ALIGN_TRACE
#define AT_DECL unsigned long long at_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define AT_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); at_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define AT_DECL
#define AT_MARK(i)
#endif

__global__ void __launch_bounds__(1024)
align_bwd_adam_mfma_kernel(AlignBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    __shared__ double sd[16];
    __shared__ float sf[16];
    __shared__ int s_arg;
    __shared__ float s_gd[64], s_sb[64];
    __shared__ double sc[16 * 64];
    const int K = a.K, M = a.M, D = a.D, KM = a.KM, KMp = a.KMp;
    const int j = blockIdx.x, d0 = j * 32, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int PITCH = KMp * 4 + 16;
    AT_DECL;
    AT_MARK(0);
    float* P = reinterpret_cast<float*>(sm);
    float* s1 = P + K * K;
    float* s2 = s1 + K;
    float* dg = s2 + K;
    float* Ls = dg + K;
    unsigned char* Sop = sm + ((size_t)(2 * K * K + 3 * K) * 4 + 15) / 16 * 16;
    unsigned char* Aop = Sop + 64 * PITCH;
    // ---- every global read of the kernel is requested here, before the first wait: this block's columns of the state operand,
    //      the joint / U rows this wave turns into weights, the Adam state of this thread's elements ----
    constexpr int kWR = 4, kWC = 6;                                  // rows per wave (K <= 64), 64-element chunks per row (KMp <= 384)
    {
        const int cpr = KMp / 4;                                     // 16-byte chunks per S row
        const uint4* src = reinterpret_cast<const uint4*>(a.ST + (size_t)j * 64 * KMp * 4);
        for (int c = tid; c < 64 * cpr; c += 1024) {
            const int row = c / cpr, cc = c - row * cpr;
            *reinterpret_cast<uint4*>(Sop + row * PITCH + cc * 16) = src[c];
        }
    }
    float jr[kWR][kWC], ur[kWR][kWC];
#pragma unroll
    for (int i = 0; i < kWR; ++i)
#pragma unroll
        for (int c = 0; c < kWC; ++c) {
            const int b = wv + 16 * i, r = lane + 64 * c;
            jr[i][c] = 0.f; ur[i][c] = 0.f;
            if (b < K && r < KM) { jr[i][c] = a.joint[(size_t)b * KM + r]; ur[i][c] = a.U[(size_t)b * KM + r]; }
        }
    float r_am[2], r_av[2], r_res[2], r_t0[2], r_t[2], r_nrm[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + 1024 * u;
        r_am[u] = r_av[u] = r_res[u] = r_t0[u] = r_t[u] = 0.f; r_nrm[u] = 1.f;
        if (e < K * 32) {
            const size_t o = (size_t)(e >> 5) * D + d0 + (e & 31);
            r_am[u] = a.am[o]; r_av[u] = a.av[o]; r_res[u] = a.res[o]; r_t0[u] = a.text0[o]; r_t[u] = a.t[o];
            r_nrm[u] = a.nrm[e >> 5];
        }
    }
    AT_MARK(1);
    // ---- loss statistics: the code of align_loss_kernel (ua_align_common.h), L staged in LDS ----
    const ua::AlignStats st = ua::align_stats(a.L, Ls, K, P, s1, s2, dg, sd, sf, &s_arg, sc);
    AT_MARK(2);
    if (j == 0 && tid == 0 && a.loss_out) *a.loss_out = (float)(-st.tot / (double)K);
    const float mx = st.mx, corr = st.corr;
    const int amax = st.amax;
    // ---- weights w[b,r] = dLoss/dL[b,k] * exp(J[b,r] - L[b,k]) as a split operand; ge_b . e_b = sum_r w U ----
#pragma unroll
    for (int i = 0; i < kWR; ++i) {
        const int b = wv + 16 * i;
        if (b < K) {                                                  // wave-uniform
            float Gl = 0.f;
            if (lane < K) { const int ik = b * K + lane; Gl = P[ik] / mx; if (ik == amax) Gl += corr; }
            const float s = pow2_scale(wave_max(fabsf(Gl)));          // |w| <= max_k |G[b,k]|
            double gd = 0.0;
#pragma unroll
            for (int c = 0; c < kWC; ++c) {
                const int r = lane + 64 * c;
                if (r < KMp) {
                    float w = 0.f;
                    if (r < KM) {
                        const int ik = b * K + r / M;
                        float G = P[ik] / mx;
                        if (ik == amax) G += corr;
                        w = G * expf(jr[i][c] - Ls[ik]);              // d logsumexp / d joint = softmax
                        gd += (double)(w * ur[i][c]);
                    }
                    const float x = w * s;
                    const _Float16 h = (_Float16)x;
                    unsigned char* q = Aop + b * PITCH + (r >> 5) * 128 + (r & 31) * 2;
                    *reinterpret_cast<_Float16*>(q) = h;
                    *reinterpret_cast<_Float16*>(q + 64) = (_Float16)(x - (float)h);
                }
            }
            gd = wave_sum(gd);
            if (lane == 0) { s_gd[b] = (float)gd; s_sb[b] = 1.0f / s; }
        }
    }
    __syncthreads();
    AT_MARK(3);
    // ---- G[b, col] = sum_r w[b,r] S[col, r]: one 16 x 16 block per wave (query block mb, column block nb) ----
    const int nMB = (K + 15) >> 4;
    const int mb = wv >> 2, nb = wv & 3, lr = lane & 15, kg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wv < nMB * 4) {                                               // wave-uniform
        const unsigned char* pa = Aop + min(mb * 16 + lr, K - 1) * PITCH + kg * 16;
        const unsigned char* pb = Sop + (nb * 16 + lr) * PITCH + kg * 16;
        for (int ks = 0; ks < KMp / 32; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pa + ks * 128), al = *reinterpret_cast<const f16x8*>(pa + ks * 128 + 64);
            const f16x8 bh = *reinterpret_cast<const f16x8*>(pb + ks * 128), bl = *reinterpret_cast<const f16x8*>(pb + ks * 128 + 64);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        }
    }
    AT_MARK(4);
    __syncthreads();                                                  // every wave is done reading the weight operand
    float* Gs = reinterpret_cast<float*>(Aop);                        // [K][65]: G1 of 32 features | G2 of 32 features
    if (wv < nMB * 4) {
        const float ci = a.st_inv[j * 64 + nb * 16 + lr];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int b = mb * 16 + 4 * kg + jj;                      // accumulator register jj: output row 4*kg + jj, column lr
            if (b < K) Gs[b * 65 + nb * 16 + lr] = (acc[jj] * s_sb[b]) * ci;
        }
    }
    __syncthreads();
    AT_MARK(5);
    // ---- Adam on this block's 32 columns of every row (torch.optim.adam._single_tensor_adam; scalars as ua_residual_adam) ----
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + 1024 * u;
        if (e < K * 32) {
            const int b = e >> 5, dl = e & 31;
            const size_t o = (size_t)b * D + d0 + dl;
            const float eb = r_t[u] / r_nrm[u];
            const float ge = Gs[b * 65 + dl] - eb * Gs[b * 65 + 32 + dl];
            const float g = (ge - eb * s_gd[b]) / r_nrm[u];           // d(t/|t|)/dt applied to grad_e
            float m = r_am[u], v = r_av[u];
            m = m + a.w1 * (g - m);                                   // exp_avg.lerp_(grad, 1-beta1)
            v = v * a.beta2 + a.omb2 * (g * g);                       // mul_(beta2).addcmul_(g,g,1-beta2)
            const float den = sqrtf(v) / a.bc2_sqrt + a.adam_eps;
            const float r = r_res[u] + ((-a.step_size) * m) / den;    // addcdiv_(m, den, value=-step_size)
            a.am[o] = m; a.av[o] = v; a.res[o] = r;
            a.t[o] = r_t0[u] + r;
            if (a.ge_out) a.ge_out[o] = ge;
        }
    }
    AT_MARK(6);
#ifdef UA_ALIGN_TRACE
    if (j == 1 && tid == 0 && a.loss_out) for (int i = 0; i < 7; ++i) a.loss_out[1 + i] = (float)(at_t[i] - at_t[0]);
#endif
}

}  // namespace

extern "C" int ua_align_mfma_supported(int32_t K, int32_t M, int32_t D) { return envelope_ok(K, M, D) ? 1 : 0; }

extern "C" size_t ua_align_mfma_workspace_bytes(int32_t K, int32_t M, int32_t D) {
    if (!envelope_ok(K, M, D)) return 0;
    return align_ws_layout(nullptr, K, M, D).total;
}

extern "C" int ua_align_mfma_prepare(const float* mu, const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(workspace);
    if (!envelope_ok(K, M, D)) return UA_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return UA_EALIGN;
    const AlignWs w = align_ws_layout(reinterpret_cast<unsigned char*>(workspace), K, M, D);
    if (workspace_bytes < w.total) return UA_EWORKSPACE;
    const size_t lds = (size_t)64 * (w.KMp + 1) * 4;
    const int KM = K * M;
    const dim3 grid(D / 32 + (KM + 15) / 16);
    hipStream_t st = ua::as_stream(stream);
#define UA_PREP(N)                                                                                                                  \
    {                                                                                                                               \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(align_prepare_kernel<N>),                          \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH; \
        hipLaunchKernelGGL((align_prepare_kernel<N>), grid, dim3(1024), lds, st, mu, var, pi, KM, w.KMp, eps, w.f.Wop, w.f.rc, w.f.colscale, \
                           w.ST, w.st_inv);                                                                                          \
    }
    switch (D / 256) {
        case 1: UA_PREP(1); break; case 2: UA_PREP(2); break; case 3: UA_PREP(3); break; case 4: UA_PREP(4); break;
        case 5: UA_PREP(5); break; case 6: UA_PREP(6); break; case 7: UA_PREP(7); break; case 8: UA_PREP(8); break;
        default: return UA_EUNSUPPORTED;
    }
#undef UA_PREP
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_align_forward_mfma(const float* t, int32_t K, int32_t M, int32_t D, const void* workspace, size_t workspace_bytes,
                                     float* L, float* joint, float* U, float* nrm, void* stream) {
    UA_CHECK_PTR(t); UA_CHECK_PTR(workspace); UA_CHECK_PTR(L); UA_CHECK_PTR(joint); UA_CHECK_PTR(U); UA_CHECK_PTR(nrm);
    UA_CHECK_ALIGN16(t);
    if (!envelope_ok(K, M, D)) return UA_EUNSUPPORTED;
    const AlignWs w = align_ws_layout(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(workspace)), K, M, D);
    if (workspace_bytes < w.total) return UA_EWORKSPACE;
    const int KM = K * M;
    const dim3 grid((KM + 15) / 16, (K + 15) / 16);
    hipStream_t st = ua::as_stream(stream);
#define UA_FWD(N) hipLaunchKernelGGL((align_fwd_mfma_kernel<N>), grid, dim3(512), 0, st, t, w.f.Wop, w.f.rc, w.f.colscale, K, KM, M, joint, U, L, nrm)
    switch (D / 256) {
        case 1: UA_FWD(1); break; case 2: UA_FWD(2); break; case 3: UA_FWD(3); break; case 4: UA_FWD(4); break;
        case 5: UA_FWD(5); break; case 6: UA_FWD(6); break; case 7: UA_FWD(7); break; case 8: UA_FWD(8); break;
        default: return UA_EUNSUPPORTED;
    }
#undef UA_FWD
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_align_step_mfma(const float* L, const float* joint, const float* U, const float* nrm, const float* text0,
                                  float* res, float* adam_m, float* adam_v, float* t, int32_t K, int32_t M, int32_t D, int32_t step,
                                  double lr, double beta1, double beta2, double adam_eps, const void* workspace, size_t workspace_bytes,
                                  float* loss, float* grad_e, void* stream) {
    UA_CHECK_PTR(L); UA_CHECK_PTR(joint); UA_CHECK_PTR(U); UA_CHECK_PTR(nrm); UA_CHECK_PTR(text0); UA_CHECK_PTR(res);
    UA_CHECK_PTR(adam_m); UA_CHECK_PTR(adam_v); UA_CHECK_PTR(t); UA_CHECK_PTR(workspace);
    if (step <= 0) return UA_EINVAL;
    if (!envelope_ok(K, M, D)) return UA_EUNSUPPORTED;
    const AlignWs w = align_ws_layout(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(workspace)), K, M, D);
    if (workspace_bytes < w.total) return UA_EWORKSPACE;
    AlignBwdArgs a{};
    a.L = L; a.joint = joint; a.U = U; a.nrm = nrm; a.t = t; a.text0 = text0; a.res = res; a.am = adam_m; a.av = adam_v;
    a.ST = w.ST; a.st_inv = w.st_inv; a.loss_out = loss; a.ge_out = grad_e;
    a.K = K; a.M = M; a.D = D; a.KM = K * M; a.KMp = w.KMp;
    // scalar bookkeeping exactly as ua_residual_adam (torch.optim.adam._single_tensor_adam, python floats = double)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.adam_eps = (float)adam_eps;
    const size_t lds = bwd_lds_bytes(K, w.KMp);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(align_bwd_adam_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return UA_ELAUNCH;
    hipLaunchKernelGGL(align_bwd_adam_mfma_kernel, dim3(D / 32), dim3(1024), lds, ua::as_stream(stream), a);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
