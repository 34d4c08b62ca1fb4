// K x K statistics of the text-alignment loss (Uni_Adapter.py:240-253 and its gradient), shared by
// align_loss_kernel (ua_logits.hip) and the fused align_step_kernel (ua_gmm.hip) so that both
// associate every sum identically.
//   Ln = L / max(L);  P = exp(exp(Ln));  loss = -mean(diag P / rowsum) - mean(diag P / colsum)
// One workgroup of 1024 threads; on return (after an internal barrier) the LDS arrays hold
//   P[i] = dLoss/dLn_i,  s1 = row sums, s2 = column sums, dg = diagonal of P
// and every thread holds the scalars of AlignStats.  Generic form (any K): row/column sums one wavefront per sum, lanes
// strided over the K terms, fp64 butterfly - 7 barriers in total; K <= 64 takes the wave-per-row form below.
#pragma once
#include "ua_common.h"

namespace ua {

struct AlignStats {
    float mx;        // max(L)
    int amax;        // first index attaining it
    float corr;      // -sum(dLn * L) / mx^2 : the extra gradient of the max() entry
    double tot;      // sum_i dg/s1 + dg/s2  (loss = -tot / K)
};

// K <= 64: wave-per-row form.  Row r of the K x K matrix lives in the lanes of one wave (lane = column; wave w owns rows
// w, w+16, w+32, w+48), so a row sum is one wavefront reduction on registers, a column sum is a per-lane running sum over the
// wave's rows plus a 16-term cross-wave sum in a fixed order, and the whole statistic needs three workgroup barriers
// (maximum | row / column partials | sum of dLn*L) instead of seven, and no strided LDS walks.
// sc (>= 16*64 doubles) is extra LDS scratch of this form.
__device__ __forceinline__ AlignStats align_stats_rows(const float* __restrict__ Lg, float* Ls, int K, float* P, float* s1, float* s2,
                                                       float* dg, double* sd, float* sf, int* s_arg, double* sc) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool col_ok = lane < K;
    AlignStats st;
    float l[4], E[4], Pv[4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wv + 16 * i;
        l[i] = -INFINITY;
        if (r < K && col_ok) { l[i] = Lg[r * K + lane]; if (Ls) Ls[r * K + lane] = l[i]; }
        mx = fmaxf(mx, l[i]);
    }
    if (tid == 0) *s_arg = K * K;
    mx = wave_max(mx);
    if (lane == 0) sf[wv] = mx;
    __syncthreads();                                                                // 1
    mx = sf[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sf[i]);
    st.mx = mx;
    double colp = 0.0;                                                              // this lane's column, summed over the wave's rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wv + 16 * i;
        E[i] = 0.f; Pv[i] = 0.f;
        if (r < K) {                                                                // wave-uniform
            if (col_ok) {
                if (l[i] == mx) atomicMin(s_arg, r * K + lane);                     // first arg-max
                E[i] = expf(l[i] / mx);                                             // :248-249
                Pv[i] = expf(E[i]);
            }
            const double rs = wave_sum((double)Pv[i]);                              // :251 (lanes >= K contribute 0)
            colp += (double)Pv[i];
            if (lane == 0) s1[r] = (float)rs;
            if (lane == r) dg[r] = Pv[i];
        }
    }
    sc[wv * 64 + lane] = colp;
    __syncthreads();                                                                // 2
    st.amax = *s_arg;
    double cs = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) cs += sc[i * 64 + lane];                           // :252, fixed order
    const float s2c = (float)cs;
    const float dgc = col_ok ? dg[lane] : 0.f, s1c = col_ok ? s1[lane] : 1.f;
    if (wv == 0 && col_ok) s2[lane] = s2c;
    double part = 0.0;                                                              // every wave forms the same total
    if (col_ok) part = (double)(dgc / s1c) + (double)(dgc / s2c);
    st.tot = wave_sum(part);                                                        // :253
    // dLoss/dP -> dLoss/dLn ; S = sum dLn * L (for the max() path)
    const float invK = 1.f / (float)K;
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wv + 16 * i;
        if (r < K && col_ok) {
            const float s1r = s1[r], dgr = dg[r];                                   // (written before barrier 2)
            const float eye = (r == lane) ? 1.f : 0.f;
            const float dP = -invK * (eye / s1r - dgr / (s1r * s1r)) - invK * (eye / s2c - dgc / (s2c * s2c));
            const float dLn = dP * Pv[i] * E[i];
            sacc += (double)dLn * (double)l[i];
            P[r * K + lane] = dLn;                                                  // P holds dLoss/dLn
        }
    }
    const double S = block_sum(sacc, sd);                                           // (2 barriers; P / s2 writes are visible after them)
    st.corr = (float)(-S / ((double)mx * (double)mx));
    return st;
}

// Lg: L in global memory (K*K); Ls: optional LDS copy to fill and use (nullptr: re-read Lg).
// sd (>= 16 doubles), sf (>= 16 floats), s_arg: LDS scratch.
__device__ __forceinline__ AlignStats align_stats_generic(const float* __restrict__ Lg, float* Ls, int K, float* P, float* s1, float* s2,
                                                  float* dg, double* sd, float* sf, int* s_arg) {
    const int n = K * K, tid = threadIdx.x, nt = 1024, wv = tid >> 6, lane = tid & 63;
    const float* Lr = Ls ? Ls : Lg;
    AlignStats st;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += nt) { const float l = Lg[i]; if (Ls) Ls[i] = l; mx = fmaxf(mx, l); }
    if (tid == 0) *s_arg = n;
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) sf[wv] = mx;
    __syncthreads();
    mx = sf[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sf[i]);
    st.mx = mx;
    for (int i = tid; i < n; i += nt) {
        const float l = Lr[i];
        if (l == mx) atomicMin(s_arg, i);                                           // first arg-max
        P[i] = expf(expf(l / mx));                                                  // :248-249
    }
    __syncthreads();
    st.amax = *s_arg;
    for (int r = wv; r < 2 * K; r += 16) {                                          // :251-252
        double s = 0.0;
        if (r < K) { for (int j = lane; j < K; j += 64) s += (double)P[r * K + j]; }
        else { const int c = r - K; for (int j = lane; j < K; j += 64) s += (double)P[j * K + c]; }
        s = wave_sum(s);
        if (lane == 0) { if (r < K) s1[r] = (float)s; else s2[r - K] = (float)s; }
    }
    for (int i = tid; i < K; i += nt) dg[i] = P[i * K + i];
    __syncthreads();
    double part = 0.0;                                                              // every wave forms the same total
    for (int i = lane; i < K; i += 64) part += (double)(dg[i] / s1[i]) + (double)(dg[i] / s2[i]);
    st.tot = wave_sum(part);                                                        // :253
    // dLoss/dP -> dLoss/dLn ; S = sum dLn * L (for the max() path)
    const float invK = 1.f / (float)K;
    double sacc = 0.0;
    for (int i = tid; i < n; i += nt) {
        const int r = (int)((unsigned)i / (unsigned)K), c = i - r * K;
        const float eye = (r == c) ? 1.f : 0.f;
        const float dP = -invK * (eye / s1[r] - dg[r] / (s1[r] * s1[r])) - invK * (eye / s2[c] - dg[c] / (s2[c] * s2[c]));
        const float l = Lr[i];
        const float E = expf(l / mx);
        const float dLn = dP * P[i] * E;
        sacc += (double)dLn * (double)l;
        P[i] = dLn;                                  // reuse LDS: P now holds dLoss/dLn
    }
    const double S = block_sum(sacc, sd);            // (2 barriers; P writes are visible after them)
    st.corr = (float)(-S / ((double)mx * (double)mx));
    return st;
}


// sc: >= 16*64 doubles of LDS (only used for K <= 64)
__device__ __forceinline__ AlignStats align_stats(const float* __restrict__ Lg, float* Ls, int K, float* P, float* s1, float* s2,
                                                  float* dg, double* sd, float* sf, int* s_arg, double* sc) {
    if (K <= 64) return align_stats_rows(Lg, Ls, K, P, s1, s2, dg, sd, sf, s_arg, sc);          // workgroup-uniform
    return align_stats_generic(Lg, Ls, K, P, s1, s2, dg, sd, sf, s_arg);
}

}  // namespace ua
