// K x K statistics of the text-alignment loss (Uni_Adapter.py:240-253 and its gradient), shared by
// align_loss_kernel (ua_logits.hip) and the fused align_step_kernel (ua_gmm.hip) so that both
// associate every sum identically.
//   Ln = L / max(L);  P = exp(exp(Ln));  loss = -mean(diag P / rowsum) - mean(diag P / colsum)
// One workgroup of 1024 threads; on return (after an internal barrier) the LDS arrays hold
//   P[i] = dLoss/dLn_i,  s1 = row sums, s2 = column sums, dg = diagonal of P
// and every thread holds the scalars of AlignStats.  Row/column sums: one wavefront per sum, lanes
// strided over the K terms, fp64 butterfly - no serial K-long loops, 7 barriers in total.
#pragma once
#include "ua_common.h"

namespace ua {

struct AlignStats {
    float mx;        // max(L)
    int amax;        // first index attaining it
    float corr;      // -sum(dLn * L) / mx^2 : the extra gradient of the max() entry
    double tot;      // sum_i dg/s1 + dg/s2  (loss = -tot / K)
};

// Lg: L in global memory (K*K); Ls: optional LDS copy to fill and use (nullptr: re-read Lg).
// sd (>= 16 doubles), sf (>= 16 floats), s_arg: LDS scratch.
__device__ __forceinline__ AlignStats align_stats(const float* __restrict__ Lg, float* Ls, int K, float* P, float* s1, float* s2,
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

}  // namespace ua
