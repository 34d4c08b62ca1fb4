// C = A @ W^T (+ fused epilogue) on the CDNA4 matrix cores with EXACT fp32 arithmetic
// (v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bit-identical to an fmaf chain).
//
// Serves every Linear / Conv1d(k=1) of the Uni3D encoder: mini-PointNet convs
// (models/point_encoder.py:133-144), encoder2trans / pos_embed / trans2embed (:175-186) and the
// q/k/v/proj/fc1/fc2 projections of the 24 timm EvaBlocks (:217-218) - >99 % of the path's FLOPs.
// The reference runs Uni3D in fp32 (utils/load_models.py:51-61) and the parity bar is 1e-3 on
// 100*cos logits, so the contraction stays fp32 (bf16 MFMA would not hold it over 24 layers).
//
// Tiling: 128x128 output tile per 256-thread workgroup, 4 wavefronts as 2x2, each wave a 64x64
// sub-tile = 2x2 MFMA blocks of 32x32 (64 accumulator VGPRs).  K is consumed in 32-wide slabs
// that are double-buffered in LDS ([128][36] floats: the +4 pad makes every ds_read_b128 of a
// 16-lane group hit 16 distinct 16-byte bank slots).  A and W are both K-contiguous, so both
// operands use the same fragment read: lane (i = lane&31, h = lane>>5) fetches row i, columns
// 8c+4h..8c+4h+3 with ONE ds_read_b128 and feeds them to four consecutive MFMAs - the k index
// inside a slab is permuted identically for A and W, which leaves the sum unchanged.
// The next slab is fetched global->VGPR while the current one is multiplied (the f32 MFMA is
// 64 cycles per instruction, so 64 MFMAs per slab hide the staging entirely).
// Workgroup ids are remapped so that each XCD (blockIdx % 8) owns a contiguous band of row
// panels: the A panel of a band is fetched from HBM once and re-used from that XCD's L2 by all
// column tiles.
#include "ua_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = BK + 4;                       // floats per LDS row
constexpr int TILE_F = BM * LDS_LD;                  // floats per operand tile

struct GemmArgs {
    const float* A; const float* W; float* C;
    long long lda, ldw, ldc;
    int M, N, K;
    const float* bias; const float* scale; const float* shift;
    const float* rga; long long ldg; int group_rows;
    int act;
    const float* residual; long long ldr;
    float* pool; long long ldp;
    int tiles_n, n_tiles;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// Branch-free guarded 16-byte load: the address is clamped into the matrix (always legal), the
// value is zeroed by a select when the element is outside (rows >= nrows, k >= K).  Out-of-range
// rows only ever feed accumulators that are never stored; out-of-range k must contribute 0.
// (The select is applied when the registers are stored to LDS, AFTER the MFMAs of the current
// slab, so the loads stay in flight behind the matrix work.)
__device__ __forceinline__ float4 ld_clamped(const float* base, long long ld, int row, int nrows, int k, int K) {
    const int rc = min(row, nrows - 1), kc = min(k, K - 4);
    return *reinterpret_cast<const float4*>(base + (size_t)rc * ld + kc);
}
__device__ __forceinline__ float4 mask4(float4 v, bool ok) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// K loop of one workgroup.  FULL = interior tile: branch-free.  !FULL = edge tile: 32x32 MFMA
// blocks that lie completely outside the matrix are skipped (scalar predicates `on`).
template <bool FULL>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, int m0, int n0, int tid, int wm, int wn,
                                              int li, int lh, const bool (&on)[2][2], f32x16 (&acc)[2][2]) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2 buffers][A tile | W tile]
    // staging map: 4 float4 per thread per operand; f = tid + 256*i -> row f/8, k-quad f%8
    float4 ra[4], rw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kq = (f & 7) * 4;
            ra[i] = ld_clamped(g.A, g.lda, m0 + row, g.M, k0 + kq, g.K);
            rw[i] = ld_clamped(g.W, g.ldw, n0 + row, g.N, k0 + kq, g.K);
        }
    };
    auto stash = [&](int buf, int k0) {
        float* a = lds + buf * 2 * TILE_F;
        float* w = a + TILE_F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kq = (f & 7) * 4;
            const bool k_ok = (k0 + kq) < g.K;
            *reinterpret_cast<float4*>(a + row * LDS_LD + kq) = mask4(ra[i], k_ok && (m0 + row) < g.M);
            *reinterpret_cast<float4*>(w + row * LDS_LD + kq) = mask4(rw[i], k_ok && (n0 + row) < g.N);
        }
    };
    const int nk = (g.K + BK - 1) / BK;
    fetch(0);
    stash(0, 0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        fetch((t + 1) * BK);        // unconditional (addresses are clamped): keeps the loop body straight-line
        const float* a = lds + buf * 2 * TILE_F + (wm * 64 + li) * LDS_LD + lh * 4;
        const float* w = lds + buf * 2 * TILE_F + TILE_F + (wn * 64 + li) * LDS_LD + lh * 4;
        // fragment registers are double-buffered: chunk c+1 is read from LDS while chunk c's 16
        // MFMAs (1024 cycles) run, so no MFMA waits on an LDS read it was just issued behind
        float4 fa[2][2], fw[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[0][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD);
            fw[0][i] = *reinterpret_cast<const float4*>(w + i * 32 * LDS_LD);
        }
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[nxt][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS_LD + (c + 1) * 8);
                    fw[nxt][i] = *reinterpret_cast<const float4*>(w + i * 32 * LDS_LD + (c + 1) * 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the prefetch reads ABOVE this chunk's MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (FULL || on[i][j]) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fw[cur][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fw[cur][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fw[cur][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fw[cur][j].w, acc[i][j], 0, 0, 0);
                    }
        }
        stash(buf ^ 1, (t + 1) * BK);   // past the last slab this stores zeros nobody reads
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256, 2) gemm_f32_mfma_kernel(GemmArgs g) {
    // ---- XCD-aware tile id (bijective for any n_tiles) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = g.n_tiles >> 3, r = g.n_tiles & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Interior tiles (block-uniform test on kernel arguments) take the branch-free main loop.  Edge
    // tiles - here the thin tile that holds the CLS rows and the last column tile of the 2730-wide
    // layers - skip the MFMA blocks outside the matrix; the predicates are made scalar with
    // readfirstlane so the skip is an s_cbranch, not an exec mask.
    const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    const int swm = __builtin_amdgcn_readfirstlane(wm), swn = __builtin_amdgcn_readfirstlane(wn);
    bool on[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            on[i][j] = ((m0 + swm * 64 + i * 32) < g.M) && ((n0 + swn * 64 + j * 32) < g.N);
    if (full) gemm_mainloop<true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
    else      gemm_mainloop<false>(g, m0, n0, tid, wm, wn, li, lh, on, acc);

    // ---- epilogue ----------------------------------------------------------------------------
    // acc[i][j][e] holds row (e&3)+8(e>>2)+4*lh, column li of a 32x32 block: a lane owns single
    // floats of 64 different rows.  The wave's 64x64 sub-tile is transposed through LDS (free after
    // the main loop) so that every lane then owns float4s of 4 consecutive columns: bias / BN affine /
    // activation / residual / max-pool all run on float4s and C is written with 16-byte stores in
    // 256-byte row segments (64 store instructions per lane become 16).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EP_LD = 68;                                        // floats per staged row
    float* stage = lds + wid * 64 * EP_LD;                           // 4 x 17,408 B <= 73,728 B
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * EP_LD + j * 32 + li] = acc[i][j][e];
    // (no barrier: each wave reads back only what it wrote; LDS ops of one wave complete in order)
    const bool swiglu = g.act == 3;
    const int lr = lane >> 4, c4 = (lane & 15) * 4;                  // row within a 4-row pass, first column
    const int ncol0 = n0 + wn * 64 + c4;                             // input column of .x
    const int nout0 = swiglu ? (n0 + wn * 64) / 2 + c4 : ncol0;      // output column of .x
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
    float4 pm = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const int mbase = m0 + wm * 64;
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int r = p * 4 + lr, m = mbase + r;
        float4 v = *reinterpret_cast<const float4*>(stage + r * EP_LD + c4);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
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
            pm.x = fmaxf(pm.x, v.x); pm.y = fmaxf(pm.y, v.y); pm.z = fmaxf(pm.z, v.z); pm.w = fmaxf(pm.w, v.w);
        }
    }
    if (g.pool) {                                                    // max over this wave's 64 rows
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            pm.x = fmaxf(pm.x, __shfl_xor(pm.x, o)); pm.y = fmaxf(pm.y, __shfl_xor(pm.y, o));
            pm.z = fmaxf(pm.z, __shfl_xor(pm.z, o)); pm.w = fmaxf(pm.w, __shfl_xor(pm.w, o));
        }
        if (lr == 0 && mbase < g.M) {
            float* q = g.pool + (size_t)(mbase / 64) * g.ldp + nout0;
            if (nout0 + 3 < n_out) *reinterpret_cast<float4*>(q) = pm;
            else { if (nout0 < n_out) q[0] = pm.x; if (nout0 + 1 < n_out) q[1] = pm.y; if (nout0 + 2 < n_out) q[2] = pm.z; }
        }
    }
}

}  // namespace

extern "C" int ua_gemm(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                       int32_t M, int32_t N, int32_t K, const ua_gemm_epilogue* ep, void* stream) {
    UA_CHECK_PTR(A); UA_CHECK_PTR(W);
    UA_CHECK_ALIGN16(A); UA_CHECK_ALIGN16(W);
    if (M <= 0 || N <= 0 || K <= 0) return UA_EINVAL;
    if (K % 4 || lda % 4 || ldw % 4) return UA_EALIGN;
    if (lda < K || ldw < K) return UA_EINVAL;
    GemmArgs g{};
    g.A = A; g.W = W; g.C = C; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    if (ep) {
        g.bias = ep->bias; g.scale = ep->scale; g.shift = ep->shift;
        g.rga = ep->rowgroup_add; g.ldg = ep->ldg; g.group_rows = ep->group_rows;
        g.act = ep->act; g.residual = ep->residual; g.ldr = ep->ldr;
        g.pool = ep->pool_out; g.ldp = ep->ldp;
        if (g.act < 0 || g.act > 3) return UA_EINVAL;
        if (g.rga && g.group_rows <= 0) return UA_EINVAL;
        if (g.pool && (ep->pool_rows != 64 || M % 64 != 0)) return UA_EUNSUPPORTED;
        if (g.act == 3 && (N % 64 != 0 || g.rga || g.scale || g.shift)) return UA_EUNSUPPORTED;
    }
    if (!C && !g.pool) return UA_EINVAL;
    const int n_out = g.act == 3 ? N / 2 : N;
    if (C && ldc < n_out) return UA_EINVAL;
    // float4 epilogue: row starts of C / residual / pool / rowgroup_add must be 16-byte aligned
    if (C && (ldc % 4 || (reinterpret_cast<uintptr_t>(C) & 15))) return UA_EALIGN;
    if (g.residual && (g.ldr % 4 || (reinterpret_cast<uintptr_t>(g.residual) & 15))) return UA_EALIGN;
    if (g.pool && (g.ldp % 4 || (reinterpret_cast<uintptr_t>(g.pool) & 15))) return UA_EALIGN;
    if (g.rga && (g.ldg % 4 || (reinterpret_cast<uintptr_t>(g.rga) & 15))) return UA_EALIGN;
    if ((g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 15)) || (g.scale && (reinterpret_cast<uintptr_t>(g.scale) & 15)) ||
        (g.shift && (reinterpret_cast<uintptr_t>(g.shift) & 15))) return UA_EALIGN;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    g.tiles_n = tiles_n; g.n_tiles = tiles_m * tiles_n;
    const size_t lds = (size_t)4 * TILE_F * sizeof(float);           // 73,728 B
    static bool attr_set = false;                                      // idempotent, benign race
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_mfma_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3(g.n_tiles), dim3(256), lds, ua::as_stream(stream), g);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
