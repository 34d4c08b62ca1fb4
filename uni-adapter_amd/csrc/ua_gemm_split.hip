// C = A @ W^T with fp32-GEMM accuracy on the f16 matrix cores of gfx950 (3-term error-compensated
// split, fp32 accumulate).  Same call sites as ua_gemm.hip (every Linear / Conv1d(k=1) of the Uni3D
// encoder); see include/uniadapter.h: ua_gemm_split / ua_split_f16 for the H32L32 operand format.
//
// Why: gfx950 has no TF32/xf32 MFMA, and v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TFLOP/s).
// v_mfma_f32_32x32x16_f16 runs 16x faster.  With x*s = hi + lo (two fp16, 22 significant bits) the
// product a*w = ah*wh + ah*wl + al*wh + O(2^-22) needs three f16 MFMAs: 16/3 = 5.3x the f32-MFMA
// roofline at the accuracy of an fp32 GEMM (products of fp16 are exact in fp32; accumulation is
// fp32 in both cases and dominates the error for K >= 64).
//
// Tiling: 256x256 output tile per 512-thread workgroup (8 waves as 2x4, each 128x64 = 4x2 MFMA
// blocks, 128 accumulator VGPRs), K consumed in 32-column slabs.  A slab row of one operand is one
// 128-byte line of the H32L32 matrix (64 B hi | 64 B lo) and lands in a 144-byte LDS row (the
// 16-byte pad makes the ds_read_b128 fragment reads conflict-free).  Two LDS buffers of
// 2 x 256 x 144 B = 144 KiB: one workgroup per CU, two waves per SIMD.  Per slab a wave issues
// 24 ds_read_b128 and 48 MFMAs (1536 matrix-pipe cycles), the workgroup stages 64 KiB.
// Full row panels are balanced over the XCDs; the thin CLS-row tiles come last (1/8 round).
#include "ua_gemm_common.h"

namespace {

using ua::f32x16;
using ua::f16x8;
using ua::EpiArgs;

constexpr int SBM = 256, SBN = 256, SBK = 32;
constexpr int S_ROW = 144;                           // LDS bytes per slab row
constexpr int S_TILE = SBM * S_ROW;                  // 36,864 B per operand tile

struct SplitArgs {
    const unsigned char* A; const unsigned char* W;
    long long lda, ldw;                              // row pitch in columns (4 bytes per column)
    int M, N, K;
    EpiArgs ep;
    int tiles_n, n_tiles;
};

__device__ __forceinline__ uint4 ld_row_clamped(const unsigned char* base, long long ld, int row, int nrows, int k0, int c) {
    const int rc = min(row, nrows - 1);
    return *reinterpret_cast<const uint4*>(base + ((size_t)rc * ld + k0) * 4 + c * 16);
}

template <bool FULL>
__device__ __forceinline__ void split_mainloop(const SplitArgs& g, int m0, int n0, int tid, int wm, int wn, int li, int lh,
                                               const bool (&on)[4][2], f32x16 (&acc)[4][2]) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];   // [2 buffers][A tile | W tile]
    uint4 ra[4], rw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 512 * i, row = f >> 3, c = f & 7;
            ra[i] = ld_row_clamped(g.A, g.lda, m0 + row, g.M, k0, c);
            rw[i] = ld_row_clamped(g.W, g.ldw, n0 + row, g.N, k0, c);
        }
    };
    auto stash = [&](int buf) {
        unsigned char* a = slds + buf * 2 * S_TILE;
        unsigned char* w = a + S_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 512 * i, row = f >> 3, c = f & 7;
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(a + row * S_ROW + c * 16) = (FULL || (m0 + row) < g.M) ? ra[i] : z;
            *reinterpret_cast<uint4*>(w + row * S_ROW + c * 16) = (FULL || (n0 + row) < g.N) ? rw[i] : z;
        }
    };
    const int nk = g.K / SBK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        fetch(min(t + 1, nk - 1) * SBK);                       // unconditional (last one re-reads slab nk-1)
        const unsigned char* a = slds + buf * 2 * S_TILE + (wm * 128 + li) * S_ROW + lh * 16;
        const unsigned char* w = slds + buf * 2 * S_TILE + S_TILE + (wn * 64 + li) * S_ROW + lh * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // two k16 steps per slab
            f16x8 ah[4], al[4], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(a + i * 32 * S_ROW + s * 32);
                al[i] = *reinterpret_cast<const f16x8*>(a + i * 32 * S_ROW + 64 + s * 32);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wh[j] = *reinterpret_cast<const f16x8*>(w + j * 32 * S_ROW + s * 32);
                wl[j] = *reinterpret_cast<const f16x8*>(w + j * 32 * S_ROW + 64 + s * 32);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (FULL || on[i][j]) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
                    }
        }
        stash(buf ^ 1);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512, 2) gemm_split_kernel(SplitArgs g) {
    // XCD-aware tile order.  Blocks are dealt to the 8 XCDs round-robin by blockIdx; the FULL row
    // panels are chunked so that every XCD owns a contiguous band (A-panel reuse in its L2) and an
    // equal number of full tiles; the tiles of a partial last row panel (here: the 32 CLS rows, 1/8
    // of the MFMA work each) come last in block order, one per XCD, so they add 1/8 of a round
    // instead of making XCDs 0..r-1 run a second full round.
    const int bid = blockIdx.x;
    const int n_full = (g.M / SBM) * g.tiles_n;
    int tile = bid;
    if (bid < n_full) {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = n_full >> 3, r = n_full & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * SBM, n0 = tn * SBN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool full = (m0 + SBM <= g.M) && (n0 + SBN <= g.N);
    const int swm = __builtin_amdgcn_readfirstlane(wm), swn = __builtin_amdgcn_readfirstlane(wn);
    bool on[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            on[i][j] = ((m0 + swm * 128 + i * 32) < g.M) && ((n0 + swn * 64 + j * 32) < g.N);
    if (full) split_mainloop<true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
    else      split_mainloop<false>(g, m0, n0, tid, wm, wn, li, lh, on, acc);

    // epilogue: two 64x64 halves through the wave's LDS staging area (8 x 17,408 B <= 147,456 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
    float* stage = reinterpret_cast<float*>(slds) + wid * 64 * ua::EP_LD;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (m0 + wm * 128 + h * 64 >= g.M) break;              // wave-uniform
        ua::stage_64x64(stage, acc[2 * h][0], acc[2 * h][1], acc[2 * h + 1][0], acc[2 * h + 1][1], li, lh);
        ua::epilogue_64x64(g.ep, stage, lane, m0 + wm * 128 + h * 64, n0 + wn * 64);
    }
}

// fp32 -> H32L32, one thread per 4 columns
__global__ void __launch_bounds__(256)
split_f16_kernel(const float* __restrict__ x, long long ldx, unsigned char* __restrict__ y, long long ldy,
                 int rows, int cols, int cols_pad, float scale) {
    const int c4 = cols_pad / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * c4) return;
    const int row = (int)(i / c4), c = (int)(i % c4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = x + (size_t)row * ldx + c;
    if (c + 3 < cols) v = *reinterpret_cast<const float4*>(p);
    else { if (c < cols) v.x = p[0]; if (c + 1 < cols) v.y = p[1]; if (c + 2 < cols) v.z = p[2]; }
    ua::f16x4 hi, lo;
    ua::split4(v, scale, hi, lo);
    unsigned char* q = y + (size_t)row * ldy * 4 + ua::split_col_off(c);
    *reinterpret_cast<ua::f16x4*>(q) = hi;
    *reinterpret_cast<ua::f16x4*>(q + 64) = lo;
}

}  // namespace

extern "C" int ua_split_f16(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols,
                            int32_t cols_pad, float scale, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(y); UA_CHECK_ALIGN16(x); UA_CHECK_ALIGN16(y);
    if (rows <= 0 || cols <= 0 || cols_pad < cols || !(scale > 0.f)) return UA_EINVAL;
    if (ldx % 4 || ldy % 32 || cols_pad % 32) return UA_EALIGN;
    if (ldx < cols || ldy < cols_pad) return UA_EINVAL;
    const size_t n = (size_t)rows * (cols_pad / 4);
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ua::as_stream(stream),
                       x, (long long)ldx, reinterpret_cast<unsigned char*>(y), (long long)ldy, rows, cols, cols_pad, scale);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_gemm_split(const void* A, int64_t lda, float a_scale, const void* W, int64_t ldw, float w_scale,
                             float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                             const ua_gemm_epilogue* ep, void* stream) {
    UA_CHECK_PTR(A); UA_CHECK_PTR(W);
    UA_CHECK_ALIGN16(A); UA_CHECK_ALIGN16(W);
    if (M <= 0 || N <= 0 || K <= 0 || !(a_scale > 0.f) || !(w_scale > 0.f)) return UA_EINVAL;
    if (K % 32 || lda % 32 || ldw % 32) return UA_EALIGN;
    if (lda < K || ldw < K) return UA_EINVAL;
    SplitArgs g{};
    g.A = reinterpret_cast<const unsigned char*>(A); g.W = reinterpret_cast<const unsigned char*>(W);
    g.lda = lda; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    if (int e = ua::fill_epi(g.ep, ep, M, N, C, ldc)) return e;
    g.ep.acc_scale = 1.0f / (a_scale * w_scale);
    const int tiles_m = (M + SBM - 1) / SBM, tiles_n = (N + SBN - 1) / SBN;
    g.tiles_n = tiles_n; g.n_tiles = tiles_m * tiles_n;
    const size_t lds = (size_t)4 * S_TILE;                            // 147,456 B
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_split_kernel, dim3(g.n_tiles), dim3(512), lds, ua::as_stream(stream), g);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
