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
// 128-byte line of the H32L32 matrix (64 B hi | 64 B lo), copied global->LDS by LDS-DMA into an
// XOR-swizzled image of unpadded 128-byte rows (see split_mainloop).  Two LDS buffers of
// 2 x 256 x 128 B = 128 KiB: one workgroup per CU, two waves per SIMD.  Per slab a wave issues
// 8 LDS-DMA copies, 24 ds_read_b128 and 48 MFMAs (1536 matrix-pipe cycles); the workgroup stages 64 KiB.
// Full row panels are balanced over the XCDs; the thin CLS-row tiles come last (1/8 round).
#include "ua_gemm_common.h"
#include <cstdlib>

namespace {

using ua::f32x16;
using ua::f16x8;
using ua::EpiArgs;

constexpr int SBM = 256, SBN = 256, SBK = 32;
constexpr int S_ROW = 128;                           // LDS bytes per slab row: 8 chunks of 16 B (4 hi | 4 lo)
constexpr int S_TILE = SBM * S_ROW;                  // 32,768 B per operand tile
constexpr int S_LDS = 8 * 64 * ua::EP_LD * 4;        // 139,264 B: max(main loop 131,072, epilogue staging)

struct SplitArgs {
    const unsigned char* A; const unsigned char* W;
    long long lda, ldw;                              // row pitch in columns (4 bytes per column)
    int M, N, K;
    EpiArgs ep;
    int tiles_n, n_tiles;
    int stagger;                                     // waves 4-7 run half a slab behind waves 0-3
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// Staging is LDS-DMA (global_load_lds_dwordx4): every wave-instruction copies 64 x 16 B from per-lane
// global addresses to 1 KiB of CONTIGUOUS LDS (wave-uniform base + lane*16) - no staging VGPRs, no
// ds_write.  The LDS image therefore has unpadded 128-byte rows; bank conflicts of the fragment
// reads are removed by an XOR swizzle applied on the SOURCE side: LDS chunk position p of row r
// holds global chunk p ^ ((r>>1)&7), and the reads apply the same involution.  (16 lanes of a
// ds_read_b128 group read 16 distinct rows r; (r&1, (r>>1)&7 ^ c) are 16 distinct 16-byte slots of the
// 256-byte bank row.)  Rows beyond M / N are clamped to the last row: they only feed accumulators
// that are never stored, so no zero-fill is needed (K is a multiple of 32).
template <bool FULL, bool LATE>
__device__ __forceinline__ void split_mainloop(const SplitArgs& g, int m0, int n0, int tid, int wm, int wn, int li, int lh,
                                               const bool (&on)[4][2], f32x16 (&acc)[4][2]) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];   // [2 buffers][A tile | W tile]
    const int lane = tid & 63;
    const unsigned char* ga[4];
    const unsigned char* gw[4];
    int ldo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 512 * i, row = f >> 3, p = f & 7, c = p ^ ((row >> 1) & 7);
        ga[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda * 4 + c * 16;
        gw[i] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw * 4 + c * 16;
        ldo[i] = (f - lane) * 16;                              // wave-uniform LDS byte offset of this copy
    }
    auto issue = [&](int buf, int k0) {
        unsigned char* a = slds + buf * 2 * S_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + (size_t)k0 * 4), (lds_ptr_t)(a + ldo[i]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + (size_t)k0 * 4), (lds_ptr_t)(a + S_TILE + ldo[i]), 16, 0, 0);
        }
    };
    // fragment read offsets: lane (row li of a 32-row block, k-half lh), k16 step s: hi chunk 2s+lh, lo chunk 4+2s+lh
    const int sw = (li >> 1) & 7;                              // row-dependent swizzle ((row>>1)&7; block offsets are multiples of 32)
    f16x8 ah[4], al[4], wh[2], wl[2];
    auto read_frags = [&](int buf, int s) {
        const unsigned char* a = slds + buf * 2 * S_TILE + (wm * 128 + li) * S_ROW;
        const unsigned char* w = slds + buf * 2 * S_TILE + S_TILE + (wn * 64 + li) * S_ROW;
        const int ph = ((2 * s + lh) ^ sw) * 16, pl = ((4 + 2 * s + lh) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(a + i * 32 * S_ROW + ph);
            al[i] = *reinterpret_cast<const f16x8*>(a + i * 32 * S_ROW + pl);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wh[j] = *reinterpret_cast<const f16x8*>(w + j * 32 * S_ROW + ph);
            wl[j] = *reinterpret_cast<const f16x8*>(w + j * 32 * S_ROW + pl);
        }
    };
    auto mfma_block = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (FULL || on[i][j]) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
                }
    };
    const int nk = g.K / SBK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Between two barriers every wave does 2 fragment reads + 2 x 24 MFMAs.  !LATE (waves 0-3):
    // [R(t,0) M(t,0) R(t,1) M(t,1)].  LATE (waves 4-7, the second wave of every SIMD) runs half a slab
    // behind: [M(t-1,1) R(t,0) M(t,0) R(t,1)] - its operands for the first MFMA block are already in
    // registers when the barrier opens, so the matrix pipe has work while the early wave waits for LDS,
    // and the early wave's last MFMA block covers the late wave's reads (the two waves of a SIMD no
    // longer hit their LDS/barrier waits together).
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) issue(buf ^ 1, (t + 1) * SBK);          // in flight behind this slab's MFMAs
        if (!LATE) {
            read_frags(buf, 0); mfma_block();
            read_frags(buf, 1); mfma_block();
        } else {
            if (t > 0) mfma_block();                           // M(t-1, 1), operands read before the barrier
            read_frags(buf, 0); mfma_block();
            read_frags(buf, 1);
        }
        // the DMA of slab t+1 must have landed (this wave's part) before anyone reads it, and every
        // wave must be done reading slab t before slab t+2 overwrites its buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (LATE) mfma_block();                                    // M(nk-1, 1)
}

__global__ void __launch_bounds__(512, 2) gemm_split_kernel(SplitArgs g) {
    // XCD-aware tile order.  Blocks are dealt to the 8 XCDs round-robin by blockIdx.  The tiles of a
    // partial last row panel (here the 32 CLS rows: almost no MFMA work but a full K walk) take the
    // FIRST blockIdx values, so they overlap with the first round instead of forming a tail; the
    // full tiles follow, chunked so that every XCD (blockIdx % 8) owns a contiguous band of row
    // panels (A-panel reuse in its L2) and all XCDs get the same number of full tiles (+-1).
    const int bid = blockIdx.x;
    const int n_full = (g.M / SBM) * g.tiles_n;
    const int n_thin = g.n_tiles - n_full;
    int tile;
    if (bid < n_thin) {
        tile = n_full + bid;
    } else {
        const int xcd = bid & 7, t8 = n_thin & 7;
        auto first = [&](int x) { return n_thin + ((x - t8 + 8) & 7); };            // first full-tile block on XCD x
        auto count = [&](int x) { const int f = first(x); return f >= g.n_tiles ? 0 : (g.n_tiles - f + 7) >> 3; };
        int start = 0;
        for (int x = 0; x < xcd; ++x) start += count(x);
        tile = start + ((bid - first(xcd)) >> 3);
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
    const bool late = g.stagger && __builtin_amdgcn_readfirstlane(wid) >= 4;
    if (full) { if (late) split_mainloop<true, true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
                else      split_mainloop<true, false>(g, m0, n0, tid, wm, wn, li, lh, on, acc); }
    else      { if (late) split_mainloop<false, true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
                else      split_mainloop<false, false>(g, m0, n0, tid, wm, wn, li, lh, on, acc); }

    // epilogue: two 64x64 halves through the wave's LDS staging area (8 x 17,408 B = 139,264 B)
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
    static const int stagger = [] { const char* e = getenv("UA_SPLIT_STAGGER"); return e ? atoi(e) : 1; }();
    g.stagger = stagger;
    const size_t lds = (size_t)S_LDS;                                 // 139,264 B
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
