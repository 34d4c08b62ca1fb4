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
#include <type_traits>

namespace {

using ua::f32x16;
using ua::f16x8;
using ua::EpiArgs;

constexpr int SBM = 256, SBN = 256, SBK = 32;
constexpr int S_ROW = 128;                           // LDS bytes per slab row: 8 chunks of 16 B (4 hi | 4 lo)
constexpr int S_TILE = SBM * S_ROW;                  // 32,768 B per operand tile
constexpr int S_LDS = 8 * 64 * ua::EP_LD * 4;        // 139,264 B: max(main loop 131,072, epilogue staging)
constexpr int S_LDS_STATS = 8 * 128 * 8;             // + 8,192 B: (sum, sumsq) partials of every wave's 128 rows (row statistics requested)
constexpr int S_LDS_FOLD = 256 * 8 + 256 * 4;        // + 3,072 B: (mean, rstd) of the tile's rows and the column sums of its columns (folded LayerNorm)

struct SplitArgs {
    const unsigned char* A; const unsigned char* W;
    long long lda, ldw;                              // row pitch in columns (4 bytes per column)
    int M, N, K;
    EpiArgs ep;
    int tiles_n, n_tiles;
    int stagger;                                     // waves 4-7 run half a slab behind waves 0-3
    int block_order;                                 // 2D-blocked tile walk (L2 locality)
    int br, bc;                                      // its block: br row panels x bc column tiles
#ifdef UA_SPLIT_TRACE
    unsigned long long* trace;                       // [block][wave][6] cycle accumulators (tools/gemm_trace.py)
    unsigned long long* timeline;                    // [block][8] s_memrealtime (100 MHz): start, main loop end, block end, full?, 4 epilogue stamps (wave 0)
#endif
};

#ifdef UA_SPLIT_TRACE
#define TR_DECL unsigned long long tr_t = 0, tr_acc[6] = {0, 0, 0, 0, 0, 0}
#define TR_MARK(slot) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); \
                           tr_acc[slot] += n_ - tr_t; tr_t = n_; } while (0)
#define TR_START do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr_t = __builtin_readcyclecounter(); } while (0)
#else
#define TR_DECL
#define TR_MARK(slot)
#define TR_START
#endif

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
    TR_DECL;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TR_START;
    // Between two barriers every wave does 2 fragment reads + 2 x 24 MFMAs.  !LATE (waves 0-3):
    // [R(t,0) M(t,0) R(t,1) M(t,1)].  LATE (waves 4-7, the second wave of every SIMD) runs half a slab
    // behind: [M(t-1,1) R(t,0) M(t,0) R(t,1)] - its operands for the first MFMA block are already in
    // registers when the barrier opens, so the matrix pipe has work while the early wave waits for LDS,
    // and the early wave's last MFMA block covers the late wave's reads (the two waves of a SIMD no
    // longer hit their LDS/barrier waits together).
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) issue(buf ^ 1, (t + 1) * SBK);          // in flight behind this slab's MFMAs
        TR_MARK(0);                                            // DMA issue
        if (!LATE) {
            read_frags(buf, 0); TR_MARK(1); mfma_block(); TR_MARK(2);
            read_frags(buf, 1); TR_MARK(1); mfma_block(); TR_MARK(2);
        } else {
            if (t > 0) mfma_block();                           // M(t-1, 1), operands read before the barrier
            TR_MARK(2);
            read_frags(buf, 0); TR_MARK(1); mfma_block(); TR_MARK(2);
            read_frags(buf, 1); TR_MARK(1);
        }
        // the DMA of slab t+1 must have landed (this wave's part) before anyone reads it, and every
        // wave must be done reading slab t before slab t+2 overwrites its buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TR_MARK(3);                                            // DMA wait
        __syncthreads();
        TR_MARK(4);                                            // barrier wait
    }
#ifdef UA_SPLIT_TRACE
    if (g.trace && (tid & 63) == 0) {
        unsigned long long* o = g.trace + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 6;
        for (int i = 0; i < 5; ++i) o[i] = tr_acc[i];
        o[5] = LATE;
    }
#endif
    if (LATE) mfma_block();                                    // M(nk-1, 1)
}


// ---------------------------------------------------------------------------------------------
// 8-phase main loop (default).  Per 32-column K-tile each wave runs four phases, one 64x32
// quadrant (a, b) of its 128x64 output block each, in the order (0,0) (0,1) (1,1) (1,0):
//
//     phase p:   ds_read the fragments that change | 2 LDS-DMA pieces of ONE half-tile | [counted waits]
//                s_barrier | 12 MFMAs (2 row blocks x 2 k16-steps x 3 split products) | s_barrier
//
// The LDS holds two K-tile buffers of four half-tiles (A0 A1 B0 B1, 128 rows x 128 B = 16 KiB each;
// half a of A = the rows a*64..a*64+63 of BOTH wave rows, half b of B = the columns b*32..b*32+31 of
// all four wave columns), so a quadrant needs exactly one A half and one B half and a half-tile is
// dead - and can be restaged with the K-tile two ahead - as soon as its quadrant's fragments are
// in registers:   K-tile t, phase 0: reads B0,A0   stages A1(t+1)
//                           phase 1: reads B1      stages B0(t+2)
//                           phase 2: reads A1      stages A0(t+2)
//                           phase 3: -             stages B1(t+2), s_waitcnt vmcnt(6)
// vmcnt(6) leaves the three youngest half-tiles (6 pieces per wave) in flight and retires A1(t+1)
// and everything older, i.e. the whole of K-tile t+1, one phase (two barriers) before its first
// read.  A half-tile is restaged at least two phases after its last read; B0, restaged one phase
// after, has its reads retired by lgkmcnt(8) before phase 0's first barrier.  Waves 4-7 run one
// barrier behind waves 0-3 (one extra s_barrier up front, matched at the end by waves 0-3), so on
// every SIMD one wave is in its MFMA section while its partner reads LDS and issues DMA: the
// matrix pipe never waits for a barrier, an LDS burst or a DMA issue of its own wave.
// ---------------------------------------------------------------------------------------------
constexpr int S_HALF = 128 * S_ROW;                  // 16,384 B half-tile
constexpr int S_BUF = 4 * S_HALF;                    // 65,536 B per K-tile buffer: A0 | A1 | B0 | B1

// MF = 32: v_mfma_f32_32x32x16_f16, accumulators f32x16 acc[4][2] (32x32 blocks)
// MF = 16: v_mfma_f32_16x16x32_f16, accumulators f32x4 acc[8][4] (16x16 blocks; one MFMA covers the whole
//          32-column K-tile).  Same LDS traffic, registers and matrix-pipe cycles; the chip holds a
//          higher clock on the 16x16x32 shape under load.
// TERMS = 3: hi*hi + hi*lo + lo*hi (fp32-class).  TERMS = 1: hi*hi only - plain fp16 operands (the lo halves of the
// lines are staged but never read); MF 16 only (the MFMA similarity kernel's fp16 roofline variant).
template <bool FULL, int MF, int TERMS, typename Acc>
__device__ __forceinline__ void split_mainloop8(const SplitArgs& g, int m0, int n0, int tid, int wm, int wn,
                                                const bool (&on)[4][2], Acc& acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
    const int lane = tid & 63;
    const int r8 = tid >> 3;                                       // LDS row of this lane's chunk inside a 64-row piece
    const int cs = (tid & 7) ^ ((r8 >> 1) & 7);                    // source chunk (swizzle on the source side)
    // source = buffer resource of the tile's row panel (SGPRs) + 32-bit lane offset + scalar K offset:
    // buffer_load_dwordx4 ... offen lds.  num_records bounds the panel, so a bad offset reads zeros.
    auto panel = [](const unsigned char* base, long long rows_left, long long ld, int K) {
        const long long ext = (rows_left - 1) * ld * 4 + (long long)K * 4;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(ext > 0x7fffffffLL ? 0x7fffffffLL : ext), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t abase = panel(g.A + (size_t)m0 * g.lda * 4, min(g.M - m0, SBM), g.lda, g.K);
    const __amdgpu_buffer_rsrc_t wbase_g = panel(g.W + (size_t)n0 * g.ldw * 4, min(g.N - n0, SBN), g.ldw, g.K);
    unsigned pa[2][2];                                             // [half a][piece i = wave row]
    unsigned pw[2][2];                                             // [half b][piece i]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 128 + h * 64 + r8;                                        // LDS row i*64+r8 of A half h
            const int col = ((r8 >> 5) + 2 * i) * 64 + h * 32 + (r8 & 31);                // LDS row i*64+r8 of B half h
            pa[h][i] = (unsigned)min(row, g.M - 1 - m0) * (unsigned)(g.lda * 4) + cs * 16;
            pw[h][i] = (unsigned)min(col, g.N - 1 - n0) * (unsigned)(g.ldw * 4) + cs * 16;
        }
    const int wbase = __builtin_amdgcn_readfirstlane((tid - lane) * 16);                 // wave-uniform LDS offset of piece 0
    auto stage = [&](int off, __amdgpu_buffer_rsrc_t base, const unsigned (&src)[2], int kt) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(base, (lds_ptr_t)(slds + off + wbase), 16, src[0], kt * S_ROW, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(base, (lds_ptr_t)(slds + off + wbase + S_HALF / 2), 16, src[1], kt * S_ROW, 0, 0);
    };
    // fragment addressing.  MF 32: lane = (row li of a 32-row block, k-half lh), two k16 steps per K-tile
    // (hi chunk 2s+lh, lo chunk 4+2s+lh).  MF 16: lane = (row lane&15 of a 16-row block, k-group lane>>4),
    // one k32 step (hi chunk kg, lo chunk 4+kg).  fr[0..3] = chunk byte offsets {hi0, lo0, hi1, lo1} (MF 32)
    // or {hi, lo, -, -} (MF 16); the row-dependent swizzle is the same function of the LDS row.
    const int lrow = MF == 32 ? (lane & 31) : (lane & 15);
    const int lk = MF == 32 ? (lane >> 5) : (lane >> 4);
    const int sw = (lrow >> 1) & 7;
    const int ph0 = ((MF == 32 ? 0 + lk : lk) ^ sw) * 16, pl0 = ((4 + lk) ^ sw) * 16;
    const int ph1 = ((2 + lk) ^ sw) * 16, pl1 = ((6 + lk) ^ sw) * 16;          // MF 32 only
    const int aoff = (wm * 64 + lrow) * S_ROW, boff = 2 * S_HALF + (wn * 32 + lrow) * S_ROW;
    f16x8 ah[2][2], al[2][2];                // MF 32: [row block i2][k16 step]; MF 16: [ib2 >> 1][ib2 & 1]
    f16x8 wh[2][2], wl[2][2];                // MF 32: [half b][k16 step];      MF 16: [half b][col block jb2]
    auto readA = [&](const unsigned char* buf, int a) {
        const unsigned char* q = buf + a * S_HALF + aoff;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            if constexpr (MF == 32) {
                ah[i2][0] = *reinterpret_cast<const f16x8*>(q + i2 * 32 * S_ROW + ph0);
                al[i2][0] = *reinterpret_cast<const f16x8*>(q + i2 * 32 * S_ROW + pl0);
                ah[i2][1] = *reinterpret_cast<const f16x8*>(q + i2 * 32 * S_ROW + ph1);
                al[i2][1] = *reinterpret_cast<const f16x8*>(q + i2 * 32 * S_ROW + pl1);
            } else {
                ah[i2][0] = *reinterpret_cast<const f16x8*>(q + (2 * i2) * 16 * S_ROW + ph0);
                al[i2][0] = *reinterpret_cast<const f16x8*>(q + (2 * i2) * 16 * S_ROW + pl0);
                ah[i2][1] = *reinterpret_cast<const f16x8*>(q + (2 * i2 + 1) * 16 * S_ROW + ph0);
                al[i2][1] = *reinterpret_cast<const f16x8*>(q + (2 * i2 + 1) * 16 * S_ROW + pl0);
            }
        }
    };
    auto readB = [&](const unsigned char* buf, int b) {
        const unsigned char* q = buf + b * S_HALF + boff;
        if constexpr (MF == 32) {
            wh[b][0] = *reinterpret_cast<const f16x8*>(q + ph0); wl[b][0] = *reinterpret_cast<const f16x8*>(q + pl0);
            wh[b][1] = *reinterpret_cast<const f16x8*>(q + ph1); wl[b][1] = *reinterpret_cast<const f16x8*>(q + pl1);
        } else {
            wh[b][0] = *reinterpret_cast<const f16x8*>(q + ph0); wl[b][0] = *reinterpret_cast<const f16x8*>(q + pl0);
            wh[b][1] = *reinterpret_cast<const f16x8*>(q + 16 * S_ROW + ph0); wl[b][1] = *reinterpret_cast<const f16x8*>(q + 16 * S_ROW + pl0);
        }
    };
#define UA_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0)
#define UA_MFMA16(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, C_, 0, 0, 0)
    auto mfma_quadrant = [&](int a, int b) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MF == 32) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // the two row blocks alternate so that dependent MFMAs are one instruction apart
                if (FULL || on[2 * a][b])     UA_MFMA(ah[0][s], wh[b][s], acc[2 * a][b]);
                if (FULL || on[2 * a + 1][b]) UA_MFMA(ah[1][s], wh[b][s], acc[2 * a + 1][b]);
                if (FULL || on[2 * a][b])     UA_MFMA(ah[0][s], wl[b][s], acc[2 * a][b]);
                if (FULL || on[2 * a + 1][b]) UA_MFMA(ah[1][s], wl[b][s], acc[2 * a + 1][b]);
                if (FULL || on[2 * a][b])     UA_MFMA(al[0][s], wh[b][s], acc[2 * a][b]);
                if (FULL || on[2 * a + 1][b]) UA_MFMA(al[1][s], wh[b][s], acc[2 * a + 1][b]);
            }
        } else {
            // 4 row blocks x 2 column blocks of 16; the three split products of one block are 8 MFMAs apart
#pragma unroll
            for (int term = 0; term < TERMS; ++term)
#pragma unroll
                for (int ib2 = 0; ib2 < 4; ++ib2)
#pragma unroll
                    for (int jb2 = 0; jb2 < 2; ++jb2)
                        if (FULL || on[2 * a + (ib2 >> 1)][b]) {
                            const f16x8& av = term == 2 ? al[ib2 >> 1][ib2 & 1] : ah[ib2 >> 1][ib2 & 1];
                            const f16x8& bv = term == 1 ? wl[b][jb2] : wh[b][jb2];
                            UA_MFMA16(av, bv, acc[4 * a + ib2][2 * b + jb2]);
                        }
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define UA_SB() __builtin_amdgcn_sched_barrier(0)
#define UA_BAR() do { UA_SB(); __builtin_amdgcn_s_barrier(); UA_SB(); } while (0)
    const int nk = g.K / SBK;
    // prologue: K-tile 0 complete, K-tile 1 minus A1 in flight (the steady-state staging order)
    stage(2 * S_HALF, wbase_g, pw[0], 0); stage(0, abase, pa[0], 0); stage(3 * S_HALF, wbase_g, pw[1], 0); stage(S_HALF, abase, pa[1], 0);
    if (nk > 1) {
        stage(S_BUF + 2 * S_HALF, wbase_g, pw[0], 1); stage(S_BUF, abase, pa[0], 1); stage(S_BUF + 3 * S_HALF, wbase_g, pw[1], 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    UA_BAR();
    if (wm == 1) UA_BAR();                                         // waves 4-7 trail by one barrier
    for (int t = 0; t < nk; ++t) {
        const int d = t & 1;
        const unsigned char* buf = slds + d * S_BUF;
        const bool s1 = t + 1 < nk, s2 = t + 2 < nk;
        // ---- phase 0: quadrant (0,0)
        readB(buf, 0); UA_SB(); readA(buf, 0); UA_SB();
        if (s1) stage((d ^ 1) * S_BUF + S_HALF, abase, pa[1], t + 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");         // B0 reads done: it is restaged next phase
        UA_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        UA_BAR();
        // ---- phase 1: quadrant (0,1)
        readB(buf, 1); UA_SB();
        if (s2) stage(d * S_BUF + 2 * S_HALF, wbase_g, pw[0], t + 2);
        UA_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        UA_BAR();
        // ---- phase 2: quadrant (1,1)
        readA(buf, 1); UA_SB();
        if (s2) stage(d * S_BUF, abase, pa[0], t + 2);
        UA_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        UA_BAR();
        // ---- phase 3: quadrant (1,0)
        if (s2) {
            stage(d * S_BUF + 3 * S_HALF, wbase_g, pw[1], t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // K-tile t+1 has landed (this wave's pieces)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        UA_BAR();
        mfma_quadrant(1, 0);
        UA_BAR();
    }
    if (wm == 0) UA_BAR();                                         // re-join the two wave groups
#undef UA_SB
#undef UA_BAR
#undef UA_MFMA
#undef UA_MFMA16
}

template <int SCHED, int TERMS>
__device__ __forceinline__ void split_tile(const SplitArgs& g, const int bid, const int zero) {
    // XCD-aware tile order.  Blocks are dealt to the 8 XCDs round-robin by blockIdx.  The tiles of a
    // partial last row panel (here the 32 CLS rows: almost no MFMA work but a full K walk) take the
    // FIRST blockIdx values, so they overlap with the first round instead of forming a tail; the
    // full tiles follow, chunked so that every XCD (blockIdx % 8) owns a contiguous band of row
    // panels (A-panel reuse in its L2) and all XCDs get the same number of full tiles (+-1).
#ifdef UA_SPLIT_TRACE
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ct0 = __builtin_amdgcn_s_memtime();
#endif
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
    int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    if (g.block_order && g.tiles_n >= 8 && tile < n_full) {          // (narrow outputs are already compact; measured: no gain below 8 column tiles)
        // 2D-blocked walk of the full tiles: the ~32 workgroups an XCD runs at a time cover 4 row panels x 8 column
        // tiles (12 operand panels per K-tile step in its L2) instead of 1-3 row panels x all column tiles (up to 24)
        const int BR = g.br, BC = g.bc;
        const int rows_full = g.M / SBM;
        const int rb = tile / (BR * g.tiles_n);
        const int h = min(BR, rows_full - rb * BR);                         // row panels in this row block
        const int o1 = tile - rb * BR * g.tiles_n;
        const int ncb = (g.tiles_n + BC - 1) / BC;
        const int cb = min(o1 / (h * BC), ncb - 1);
        const int o2 = o1 - cb * h * BC;
        const int bc = min(BC, g.tiles_n - cb * BC);                        // column tiles in this column block
        tm = rb * BR + o2 / bc;
        tn = cb * BC + o2 % bc;
    }
    const int m0 = tm * SBM, n0 = tn * SBN;
    // (`zero` is 0: the literal in the one-tile kernel; in the persistent kernel an opaque per-iteration SGPR, so that nothing derived
    //  from the lane id is hoisted out of the tile loop and kept in registers across the whole body)
    const int tid = threadIdx.x + zero, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int li = lane & 31, lh = lane >> 5;

    // SCHED 16: 8-phase loop on the 16x16x32 MFMA (acc 8x4 blocks of 16x16); SCHED 8 / 2: 32x32x16 MFMA
    constexpr bool B16 = SCHED == 16;
    typedef typename std::conditional<B16, ua::f32x4[8][4], f32x16[4][2]>::type AccT;
    AccT acc;
    if constexpr (B16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = ua::f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    const bool full = (m0 + SBM <= g.M) && (n0 + SBN <= g.N);
    if (B16 && (g.ep.ln_mr || g.ep.ln_in)) {
        // folded LayerNorm: the tile's per-row (mean, rstd) - or the producer's (sum, sumsq) partials they are made of - and
        // per-column sums go straight to LDS beyond the main loop's buffers (buffer_load ... lds, like the operand tiles):
        // no VGPRs, no wait here - the main loop's first vmcnt wait + barrier covers them.  Rows / columns past the matrix
        // read as zeros (num_records).
        extern __shared__ __attribute__((aligned(16))) unsigned char slds0[];
        const int wb = __builtin_amdgcn_readfirstlane(wid) * 256;                  // this wave's 64 dwords
        const int rows_b = min(g.M - m0, SBM) * 8;
        if (g.ep.ln_mr) {
            const __amdgpu_buffer_rsrc_t r_mr = __builtin_amdgcn_make_buffer_rsrc((void*)(g.ep.ln_mr + m0), 0, rows_b, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_mr, (lds_ptr_t)(slds0 + S_LDS + S_LDS_STATS + wb), 4, wb + lane * 4, 0, 0, 0);
        } else {
            // (the partials wait in the region this tile's own output statistics use later, after they have been consumed)
            for (int p = 0; p < g.ep.ln_parts; ++p) {
                const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void*)(g.ep.ln_in + (size_t)p * g.ep.ln_in_ld + m0), 0, rows_b, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (lds_ptr_t)(slds0 + S_LDS + p * 2048 + wb), 4, wb + lane * 4, 0, 0, 0);
            }
        }
        if (wid < 4) {
            const __amdgpu_buffer_rsrc_t r_lc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.ep.ln_c + n0), 0, min(g.N - n0, SBN) * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_lc, (lds_ptr_t)(slds0 + S_LDS + S_LDS_STATS + 256 * 8 + wb), 4, wb + lane * 4, 0, 0, 0);
        }
    }
    const int swm = __builtin_amdgcn_readfirstlane(wm), swn = __builtin_amdgcn_readfirstlane(wn);
    bool on[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            on[i][j] = ((m0 + swm * 128 + i * 32) < g.M) && ((n0 + swn * 64 + j * 32) < g.N);
    if constexpr (SCHED == 16) {
        if (full) split_mainloop8<true, 16, TERMS>(g, m0, n0, tid, swm, swn, on, acc);
        else      split_mainloop8<false, 16, TERMS>(g, m0, n0, tid, swm, swn, on, acc);
    } else if constexpr (SCHED == 8) {
        if (full) split_mainloop8<true, 32, 3>(g, m0, n0, tid, swm, swn, on, acc);
        else      split_mainloop8<false, 32, 3>(g, m0, n0, tid, swm, swn, on, acc);
    } else {
        const bool late = g.stagger && __builtin_amdgcn_readfirstlane(wid) >= 4;
        if (full) { if (late) split_mainloop<true, true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
                    else      split_mainloop<true, false>(g, m0, n0, tid, wm, wn, li, lh, on, acc); }
        else      { if (late) split_mainloop<false, true>(g, m0, n0, tid, wm, wn, li, lh, on, acc);
                    else      split_mainloop<false, false>(g, m0, n0, tid, wm, wn, li, lh, on, acc); }
    }

#ifdef UA_SPLIT_TRACE
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ct1 = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: two 64x64 halves through the wave's LDS staging area (8 x 17,408 B = 139,264 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
    float* stage = reinterpret_cast<float*>(slds) + wid * 64 * ua::EP_LD;
    float2* part = g.ep.stats ? reinterpret_cast<float2*>(slds + S_LDS) + wid * 128 : nullptr;     // block-uniform
    const bool fold = B16 && (g.ep.ln_mr != nullptr || g.ep.ln_in != nullptr);                     // block-uniform
    if (B16 && g.ep.ln_in) {
        // (mean, rstd) of the tile's rows from the producer's partials: ua_row_stats_finalize's arithmetic, tiles in index order
        if (tid < SBM) {
            const float2* pin = reinterpret_cast<const float2*>(slds + S_LDS) + tid;
            double sm = 0.0, sq = 0.0;
            for (int p = 0; p < g.ep.ln_parts; ++p) { const float2 v = pin[p * 256]; sm += (double)v.x; sq += (double)v.y; }
            const double mean = sm * g.ep.ln_inv_n;
            const double var = fmax(sq * g.ep.ln_inv_n - mean * mean, 0.0);
            reinterpret_cast<float2*>(slds + S_LDS + S_LDS_STATS)[tid] = make_float2((float)mean, (float)(1.0 / sqrt(var + g.ep.ln_eps)));
        }
        __syncthreads();
    }
    const float2* s_mr = reinterpret_cast<const float2*>(slds + S_LDS + S_LDS_STATS) + wm * 128;
    const float* s_lc = reinterpret_cast<const float*>(slds + S_LDS + S_LDS_STATS + 256 * 8) + wn * 64;
#ifdef UA_SPLIT_TRACE
    unsigned long long es[4] = {0, 0, 0, 0};
#define UA_ES(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); es[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define UA_ES(i)
#endif
    // (two explicit halves: the accumulator blocks must be indexed statically)
    if (m0 + wm * 128 < g.M) {                                 // wave-uniform
        if constexpr (B16) {
            if (fold) ua::stage_64x64_b16_fold(stage, acc, 0, lane, g.ep.acc_scale, s_mr, s_lc);
            else ua::stage_64x64_b16(stage, acc, 0, lane);
        } else ua::stage_64x64(stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], li, lh);
        UA_ES(0);
#if defined(UA_SPLIT_TRACE) && defined(UA_EPI_STORE_ONLY)
        {   // ablation (tools/gemm_trace.py --store-only): the 16 row stores of the sub-tile without LDS reads or arithmetic
            const int lr = lane >> 4, c4 = (lane & 15) * 4;
            float* cp = g.ep.C + (size_t)(m0 + wm * 128 + lr) * g.ep.ldc + n0 + wn * 64 + c4;
            const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
#pragma unroll
            for (int p = 0; p < 16; ++p) *reinterpret_cast<float4*>(cp + (size_t)p * 4 * g.ep.ldc) = v;
        }
#else
        ua::epilogue_64x64(g.ep, stage, lane, m0 + wm * 128, n0 + wn * 64, part, fold);
#endif
        UA_ES(1);
    }
    if (m0 + wm * 128 + 64 < g.M) {
        if constexpr (B16) {
            if (fold) ua::stage_64x64_b16_fold(stage, acc, 4, lane, g.ep.acc_scale, s_mr + 64, s_lc);
            else ua::stage_64x64_b16(stage, acc, 4, lane);
        } else ua::stage_64x64(stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], li, lh);
        UA_ES(2);
        ua::epilogue_64x64(g.ep, stage, lane, m0 + wm * 128 + 64, n0 + wn * 64, part ? part + 64 : nullptr, fold);
        UA_ES(3);
    }
#undef UA_ES
    if (g.ep.stats) {
        // row statistics of this tile: the four column waves' partials are added in wave order (deterministic)
        __syncthreads();
        if (tid < SBM && m0 + tid < g.M) {
            const float2* pp = reinterpret_cast<const float2*>(slds + S_LDS) + (tid >> 7) * 4 * 128 + (tid & 127);
            const float2 a = pp[0], b = pp[128], c = pp[256], d = pp[384];
            g.ep.stats[(size_t)tn * g.ep.ld_stats + m0 + tid] = make_float2((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y));
        }
    }
#ifdef UA_SPLIT_TRACE
    if (g.timeline) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long rt_st = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = g.timeline + (size_t)bid * 10;
            o[0] = rt0; o[1] = rt1; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = full ? 1 : 0;
            o[4] = es[0]; o[5] = es[1]; o[6] = es[2]; o[7] = es[3]; o[8] = rt_st; o[9] = ct1 - ct0;
        }
    }
#endif
}

template <int SCHED, int TERMS = 3>
__global__ void __launch_bounds__(512, 2) gemm_split_kernel(SplitArgs g) {
    split_tile<SCHED, TERMS>(g, blockIdx.x, 0);
}

// Persistent form: gridDim.x (a multiple of 8, at most one workgroup per CU) resident workgroups walk the tile list in the order
// the hardware dispatcher would have dealt it (virtual block id = blockIdx + i * gridDim keeps blockIdx % 8, i.e. the XCD band of
// split_tile's tile order), so a CU goes from one tile's last store to the next tile's first DMA without waiting for a new
// workgroup to be dispatched.
template <int SCHED, int TERMS = 3>
__global__ void __launch_bounds__(512, 2) gemm_split_persistent_kernel(SplitArgs g) {
    for (int vb = blockIdx.x; vb < g.n_tiles; vb += gridDim.x) {
        int zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        split_tile<SCHED, TERMS>(g, vb, zero);
        __syncthreads();                       // every wave is done with the staging area before the next tile's DMA overwrites it
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

static int g_sched = -1;
static int g_block_order = -1;
static int g_persist = -1;
extern "C" int ua_gemm_split_set_persistent(int32_t mode) { g_persist = mode; return UA_OK; }
#ifdef UA_SPLIT_TRACE
extern "C" void ua_gemm_split_set_schedule(int v) { g_sched = v; }
extern "C" void ua_gemm_split_set_tile_order(int v) { g_block_order = v; }
static unsigned long long* g_trace = nullptr;
static unsigned long long* g_timeline = nullptr;
extern "C" void ua_gemm_split_set_trace(unsigned long long* p, unsigned long long* tl) { g_trace = p; g_timeline = tl; }
#endif

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
    if (g_sched < 0) { const char* e = getenv("UA_SPLIT_SCHEDULE"); g_sched = e ? atoi(e) : 16; }              // 16 (default) | 8 | 2: A/B aid
    const int sched = g_sched;
    if (g_block_order < 0) { const char* e = getenv("UA_SPLIT_TILE_ORDER"); g_block_order = e ? atoi(e) : 1; }
    g.block_order = g_block_order;
    static const int env_br = [] { const char* e = getenv("UA_SPLIT_BR"); return e ? atoi(e) : 4; }();
    static const int env_bc = [] { const char* e = getenv("UA_SPLIT_BC"); return e ? atoi(e) : 8; }();
    g.br = env_br; g.bc = env_bc;
    g.stagger = stagger;
#ifdef UA_SPLIT_TRACE
    g.trace = g_trace; g.timeline = g_timeline;
#endif
    if ((g.ep.ln_mr || g.ep.ln_in) && sched != 16) return UA_EUNSUPPORTED;            // (the A/B schedules have no folded-LayerNorm staging)
    const size_t lds = (size_t)S_LDS + (g.ep.stats || g.ep.ln_mr || g.ep.ln_in ? S_LDS_STATS + S_LDS_FOLD : 0);   // 139,264 B (+ 11,264)
    static bool attr_set = false;
    if (!attr_set) {
        const int lds_max = S_LDS + S_LDS_STATS + S_LDS_FOLD;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) return UA_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) return UA_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) return UA_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_persistent_kernel<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) return UA_ELAUNCH;
        attr_set = true;
    }
    // UA_SPLIT_PERSIST=1 / ua_gemm_split_set_persistent(1): resident workgroups walk the tile list (launches of more than one round only)
    if (g_persist < 0) { const char* e = getenv("UA_SPLIT_PERSIST"); g_persist = e ? atoi(e) : 0; }
    const int persist = g_persist;
    static const int n_cu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                                 return n >= 8 ? n / 8 * 8 : 8; }();
    // mode 2 (default): only where it measured faster - launches of >= 8 rounds of tiles (fc1 of a >= 62-cloud forward -3 %; 1-6 rounds 0..+3 %)
    if (sched == 16 && persist && g.n_tiles > n_cu && (persist == 1 || g.n_tiles >= 8 * n_cu)) {
        hipLaunchKernelGGL(gemm_split_persistent_kernel<16>, dim3(n_cu), dim3(512), lds, ua::as_stream(stream), g);
        UA_LAUNCH_CHECK();
        return UA_OK;
    }
    if (sched == 16) hipLaunchKernelGGL(gemm_split_kernel<16>, dim3(g.n_tiles), dim3(512), lds, ua::as_stream(stream), g);
    else if (sched == 8) hipLaunchKernelGGL(gemm_split_kernel<8>, dim3(g.n_tiles), dim3(512), lds, ua::as_stream(stream), g);
    else            hipLaunchKernelGGL(gemm_split_kernel<2>, dim3(g.n_tiles), dim3(512), lds, ua::as_stream(stream), g);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

// Row statistics of the folded LayerNorm: (sum, sumsq) partials per column tile -> (mean, rstd) per row
__global__ void __launch_bounds__(256)
row_stats_finalize_kernel(const float2* __restrict__ partials, int parts, long long ld, int rows, double inv_n, double eps,
                          float2* __restrict__ mean_rstd) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0, q = 0.0;
    for (int t = 0; t < parts; ++t) { const float2 v = partials[(size_t)t * ld + r]; s += (double)v.x; q += (double)v.y; }
    const double mean = s * inv_n;
    const double var = fmax(q * inv_n - mean * mean, 0.0);           // biased variance (F.layer_norm)
    mean_rstd[r] = make_float2((float)mean, (float)(1.0 / sqrt(var + eps)));
}

extern "C" int ua_row_stats_finalize(const float* partials, int32_t parts, int64_t ld_stats, int32_t rows, int32_t n, float eps,
                                     float* mean_rstd, void* stream) {
    UA_CHECK_PTR(partials); UA_CHECK_PTR(mean_rstd);
    if (parts <= 0 || rows <= 0 || n <= 0 || ld_stats < rows || !(eps >= 0.f)) return UA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(partials) & 7) || (reinterpret_cast<uintptr_t>(mean_rstd) & 7)) return UA_EALIGN;
    hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, ua::as_stream(stream),
                       reinterpret_cast<const float2*>(partials), parts, (long long)ld_stats, rows, 1.0 / n, (double)eps,
                       reinterpret_cast<float2*>(mean_rstd));
    UA_LAUNCH_CHECK();
    return UA_OK;
}

// =============================================================================================
// MFMA (GEMM-form) similarity kernel: DOTA_mix.predict / _log_likelihood for a BATCH of queries
// (dota_mixture.py:121-156,236-267 with B queries; Uni_Adapter.py:219-244 with the B = K class
// embeddings of the alignment matrix).  SURVEY section 2.1 N14 / section 8(d) "batched similarity":
//
//   sum_d (x_d - mu_d)^2 / v_d  =  sum_d x_d^2 (1/v_d)  +  sum_d x_d (-2 mu_d / v_d)  +  sum_d mu_d^2 / v_d
//
//   joint[b, n] = rc[n] - 1/2 * ( [x_b^2 | x_b] . [1/v_n | -2 mu_n / v_n] ),   n = k*M + m,
//   rc[n]       = ln(pi_n + 1e-10) - 1/2 sum_d ln v_nd - 1/2 sum_d mu_nd^2 / v_nd,
//   scores[b,k] = logsumexp_m joint[b, k*M + m]
//
// i.e. a (B x 2D) . (2D x KM) product on the matrix cores, with both operands in the H32L32 split
// format (fp32-class: hi*hi + hi*lo + lo*hi) or - `terms` = 1 - as plain fp16 (hi only), the row
// constant and the -1/2 as the GEMM epilogue's per-column shift / scale, and the logsumexp over the M
// modes fused into the epilogue (M = 4, 8, 16; otherwise a second small launch).  Every operand ROW
// carries its own power-of-two scale (1/v reaches 1e4 and more, x^2 goes down to 1e-6), folded into
// the epilogue's per-row / per-column factors, so fp16 range is never an issue.
// The state-side operand is built by gmm_operand_kernel (one pass over mu, var: 8 KMD bytes read,
// 8 KMD written) and can be reused for any number of query batches while the state is unchanged.
// =============================================================================================
namespace {

using ua::pow2_scale; using ua::sim_col; using ua::store_split4;

// one wave per state row n = k*M + m; block = 4 waves
template <int VEC>
__global__ void __launch_bounds__(256)
gmm_operand_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ pi, int KM, float eps,
                   unsigned char* __restrict__ Wop, float* __restrict__ rc, float* __restrict__ colscale) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= KM) return;                                                     // wave-uniform
    ua::gmm_operand_row<VEC>(mu, var, pi, n, eps, lane, Wop, rc, colscale);
}

// one wave per query row b: A row = [x^2 | x] (split), row_scale[b] = 1/s_b
template <int VEC>
__global__ void __launch_bounds__(256)
gmm_query_operand_kernel(const float* __restrict__ x, int B, unsigned char* __restrict__ Aop, float* __restrict__ row_scale) {
    constexpr int D = 256 * VEC;
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float4* px = reinterpret_cast<const float4*>(x + (size_t)b * D);
    float4 xv[VEC];
    float amax = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        xv[v] = px[v * 64 + lane];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(xv[v].x), fabsf(xv[v].y)), fmaxf(fabsf(xv[v].z), fabsf(xv[v].w))));
    }
    amax = ua::wave_max(amax);
    const float s = pow2_scale(fmaxf(amax, amax * amax));
    unsigned char* row = Aop + (size_t)b * (2 * D) * 4;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int c = (v * 64 + lane) * 4;
        store_split4(row, sim_col(0, c), make_float4(xv[v].x * xv[v].x, xv[v].y * xv[v].y, xv[v].z * xv[v].z, xv[v].w * xv[v].w), s);
        store_split4(row, sim_col(1, c), xv[v], s);
    }
    if (lane == 0) row_scale[b] = 1.0f / s;
}

// logsumexp over the M modes for mode counts the fused epilogue does not cover: joint (B, K*M) -> scores (B, K)
__global__ void __launch_bounds__(256)
lse_modes_kernel(const float* __restrict__ joint, int BK, int M, float* __restrict__ scores) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BK) return;
    const float* j = joint + (size_t)i * M;
    float mx = j[0];
    for (int m = 1; m < M; ++m) mx = fmaxf(mx, j[m]);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += expf(j[m] - mx);
    scores[i] = mx + logf(s);
}

// ---------------------------------------------------------------------------------------------
// Small-problem form of the MFMA similarity (the B = K alignment queries of residual learning, DOTA_mix.predict on a
// small batch): the 256x256-tile GEMM above would run such a product on a handful of workgroups with 64 dependent
// K-steps each.  Here one workgroup owns a 16-query x 16-state-row block of the joint matrix and its 8 waves split
// the 2D-long contraction (wave w takes K-steps w*nk/8 ..): every wave issues all its operand loads (straight from the
// H32L32 rows into MFMA fragments - lane = (row l&15, k-group l>>4), one 16-byte chunk per operand half and K-step, no
// LDS staging), runs nk/8 x TERMS v_mfma_f32_16x16x32_f16, and the 8 partial blocks are summed in fp64 through LDS in a
// fixed order.  Each partial accumulates only 1/8 of the contraction (magnitude ~6e2 instead of ~5e3) over nk/8 x 3
// roundings and the cross-wave sum is exact to fp64, so the result is within ~3e-4 of fp64 on raw scores of ~3e3 -
// the accuracy of the fp64-summing VALU kernel, at the arithmetic cost of 24 MFMAs per wave.
// grid (ceil(KM/16), ceil(B/16)), block 512.
// ---------------------------------------------------------------------------------------------
template <int TERMS, int NKW>      // NKW = K-steps per wave = (2D/32)/8
__global__ void __launch_bounds__(512)
gmm_loglik_mfma_small_kernel(const unsigned char* __restrict__ Aop, const unsigned char* __restrict__ Wop,
                             const float* __restrict__ rc, const float* __restrict__ colscale, const float* __restrict__ row_scale,
                             int B, int KM, int M, int lse_fused, float* __restrict__ joint, float* __restrict__ scores) {
    __shared__ double s_part[8][256];
    __shared__ float s_j[16][17];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int lr = lane & 15, kg = lane >> 4;
    const size_t pitch = (size_t)NKW * 8 * 128;                                     // bytes per operand row (2D columns x 4 B)
    const unsigned char* pa = Aop + (size_t)min(b0 + lr, B - 1) * pitch + (size_t)w * NKW * 128 + kg * 16;
    const unsigned char* pw = Wop + (size_t)min(n0 + lr, KM - 1) * pitch + (size_t)w * NKW * 128 + kg * 16;
    f16x8 ah[NKW], wh[NKW], al[NKW], wl[NKW];
#pragma unroll
    for (int t = 0; t < NKW; ++t) {
        ah[t] = *reinterpret_cast<const f16x8*>(pa + t * 128);
        wh[t] = *reinterpret_cast<const f16x8*>(pw + t * 128);
        if (TERMS == 3) {
            al[t] = *reinterpret_cast<const f16x8*>(pa + t * 128 + 64);
            wl[t] = *reinterpret_cast<const f16x8*>(pw + t * 128 + 64);
        }
    }
    ua::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKW; ++t) {
        // W rows as the MFMA's A operand (output rows = state rows), queries as its B operand (output columns = queries):
        // acc[j] = block[state row 4*kg + j][query lr]
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], ah[t], acc, 0, 0, 0);
        if (TERMS == 3) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], ah[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], al[t], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_part[w][lane * 4 + j] = (double)acc[j];
    __syncthreads();
    if (tid < 256) {
        // thread = (query q = tid >> 4, state row c = tid & 15): element of lane (c >> 2) * 16 + q, j = c & 3
        const int q = tid >> 4, c = tid & 15;
        const int idx = ((c >> 2) * 16 + q) * 4 + (c & 3);
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += s_part[i][idx];
        const int b = b0 + q, n = n0 + c;
        float jv = 0.f;
        if (b < B && n < KM) {
            jv = (float)((double)rc[n] + ((double)colscale[n] * (double)row_scale[b]) * v);
            if (joint) joint[(size_t)b * KM + n] = jv;
        }
        s_j[q][c] = jv;
    }
    if (!lse_fused || !scores) return;                                              // workgroup-uniform
    __syncthreads();
    if (tid < 256) {
        const int q = tid >> 4, c = tid & 15;
        const int b = b0 + q, n = n0 + c;
        if (c % M == 0 && b < B && n < KM) {                                        // M divides 16: the group lies inside the block
            float mx = s_j[q][c];
            for (int m = 1; m < M; ++m) mx = fmaxf(mx, s_j[q][c + m]);
            float sm = 0.f;
            for (int m = 0; m < M; ++m) sm += expf(s_j[q][c + m] - mx);
            scores[(size_t)b * (KM / M) + n / M] = mx + logf(sm);                   // torch.logsumexp
        }
    }
}

using ua::MfmaWs; using ua::mfma_ws_layout;

inline bool lse_fusable(int M) { return M == 4 || M == 8 || M == 16; }     // both paths fuse these (the small path also M = 1, 2)

template <typename F>
int dispatch_vec_op(int D, F&& f) {
    switch (D / 256) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 5: return f(std::integral_constant<int, 5>{});
        case 6: return f(std::integral_constant<int, 6>{});
        case 7: return f(std::integral_constant<int, 7>{});
        case 8: return f(std::integral_constant<int, 8>{});
        default: return UA_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" size_t ua_gmm_loglik_mfma_workspace_bytes(int32_t B, int32_t K, int32_t M, int32_t D) {
    if (B <= 0 || K <= 0 || M <= 0 || D <= 0) return 0;
    return mfma_ws_layout(nullptr, B, K, M, D, !lse_fusable(M)).total;
}

extern "C" int ua_gmm_mfma_operand(const float* mu, const float* var, const float* pi, int32_t K, int32_t M, int32_t D, float eps,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    UA_CHECK_PTR(mu); UA_CHECK_PTR(var); UA_CHECK_PTR(pi); UA_CHECK_PTR(workspace);
    UA_CHECK_ALIGN16(mu); UA_CHECK_ALIGN16(var);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return UA_EALIGN;
    if (K <= 0 || M <= 0 || D <= 0) return UA_EINVAL;
    if (D % 256 != 0 || D > 2048) return UA_EUNSUPPORTED;
    const MfmaWs w = mfma_ws_layout(reinterpret_cast<unsigned char*>(workspace), 1, K, M, D, false);
    if (workspace_bytes < w.total) return UA_EWORKSPACE;
    const int KM = K * M;
    return dispatch_vec_op(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        hipLaunchKernelGGL(gmm_operand_kernel<VEC>, dim3((KM + 3) / 4), dim3(256), 0, ua::as_stream(stream), mu, var, pi, KM, eps,
                           w.Wop, w.rc, w.colscale);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
}

extern "C" int ua_gmm_loglik_mfma(const float* x, int32_t B, int32_t K, int32_t M, int32_t D, int32_t terms,
                                  float* scores, float* joint, void* workspace, size_t workspace_bytes, void* stream) {
    UA_CHECK_PTR(x); UA_CHECK_PTR(workspace); UA_CHECK_ALIGN16(x);
    if (!scores && !joint) return UA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return UA_EALIGN;
    if (B <= 0 || K <= 0 || M <= 0 || D <= 0 || (terms != 1 && terms != 3)) return UA_EINVAL;
    if (D % 256 != 0 || D > 2048) return UA_EUNSUPPORTED;
    const bool fused = lse_fusable(M);
    const MfmaWs w = mfma_ws_layout(reinterpret_cast<unsigned char*>(workspace), B, K, M, D, !fused);
    if (workspace_bytes < w.total) return UA_EWORKSPACE;
    if (joint && (reinterpret_cast<uintptr_t>(joint) & 15u) != 0) return UA_EALIGN;
    hipStream_t st = ua::as_stream(stream);
    const int KM = K * M, K2 = 2 * D;
    int rc = dispatch_vec_op(D, [&](auto vec) {
        constexpr int VEC = decltype(vec)::value;
        hipLaunchKernelGGL(gmm_query_operand_kernel<VEC>, dim3((B + 3) / 4), dim3(256), 0, st, x, B, w.Aop, w.row_scale);
        UA_LAUNCH_CHECK();
        return UA_OK;
    });
    if (rc) return rc;
    // small problems: 16 x 16 blocks with the contraction split over the 8 waves of a workgroup (accurate and parallel);
    // large ones (the BASELINE configs[4] roofline shapes): the 256 x 256 tile GEMM
    static const int force_path = [] { const char* e = getenv("UA_MFMA_SIM_PATH"); return e ? atoi(e) : 0; }();   // 1 small, 2 tiles (A/B aid)
    const bool small = (KM % 4 != 0) || (force_path ? force_path == 1 : ((long long)B * KM <= (1ll << 19) || B <= 32));   // (the tile path stores float4 rows)
    if (small) {
        const bool sfused = scores && (16 % M == 0);
        float* jsmall = joint ? joint : (sfused ? nullptr : w.joint ? w.joint : nullptr);
        if (!sfused && scores && !jsmall) return UA_EWORKSPACE;
        const dim3 grid((KM + 15) / 16, (B + 15) / 16);
        const int nkw = K2 / 32 / 8;
#define UA_SMALL(T, N) hipLaunchKernelGGL((gmm_loglik_mfma_small_kernel<T, N>), grid, dim3(512), 0, st, w.Aop, w.Wop, w.rc, w.colscale, \
                                          w.row_scale, B, KM, M, sfused ? 1 : 0, jsmall, scores)
        if (terms == 3) {
            switch (nkw) { case 2: UA_SMALL(3, 2); break; case 4: UA_SMALL(3, 4); break; case 6: UA_SMALL(3, 6); break; case 8: UA_SMALL(3, 8); break;
                           case 10: UA_SMALL(3, 10); break; case 12: UA_SMALL(3, 12); break; case 14: UA_SMALL(3, 14); break; case 16: UA_SMALL(3, 16); break;
                           default: return UA_EUNSUPPORTED; }
        } else {
            switch (nkw) { case 2: UA_SMALL(1, 2); break; case 4: UA_SMALL(1, 4); break; case 6: UA_SMALL(1, 6); break; case 8: UA_SMALL(1, 8); break;
                           case 10: UA_SMALL(1, 10); break; case 12: UA_SMALL(1, 12); break; case 14: UA_SMALL(1, 14); break; case 16: UA_SMALL(1, 16); break;
                           default: return UA_EUNSUPPORTED; }
        }
#undef UA_SMALL
        UA_LAUNCH_CHECK();
        if (!sfused && scores) {
            hipLaunchKernelGGL(lse_modes_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, jsmall, B * K, M, scores);
            UA_LAUNCH_CHECK();
        }
        return UA_OK;
    }
    SplitArgs g{};
    g.A = w.Aop; g.W = w.Wop; g.lda = K2; g.ldw = K2; g.M = B; g.N = KM; g.K = K2;
    g.ep = EpiArgs{};
    g.ep.M = B; g.ep.N = KM; g.ep.acc_scale = 1.0f;
    g.ep.scale = w.colscale; g.ep.shift = w.rc; g.ep.row_scale = w.row_scale;
    float* jout = joint ? joint : (fused ? nullptr : w.joint);
    g.ep.C = jout; g.ep.ldc = KM;
    if (fused && scores) { g.ep.act = 4; g.ep.lse = scores; g.ep.ld_lse = K; g.ep.lse_group = M; }
    const int tiles_m = (B + SBM - 1) / SBM, tiles_n = (KM + SBN - 1) / SBN;
    g.tiles_n = tiles_n; g.n_tiles = tiles_m * tiles_n;
    g.block_order = 1; g.stagger = 1; g.br = 4; g.bc = 8;
    const size_t lds = (size_t)S_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<16, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH;
        attr_set = true;
    }
    if (terms == 3) hipLaunchKernelGGL((gemm_split_kernel<16, 3>), dim3(g.n_tiles), dim3(512), lds, st, g);
    else            hipLaunchKernelGGL((gemm_split_kernel<16, 1>), dim3(g.n_tiles), dim3(512), lds, st, g);
    UA_LAUNCH_CHECK();
    if (!fused && scores) {
        hipLaunchKernelGGL(lse_modes_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, jout, B * K, M, scores);
        UA_LAUNCH_CHECK();
    }
    return UA_OK;
}
