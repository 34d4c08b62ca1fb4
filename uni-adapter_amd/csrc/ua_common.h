// Shared helpers for the gfx950 kernels of libuniadapter_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uniadapter.h"

#define UA_CHECK_PTR(p)      do { if ((p) == nullptr) return UA_EINVAL; } while (0)
#define UA_CHECK_ALIGN16(p)  do { if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return UA_EALIGN; } while (0)
#define UA_LAUNCH_CHECK()    do { if (hipGetLastError() != hipSuccess) return UA_ELAUNCH; } while (0)

namespace ua {

constexpr int kWave = 64;

// Wavefront (64-lane) reductions on DPP: a `__shfl_xor` butterfly is six ds_bpermute round trips
// (twelve for a double) of ~100 cycles each in a dependent chain; the DPP forms ride on the VALU
// operand path.  Steps: quad xor 1, quad xor 2, row_half_mirror, row_mirror (after which every lane
// of a 16-lane row holds the row total), row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3;
// lane 63 then holds the total, which is broadcast through an SGPR.  All 64 lanes must be active.
// The association ((..pairs..quads..rows) is fixed, so results are reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    return __hiloint2double(dpp_i<CTRL, ROW_MASK>(__double2hiint(v)), dpp_i<CTRL, ROW_MASK>(__double2loint(v)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<0xB1, 0xF>(v);          // quad_perm [1,0,3,2]
    v += dpp_d<0x4E, 0xF>(v);          // quad_perm [2,3,0,1]
    v += dpp_d<0x141, 0xF>(v);         // row_half_mirror
    v += dpp_d<0x140, 0xF>(v);         // row_mirror
    v += dpp_d<0x142, 0xA>(v);         // row_bcast15 -> rows 1, 3
    v += dpp_d<0x143, 0xC>(v);         // row_bcast31 -> rows 2, 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xF>(v);
    v += dpp_f<0x4E, 0xF>(v);
    v += dpp_f<0x141, 0xF>(v);
    v += dpp_f<0x140, 0xF>(v);
    v += dpp_f<0x142, 0xA>(v);
    v += dpp_f<0x143, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1, 0xF>(v));
    v = fmaxf(v, dpp_f<0x4E, 0xF>(v));
    v = fmaxf(v, dpp_f<0x141, 0xF>(v));
    v = fmaxf(v, dpp_f<0x140, 0xF>(v));
    // rows not written by a row_bcast receive 0 from the DPP move: take the maximum only where it was written
    const float b15 = dpp_f<0x142, 0xA>(v);
    if ((threadIdx.x & 16) != 0) v = fmaxf(v, b15);                  // rows 1 and 3
    const float b31 = dpp_f<0x143, 0xC>(v);
    if ((threadIdx.x & 32) != 0) v = fmaxf(v, b31);                  // rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Block-wide sum of doubles for blocks of up to 1024 threads; `scratch` holds >= 16 doubles.
// Every thread returns the total.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace ua
