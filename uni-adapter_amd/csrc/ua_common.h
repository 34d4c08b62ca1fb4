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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
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
