// Checks on the GPU that log_normal / div_nr of csrc/ua_gmm.hip agree with logf and the IEEE division:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/_probe/logcheck.hip -o tools/_probe/logcheck && tools/_probe/logcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__device__ __forceinline__ float log_normal(float x) {
    const float y = __builtin_amdgcn_logf(x);
    const float c = __uint_as_float(0x3f317217u), cc = __uint_as_float(0x3377d1cfu);
    const float r = y * c;
    return r + __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
}
__device__ __forceinline__ float div_nr(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);
}
__global__ void check(unsigned long long* out, uint32_t n_per_thread) {
    // out[0]: log mismatches, out[1]: div mismatches, out[2]: div mismatches > 1 ulp, out[3]: tested
    uint32_t s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long lm = 0, dm = 0, d2 = 0;
    for (uint32_t i = 0; i < n_per_thread; ++i) {
        s = s * 1664525u + 1013904223u; const uint32_t a = s;
        s = s * 1664525u + 1013904223u; const uint32_t b = s;
        // x: every normal float in [2^-27, 2^20) (covers 1e-8 .. 1e6); t: numerators in [2^-40, 2^4)
        const float x = __uint_as_float(((a % (47u << 23)) + (100u << 23)));
        const float t = __uint_as_float(((b % (44u << 23)) + (87u << 23)));
        if (__float_as_uint(logf(x)) != __float_as_uint(log_normal(x))) ++lm;
        const float q0 = t / x, q1 = div_nr(t, x);
        if (__float_as_uint(q0) != __float_as_uint(q1)) { ++dm; if (fabsf(q0 - q1) > fabsf(q0) * 1.2e-7f) ++d2; }
    }
    atomicAdd(&out[0], lm); atomicAdd(&out[1], dm); atomicAdd(&out[2], d2); atomicAdd(&out[3], (unsigned long long)n_per_thread);
}
int main() {
    unsigned long long* d; unsigned long long h[4] = {0, 0, 0, 0};
    hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    check<<<1024, 256>>>(d, 4096);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("tested %llu pairs: logf != log_normal: %llu; t/x != div_nr: %llu (of which > 1 ulp apart: %llu)\n", h[3], h[0], h[1], h[2]);
    return (h[0] == 0 && h[2] == 0) ? 0 : 1;
}
