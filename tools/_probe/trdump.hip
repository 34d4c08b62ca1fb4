// empirical probe of ds_read_b64_tr_b16 semantics: LDS holds row-major 16-bit values v = row*64 + col
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // group g: rows 4g..4g+3, columns 0..15
    auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * 64 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
}
