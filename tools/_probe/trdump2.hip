#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int VROW = 320;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) unsigned char s_v[32 * VROW];
    // element at (key, halfindex) = key*256 + halfindex  (halfindex 0..127 within the 256-byte row)
    for (int i = threadIdx.x; i < 32 * 128; i += 64) { int key = i / 128, hidx = i % 128; *(short*)&s_v[key * VROW + hidx * 2] = (short)(key * 256 + hidx); }
    __syncthreads();
    const int lane = threadIdx.x;
    const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned char* vp = &s_v[(4 * (gq >> 1) + tq) * VROW + (16 * (gq & 1) + 4 * tp) * 2];
    typedef __attribute__((address_space(3))) s16x4* lp_t;
    int n = 0;
    for (int ks = 0; ks < 2; ++ks) for (int db = 0; db < 2; ++db) {
        const unsigned char* r = vp + (16 * ks) * VROW + db * 128;
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 8 * VROW));
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 64));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(r + 8 * VROW + 64));
        for (int j = 0; j < 4; ++j) { out[(n * 64 + lane) * 16 + j] = a0[j]; out[(n * 64 + lane) * 16 + 4 + j] = a1[j]; out[(n * 64 + lane) * 16 + 8 + j] = b0[j]; out[(n * 64 + lane) * 16 + 12 + j] = b1[j]; }
        ++n;
    }
}
int main() {
    short* d; hipMalloc(&d, 4 * 64 * 16 * 2);
    k<<<1, 64>>>(d);
    static short h[4 * 64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int n = 0; n < 4; ++n) { int ks = n >> 1, db = n & 1;
        for (int lane = 0; lane < 64; ++lane) { int hh = lane >> 5, dw = (lane & 31) + 32 * db;
            for (int i = 0; i < 8; ++i) for (int lo = 0; lo < 2; ++lo) {
                int key = (i & 3) + 8 * (i >> 2) + 4 * hh + 16 * ks;
                int hidx = (dw >> 5) * 64 + lo * 32 + (dw & 31);
                int want = key * 256 + hidx, got = h[(n * 64 + lane) * 16 + lo * 8 + i];
                if (want != got) { if (bad < 8) printf("n%d lane%d i%d lo%d want key%d hidx%d got key%d hidx%d\n", n, lane, i, lo, key, hidx, got / 256, got % 256); ++bad; }
            } } }
    printf("bad=%d\n", bad);
}
