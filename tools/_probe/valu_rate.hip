// VALU issue rates on gfx950, per SIMD, for the instruction kinds of the cache-step kernels (csrc/ua_gmm.hip):
//   hipcc -O3 --offload-arch=gfx950 tools/_probe/valu_rate.hip -o tools/_probe/valu_rate && tools/_probe/valu_rate
// One workgroup per CU with W waves per SIMD (W = 1, 2, 4); every wave runs N independent instructions of one kind in an
// unrolled loop (8 independent chains, so dependency latency is hidden); cycles from s_memtime around the loop.
// Reported: cycles per instruction per SIMD = (cycles of the loop) / (instructions per wave * W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(1024) rate(uint64_t* out, float seed, int iters) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float b = seed * 0.5f + 1.0f;
    const f2 pb = {b, b + 1};
    const double db = b;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##n) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 1) {
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##n) : "v"(pb));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 2) {
#define X(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(pb));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 3) {
#define X(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "v"(pb));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 4) {
#define X(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##n) : "v"(db));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 5) {
#define X(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##n) : "v"(a##n));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 6) {
#define X(n) asm volatile("v_log_f32 %0, %0" : "+v"(a##n));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 7) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##n));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 8) {
#define X(n) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 9) {
#define X(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 10) {      // DPP add (the wave reductions)
#define X(n) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a##n));
            REP8(X) REP8(X)
#undef X
        } else if (KIND == 11) {      // mixed: pk_fma + plain fma alternating
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %3, %3" : "+v"(p##n), "+v"(a##n) : "v"(pb), "v"(b));
            REP8(X)
#undef X
        } else if (KIND == 12) {
#define X(n) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##n) : "v"(db));
            REP8(X) REP8(X)
#undef X
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    if (s == 12345.678f || ds == 1.2345) out[0] = 1;              // keep everything live
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, uint64_t* d) {
    const int iters = 2000, per_iter = 16;
    printf("%-28s", name);
    for (int W : {1, 2, 4}) {
        std::vector<uint64_t> h(1 + 256 * 16);
        hipMemset(d, 0, h.size() * 8);
        rate<KIND><<<256, 256 * W>>>(d, 1.0f, iters);
        rate<KIND><<<256, 256 * W>>>(d, 1.0f, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * W; ++w) v.push_back((double)h[1 + b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        printf("  W=%d: %6.2f cyc/instr/SIMD (%6.2f per wave)", W, med / (iters * per_iter) / W, med / (iters * per_iter));
    }
    printf("\n");
}

int main() {
    uint64_t* d; hipMalloc(&d, (1 + 256 * 16) * 8);
    run<0>("v_fma_f32", d);
    run<9>("v_mul_f32", d);
    run<8>("v_max_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<3>("v_pk_add_f32", d);
    run<11>("pk_fma + fma alternating", d);
    run<4>("v_add_f64", d);
    run<12>("v_fma_f64", d);
    run<5>("v_cvt_f64_f32", d);
    run<6>("v_log_f32", d);
    run<7>("v_rcp_f32", d);
    run<10>("v_add_f32 dpp quad_perm", d);
    return 0;
}
