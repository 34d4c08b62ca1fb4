// Point grouping for gfx950: farthest-point sampling and kNN + neighbourhood gather.
//
// Reference semantics: models/point_encoder.py:7-14 (fps -> pointnet2_ops CUDA, third-party),
// :17-49 (square_distance / knn_point), :99-127 (Group.forward);
// pure-torch FPS flavour: models/ulip/pointbert/misc.py:40-60.
//
// FPS is G strictly sequential arg-max steps, so it is latency-bound: one wavefront per cloud
// keeps all points and their running min-distance in registers (N/64 per lane), the arg-max is
// a 64-bit-key wavefront butterfly (distance bits | inverted index -> lowest index wins ties) and
// there is no barrier anywhere in the loop.  Clouds of a chunk run on different CUs.
//
// kNN: one wavefront per centre.  The cloud sits in LDS; each lane owns N/64 distances in
// registers.  The S-th smallest distance is found by bisection on the order-preserving integer
// image of the float (32 wavefront-wide counting steps, no sort), ties at the threshold go to the
// lowest indices, and the neighbourhood is written in ascending index order through ballot ranks.
#include "ua_common.h"

namespace {

// 64-bit key maximum over the wavefront on DPP (see ua_common.h: the shuffle butterfly costs twelve
// ds_bpermute round trips per step of the 512-step FPS chain).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)ua::dpp_i<CTRL, ROW_MASK>((int)(unsigned)v);
    const unsigned hi = (unsigned)ua::dpp_i<CTRL, ROW_MASK>((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64<0xB1, 0xF>(v); v = t > v ? t : v;
    t = dpp_u64<0x4E, 0xF>(v); v = t > v ? t : v;
    t = dpp_u64<0x141, 0xF>(v); v = t > v ? t : v;
    t = dpp_u64<0x140, 0xF>(v); v = t > v ? t : v;
    t = dpp_u64<0x142, 0xA>(v); v = t > v ? t : v;          // rows that are not written receive 0 = the identity of max
    t = dpp_u64<0x143, 0xC>(v); v = t > v ? t : v;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------
// FPS: grid B, block 64.  PPL = points per lane; point j*64+lane lives in slot j.
// ------------------------------------------------------------------------------------------
template <int PPL>
__global__ void __launch_bounds__(64)
fps_kernel(const float* __restrict__ xyz, int N, int G, int start, float skip_eps,
           int* __restrict__ center_idx, float* __restrict__ centers) {
    extern __shared__ __attribute__((aligned(16))) float s_xyz[];      // N*3
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = lane; i < N * 3; i += 64) s_xyz[i] = p[i];
    __syncthreads();
    float px[PPL], py[PPL], pz[PPL], md[PPL];
    bool live[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int i = j * 64 + lane;
        const bool in = i < N;
        px[j] = in ? s_xyz[3 * i] : 0.f; py[j] = in ? s_xyz[3 * i + 1] : 0.f; pz[j] = in ? s_xyz[3 * i + 2] : 0.f;
        md[j] = 1e10f;
        const float mag = (px[j] * px[j] + py[j] * py[j]) + pz[j] * pz[j];
        live[j] = in && !(skip_eps >= 0.f && mag <= skip_eps);
    }
    int far = start;
    for (int g = 0; g < G; ++g) {
        const float cx = s_xyz[3 * far], cy = s_xyz[3 * far + 1], cz = s_xyz[3 * far + 2];
        if (lane == 0) {
            if (center_idx) center_idx[(size_t)b * G + g] = far;
            float* c = centers + ((size_t)b * G + g) * 3;
            c[0] = cx; c[1] = cy; c[2] = cz;
        }
        unsigned long long best = 0ull;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (live[j]) {
                md[j] = fminf(md[j], d);
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(md[j]) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(j * 64 + lane));
                best = key > best ? key : best;
            }
        }
        best = wave_max_u64(best);
        far = best == 0ull ? 0 : (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    }
}

// ------------------------------------------------------------------------------------------
// kNN + gather: grid (ceil(G/4), B), block 256 (one wave per centre).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int PPL>
__global__ void __launch_bounds__(256)
knn_group_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb, const float* __restrict__ centers,
                 int N, int G, int S, float* __restrict__ feat, int ldf, int* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // xyz N*3 | |p|^2 N
    float* s_xyz = smem;
    float* s_pn = smem + 3 * N;
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N * 3; i += 256) s_xyz[i] = p[i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = s_xyz[3 * i], y = s_xyz[3 * i + 1], z = s_xyz[3 * i + 2];
        s_pn[i] = (x * x + y * y) + z * z;                               // torch.sum(dst**2,-1)  :48
    }
    __syncthreads();
    const int g = blockIdx.x * 4 + w;
    if (g >= G) return;
    const float* c = centers + ((size_t)b * G + g) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float cn = (cx * cx + cy * cy) + cz * cz;                      // :47
    unsigned key[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int i = j * 64 + lane;
        if (i < N) {
            const float dot = fmaf(cz, s_xyz[3 * i + 2], fmaf(cy, s_xyz[3 * i + 1], cx * s_xyz[3 * i]));
            float d = -2.f * dot;                                        // :46
            d += cn;                                                     // :47
            d += s_pn[i];                                                // :48
            key[j] = ordered_key(d);
        } else {
            key[j] = 0xFFFFFFFFu;
        }
    }
    // smallest T with count(key <= T) >= S
    unsigned lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) cnt += key[j] <= mid;
        cnt = wave_sum_i(cnt);
        if (cnt >= S) hi = mid; else lo = mid + 1;
    }
    const unsigned T = lo;
    int below = 0;
#pragma unroll
    for (int j = 0; j < PPL; ++j) below += key[j] < T;
    below = wave_sum_i(below);
    int need_eq = S - below;                                             // ties at T: lowest indices first
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int base = 0;
    const size_t grp = ((size_t)b * G + g) * S;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int i = j * 64 + lane;
        const bool eq = key[j] == T && i < N;
        const unsigned long long eqm = __ballot(eq);
        const int eq_rank = __popcll(eqm & lt_mask);
        const bool sel = (key[j] < T) || (eq && eq_rank < need_eq);
        need_eq -= min(need_eq, __popcll(eqm));
        const unsigned long long sm = __ballot(sel);
        if (sel) {
            const int pos = base + __popcll(sm & lt_mask);
            float* o = feat + (grp + pos) * (size_t)ldf;
            o[0] = s_xyz[3 * i] - cx; o[1] = s_xyz[3 * i + 1] - cy; o[2] = s_xyz[3 * i + 2] - cz;   // :124
            if (rgb) { const float* q = rgb + ((size_t)b * N + i) * 3; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; }   // :126
            else { o[3] = 0.f; o[4] = 0.f; o[5] = 0.f; }          // xyz-only variant (models/ulip/pointbert/dvae.py:152-181)
            for (int t = 6; t < ldf; ++t) o[t] = 0.f;
            if (idx_out) idx_out[grp + pos] = i;
        }
        base += __popcll(sm);
    }
}

// ------------------------------------------------------------------------------------------
// Ball query + gather (OpenShape PPAT set abstraction: models/openshape/pointnet_util.py:89-146):
// the first S points IN INDEX ORDER with squared distance <= r^2 to the centre (square_distance's
// expanded form, :20-42), padded with the first hit; row = [p - centre (3) | point features (F)].
// One wavefront per centre, grid (ceil(G/4), B), block 256; the cloud (xyz and |p|^2) sits in LDS.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ball_group_kernel(const float* __restrict__ xyz, const float* __restrict__ pf, const float* __restrict__ centers,
                  int N, int G, int S, int F, float r2, float* __restrict__ feat, int ldf, int* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // xyz N*3 | |p|^2 N
    float* s_xyz = smem;
    float* s_pn = smem + 3 * N;
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N * 3; i += 256) s_xyz[i] = p[i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = s_xyz[3 * i], y = s_xyz[3 * i + 1], z = s_xyz[3 * i + 2];
        s_pn[i] = (x * x + y * y) + z * z;
    }
    __syncthreads();
    const int g = blockIdx.x * 4 + w;
    if (g >= G) return;
    const float* c = centers + ((size_t)b * G + g) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float cn = (cx * cx + cy * cy) + cz * cz;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const size_t grp = ((size_t)b * G + g) * S;
    int found = 0, first = -1;
    auto emit = [&](int pos, int i) {                                    // one lane writes row `pos` from point i
        float* o = feat + (grp + pos) * (size_t)ldf;
        o[0] = s_xyz[3 * i] - cx; o[1] = s_xyz[3 * i + 1] - cy; o[2] = s_xyz[3 * i + 2] - cz;        // :129
        const float* q = pf + ((size_t)b * N + i) * F;
        for (int t = 0; t < F; ++t) o[3 + t] = q[t];                                                  // :133-134
        for (int t = 3 + F; t < ldf; ++t) o[t] = 0.f;
        if (idx_out) idx_out[grp + pos] = i;
    };
    for (int i0 = 0; i0 < N && found < S; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < N) {
            const float dot = fmaf(cz, s_xyz[3 * i + 2], fmaf(cy, s_xyz[3 * i + 1], cx * s_xyz[3 * i]));
            float d = -2.f * dot;                                        // :39
            d += cn;                                                     // :40 (src = query centre)
            d += s_pn[i];                                                // :41
            in = !(d > r2);                                              // group_idx[sqrdists > radius**2] = N  (:104)
        }
        const unsigned long long m = __ballot(in);
        if (first < 0 && m) first = i0 + __builtin_ctzll(m);
        const int pos = found + __popcll(m & lt_mask);
        if (in && pos < S) emit(pos, i);
        found += __popcll(m);
    }
    found = min(found, S);
    if (first < 0) first = 0;                                            // cannot happen for a centre taken from the cloud
    for (int pos = found + lane; pos < S; pos += 64) emit(pos, first);   // :106-108 pad with the first hit
}

}  // namespace

extern "C" int ua_fps(const float* xyz, int32_t B, int32_t N, int32_t G, int32_t start, float skip_eps,
                      int32_t* center_idx, float* centers, void* stream) {
    UA_CHECK_PTR(xyz); UA_CHECK_PTR(centers);
    if (B <= 0 || N <= 0 || G <= 0 || start < 0 || start >= N) return UA_EINVAL;
    if (N > 4096) return UA_EUNSUPPORTED;
    const size_t lds = (size_t)N * 3 * sizeof(float);
    hipStream_t st = ua::as_stream(stream);
    if (N <= 1024)      hipLaunchKernelGGL(fps_kernel<16>, dim3(B), dim3(64), lds, st, xyz, N, G, start, skip_eps, center_idx, centers);
    else if (N <= 2048) hipLaunchKernelGGL(fps_kernel<32>, dim3(B), dim3(64), lds, st, xyz, N, G, start, skip_eps, center_idx, centers);
    else                hipLaunchKernelGGL(fps_kernel<64>, dim3(B), dim3(64), lds, st, xyz, N, G, start, skip_eps, center_idx, centers);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_knn_group(const float* xyz, const float* rgb, const float* centers,
                            int32_t B, int32_t N, int32_t G, int32_t S,
                            float* feat, int32_t ldf, int32_t* idx, void* stream) {
    UA_CHECK_PTR(xyz); UA_CHECK_PTR(centers); UA_CHECK_PTR(feat);       // rgb may be NULL (xyz-only groups)
    if (B <= 0 || N <= 0 || G <= 0 || S <= 0 || S > N || ldf < 6) return UA_EINVAL;
    if (N > 4096 || S > 256) return UA_EUNSUPPORTED;
    const size_t lds = (size_t)N * 4 * sizeof(float);
    dim3 grid((G + 3) / 4, B);
    hipStream_t st = ua::as_stream(stream);
    if (lds > 48 * 1024) {
        // dynamic LDS above the default cap needs the attribute (N > 3072)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_group_kernel<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UA_ELAUNCH;
    }
    if (N <= 1024)      hipLaunchKernelGGL(knn_group_kernel<16>, grid, dim3(256), lds, st, xyz, rgb, centers, N, G, S, feat, ldf, idx);
    else if (N <= 2048) hipLaunchKernelGGL(knn_group_kernel<32>, grid, dim3(256), lds, st, xyz, rgb, centers, N, G, S, feat, ldf, idx);
    else                hipLaunchKernelGGL(knn_group_kernel<64>, grid, dim3(256), lds, st, xyz, rgb, centers, N, G, S, feat, ldf, idx);
    UA_LAUNCH_CHECK();
    return UA_OK;
}

extern "C" int ua_ball_group(const float* xyz, const float* point_feat, const float* centers,
                             int32_t B, int32_t N, int32_t G, int32_t S, int32_t F, float radius,
                             float* feat, int32_t ldf, int32_t* idx, void* stream) {
    UA_CHECK_PTR(xyz); UA_CHECK_PTR(point_feat); UA_CHECK_PTR(centers); UA_CHECK_PTR(feat);
    if (B <= 0 || N <= 0 || G <= 0 || S <= 0 || F < 0 || !(radius > 0.f) || ldf < 3 + F) return UA_EINVAL;
    if (N > 8192) return UA_EUNSUPPORTED;
    const size_t lds = (size_t)N * 4 * sizeof(float);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(ball_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return UA_ELAUNCH;
    hipLaunchKernelGGL(ball_group_kernel, dim3((G + 3) / 4, B), dim3(256), lds, ua::as_stream(stream),
                       xyz, point_feat, centers, N, G, S, F, radius * radius, feat, ldf, idx);
    UA_LAUNCH_CHECK();
    return UA_OK;
}
