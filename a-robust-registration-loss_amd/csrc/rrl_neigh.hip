// rrl_neigh.hip -- pseudo-triangle builder: farthest-point sampling and
// k = 3 nearest neighbours on the GPU, replacing the reference's sequential torch loop
// (code/utils.py:275-296, farthest_point_sample) and sklearn KDTree query (code/loss.py:473-485).
// Preprocessing outside the timed loss path; correctness first, but no host round trips:
// the S sequential FPS iterations run inside ONE workgroup with one barrier per iteration.
#include "rrl_common.h"

// (value, index) key: larger distance wins, then the SMALLER index (torch.max's first occurrence)
__device__ __forceinline__ unsigned long long fps_key(float d, int i) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
}

__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v, int ctrl_sel) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    unsigned plo, phi;
    switch (ctrl_sel) {  // compile-time after unrolling
        case 0: plo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); phi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); break;
        case 1: plo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); phi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); break;
        case 2: plo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); phi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); break;
        default: plo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); phi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); break;
    }
    return ((unsigned long long)phi << 32) | plo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // butterfly inside rows of 16 (DPP), then the 4 rows via SGPRs
        unsigned long long p = dpp_u64(v, s);
        v = p > v ? p : v;
    }
    unsigned long long best = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 16 * r);
        unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), 16 * r);
        unsigned long long x = ((unsigned long long)hi << 32) | lo;
        best = x > best ? x : best;
    }
    return best;
}

// Farthest-point sampling, code/utils.py:275-296: distance = 1e10; repeat S times: emit the
// current point, dist = ((dx^2 + dy^2) + dz^2) to it (fp32, no FMA), distance = min(distance,
// dist), next = argmax(distance) (first occurrence).  One 1024-lane workgroup per cloud; points
// and running distances live in LDS (16 B per point) when they fit, else in global scratch.
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void fps_kernel(const float *__restrict__ pts,
                                                   const int32_t *__restrict__ start,
                                                   int32_t *__restrict__ out, float *__restrict__ dist_g,
                                                   int n, int S) {
    extern __shared__ __attribute__((aligned(16))) float4 cache[];  // IN_LDS: (x, y, z, distance)
    __shared__ unsigned long long wbest[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = pts + (size_t)b * n * 3;
    float *dg = dist_g + (size_t)b * n;
    for (int i = tid; i < n; i += 1024) {
        if (IN_LDS) cache[i] = make_float4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 1e10f);
        else dg[i] = 1e10f;
    }
    int far = start[b];
    __syncthreads();
    for (int it = 0; it < S; ++it) {
        if (tid == 0) out[(size_t)b * S + it] = far;
        float cx, cy, cz;
        if (IN_LDS) { const float4 c = cache[far]; cx = c.x; cy = c.y; cz = c.z; }
        else { cx = p[3 * far]; cy = p[3 * far + 1]; cz = p[3 * far + 2]; }
        unsigned long long best = 0;
        for (int i = tid; i < n; i += 1024) {
            float x, y, z, d;
            if (IN_LDS) { const float4 q = cache[i]; x = q.x; y = q.y; z = q.z; d = q.w; }
            else { x = p[3 * i]; y = p[3 * i + 1]; z = p[3 * i + 2]; d = dg[i]; }
            const float dx = x - cx, dy = y - cy, dz = z - cz;
            float s = dx * dx;
            s = s + dy * dy;
            s = s + dz * dz;
            if (s < d) {  // distance[mask] = dist[mask]
                d = s;
                if (IN_LDS) cache[i].w = d; else dg[i] = d;
            }
            const unsigned long long k = fps_key(d, i);
            best = k > best ? k : best;
        }
        best = wave_max_u64(best);
        if (lane == 0) wbest[it & 1][wave] = best;
        __syncthreads();
        unsigned long long all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned long long x = wbest[it & 1][w];
            all = x > all ? x : all;
        }
        far = (int)(0xffffffffu - (unsigned)all);
    }
}

extern "C" int rrl_fps(const float *pts, const int32_t *start, int32_t *out_idx, float *dist_scratch,
                       int B, int n, int S, void *stream) {
    if (!pts || !start || !out_idx || !dist_scratch || B < 0 || n <= 0 || S < 0 || S > n) return RRL_E_ARG;
    if (B == 0 || S == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)n * 16 <= 128 * 1024)
        hipLaunchKernelGGL(fps_kernel<true>, dim3((unsigned)B), dim3(1024), (size_t)n * 16, s, pts, start,
                           out_idx, dist_scratch, n, S);
    else
        hipLaunchKernelGGL(fps_kernel<false>, dim3((unsigned)B), dim3(1024), 0, s, pts, start, out_idx,
                           dist_scratch, n, S);
    RRL_LAUNCH_CHECK();
    return 0;
}

// k nearest neighbours (k = 3) of S query points among the n points of their cloud, by brute
// force in double precision (what sklearn's KDTree compares), ties to the lower index.  Lane =
// query, the candidate point is wave-uniform (scalar loads).
typedef const float __attribute__((address_space(4))) * kptr_n;

__global__ __launch_bounds__(256) void knn3_kernel(const float *__restrict__ pts,
                                                   const int32_t *__restrict__ query_idx,
                                                   int32_t *__restrict__ nn, int n, int S) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const float *p = pts + (size_t)b * n * 3;
    const int qi = q < S ? query_idx[(size_t)b * S + q] : 0;
    const double qx = p[3 * qi], qy = p[3 * qi + 1], qz = p[3 * qi + 2];
    double d0 = 1e300, d1 = 1e300, d2 = 1e300;
    int i0 = 0, i1 = 0, i2 = 0;
    kptr_n tp = (kptr_n)(uintptr_t)p;
    for (int j = 0; j < n; ++j, tp += 3) {
        const double dx = qx - (double)tp[0], dy = qy - (double)tp[1], dz = qz - (double)tp[2];
        const double d = (dx * dx + dy * dy) + dz * dz;
        if (d < d2) {
            if (d < d1) {
                d2 = d1; i2 = i1;
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            } else { d2 = d; i2 = j; }
        }
    }
    if (q < S) {
        int32_t *o = nn + ((size_t)b * S + q) * 3;
        o[0] = i0; o[1] = i1; o[2] = i2;
    }
}

extern "C" int rrl_knn3(const float *pts, const int32_t *query_idx, int32_t *nn, int B, int n, int S,
                        void *stream) {
    if (!pts || !query_idx || !nn || B < 0 || n <= 0 || S < 0) return RRL_E_ARG;
    if (B == 0 || S == 0) return 0;
    hipLaunchKernelGGL(knn3_kernel, dim3((unsigned)((S + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, pts, query_idx, nn, n, S);
    RRL_LAUNCH_CHECK();
    return 0;
}
