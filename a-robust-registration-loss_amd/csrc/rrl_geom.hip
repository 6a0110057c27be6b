// rrl_geom.hip -- K6 rigid apply, K7 Chamfer monitor, K8 line sampler.
//   K6 code/loss.py:458-463 (+ the RPM/DCP/FMR layouts, see include/rrl.h)   HBM-bound
//   K7 code/loss.py:38-52, 236-252                                           VALU-bound, N*M pairs
//   K8 code/loss.py:265-432                                                  tiny
#include <stdlib.h>

#include "rrl_ws.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

// ---------------------------------------------------------------------------------------
// K6 rigid apply
// ---------------------------------------------------------------------------------------
struct Mat {
    float r[9], t[3];
};

__device__ __forceinline__ Mat load_mat(const float *__restrict__ R, const float *__restrict__ t,
                                        int b, int transpose_r) {
    Mat m;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)  // m.r[i*3+j] multiplies x_i into y_j
            m.r[i * 3 + j] = transpose_r ? R[b * 9 + j * 3 + i] : R[b * 9 + i * 3 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) m.t[j] = t ? t[b * 3 + j] : 0.0f;
    return m;
}

template <bool CF>
__global__ __launch_bounds__(256) void rigid_fwd_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ R,
                                                        const float *__restrict__ t,
                                                        float *__restrict__ y, int n,
                                                        int transpose_r) {
    const int b = blockIdx.y;
    const Mat m = load_mat(R, t, b, transpose_r);
    const size_t base = (size_t)b * n * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = CF ? x[base + (size_t)c * n + i] : x[base + 3 * (size_t)i + c];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = fmaf(v[2], m.r[6 + j], fmaf(v[1], m.r[3 + j], v[0] * m.r[j])) + m.t[j];
            if (CF) y[base + (size_t)j * n + i] = s; else y[base + 3 * (size_t)i + j] = s;
        }
    }
}

extern "C" int rrl_rigid_apply_fwd(const float *x, const float *R, const float *t, float *y, int B,
                                   int n, int transpose_r, int channel_first, void *stream) {
    if (!x || !R || !t || !y || B < 0 || n < 0) return RRL_E_ARG;
    if (B == 0 || n == 0) return 0;
    unsigned gx = (unsigned)((n + 255) / 256);
    if (gx > 2048) gx = 2048;
    dim3 grid(gx, (unsigned)B);
    if (channel_first)
        hipLaunchKernelGGL(rigid_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, R, t,
                           y, n, transpose_r);
    else
        hipLaunchKernelGGL(rigid_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, R,
                           t, y, n, transpose_r);
    RRL_LAUNCH_CHECK();
    return 0;
}

#define RIGID_BWD_PTS 16384  // points per 1024-lane workgroup in the backward reduction

extern "C" int rrl_rigid_bwd_blocks(int n) { return n > 0 ? (n + RIGID_BWD_PTS - 1) / RIGID_BWD_PTS : 0; }

// y_j = sum_i x_i m[i][j] + t_j  =>  gx_i = sum_j gy_j m[i][j],  gm[i][j] = sum x_i gy_j,
// gt_j = sum gy_j.  Wave shuffle reduction, then a fixed-order cross-wave / cross-block sum
// (deterministic: no float atomics).
template <bool CF>
__global__ __launch_bounds__(1024) void rigid_bwd_kernel(const float *__restrict__ x,
                                                         const float *__restrict__ R,
                                                         const float *__restrict__ gy,
                                                         float *__restrict__ gx,
                                                         float *__restrict__ partial,
                                                         float *__restrict__ gR,
                                                         float *__restrict__ gt, int n,
                                                         int transpose_r) {
    __shared__ float red[16][12];
    const int b = blockIdx.y;
    const Mat m = load_mat(R, nullptr, b, transpose_r);
    const size_t base = (size_t)b * n * 3;
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    const int i0 = blockIdx.x * RIGID_BWD_PTS;
    for (int i = i0 + threadIdx.x; i < min(n, i0 + RIGID_BWD_PTS); i += 1024) {
        float v[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = CF ? x[base + (size_t)c * n + i] : x[base + 3 * (size_t)i + c];
            g[c] = CF ? gy[base + (size_t)c * n + i] : gy[base + 3 * (size_t)i + c];
        }
        if (gx) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = fmaf(g[2], m.r[c * 3 + 2], fmaf(g[1], m.r[c * 3 + 1], g[0] * m.r[c * 3]));
                if (CF) gx[base + (size_t)c * n + i] = s; else gx[base + 3 * (size_t)i + c] = s;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[c * 3 + j] = fmaf(v[c], g[j], acc[c * 3 + j]);
            acc[9 + c] += g[c];
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 12; ++q) red[wave][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 12) {
        const int q = threadIdx.x;
        float s = 0.0f;
        for (int w = 0; w < 16; ++w) s += red[w][q];  // fixed order
        if (gridDim.x > 1) {
            partial[((size_t)b * gridDim.x + blockIdx.x) * 12 + q] = s;
        } else if (q < 9) {  // single workgroup per sample: write the result directly
            int i = q / 3, j = q % 3;
            gR[b * 9 + (transpose_r ? j * 3 + i : i * 3 + j)] = s;
        } else {
            gt[b * 3 + (q - 9)] = s;
        }
    }
}

__global__ void rigid_bwd_finalize_kernel(const float *__restrict__ partial, float *__restrict__ gR,
                                          float *__restrict__ gt, int nblk, int transpose_r) {
    const int b = blockIdx.x, q = threadIdx.x;
    if (q >= 12) return;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)partial[((size_t)b * nblk + k) * 12 + q];
    if (q < 9) {
        int i = q / 3, j = q % 3;  // gm[i][j]; m[i][j] = R[i][j] or R[j][i]
        gR[b * 9 + (transpose_r ? j * 3 + i : i * 3 + j)] = (float)s;
    } else {
        gt[b * 3 + (q - 9)] = (float)s;
    }
}

extern "C" int rrl_rigid_apply_bwd(const float *x, const float *R, const float *gy, float *gx,
                                   float *gR, float *gt, float *partial, int B, int n,
                                   int transpose_r, int channel_first, void *stream) {
    if (!x || !R || !gy || !gR || !gt || !partial || B < 0 || n < 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = rrl_rigid_bwd_blocks(n);
    if (nblk > 0) {
        dim3 grid((unsigned)nblk, (unsigned)B);
        if (channel_first)
            hipLaunchKernelGGL(rigid_bwd_kernel<true>, grid, dim3(1024), 0, s, x, R, gy, gx, partial,
                               gR, gt, n, transpose_r);
        else
            hipLaunchKernelGGL(rigid_bwd_kernel<false>, grid, dim3(1024), 0, s, x, R, gy, gx,
                               partial, gR, gt, n, transpose_r);
        RRL_LAUNCH_CHECK();
    }
    if (nblk != 1) {  // 0 blocks: zeros; > 1: fixed-order sum of the per-block partials
        hipLaunchKernelGGL(rigid_bwd_finalize_kernel, dim3((unsigned)B), dim3(64), 0, s, partial, gR,
                           gt, nblk, transpose_r);
        RRL_LAUNCH_CHECK();
    }
    return 0;
}

// Backward tail of the fused training op in ONE launch: rigid-apply backward of the accumulated
// triangle gradient g1 (cleared again on the way, so the next backward on this workspace starts
// from zero) over REG_BWD_PTS points per 256-lane workgroup -- a single workgroup per sample is
// bound by one CU's memory pipe --, per-workgroup partial sums, and, by the LAST workgroup of the
// launch to finish (ticket counter), the fixed-order reduction to dR / dt per sample and the
// 14-float shard payload.  Deterministic: no float atomics, fixed summation order.
#define REG_BWD_PTS 1024

__device__ __forceinline__ float ld_agent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void reg_bwd_kernel(const float *__restrict__ x,
                                                      const float *__restrict__ R,
                                                      float *__restrict__ g1, float *__restrict__ gx,
                                                      float *__restrict__ partial,
                                                      float *__restrict__ gR, float *__restrict__ gt,
                                                      float *__restrict__ payload,
                                                      const float *__restrict__ loss,
                                                      const int32_t *__restrict__ info,
                                                      int32_t *__restrict__ done, int n, int B,
                                                      int transpose_r) {
    __shared__ float red[4][12];
    __shared__ int ticket;
    __shared__ double psum[14];
    const int b = blockIdx.y, nblk = gridDim.x;
    const Mat m = load_mat(R, nullptr, b, transpose_r);
    const size_t base = (size_t)b * n * 3;
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    const int i0 = blockIdx.x * REG_BWD_PTS, i1 = min(n, i0 + REG_BWD_PTS);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        float v[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = x[base + 3 * (size_t)i + c];
            g[c] = g1[base + 3 * (size_t)i + c];
            g1[base + 3 * (size_t)i + c] = 0.0f;
        }
        if (gx) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                gx[base + 3 * (size_t)i + c] =
                    fmaf(g[2], m.r[c * 3 + 2], fmaf(g[1], m.r[c * 3 + 1], g[0] * m.r[c * 3]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[c * 3 + j] = fmaf(v[c], g[j], acc[c * 3 + j]);
            acc[9 + c] += g[c];
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 12; ++q) red[wave][q] = acc[q];
    __syncthreads();
    // Cross-workgroup hand-over WITHOUT __threadfence(): an agent-scope fence writes back the
    // whole L2 of the XCD (several microseconds here).  Instead the 12 partials go out as
    // agent-scope (write-through) atomic stores from wave 0, which waits for their completion
    // before its lane 0 takes the ticket; the last workgroup reads them with agent-scope loads.
    // This is NOT the HIP memory model's release/acquire pairing: it relies on gfx9 behaviour (stores
    // are counted by vmcnt, agent-scope accesses are served by the device-coherent level), hence the
    // target check; tests/test_gpu_parity.py stresses the d/dsrc path over many runs.
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "reg_bwd_kernel's ticket hand-over is written for gfx942 / gfx950 (vmcnt-counted stores)"
#endif
    if (threadIdx.x < 12) {
        const int q = threadIdx.x;
        __hip_atomic_store(&partial[((size_t)b * nblk + blockIdx.x) * 12 + q],
                           (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0)
            ticket = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 14) psum[threadIdx.x] = 0.0;
    __syncthreads();
    if (ticket != nblk * B - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the last workgroup: per-sample sums over the workgroups in index order (double accumulator),
    // then the payload over the samples in index order
    for (int k0 = 0; k0 < B; k0 += 16) {
        const int k = k0 + (int)(threadIdx.x / 12), q = threadIdx.x % 12;
        float r = 0.0f;
        const bool on = threadIdx.x < 192 && k < B;
        if (on) {
            double s = 0.0;
            for (int j = 0; j < nblk; ++j) s += (double)ld_agent(&partial[((size_t)k * nblk + j) * 12 + q]);
            r = (float)s;
            if (q < 9) {
                int i = q / 3, jj = q % 3;
                gR[k * 9 + (transpose_r ? jj * 3 + i : i * 3 + jj)] = r;
            } else {
                gt[k * 3 + (q - 9)] = r;
            }
        }
        if (payload) {  // samples in index order: 16 per pass, serialised by a single lane per slot
            __shared__ float stage[16][12];
            if (on) stage[threadIdx.x / 12][q] = r;
            __syncthreads();
            if (threadIdx.x < 12) {
                double s = psum[2 + threadIdx.x];
                for (int kk = 0; kk < min(16, B - k0); ++kk) s += (double)stage[kk][threadIdx.x];
                psum[2 + threadIdx.x] = s;
            }
            __syncthreads();
        }
    }
    if (!payload) return;
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int k = 0; k < B; ++k) s += info[k * 4] > 0 ? (threadIdx.x == 0 ? (double)loss[k] : 1.0) : 0.0;
        payload[threadIdx.x] = (float)s;
    } else if (threadIdx.x < 14) {
        // psum[2..10] = dR entries in (i, j) order of m; the payload wants gR's memory order
        const int q = threadIdx.x - 2;
        if (q < 9) {
            int i = q / 3, jj = q % 3;
            payload[2 + (transpose_r ? jj * 3 + i : i * 3 + jj)] = (float)psum[2 + q];
        } else {
            payload[2 + q] = (float)psum[2 + q];
        }
    }
}

// the fused tail applies when the clouds take the sorted prepare path (whose build kernel clears
// G1 in the forward)
int rrl_sort_capacity(void);
int rrl_fused_backward(int B, int N, int M) { return B > 0 && N > 0 && (N > M ? N : M) <= rrl_sort_capacity(); }

int rrl_launch_reg_bwd(const float *src, const float *R, float *g1, float *grad_src, float *partial,
                       float *gR, float *gt, float *payload, const float *loss, const int32_t *info,
                       int32_t *done, int B, int N, int transpose_r, hipStream_t s) {
    const int n = 3 * N;
    hipLaunchKernelGGL(reg_bwd_kernel, dim3((unsigned)((n + REG_BWD_PTS - 1) / REG_BWD_PTS), (unsigned)B),
                       dim3(256), 0, s, src, R, g1, grad_src, partial, gR, gt, payload, loss, info, done, n,
                       B, transpose_r);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K7 Chamfer: nearest target of every query.  Lane = query point (registers), targets are
// wave-uniform and stream through the scalar cache.  A (query tile, target chunk) workgroup
// folds its partial result into a u64 key (dist bits << 32 | index) with atomicMin: smallest
// distance, then smallest index == torch.min's first occurrence; order independent.
// ---------------------------------------------------------------------------------------
// Both directions run in ONE launch (blockIdx.z = 2 b + direction) and the target chunk adapts to
// the problem: the demo shape (B=1, 1024 x 1024) used to be 4 workgroups walking 1024 targets
// each, 81 us per direction; with 32-target chunks it is 256 workgroups and a few microseconds.
__global__ __launch_bounds__(256) void chamfer_nn_kernel(const float *__restrict__ x,
                                                         const float *__restrict__ y,
                                                         unsigned long long *__restrict__ best_x,
                                                         unsigned long long *__restrict__ best_y,
                                                         int N, int M, int chunk) {
    const int b = blockIdx.z >> 1, dir = blockIdx.z & 1;
    const float *q = dir ? y : x, *tg = dir ? x : y;
    unsigned long long *best = dir ? best_y : best_x;
    const int nq = dir ? M : N, nt = dir ? N : M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * chunk, j1 = min(nt, j0 + chunk);
    if ((int)blockIdx.x * 256 >= nq || j0 >= nt) return;  // uniform: the grid covers the larger side
    const float *qp = q + ((size_t)b * nq + (i < nq ? i : 0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    kptr tp = (kptr)(uintptr_t)(tg + ((size_t)b * nt + j0) * 3);
    float bd = INFINITY;
    int bj = j0;
    for (int j = j0; j < j1; ++j, tp += 3) {
        // code/loss.py:51: sum((x - y)**2, -1); (a0 + a1) + a2, no FMA
        float dx = qx - tp[0], dy = qy - tp[1], dz = qz - tp[2];
        float s = dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        if (s < bd) { bd = s; bj = j; }
    }
    if (i < nq) {
        unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bj;
        atomicMin(&best[(size_t)b * nq + i], key);
    }
}

__global__ __launch_bounds__(1024) void chamfer_mean_kernel(const unsigned long long *__restrict__ bx,
                                                            const unsigned long long *__restrict__ by,
                                                            float *__restrict__ value, long nx,
                                                            long ny) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < nx; i += 1024) s += (double)__uint_as_float((unsigned)(bx[i] >> 32));
    for (long i = threadIdx.x; i < ny; i += 1024) s += (double)__uint_as_float((unsigned)(by[i] >> 32));
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) value[0] = (float)(red[0] / (double)(nx + ny));
}

extern "C" int rrl_chamfer_fwd(const float *x, const float *y, uint64_t *best_x, uint64_t *best_y,
                               float *value, int B, int N, int M, void *stream) {
    if (!x || !y || !best_x || !best_y || !value || B < 0 || N < 0 || M < 0) return RRL_E_ARG;
    if (B == 0 || N == 0 || M == 0) return RRL_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    // keys start at all-ones (a kernel, never a memset node: see rrl_fill)
    int rc;
    if ((rc = rrl_fill(best_x, 0xffffffffu, sizeof(uint64_t) * (size_t)B * N, s))) return rc;
    if ((rc = rrl_fill(best_y, 0xffffffffu, sizeof(uint64_t) * (size_t)B * M, s))) return rc;
    // target chunk: as large as still gives ~2048 workgroups (32 <= chunk <= 1024)
    const int big = N > M ? N : M;
    const long qblocks = (long)2 * B * ((big + 255) / 256);
    int chunk = 1024;
    while (chunk > 32 && qblocks * ((big + chunk - 1) / chunk) < 2048) chunk >>= 1;
    hipLaunchKernelGGL(chamfer_nn_kernel,
                       dim3((unsigned)((big + 255) / 256), (unsigned)((big + chunk - 1) / chunk), (unsigned)(2 * B)),
                       dim3(256), 0, s, x, y, (unsigned long long *)best_x, (unsigned long long *)best_y, N, M, chunk);
    RRL_LAUNCH_CHECK();
    hipLaunchKernelGGL(chamfer_mean_kernel, dim3(1), dim3(1024), 0, s,
                       (const unsigned long long *)best_x, (const unsigned long long *)best_y, value,
                       (long)B * N, (long)B * M);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K9 pose kernels of the single-pair demo (code/loss.py:437-463 Reconstruction_point.Transform,
// code/LieAlgebra/se3.py:83-106 exp3, sinc.py:5-17, 91-103, 120-132; torch.optim.Adam as the demo
// uses it, test_demo_optimized_Lie_Algebra.py:42, 64-66).  The captured demo step spent ~330 of
// its ~350 graph nodes in the exponential map, its autograd and Adam written as torch ops on 6
// floats; here they are three launches.
//   xi = (w, v);  t = |w|;  W = [w]x;  R = I + s1 W + s2 W^2;  V = I + s2 W + s3 W^2;  T = V v
//   s1 = sin t / t, s2 = (1 - cos t) / t^2, s3 = (t - sin t) / t^3; the reference's Taylor
//   polynomials in t^2 inside |t| < 0.01.
// The backward evaluates the same formulas on dual numbers, one lane per (sample, parameter): the
// derivative is exact for whichever branch the value takes, with no hand-derived Jacobian.
// ---------------------------------------------------------------------------------------
struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { return {a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v)}; }
__device__ __forceinline__ Dual dconst(float c) { return {c, 0.0f}; }
__device__ __forceinline__ Dual dsqrt(Dual a) { const float r = sqrtf(a.v); return {r, a.d / (2.0f * r)}; }
__device__ __forceinline__ Dual dsin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
__device__ __forceinline__ Dual dcos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }

__device__ __forceinline__ float sc_val(float a) { return a; }
__device__ __forceinline__ float sc_val(Dual a) { return a.v; }
__device__ __forceinline__ float mk(float, float c) { return c; }
__device__ __forceinline__ Dual mk(Dual, float c) { return dconst(c); }
__device__ __forceinline__ float sc_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ Dual sc_sqrt(Dual a) { return dsqrt(a); }
__device__ __forceinline__ float sc_sin(float a) { return sinf(a); }
__device__ __forceinline__ Dual sc_sin(Dual a) { return dsin(a); }
__device__ __forceinline__ float sc_cos(float a) { return cosf(a); }
__device__ __forceinline__ Dual sc_cos(Dual a) { return dcos(a); }

// R (row-major 3x3) and T from xi; S = float (value) or Dual (value + one directional derivative)
template <typename S>
__device__ __forceinline__ void se3_exp3(const S *xi, S *R, S *T) {
    const S one = mk(xi[0], 1.0f), zero = mk(xi[0], 0.0f);
    const S a = xi[0], b = xi[1], c = xi[2];
    const S n2 = a * a + b * b + c * c;
    // |w| with a finite derivative at w = 0 (LieAlgebra/so3.py _angle)
    const S t = sc_val(n2) > 0.0f ? sc_sqrt(n2) : zero;
    const S t2 = t * t;
    S s1, s2, s3;
    if (fabsf(sc_val(t)) < 0.01f) {  // code/LieAlgebra/sinc.py:14, 100, 129
        s1 = one - t2 / mk(t, 6.0f) * (one - t2 / mk(t, 20.0f) * (one - t2 / mk(t, 42.0f)));
        s2 = mk(t, 0.5f) * (one - t2 / mk(t, 12.0f) * (one - t2 / mk(t, 30.0f) * (one - t2 / mk(t, 56.0f))));
        s3 = (one / mk(t, 6.0f)) * (one - t2 / mk(t, 20.0f) * (one - t2 / mk(t, 42.0f) * (one - t2 / mk(t, 72.0f))));
    } else {
        const S sn = sc_sin(t), cs = sc_cos(t);
        s1 = sn / t;
        s2 = (one - cs) / (t * t);
        s3 = (t - sn) / (t * t * t);
    }
    // W = [[0,-c,b],[c,0,-a],[-b,a,0]] (so3.mat), Sq = W W
    const S W[9] = {zero, -c, b, c, zero, -a, -b, a, zero};
    S Sq[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Sq[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    S V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const S id = (i == 0 || i == 4 || i == 8) ? one : zero;
        R[i] = id + s1 * W[i] + s2 * Sq[i];
        V[i] = id + s2 * W[i] + s3 * Sq[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = V[3 * i] * xi[3] + V[3 * i + 1] * xi[4] + V[3 * i + 2] * xi[5];
}

__global__ __launch_bounds__(64) void se3_exp_kernel(const float *__restrict__ xi, float *__restrict__ R,
                                                     float *__restrict__ T, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float x[6], r[9], t[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = xi[b * 6 + i];
    se3_exp3<float>(x, r, t);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[b * 9 + i] = r[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[b * 3 + i] = t[i];
}

__global__ __launch_bounds__(64) void se3_exp_bwd_kernel(const float *__restrict__ xi, const float *__restrict__ gR,
                                                         const float *__restrict__ gT, float *__restrict__ gxi, int B) {
    const int t_ = blockIdx.x * 64 + threadIdx.x;
    const int b = t_ / 6, k = t_ % 6;
    if (b >= B) return;
    Dual x[6], r[9], t[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = {xi[b * 6 + i], i == k ? 1.0f : 0.0f};
    se3_exp3<Dual>(x, r, t);
    float g = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) g += (gR ? gR[b * 9 + i] : 0.0f) * r[i].d;
#pragma unroll
    for (int i = 0; i < 3; ++i) g += (gT ? gT[b * 3 + i] : 0.0f) * t[i].d;
    gxi[b * 6 + k] = g;
}

extern "C" int rrl_se3_exp(const float *xi, float *R, float *T, int B, void *stream) {
    if (!xi || !R || !T || B < 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(se3_exp_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, xi, R, T, B);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_se3_exp_bwd(const float *xi, const float *gR, const float *gT, float *gxi, int B, void *stream) {
    if (!xi || !gxi || B < 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(se3_exp_bwd_kernel, dim3((unsigned)((6 * B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, xi,
                       gR, gT, gxi, B);
    RRL_LAUNCH_CHECK();
    return 0;
}

// One parameter's torch.optim.Adam update (no weight decay, no amsgrad) as PyTorch's CPU path evaluates it
// (torch/optim/adam.py _single_tensor_adam of the PyTorch that generated the fixtures, 2.10: exp_avg.lerp_; 1.7.1, which
// the reference pins, wrote exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1) -- the same value to an ulp): the betas, 1 - beta and eps are Python doubles cast to fp32 where they
// meet the fp32 tensors (exp_avg.lerp_ / mul_ / addcmul_), the bias corrections and step_size = lr / (1 - beta1^t) are
// DOUBLE arithmetic on the host, denom = sqrt(v) / float(sqrt(1 - beta2^t)) + eps, p += float(-step_size) * m / denom.
// (Round 4: in fp32, 1 - 0.999^t carries a relative error of 3e-5 at small t and 1.0f - 0.999f differs from
// float(0.001) by 1.3e-5 -- enough to move the demo's xi by 3e-8 per epoch and, at the reference's data scale, to
// flip labels within three epochs; tests/golden/demo_trajectory_airplane.npz.)
// lb1 / lb2 = log2(b1), log2(b2), taken on the HOST in double: beta ** step = exp2(step * log2(beta)) -- a product and one
// double exp2 instead of two double pow calls (1.2 us of the single wavefront's 6.2 us in se3_adam_step_kernel); the
// result is within ~1e-15 of pow's (|step * log2 beta| <= 15 for beta = 0.999 over 10^4 steps), far inside the float
// rounding of step_size and sqrt(bias2) that follows.
__device__ __forceinline__ float adam_update(float p, float g, float &m, float &v, float step, float lr, double b1, double b2,
                                             double eps, double lb1, double lb2) {
    const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2);
    m = m + w1 * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2f + (g * g) * w2;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const double bias1 = 1.0 - exp2((double)step * lb1), bias2 = 1.0 - exp2((double)step * lb2);
    const float step_size = (float)((double)lr / bias1), bc2 = (float)sqrt(bias2);
    const float denom = sqrtf(v) / bc2 + (float)eps;
    return p + (-step_size * m) / denom;   // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// torch.optim.Adam's update (no weight decay, no amsgrad) with every scalar on the device, so a
// captured graph takes a new learning rate per replay and skips the update when gate[0] <= 0 (the
// demo's `if loss_di is not None`, gate = the loss's bucket count).  state = [step]; one lane per
// parameter; lane 0 advances the step AFTER all lanes of the (single) workgroup have read it.
__global__ __launch_bounds__(256) void adam_gated_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v,
                                                         float *__restrict__ state, const float *__restrict__ lr,
                                                         const int32_t *__restrict__ gate, int n, double b1, double b2,
                                                         double eps, double lb1, double lb2) {
    const int i = threadIdx.x;
    const bool ok = gate == nullptr || gate[0] > 0;
    const float step = state[0] + (ok ? 1.0f : 0.0f);
    __syncthreads();
    if (ok) {
        for (int q = i; q < n; q += 256) {
            float mi = m[q], vi = v[q];
            p[q] = adam_update(p[q], g[q], mi, vi, step, lr[0], b1, b2, eps, lb1, lb2);
            m[q] = mi;
            v[q] = vi;
        }
    }
    if (i == 0) state[0] = step;
}

extern "C" int rrl_adam_gated(float *p, const float *g, float *m, float *v, float *state, const float *lr,
                              const int32_t *gate, int n, double b1, double b2, double eps, void *stream) {
    if (!p || !g || !m || !v || !state || !lr || n < 0) return RRL_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_gated_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, g, m, v, state, lr, gate, n,
                       b1, b2, eps, log2(b1), log2(b2));
    RRL_LAUNCH_CHECK();
    return 0;
}

// The pose side of one demo epoch in ONE launch (a single pose, one wave): d loss / d xi from (dL/dR, dL/dT) by the
// dual-number exponential (== se3_exp_bwd_kernel), the gated Adam update of xi (== adam_gated_kernel, same
// expressions in the same order), the exponential of the UPDATED xi into (R, T) for the next epoch (== se3_exp_kernel)
// and the epoch's log row (== log_row_kernel).  As four launches these cost ~18 us of pure launch latency per epoch
// on 6 floats; results are bit-identical to the four (tests/test_gpu_harness.py).
__global__ __launch_bounds__(64) void se3_adam_step_kernel(float *__restrict__ xi, const float *__restrict__ gR,
                                                           const float *__restrict__ gT, float *__restrict__ m,
                                                           float *__restrict__ v, float *__restrict__ state,
                                                           const float *__restrict__ lr,
                                                           const int32_t *__restrict__ gate, double b1, double b2,
                                                           double eps, double lb1, double lb2, float *__restrict__ R,
                                                           float *__restrict__ T,
                                                           float *__restrict__ gxi, const float *__restrict__ loss,
                                                           const float *__restrict__ value, float *__restrict__ table,
                                                           long long *__restrict__ cursor, long long nrows,
                                                           float *__restrict__ row, const float *__restrict__ aabb_rows,
                                                           int n_aabb_rows, float *__restrict__ box) {
    const int k = threadIdx.x;
    const bool ok = gate == nullptr || gate[0] > 0;
    // (the log row's inputs are requested with the first loads, not after the arithmetic)
    long long at = -1;
    float q0 = 0.0f, q1 = 0.0f;
    if (k == 0 && table && cursor) {
        at = cursor[0];
        q0 = loss ? loss[0] : 0.0f;
        q1 = value ? value[0] : 0.0f;
    }
    // the moved source's AABB for the NEXT epoch's sampler (code/test_demo_optimized_Lie_Algebra.py:46-51 samples against
    // the previous epoch's moved source) from the per-workgroup partial rows the loss step's records launch just left
    // (APART: min xyz, max xyz of the moved first points) -- one launch (rigid apply + AABB) less per epoch
    if (box != nullptr && k >= 8 && k < 14) {
        const int c = k - 8;
        float v = c < 3 ? INFINITY : -INFINITY;
        for (int r = 0; r < n_aabb_rows; ++r) {
            const float q = aabb_rows[(size_t)r * 8 + c];
            v = c < 3 ? fminf(v, q) : fmaxf(v, q);
        }
        box[c] = v;
    }
    const float step = state[0] + (ok ? 1.0f : 0.0f);
    float pk = k < 6 ? xi[k] : 0.0f;
    if (k < 6) {
        Dual x[6], r[9], t[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = {xi[i], i == k ? 1.0f : 0.0f};
        se3_exp3<Dual>(x, r, t);
        float g = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; ++i) g += (gR ? gR[i] : 0.0f) * r[i].d;
#pragma unroll
        for (int i = 0; i < 3; ++i) g += (gT ? gT[i] : 0.0f) * t[i].d;
        if (gxi) gxi[k] = g;
        if (ok) {
            float mi = m[k], vi = v[k];
            pk = adam_update(pk, g, mi, vi, step, lr[0], b1, b2, eps, lb1, lb2);
            m[k] = mi;
            v[k] = vi;
            xi[k] = pk;
        }
    }
    float xn[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = __shfl(pk, i);
    if (k != 0) return;
    state[0] = step;
    float r[9], t[3];
    se3_exp3<float>(xn, r, t);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = r[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = t[i];
    if (table && cursor) {
        const float q[3] = {q0, q1, ok ? 1.0f : 0.0f};
        for (int c = 0; c < 3; ++c) {
            if (row) row[c] = q[c];
            if (at >= 0 && at < nrows) table[at * 3 + c] = q[c];
        }
        cursor[0] = at + 1;
    }
}

extern "C" int rrl_se3_adam_step(float *xi, const float *gR, const float *gT, float *m, float *v, float *state,
                                 const float *lr, const int32_t *gate, double b1, double b2, double eps, float *R,
                                 float *T, float *gxi, const float *loss, const float *value, float *table,
                                 long long *cursor, long long nrows, float *row, const float *aabb_rows, int n_aabb_rows,
                                 float *box, void *stream) {
    if (!xi || !m || !v || !state || !lr || !R || !T || (box && (!aabb_rows || n_aabb_rows <= 0))) return RRL_E_ARG;
    hipLaunchKernelGGL(se3_adam_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, xi, gR, gT, m, v, state, lr,
                       gate, b1, b2, eps, log2(b1), log2(b2), R, T, gxi, loss, value, table, cursor, nrows, row, aabb_rows,
                       n_aabb_rows, box);
    RRL_LAUNCH_CHECK();
    return 0;
}

// One row of the demo's scalar log inside a captured step: table[cursor[0]] = (loss[0], value[0],
// info[0] > 0), then cursor[0] += 1 -- as torch ops (cast, cat, index_copy_, add_) four launches.
__global__ void log_row_kernel(const float *__restrict__ loss, const float *__restrict__ value,
                               const int32_t *__restrict__ info, float *__restrict__ table,
                               long long *__restrict__ cursor, long long nrows, float *__restrict__ row) {
    if (threadIdx.x != 0) return;
    const long long at = cursor[0];
    const float r[3] = {loss[0], value[0], info[0] > 0 ? 1.0f : 0.0f};
    for (int c = 0; c < 3; ++c) {
        if (row) row[c] = r[c];
        if (at >= 0 && at < nrows) table[at * 3 + c] = r[c];
    }
    cursor[0] = at + 1;
}

extern "C" int rrl_log_row(const float *loss, const float *value, const int32_t *info, float *table,
                           long long *cursor, long long nrows, float *row, void *stream) {
    if (!loss || !value || !info || !table || !cursor || nrows < 0) return RRL_E_ARG;
    hipLaunchKernelGGL(log_row_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, value, info, table, cursor,
                       nrows, row);
    RRL_LAUNCH_CHECK();
    return 0;
}

// d mean / d x_i: every minimum contributes 2 (x_i - y_j) / (B (N + M)) to its two end points
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ y,
                                                          const unsigned long long *__restrict__ bx,
                                                          const unsigned long long *__restrict__ by,
                                                          const float *__restrict__ gval,
                                                          float *__restrict__ gx,
                                                          float *__restrict__ gy, int B, int N,
                                                          int M) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N + M) return;
    int i, j;
    if (t < N) { i = t; j = (int)(unsigned)bx[(size_t)b * N + i]; }
    else { j = t - N; i = (int)(unsigned)by[(size_t)b * M + j]; }
    const float sc = 2.0f * gval[0] / ((float)B * (float)(N + M));
    const float *xp = x + ((size_t)b * N + i) * 3, *yp = y + ((size_t)b * M + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float g = (xp[c] - yp[c]) * sc;
        if (gx) atomicAdd(&gx[((size_t)b * N + i) * 3 + c], g);
        if (gy) atomicAdd(&gy[((size_t)b * M + j) * 3 + c], -g);
    }
}

extern "C" int rrl_chamfer_bwd(const float *x, const float *y, const uint64_t *best_x,
                               const uint64_t *best_y, const float *grad_value, float *gx, float *gy,
                               int B, int N, int M, void *stream) {
    if (!x || !y || !best_x || !best_y || !grad_value || B < 0 || N < 0 || M < 0) return RRL_E_ARG;
    if (B == 0 || N + M == 0) return 0;
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3((unsigned)((N + M + 255) / 256), (unsigned)B), dim3(256),
                       0, (hipStream_t)stream, x, y, (const unsigned long long *)best_x,
                       (const unsigned long long *)best_y, grad_value, gx, gy, B, N, M);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K8 line sampler
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void aabb_kernel(const float *__restrict__ v,
                                                    float *__restrict__ aabb, int n) {
    __shared__ float red[16][6];
    const int b = blockIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float f = v[((size_t)b * n + i) * 3 + c];
            mn[c] = fminf(mn[c], f);
            mx[c] = fmaxf(mx[c], f);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_down(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_down(mx[c], o));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[threadIdx.x >> 6][c] = mn[c];
            red[threadIdx.x >> 6][3 + c] = mx[c];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = red[0][threadIdx.x];
        for (int w = 1; w < 16; ++w)
            r = threadIdx.x < 3 ? fminf(r, red[w][threadIdx.x]) : fmaxf(r, red[w][threadIdx.x]);
        aabb[b * 6 + threadIdx.x] = r;
    }
}

extern "C" int rrl_aabb(const float *v, float *aabb, int B, int n, void *stream) {
    if (!v || !aabb || B < 0 || n <= 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(aabb_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, v, aabb, n);
    RRL_LAUNCH_CHECK();
    return 0;
}

// y = x R + t AND the AABB of y, one 1024-lane workgroup per sample: the demo's epoch needs both (the moved
// cloud and, for the next epoch's line sampler, its box) and at its sizes (<= a few thousand points) each is
// launch latency; rrl_rigid_apply_fwd's arithmetic, aabb_kernel's reduction.
__global__ __launch_bounds__(1024) void rigid_aabb_kernel(const float *__restrict__ x, const float *__restrict__ R,
                                                          const float *__restrict__ t, float *__restrict__ y,
                                                          float *__restrict__ aabb, int n, int transpose_r) {
    __shared__ float red[16][6];
    const int b = blockIdx.x;
    const Mat m = load_mat(R, t, b, transpose_r);
    const size_t base = (size_t)b * n * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += 1024) {
        float p[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = x[base + 3 * (size_t)i + c];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float s = fmaf(p[2], m.r[6 + j], fmaf(p[1], m.r[3 + j], p[0] * m.r[j])) + m.t[j];
            y[base + 3 * (size_t)i + j] = s;
            mn[j] = fminf(mn[j], s);
            mx[j] = fmaxf(mx[j], s);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_down(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_down(mx[c], o));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[threadIdx.x >> 6][c] = mn[c];
            red[threadIdx.x >> 6][3 + c] = mx[c];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = red[0][threadIdx.x];
        for (int w = 1; w < 16; ++w)
            r = threadIdx.x < 3 ? fminf(r, red[w][threadIdx.x]) : fmaxf(r, red[w][threadIdx.x]);
        aabb[b * 6 + threadIdx.x] = r;
    }
}

extern "C" int rrl_rigid_apply_aabb(const float *x, const float *R, const float *t, float *y, float *aabb, int B,
                                    int n, int transpose_r, void *stream) {
    if (!x || !R || !t || !y || !aabb || B < 0 || n <= 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(rigid_aabb_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, x, R, t, y, aabb, n,
                       transpose_r);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// Dense form of the scan: the full (line, triangle) tables that the reference's
// cal_intersection_batch2_points_with_line returns (code/loss.py:68-112).  The loss never
// materialises them (rrl_scan/rrl_cull keep <= 4 hits per line); this entry point exists for
// callers of that public function and for diagnostics.  Lane = triangle, line = blockIdx.y.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_scan_kernel(const float *__restrict__ tri,
                                                         const float *__restrict__ line,
                                                         float *__restrict__ norm_d,
                                                         uint8_t *__restrict__ label,
                                                         int32_t *__restrict__ status, int N, int L) {
    const int b = blockIdx.z, l = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= N) return;
    const float *ln = line + 6 * ((size_t)b * L + l);
    const float *p = tri + 9 * ((size_t)b * N + f);
    float c[9], thr, thr2, d[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = p[i];
    tri_thresholds(c, &thr, &thr2);
    bool hit = true, bad = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = sqrtf(dist_sq<float>(c[3 * k], c[3 * k + 1], c[3 * k + 2], ln[0], ln[1], ln[2], ln[3],
                                    ln[4], ln[5]));
        hit = hit && (d[k] < thr);
        bad = bad || (d[k] != d[k]);
    }
    const float sum = (d[0] + d[1]) + d[2];
    float *o = norm_d + 3 * (((size_t)b * L + l) * N + f);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = d[k] / sum;
    label[((size_t)b * L + l) * N + f] = hit ? 1 : 0;
    if (bad) atomicOr(&status[0], 1);
}

extern "C" int rrl_dense_scan(const float *tri, const float *line, float *norm_d, uint8_t *label,
                              int32_t *status, int B, int N, int L, void *stream) {
    if (!tri || !line || !norm_d || !label || !status || B < 0 || N < 0 || L < 0) return RRL_E_ARG;
    if (L > 65535 || B > 65535) return RRL_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc0 = rrl_fill(status, 0u, sizeof(int32_t), s);
    if (rc0) return rc0;
    if (B == 0 || N == 0 || L == 0) return 0;
    hipLaunchKernelGGL(dense_scan_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)L, (unsigned)B),
                       dim3(256), 0, s, tri, line, norm_d, label, status, N, L);
    RRL_LAUNCH_CHECK();
    return 0;
}

#include "rrl_sampler.h"  // the sampler's device code: shared with the launch that carries the count pass (rrl_sparse.hip)

// The accept test on CALLER-SUPPLIED lines (include/rrl.h rrl_box_accept): the same face table, the same
// face_hit and the same slab pre-test as the sampler kernels below, one lane per line walking the 24
// faces.  It is the seam through which tests compare the decision with the CPU oracle and with
// the reference's own label1 / label2 on identical candidates, and a public helper for callers
// that bring their own lines.
__global__ __launch_bounds__(256) void box_accept_kernel(const float *__restrict__ lines,
                                                         const float *__restrict__ aabb1,
                                                         const float *__restrict__ aabb2,
                                                         uint8_t *__restrict__ mask, int32_t *__restrict__ hits,
                                                         int n) {
    __shared__ __attribute__((aligned(16))) float faces[24][FACE_FLOATS];
    const int b = blockIdx.y, tid = threadIdx.x;
    float bb1[6], bb2[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { bb1[c] = aabb1[b * 6 + c]; bb2[c] = aabb2[b * 6 + c]; }
    if (tid < 24) face_entry(tid < 12 ? bb1 : bb2, tid % 12, faces[tid]);
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= n) return;
    float ln[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) ln[c] = lines[((size_t)b * n + i) * 6 + c];
    int h1 = 0, h2 = 0;
    for (int f = 0; f < 12; ++f) {
        h1 += face_hit(faces[f], ln) ? 1 : 0;
        h2 += face_hit(faces[12 + f], ln) ? 1 : 0;
    }
    float inv[3];
    slab_inv(ln, inv);
    const bool slab = slab_maybe(bb1, ln, inv) && slab_maybe(bb2, ln, inv);
    mask[(size_t)b * n + i] = (uint8_t)((h1 > 0 ? 1 : 0) | (h2 > 0 ? 2 : 0) | (slab ? 4 : 0));
    if (hits) {
        hits[((size_t)b * n + i) * 2] = h1;
        hits[((size_t)b * n + i) * 2 + 1] = h2;
    }
}

extern "C" int rrl_box_accept(const float *lines, const float *aabb1, const float *aabb2, uint8_t *mask,
                              int32_t *hits, int B, int n, void *stream) {
    if (!lines || !aabb1 || !aabb2 || !mask || B < 0 || n < 0 || B > 65535) return RRL_E_ARG;
    if (B == 0 || n == 0) return 0;
    hipLaunchKernelGGL(box_accept_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, lines, aabb1, aabb2, mask, hits, n);
    RRL_LAUNCH_CHECK();
    return 0;
}

// The reference fills its (N, 6) buffer candidate by candidate, round by round, and skips a round
// once more than N candidates were accepted (code/loss.py:365-381): slot order == candidate
// order, overflow truncated, unfilled rows zero.  Two wide launches reproduce that order:
//   sample_count_kernel: evaluates the candidates of every (tile of 1024, round, sample) and stores
//     which are accepted (one 64-bit ballot per wavefront);
//   sample_write_kernel: every workgroup counts the ballots per tile, derives its base slot from the
//     tile counts (a walk over <= rounds x tiles integers, applying the skip rule), rebuilds the ACCEPTED candidates' lines
//     (no box test: the ballots say which) and writes them at base + rank (ballot prefix), then
//     zero-fills its share of the tail.
// (A single workgroup per sample walking everything in order took 1.9 ms for 10 x 10000
// candidates; this takes a few tens of microseconds.)
__global__ __launch_bounds__(1024) void sample_count_kernel(
    const float *__restrict__ rands, const unsigned long long *__restrict__ rng_state, const float *__restrict__ r,
    const float *__restrict__ centers,
    const float *__restrict__ aabb1, const float *__restrict__ aabb2,
    unsigned long long *__restrict__ accept, int B, int n, int rounds, int prefilter, int rd0) {
    __shared__ SampleCountLds lds_;
    sample_count_body(lds_, rands, rng_state, r, centers, aabb1, aabb2, accept, B, n, rounds, prefilter, rd0, (int)blockIdx.x,
                      (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x);
}

__global__ __launch_bounds__(1024) void sample_write_kernel(
    const float *__restrict__ rands, unsigned long long *__restrict__ rng_state, const float *__restrict__ r,
    const float *__restrict__ centers,
    const float *__restrict__ aabb1, const float *__restrict__ aabb2,
    const unsigned long long *__restrict__ accept, float *__restrict__ lines, int32_t *__restrict__ filled,
    int B, int n, int rounds) {
    extern __shared__ int s_tc[];  // this sample's tile counts [rounds][ntiles] from the ballots, then their exclusive prefix
    sample_write_body<1024>(s_tc, rands, rng_state, r, centers, aabb1, aabb2, accept, lines, filled, B, n, rounds, (int)blockIdx.x,
                            (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, gridDim.x * gridDim.y * gridDim.z);
}

int rrl_sample_prefilter(void) {
    static const int prefilter = [] { const char *e = getenv("RRL_SAMPLER_PREFILTER"); return e && e[0] == '0' ? 0 : 1; }();
    return prefilter;
}

// The two passes on their own (rrl_ws.h; rrl_demo_epoch pipelines them: the count pass of epoch k + 1 rides in epoch k's
// per-line launch).  Small calls only (< 512 workgroups: one count launch for all rounds, as sample_lines_impl issues it).
int rrl_sample_count_pass(const uint64_t *rng_state, const float *r, const float *centers, const float *aabb1, const float *aabb2,
                          int32_t *tile_counts, int B, int n, int rounds, void *stream) {
    if (!rng_state || !r || !centers || !tile_counts || B <= 0 || n <= 0 || rounds <= 0 || rounds > 65535 || B > 65535) return RRL_E_ARG;
    const dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)rounds, (unsigned)B);
    if ((size_t)grid.x * rounds * B >= 512 || ((uintptr_t)tile_counts & 7) != 0) return RRL_E_ARG;
    hipLaunchKernelGGL(sample_count_kernel, grid, dim3(1024), 0, (hipStream_t)stream, (const float *)nullptr,
                       (const unsigned long long *)rng_state, r, centers, aabb1, aabb2, (unsigned long long *)tile_counts, B, n,
                       rounds, rrl_sample_prefilter(), 0);
    RRL_LAUNCH_CHECK();
    return 0;
}
int rrl_sample_write_pass(uint64_t *rng_state, const float *r, const float *centers, float *lines, int32_t *filled,
                          int32_t *tile_counts, int B, int n, int rounds, void *stream) {
    if (!rng_state || !r || !centers || !lines || !filled || !tile_counts || B <= 0 || n <= 0 || rounds <= 0) return RRL_E_ARG;
    const dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)rounds, (unsigned)B);
    const size_t lds = sizeof(int32_t) * (size_t)rounds * grid.x;
    if (lds > 96 * 1024) return RRL_E_ARG;
    hipLaunchKernelGGL(sample_write_kernel, grid, dim3(1024), lds, (hipStream_t)stream, (const float *)nullptr,
                       (unsigned long long *)rng_state, r, centers, (const float *)nullptr, (const float *)nullptr,
                       (const unsigned long long *)tile_counts, lines, filled, B, n, rounds);
    RRL_LAUNCH_CHECK();
    return 0;
}

static int sample_lines_impl(const float *rands, unsigned long long *rng_state, const float *r, const float *centers,
                             const float *aabb1, const float *aabb2, float *lines, int32_t *filled,
                             int32_t *tile_counts, int B, int n, int rounds, void *stream) {
    if ((!rands && !rng_state) || !r || !centers || !lines || !filled || !tile_counts) return RRL_E_ARG;
    if (B < 0 || n < 0 || rounds < 0 || rounds > 65535 || B > 65535) return RRL_E_ARG;
    if (B == 0 || n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (rounds == 0) {
        int rc = rrl_fill(lines, 0u, sizeof(float) * 6 * (size_t)B * n, s);
        return rc ? rc : rrl_fill(filled, 0u, sizeof(int32_t) * (size_t)B, s);
    }
    const dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)rounds, (unsigned)B);
    // scratch: one 64-bit accept ballot per wavefront of every tile (16 per tile)
    unsigned long long *accept = (unsigned long long *)tile_counts;
    if (((uintptr_t)accept & 7) != 0) return RRL_E_ARG;
    const int prefilter = rrl_sample_prefilter();
    // Big calls (>= 512 workgroups: the trainers' B x 10 rounds x 10 tiles): rounds [0, 3) first, the
    // rest as a second launch that starts by checking whether they are skipped (92 -> ~30 us at B=8,
    // 10 x 10000 candidates, radius = half the box diagonal).  Small calls (the demo: 200 workgroups,
    // never full) stay one launch: two latency-bound launches cost it 10 us of 15.
    const int first = (rounds < 3 || (size_t)grid.x * rounds * B < 512) ? rounds : 3;
    hipLaunchKernelGGL(sample_count_kernel, dim3(grid.x, (unsigned)first, grid.z), dim3(1024), 0, s, rands, rng_state, r,
                       centers, aabb1, aabb2, accept, B, n, rounds, prefilter, 0);
    if (rounds > first)
        hipLaunchKernelGGL(sample_count_kernel, dim3(grid.x, (unsigned)(rounds - first), grid.z), dim3(1024), 0, s, rands,
                           rng_state, r, centers, aabb1, aabb2, accept, B, n, rounds, prefilter, first);
    const size_t lds = sizeof(int32_t) * (size_t)rounds * grid.x;
    if (lds > 96 * 1024) return RRL_E_ARG;  // > 24576 tiles x rounds: far beyond any caller
    hipLaunchKernelGGL(sample_write_kernel, grid, dim3(1024), lds, s, rands, rng_state, r, centers, aabb1, aabb2,
                       accept, lines, filled, B, n, rounds);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_sample_lines(const float *rands, const float *r, const float *centers,
                                const float *aabb1, const float *aabb2, float *lines,
                                int32_t *filled, int32_t *tile_counts, int B, int n, int rounds,
                                void *stream) {
    if (!rands) return RRL_E_ARG;
    return sample_lines_impl(rands, nullptr, r, centers, aabb1, aabb2, lines, filled, tile_counts, B, n, rounds, stream);
}

extern "C" int rrl_sample_lines_rng(uint64_t *rng_state, const float *r, const float *centers,
                                    const float *aabb1, const float *aabb2, float *lines,
                                    int32_t *filled, int32_t *tile_counts, int B, int n, int rounds,
                                    void *stream) {
    if (!rng_state) return RRL_E_ARG;
    return sample_lines_impl(nullptr, (unsigned long long *)rng_state, r, centers, aabb1, aabb2, lines, filled, tile_counts,
                             B, n, rounds, stream);
}

// ---------------------------------------------------------------------------------------
// batch-shard payload: what one rank contributes to the all-reduce
// ---------------------------------------------------------------------------------------
// 14 wavefronts, one per payload element: the lanes stride over the samples (a fixed order per lane) and meet in a
// fixed shuffle tree -- double sums of <= a few thousand floats: exact, so the result does not depend on the association.
// (Round 5: the 14 lanes of ONE wavefront looping over all samples took 16 us at B = 64.)
__global__ __launch_bounds__(14 * 64) void shard_payload_kernel(const float *__restrict__ loss, const int32_t *__restrict__ info,
                                                               const float *__restrict__ gR, const float *__restrict__ gt,
                                                               float *__restrict__ out, int B) {
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0.0;
    for (int b = lane; b < B; b += 64) {
        if (q == 0) s += info[b * 4] > 0 ? (double)loss[b] : 0.0;
        else if (q == 1) s += info[b * 4] > 0 ? 1.0 : 0.0;
        else if (q < 11) s += gR ? (double)gR[b * 9 + (q - 2)] : 0.0;
        else s += gt ? (double)gt[b * 3 + (q - 11)] : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) out[q] = (float)s;
}

extern "C" int rrl_shard_payload(const float *loss, const void *ws, size_t ws_bytes, const float *gR,
                                 const float *gt, float *out, int B, int N, int M, int L,
                                 void *stream) {
    if (!loss || !ws || !out || B < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    hipLaunchKernelGGL(shard_payload_kernel, dim3(1), dim3(14 * 64), 0, (hipStream_t)stream, loss,
                       w.i32(ws, RRL_WS_INFO), gR, gt, out, B);
    RRL_LAUNCH_CHECK();
    return 0;
}

#ifndef RRL_BUILD_FLAGS
#define RRL_BUILD_FLAGS ""
#endif
// experimental builds (RRL_HIPCC_FLAGS, rrl_hip/build.py) carry their flags in the version string
extern "C" const char *rrl_version(void) {
    return sizeof(RRL_BUILD_FLAGS) > 1 ? "rrl_hip 0.6 (gfx950) [" RRL_BUILD_FLAGS "]" : "rrl_hip 0.6 (gfx950)";
}

// ---------------------------------------------------------------------------------------
// Test hook (include/rrl.h rrl_debug_occupy): a filler launch that HOLDS compute-unit slots -- `workgroups` x `lanes`
// threads spin until the 100 MHz wall clock has advanced by `ticks` -- so that tests can run the library's in-launch
// hand-offs (the exchange reduce's bounded spins) while another stream or process occupies most of the device.
// ---------------------------------------------------------------------------------------
__global__ void occupy_kernel(long long ticks, unsigned *sink) {
    const long long t0 = (long long)wall_clock64();
    unsigned spins = 0;
    while ((long long)wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
    }
    if (sink && spins == 0xffffffffu) sink[0] = spins;  // (keeps the loop observable)
}
extern "C" int rrl_debug_occupy(int workgroups, int lanes, long long ticks, void *stream) {
    if (workgroups <= 0 || lanes <= 0 || lanes > 1024 || ticks < 0 || ticks > 100000000ll * 10) return RRL_E_ARG;  // <= 10 s
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)workgroups), dim3((unsigned)lanes), 0, (hipStream_t)stream, ticks, (unsigned *)nullptr);
    RRL_LAUNCH_CHECK();
    return 0;
}

