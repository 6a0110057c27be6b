// rrl_sparse.hip -- K2..K5: everything after the dense scan.
//   K2 line_pair_dist   code/loss.py:115-167   (one lane per line, <= 4x4 block in registers)
//   K3 lower_median     code/loss.py:223-224   (radix select, one workgroup per sample)
//   K4 welsch_reduce    code/loss.py:20-21, 226-230
//   K5 backward         autograd of code/loss.py:170-232
// About 9 % of the lines are selected; these kernels touch O(L) data and are launch/latency
// bound (a few microseconds each) next to the O(L*(N+M)) scan.
#include "rrl_common.h"

#define FIX_SHIFT 40  // bucket sums in 2^-40 fixed point: order-independent, bit-deterministic

// sqrt(dist_sq) of the three points of triangle f and the detached weights of
// code/loss.py:92: w_k = d_k / ((d0 + d1) + d2).  Same arithmetic as the scan, so the
// distances are bit-identical to the ones that decided the label.
__device__ __forceinline__ void hit_weights(const float *__restrict__ tri, int f,
                                            const float *ln, float *w) {
    const float *p = tri + 9 * (size_t)f;
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        d[k] = sqrtf(dist_sq<float>(p[3 * k], p[3 * k + 1], p[3 * k + 2], ln[0], ln[1], ln[2],
                                    ln[3], ln[4], ln[5]));
    float s = (d[0] + d[1]) + d[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = d[k] / s;
}

// q = mean_k(w_k * P_k), code/loss.py:155-163 (a mean: 1/3 of the convex combination)
__device__ __forceinline__ void inter_point(const float *__restrict__ tri, int f, const float *w,
                                            float *q) {
    const float *p = tri + 9 * (size_t)f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = w[0] * p[c];
        s = s + w[1] * p[3 + c];
        s = s + w[2] * p[6 + c];
        q[c] = s / 3.0f;
    }
}

__device__ __forceinline__ void sort4(int *h, int n) {  // ascending, n <= 4
#pragma unroll
    for (int i = 1; i < RRL_MAX_HITS; ++i)
#pragma unroll
        for (int j = RRL_MAX_HITS - 1; j >= i; --j)
            if (j < n && h[j] < h[j - 1]) { int t = h[j]; h[j] = h[j - 1]; h[j - 1] = t; }
}

__global__ __launch_bounds__(256) void line_pair_dist_kernel(
    const float *__restrict__ tri1, const float *__restrict__ tri2, const float *__restrict__ line,
    const int32_t *__restrict__ count1, const int32_t *__restrict__ hit1,
    const int32_t *__restrict__ count2, const int32_t *__restrict__ hit2,
    uint8_t *__restrict__ kj, int32_t *__restrict__ hs1, int32_t *__restrict__ hs2,
    float *__restrict__ w1, float *__restrict__ w2, float *__restrict__ D,
    int32_t *__restrict__ bcnt, int B, int N, int M, int L, int s_m, int s_n, int e_m, int e_n,
    int pool) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const size_t gl = (size_t)b * L + l;
    const int k = count1[gl], j = count2[gl];
    const bool sel = k >= s_m && k < e_m && j >= s_n && j < e_n;
    kj[gl] = sel ? (uint8_t)(k | (j << 4)) : (uint8_t)0;
    if (!sel) return;
    float ln[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) ln[c] = line[gl * 6 + c];
    const float *t1 = tri1 + (size_t)b * N * 9, *t2 = tri2 + (size_t)b * M * 9;
    int h1[RRL_MAX_HITS], h2[RRL_MAX_HITS];
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
        h1[a] = a < k ? hit1[gl * RRL_MAX_HITS + a] : 0x7fffffff;
        h2[a] = a < j ? hit2[gl * RRL_MAX_HITS + a] : 0x7fffffff;
    }
    sort4(h1, k);  // ascending triangle index == nonzero() order (code/loss.py:125-131)
    sort4(h2, j);
    float q1[RRL_MAX_HITS][3], q2[RRL_MAX_HITS][3];
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
        if (a < k) {
            float w[3];
            hit_weights(t1, h1[a], ln, w);
            inter_point(t1, h1[a], w, q1[a]);
            hs1[gl * RRL_MAX_HITS + a] = h1[a];
#pragma unroll
            for (int c = 0; c < 3; ++c) w1[(gl * RRL_MAX_HITS + a) * 3 + c] = w[c];
        }
        if (a < j) {
            float w[3];
            hit_weights(t2, h2[a], ln, w);
            inter_point(t2, h2[a], w, q2[a]);
            hs2[gl * RRL_MAX_HITS + a] = h2[a];
#pragma unroll
            for (int c = 0; c < 3; ++c) w2[(gl * RRL_MAX_HITS + a) * 3 + c] = w[c];
        }
    }
    // D[a][b] = sum_c (q1 - q2)^2, code/loss.py:38-52
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb)
            if (a < k && bb < j) {
                float dx = q1[a][0] - q2[bb][0], dy = q1[a][1] - q2[bb][1],
                      dz = q1[a][2] - q2[bb][2];
                float s = dx * dx;
                s = s + dy * dy;
                s = s + dz * dz;
                D[gl * 16 + a * j + bb] = s;
            }
    atomicAdd(&bcnt[(pool ? 0 : b) * 16 + (k - 1) * 4 + (j - 1)], 1);
}

extern "C" int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line,
                                  const int32_t *count1, const int32_t *hit1,
                                  const int32_t *count2, const int32_t *hit2, uint8_t *kj,
                                  int32_t *hs1, int32_t *hs2, float *w1, float *w2, float *D,
                                  int32_t *bcnt, int B, int N, int M, int L, int s_m, int s_n,
                                  int e_m, int e_n, int pool, void *stream) {
    if (!tri1 || !tri2 || !line || !count1 || !hit1 || !count2 || !hit2 || !kj || !hs1 || !hs2 ||
        !w1 || !w2 || !D || !bcnt)
        return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    if (B == 0 || L == 0) return 0;
    hipLaunchKernelGGL(line_pair_dist_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream, tri1, tri2, line, count1, hit1, count2,
                       hit2, kj, hs1, hs2, w1, w2, D, bcnt, B, N, M, L, s_m, s_n, e_m, e_n, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K3 lower median: 4-pass MSB-first radix select on the bit patterns (D >= 0, so unsigned
// order == float order).  One 1024-lane workgroup per sample.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lower_median_kernel(const uint8_t *__restrict__ kj,
                                                            const float *__restrict__ D,
                                                            float *__restrict__ med,
                                                            int32_t *__restrict__ nval, int B,
                                                            int L, int pool) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_total;
    const int g = blockIdx.x;
    const int b = pool ? B - 1 : g;  // reference B>1 quirk: the last sample's median (Q2)
    const uint8_t *kjb = kj + (size_t)b * L;
    const float *Db = D + (size_t)b * L * 16;
    const int tid = threadIdx.x;
    // total number of values
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    unsigned mine = 0;
    for (int l = tid; l < L; l += 1024) {
        unsigned c = kjb[l];
        mine += (c & 15u) * (c >> 4);
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((tid & 63) == 0) atomicAdd(&hist[0], mine);
    __syncthreads();
    if (tid == 0) {
        s_total = hist[0];
        s_rank = hist[0] ? (hist[0] - 1) / 2 : 0;  // torch.median: sorted[(n-1)/2]
        s_prefix = 0;
    }
    __syncthreads();
    const unsigned total = s_total;
    if (total == 0) {
        if (tid == 0) { med[g] = 0.0f; nval[g] = 0; }
        return;
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const unsigned himask = pass ? (0xffffffffu << (shift + 8)) : 0u;
        const unsigned prefix = s_prefix;
        __syncthreads();
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int l = tid; l < L; l += 1024) {
            unsigned c = kjb[l];
            int nv = (int)((c & 15u) * (c >> 4));
            for (int i = 0; i < nv; ++i) {
                unsigned u = __float_as_uint(Db[(size_t)l * 16 + i]);
                if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned r = s_rank, acc = 0;
            int bin = 0;
            for (; bin < 256; ++bin) {
                if (acc + hist[bin] > r) break;
                acc += hist[bin];
            }
            s_rank = r - acc;
            s_prefix = prefix | ((unsigned)bin << shift);
        }
        __syncthreads();
    }
    if (tid == 0) {
        med[g] = __uint_as_float(s_prefix);
        nval[g] = (int32_t)total;
    }
}

extern "C" int rrl_lower_median(const uint8_t *kj, const float *D, float *med, int32_t *nval,
                                int B, int L, int pool, void *stream) {
    if (!kj || !D || !med || !nval || B < 0 || L < 0) return RRL_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(lower_median_kernel, dim3((unsigned)(pool ? 1 : B)), dim3(1024), 0,
                       (hipStream_t)stream, kj, D, med, nval, B, L, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K4 Welsch + symmetric min/mean
// ---------------------------------------------------------------------------------------
// Welsch1(x, c) = 1 - exp(-(x / c) / 2), code/loss.py:20-21
__device__ __forceinline__ float welsch(float d, float med) {
    return 1.0f - expf(-(d / med) / 2.0f);
}

// Row/column minima of the k x j Welsch block with first-occurrence argmin (torch.min,
// SURVEY.md Q11).  Wl[a*4+b].
__device__ __forceinline__ void welsch_block(const float *__restrict__ Dl, int k, int j,
                                             float med, float *Wl, int *arg_b, int *arg_a) {
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
#pragma unroll
        for (int b = 0; b < RRL_MAX_HITS; ++b)
            Wl[a * 4 + b] = (a < k && b < j) ? welsch(Dl[a * j + b], med) : INFINITY;
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
        int m = 0;
#pragma unroll
        for (int b = 1; b < RRL_MAX_HITS; ++b)
            if (Wl[a * 4 + b] < Wl[a * 4 + m]) m = b;
        arg_b[a] = m;
    }
#pragma unroll
    for (int b = 0; b < RRL_MAX_HITS; ++b) {
        int m = 0;
#pragma unroll
        for (int a = 1; a < RRL_MAX_HITS; ++a)
            if (Wl[a * 4 + b] < Wl[m * 4 + b]) m = a;
        arg_a[b] = m;
    }
}

__global__ __launch_bounds__(256) void welsch_fwd_kernel(const uint8_t *__restrict__ kj,
                                                         const float *__restrict__ D,
                                                         const float *__restrict__ med,
                                                         int64_t *__restrict__ bsum, int B, int L,
                                                         int pool) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const size_t gl = (size_t)b * L + l;
    const unsigned c = kj[gl];
    if (!c) return;
    const int k = c & 15, j = c >> 4, g = pool ? 0 : b;
    float Wl[16];
    int arg_b[4], arg_a[4];
    welsch_block(D + gl * 16, k, j, med[g], Wl, arg_b, arg_a);
    float row = 0.0f, col = 0.0f;
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
        if (a < k) row += Wl[a * 4 + arg_b[a]];
#pragma unroll
    for (int bb = 0; bb < RRL_MAX_HITS; ++bb)
        if (bb < j) col += Wl[arg_a[bb] * 4 + bb];
    // Wl in [0,1]; <= 4 terms; 2^-40 fixed point keeps 2^-38 relative resolution per line
    const int bi = (k - 1) * 4 + (j - 1);
    long long fr = (long long)((double)row * (double)(1ll << FIX_SHIFT) + 0.5);
    long long fc = (long long)((double)col * (double)(1ll << FIX_SHIFT) + 0.5);
    atomicAdd((unsigned long long *)&bsum[((size_t)g * 16 + bi) * 2 + 0], (unsigned long long)fr);
    atomicAdd((unsigned long long *)&bsum[((size_t)g * 16 + bi) * 2 + 1], (unsigned long long)fc);
}

extern "C" int rrl_welsch_reduce_fwd(const uint8_t *kj, const float *D, const float *med,
                                     int64_t *bsum, int B, int L, int pool, void *stream) {
    if (!kj || !D || !med || !bsum || B < 0 || L < 0) return RRL_E_ARG;
    if (B == 0 || L == 0) return 0;
    hipLaunchKernelGGL(welsch_fwd_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream, kj, D, med, bsum, B, L, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// loss = ( sum_{non-empty (k,j), k-major} exp(-0.5|k-j|) * (mean_row + mean_col) ) / C
// code/loss.py:215-217, 226-230
__global__ void loss_finalize_kernel(const int64_t *__restrict__ bsum,
                                     const int32_t *__restrict__ bcnt, float *__restrict__ loss,
                                     int32_t *__restrict__ nbuckets, int G, int s_m, int s_n,
                                     int e_m, int e_n) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float acc = 0.0f;
    int C = 0;
    for (int k = s_m; k < e_m; ++k)
        for (int j = s_n; j < e_n; ++j) {
            const int bi = (k - 1) * 4 + (j - 1);
            const int S = bcnt[g * 16 + bi];
            if (S == 0) continue;
            const double sc = 1.0 / (double)(1ll << FIX_SHIFT);
            float mrow = (float)((double)bsum[((size_t)g * 16 + bi) * 2 + 0] * sc / ((double)S * k));
            float mcol = (float)((double)bsum[((size_t)g * 16 + bi) * 2 + 1] * sc / ((double)S * j));
            float wkj = expf(-0.5f * (float)abs(k - j));
            acc = acc + wkj * (mrow + mcol);
            ++C;
        }
    nbuckets[g] = C;
    loss[g] = C ? acc / (float)C : 0.0f;
}

extern "C" int rrl_loss_finalize(const int64_t *bsum, const int32_t *bcnt, float *loss,
                                 int32_t *nbuckets, int G, int s_m, int s_n, int e_m, int e_n,
                                 void *stream) {
    if (!bsum || !bcnt || !loss || !nbuckets || G < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    if (G == 0) return 0;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, bsum, bcnt, loss, nbuckets, G, s_m, s_n, e_m, e_n);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K5 backward.  dL/dD[a][b] = gout * w_kj / C * exp(-D/(2 med)) / (2 med)
//                              * ( [b = argmin_b(a)] / (S k) + [a = argmin_a(b)] / (S j) )
// dL/dq1[a] = sum_b 2 (q1_a - q2_b) dL/dD;  dL/dP1[f_a][kk] += w_kk / 3 * dL/dq1[a];
// weights, median and labels carry no gradient (code/loss.py:112, 224).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void welsch_bwd_kernel(
    const float *__restrict__ tri1, const float *__restrict__ tri2, const uint8_t *__restrict__ kj,
    const int32_t *__restrict__ hs1, const int32_t *__restrict__ hs2, const float *__restrict__ w1,
    const float *__restrict__ w2, const float *__restrict__ D, const float *__restrict__ med,
    const int32_t *__restrict__ bcnt, const int32_t *__restrict__ nbuckets,
    const float *__restrict__ grad_loss, float *__restrict__ g1, float *__restrict__ g2, int B,
    int N, int M, int L, int pool) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const size_t gl = (size_t)b * L + l;
    const unsigned c = kj[gl];
    if (!c) return;
    const int k = c & 15, j = c >> 4, g = pool ? 0 : b;
    const int C = nbuckets[g];
    if (C == 0) return;
    const float m = med[g];
    float Wl[16];
    int arg_b[4], arg_a[4];
    welsch_block(D + gl * 16, k, j, m, Wl, arg_b, arg_a);
    const int S = bcnt[g * 16 + (k - 1) * 4 + (j - 1)];
    const float wkj = expf(-0.5f * (float)abs(k - j));
    const float scale = grad_loss[g] * wkj / (float)C;
    const float inv_row = 1.0f / ((float)S * (float)k), inv_col = 1.0f / ((float)S * (float)j);
    const float *t1 = tri1 + (size_t)b * N * 9, *t2 = tri2 + (size_t)b * M * 9;
    float q1[RRL_MAX_HITS][3], q2[RRL_MAX_HITS][3], gq1[RRL_MAX_HITS][3], gq2[RRL_MAX_HITS][3];
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) gq1[a][cc] = gq2[a][cc] = 0.0f;
        if (a < k) inter_point(t1, hs1[gl * 4 + a], w1 + (gl * 4 + a) * 3, q1[a]);
        if (a < j) inter_point(t2, hs2[gl * 4 + a], w2 + (gl * 4 + a) * 3, q2[a]);
    }
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb)
            if (a < k && bb < j) {
                float sel = (arg_b[a] == bb ? inv_row : 0.0f) + (arg_a[bb] == a ? inv_col : 0.0f);
                if (sel != 0.0f) {
                    // dWl/dD = exp(-D/(2 med)) / (2 med) = (1 - Wl) / (2 med)
                    float gD = scale * sel * expf(-(D[gl * 16 + a * j + bb] / m) / 2.0f) / (2.0f * m);
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        float t = 2.0f * (q1[a][cc] - q2[bb][cc]) * gD;
                        gq1[a][cc] += t;
                        gq2[bb][cc] -= t;
                    }
                }
            }
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
        if (a < k) {
            float *dst = g1 + ((size_t)b * N + hs1[gl * 4 + a]) * 9;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                float wk = w1[(gl * 4 + a) * 3 + kk] / 3.0f;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) atomicAdd(dst + 3 * kk + cc, wk * gq1[a][cc]);
            }
        }
        if (g2 && a < j) {
            float *dst = g2 + ((size_t)b * M + hs2[gl * 4 + a]) * 9;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                float wk = w2[(gl * 4 + a) * 3 + kk] / 3.0f;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) atomicAdd(dst + 3 * kk + cc, wk * gq2[a][cc]);
            }
        }
    }
}

extern "C" int rrl_welsch_reduce_bwd(const float *tri1, const float *tri2, const uint8_t *kj,
                                     const int32_t *hs1, const int32_t *hs2, const float *w1,
                                     const float *w2, const float *D, const float *med,
                                     const int32_t *bcnt, const int32_t *nbuckets,
                                     const float *grad_loss, float *grad_tri1, float *grad_tri2,
                                     int B, int N, int M, int L, int pool, void *stream) {
    if (!tri1 || !tri2 || !kj || !hs1 || !hs2 || !w1 || !w2 || !D || !med || !bcnt || !nbuckets ||
        !grad_loss || !grad_tri1)
        return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (B == 0 || L == 0) return 0;
    hipLaunchKernelGGL(welsch_bwd_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream, tri1, tri2, kj, hs1, hs2, w1, w2, D, med,
                       bcnt, nbuckets, grad_loss, grad_tri1, grad_tri2, B, N, M, L, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}
