// rrl_sparse.hip -- everything after the dense scan, plus the fused forward/backward entries.
//   K2 line_pair_dist   code/loss.py:115-167   (one lane per line, <= 4x4 block in registers)
//   K3+K4 loss_reduce   code/loss.py:223-230   (radix-select median + Welsch min/mean, one
//                                               workgroup per sample, fixed-point bucket sums)
//   K5 backward         autograd of code/loss.py:170-232
// About 9 % of the lines are selected; these kernels touch O(L) data and are latency bound
// (a few microseconds each) next to the O(L*(N+M)) scan.
#include "rrl_ws.h"

#define FIX_SHIFT 40  // bucket sums in 2^-40 fixed point: order-independent, bit-deterministic

// sqrt(dist_sq) of the three points of triangle f and the detached weights of
// code/loss.py:92: w_k = d_k / ((d0 + d1) + d2).  Same arithmetic as the scan, so the
// distances are bit-identical to the ones that decided the label.
__device__ __forceinline__ void hit_weights(const float *__restrict__ tri, int f, const float *ln,
                                            float *w) {
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

// 1024 lines per workgroup.  Phase 1: every lane classifies its line and the selected ones
// (~9 %) are compacted through LDS, so that phase 2 -- the gather-heavy part -- runs on dense
// wavefronts; the compacted line ids also go to SEL[b] for the reduce and backward kernels.
__global__ __launch_bounds__(1024) void line_pair_dist_kernel(
    const float *__restrict__ tri1, const float *__restrict__ tri2, const float *__restrict__ line,
    const int32_t *__restrict__ count1, const int32_t *__restrict__ hit1,
    const int32_t *__restrict__ count2, const int32_t *__restrict__ hit2,
    uint8_t *__restrict__ kj, int32_t *__restrict__ sel_out, int32_t *__restrict__ nsel,
    int32_t *__restrict__ hs1, int32_t *__restrict__ hs2, float *__restrict__ w1,
    float *__restrict__ w2, float *__restrict__ D, float *__restrict__ vals,
    int32_t *__restrict__ nvals, int B, int N, int M, int L, int s_m, int s_n, int e_m, int e_n,
    int pool) {
    __shared__ int s_list[1024];
    __shared__ int s_wave[16];
    __shared__ int s_total, s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    {
        const int l = blockIdx.x * 1024 + tid;
        bool sel = false;
        if (l < L) {
            const size_t gl = (size_t)b * L + l;
            const int k = count1[gl], j = count2[gl];
            sel = k >= s_m && k < e_m && j >= s_n && j < e_n;
            kj[gl] = sel ? (uint8_t)(k | (j << 4)) : (uint8_t)0;
        }
        const unsigned long long mask = __ballot(sel);
        if (lane == 0) s_wave[wave] = __popcll(mask);
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int w = 0; w < 16; ++w) { int c = s_wave[w]; s_wave[w] = acc; acc += c; }
            s_total = acc;
            s_base = acc ? atomicAdd(&nsel[b], acc) : 0;
        }
        __syncthreads();
        if (sel) s_list[s_wave[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = l;
        __syncthreads();
    }
    if (tid >= s_total) return;
    const int l = s_list[tid];
    sel_out[(size_t)b * L + s_base + tid] = l;
    const size_t gl = (size_t)b * L + l;
    const int k = count1[gl], j = count2[gl];
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
    // the median's input: all of this sample's D values, any order (reference B>1 quirk:
    // only the LAST sample's values define the median)
    const bool feeds_median = !pool || b == B - 1;
    int pos = feeds_median ? atomicAdd(&nvals[b], k * j) : 0;
    float *vb = vals + (size_t)b * L * 16;
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
                if (feeds_median) vb[pos + a * j + bb] = s;
            }
}

extern "C" int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line,
                                  void *ws, size_t ws_bytes, int B, int N, int M, int L, int s_m,
                                  int s_n, int e_m, int e_n, int pool, void *stream) {
    if (!tri1 || !tri2 || !line || !ws || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0 || L == 0) return 0;
    hipLaunchKernelGGL(line_pair_dist_kernel, dim3((unsigned)((L + 1023) / 1024), (unsigned)B),
                       dim3(1024), 0, (hipStream_t)stream, tri1, tri2, line,
                       w.i32(ws, RRL_WS_COUNT1), w.i32(ws, RRL_WS_HIT1), w.i32(ws, RRL_WS_COUNT2),
                       w.i32(ws, RRL_WS_HIT2), w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL),
                       w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), w.i32(ws, RRL_WS_HS2),
                       w.f32(ws, RRL_WS_W1), w.f32(ws, RRL_WS_W2), w.f32(ws, RRL_WS_D),
                       w.f32(ws, RRL_WS_VALS), w.i32(ws, RRL_WS_NVALS), B, N, M, L, s_m, s_n, e_m,
                       e_n, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K3+K4: one 1024-lane workgroup per sample
// ---------------------------------------------------------------------------------------
// Welsch1(x, c) = 1 - exp(-(x / c) / 2), code/loss.py:20-21
__device__ __forceinline__ float welsch(float d, float med) {
    return 1.0f - expf(-(d / med) / 2.0f);
}

// Row/column minima of the k x j Welsch block with first-occurrence argmin (torch.min,
// SURVEY.md Q11).  All indices static after unrolling (no scratch).
__device__ __forceinline__ void welsch_block(const float *__restrict__ Dl, int k, int j, float med,
                                             float *rowmin, float *colmin, int *arg_b, int *arg_a) {
    float Wl[16];
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
#pragma unroll
        for (int b = 0; b < RRL_MAX_HITS; ++b)
            Wl[a * 4 + b] = (a < k && b < j) ? welsch(Dl[a * j + b], med) : INFINITY;
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a) {
        float best = Wl[a * 4];
        int m = 0;
#pragma unroll
        for (int b = 1; b < RRL_MAX_HITS; ++b)
            if (Wl[a * 4 + b] < best) { best = Wl[a * 4 + b]; m = b; }
        rowmin[a] = best;
        arg_b[a] = m;
    }
#pragma unroll
    for (int b = 0; b < RRL_MAX_HITS; ++b) {
        float best = Wl[b];
        int m = 0;
#pragma unroll
        for (int a = 1; a < RRL_MAX_HITS; ++a)
            if (Wl[a * 4 + b] < best) { best = Wl[a * 4 + b]; m = a; }
        colmin[b] = best;
        arg_a[b] = m;
    }
}

#define MED_REGS 8  // values cached in registers per lane (n <= 8192); the rest is re-read

__global__ __launch_bounds__(1024) void loss_reduce_kernel(
    const uint8_t *__restrict__ kj, const int32_t *__restrict__ sel, const int32_t *__restrict__ nsel,
    const float *__restrict__ D, const float *__restrict__ vals, const int32_t *__restrict__ nvals,
    float *__restrict__ med_out, int32_t *__restrict__ bcnt_out, int64_t *__restrict__ bsum_out,
    int32_t *__restrict__ info, float *__restrict__ loss, int B, int L, int s_m, int s_n, int e_m,
    int e_n, int pool) {
    __shared__ unsigned s_bit[32];
    __shared__ unsigned long long s_sum[32];
    __shared__ int s_cnt[16];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int bm = pool ? B - 1 : g;  // whose values define the median
    const float *v = vals + (size_t)bm * L * 16;
    const unsigned n = (unsigned)nvals[bm];
    if (tid < 32) { s_bit[tid] = 0; s_sum[tid] = 0ull; }
    if (tid < 16) s_cnt[tid] = 0;
    __syncthreads();

    // ---- lower median = element of rank (n-1)/2 (torch.median).  Bitwise MSB-first select on
    //      the bit patterns (D >= 0: unsigned order == float order).  Per bit: count the values
    //      that agree with the prefix and have the bit clear; counting is ballot/shuffle based,
    //      one LDS atomic per wave and ONE barrier per bit (no contended histogram).
    unsigned u[MED_REGS];
#pragma unroll
    for (int i = 0; i < MED_REGS; ++i) {
        unsigned idx = (unsigned)tid + 1024u * i;
        u[i] = idx < n ? __float_as_uint(v[idx]) : 0xffffffffu;  // all-ones never matches a prefix
    }
    unsigned prefix = 0, rank = n ? (n - 1) / 2 : 0;
    for (int bit = 30; bit >= 0 && n > 0; --bit) {  // bit 31 (sign) is clear for every D
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < MED_REGS; ++i) c += ((u[i] ^ prefix) >> bit) == 0u;
        for (unsigned idx = (unsigned)tid + 1024u * MED_REGS; idx < n; idx += 1024u)
            c += ((__float_as_uint(v[idx]) ^ prefix) >> bit) == 0u;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        if ((tid & 63) == 0 && c) atomicAdd(&s_bit[bit], c);
        __syncthreads();
        const unsigned zeros = s_bit[bit];
        if (rank >= zeros) { rank -= zeros; prefix |= 1u << bit; }
    }
    const float med = n ? __uint_as_float(prefix) : 0.0f;

    // ---- Welsch + symmetric min per selected line, bucket sums in LDS (fixed point)
    const int b0 = pool ? 0 : g, b1 = pool ? B : g + 1;
    for (int b = b0; b < b1; ++b) {
        const int ns = nsel[b];
        for (int i = tid; i < ns; i += 1024) {
            const size_t gl = (size_t)b * L + sel[(size_t)b * L + i];
            const unsigned c = kj[gl];
            const int k = c & 15, j = c >> 4;
            float rowmin[4], colmin[4];
            int arg_b[4], arg_a[4];
            welsch_block(D + gl * 16, k, j, med, rowmin, colmin, arg_b, arg_a);
            float row = 0.0f, col = 0.0f;
#pragma unroll
            for (int a = 0; a < RRL_MAX_HITS; ++a)
                if (a < k) row += rowmin[a];
#pragma unroll
            for (int bb = 0; bb < RRL_MAX_HITS; ++bb)
                if (bb < j) col += colmin[bb];
            // Wl in [0,1], <= 4 terms: 2^-40 fixed point keeps ~2^-38 relative resolution
            const int bi = (k - 1) * 4 + (j - 1);
            atomicAdd(&s_sum[bi * 2 + 0], (unsigned long long)((double)row * (double)(1ll << FIX_SHIFT) + 0.5));
            atomicAdd(&s_sum[bi * 2 + 1], (unsigned long long)((double)col * (double)(1ll << FIX_SHIFT) + 0.5));
            atomicAdd(&s_cnt[bi], 1);
        }
    }
    __syncthreads();

    // ---- loss = ( sum_{non-empty (k,j), k-major} exp(-|k-j|/2) (mean_row + mean_col) ) / C
    if (tid < 16) bcnt_out[g * 16 + tid] = s_cnt[tid];
    if (tid < 32) bsum_out[(size_t)g * 32 + tid] = (int64_t)s_sum[tid];
    if (tid == 0) {
        float acc = 0.0f;
        int C = 0, nselected = 0;
        for (int k = s_m; k < e_m; ++k)
            for (int j = s_n; j < e_n; ++j) {
                const int bi = (k - 1) * 4 + (j - 1);
                const int S = s_cnt[bi];
                if (S == 0) continue;
                const double sc = 1.0 / (double)(1ll << FIX_SHIFT);
                float mrow = (float)((double)s_sum[bi * 2 + 0] * sc / ((double)S * k));
                float mcol = (float)((double)s_sum[bi * 2 + 1] * sc / ((double)S * j));
                float wkj = expf(-0.5f * (float)abs(k - j));  // code/loss.py:215
                acc = acc + wkj * (mrow + mcol);
                ++C;
                nselected += S;
            }
        med_out[g] = med;
        loss[g] = C ? acc / (float)C : 0.0f;  // code/loss.py:230
        info[g * 4 + 0] = C;
        info[g * 4 + 1] = nselected;
        info[g * 4 + 2] = (int)n;
        info[g * 4 + 3] = 0;
    }
}

extern "C" int rrl_loss_reduce(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L,
                               int s_m, int s_n, int e_m, int e_n, int pool, void *stream) {
    if (!ws || !loss || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0) return 0;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3((unsigned)(pool ? 1 : B)), dim3(1024), 0,
                       (hipStream_t)stream, w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL),
                       w.i32(ws, RRL_WS_NSEL), w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_VALS),
                       w.i32(ws, RRL_WS_NVALS), w.f32(ws, RRL_WS_MED), w.i32(ws, RRL_WS_BCNT),
                       w.i64(ws, RRL_WS_BSUM), w.i32(ws, RRL_WS_INFO), loss, B, L, s_m, s_n, e_m,
                       e_n, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K5 backward.  dL/dD[a][b] = gout * w_kj / C * exp(-D/(2 med)) / (2 med)
//                              * ( [b = argmin_b(a)] / (S k) + [a = argmin_a(b)] / (S j) )
// dL/dq1[a] = sum_b 2 (q1_a - q2_b) dL/dD;  dL/dP1[f_a][kk] += w_kk / 3 * dL/dq1[a];
// weights, median and labels carry no gradient (code/loss.py:112, 224).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_bwd_kernel(
    const float *__restrict__ tri1, const float *__restrict__ tri2, const uint8_t *__restrict__ kj,
    const int32_t *__restrict__ sel, const int32_t *__restrict__ nsel,
    const int32_t *__restrict__ hs1, const int32_t *__restrict__ hs2, const float *__restrict__ w1,
    const float *__restrict__ w2, const float *__restrict__ D, const float *__restrict__ med,
    const int32_t *__restrict__ bcnt, const int32_t *__restrict__ info,
    const float *__restrict__ grad_loss, float *__restrict__ g1, float *__restrict__ g2, int B,
    int N, int M, int L, int pool) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= nsel[b]) return;  // dense wavefronts over the compacted selected lines
    const size_t gl = (size_t)b * L + sel[(size_t)b * L + i];
    const unsigned c = kj[gl];
    const int k = c & 15, j = c >> 4, g = pool ? 0 : b;
    const int C = info[g * 4];
    if (C == 0) return;
    const float m = med[g];
    float rowmin[4], colmin[4];
    int arg_b[4], arg_a[4];
    welsch_block(D + gl * 16, k, j, m, rowmin, colmin, arg_b, arg_a);
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
                    // dWl/dD = exp(-D/(2 med)) / (2 med)
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

extern "C" int rrl_loss_backward(const float *tri1, const float *tri2, const void *ws,
                                 size_t ws_bytes, const float *grad_loss, float *grad_tri1,
                                 float *grad_tri2, int B, int N, int M, int L, int pool,
                                 void *stream) {
    if (!tri1 || !tri2 || !ws || !grad_loss || !grad_tri1) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if ((size_t)B * N && (e = hipMemsetAsync(grad_tri1, 0, sizeof(float) * 9 * (size_t)B * N, s)) != hipSuccess)
        return (int)e;
    if (grad_tri2 && (size_t)B * M &&
        (e = hipMemsetAsync(grad_tri2, 0, sizeof(float) * 9 * (size_t)B * M, s)) != hipSuccess)
        return (int)e;
    if (B == 0 || L == 0) return 0;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), 0,
                       s, tri1, tri2, w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL),
                       w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), w.i32(ws, RRL_WS_HS2),
                       w.f32(ws, RRL_WS_W1), w.f32(ws, RRL_WS_W2),
                       w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_MED), w.i32(ws, RRL_WS_BCNT),
                       w.i32(ws, RRL_WS_INFO), grad_loss, grad_tri1, grad_tri2, B, N, M, L, pool);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// workspace + fused forward
// ---------------------------------------------------------------------------------------
extern "C" size_t rrl_workspace_bytes(int B, int N, int M, int L) { return WsLayout(B, N, M, L).total; }

extern "C" int rrl_workspace_layout(int B, int N, int M, int L, size_t *offsets) {
    if (!offsets || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    for (int i = 0; i < RRL_WS_FIELDS; ++i) offsets[i] = w.off[i];
    return 0;
}

extern "C" int rrl_loss_forward(const float *tri1, const float *tri2, const float *line, void *ws,
                                size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                                int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                                void *stream) {
    if (!tri1 || !tri2 || !line || !ws || !loss) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    int rc;
    if ((rc = rrl_tri_prepare(tri1, tri2, ws, ws_bytes, B, N, M, L, stream))) return rc;
    if ((rc = rrl_line_tri_scan(line, ws, ws_bytes, B, N, M, L, mode, chunk, stream))) return rc;
    if ((rc = rrl_line_pair_dist(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m, e_n,
                                 pool, stream)))
        return rc;
    return rrl_loss_reduce(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, stream);
}
