// rrl_cull.hip -- Morton sort of the prepared triangles and K1 with sphere culling
// (scan mode RRL_SCAN_CULL).  Compiled with -fno-slp-vectorize: packed fp32 issues at half
// rate on gfx950, so SLP-packing the all-VGPR exact test only adds register shuffling.
//
// The full scan evaluates every (line, triangle) pair although only ~6e-4 of them can pass even
// the first point's test.  Here:
//   tri_sort_kernel (one 1024-lane workgroup per cloud and sample)
//     orders the triangles by the 16^3 grid cell of P0, cells in Morton order (counting sort
//     in LDS), and writes, in that order, 16-byte (P0, thr2) records (P0S), their original indices (IDX) and, for every
//     group of 16 consecutive triangles, a bounding sphere of the P0s: centre c,
//     rho = max |P0 - c| and the conservative squared radius
//     R2 = ((rho + max thr)^2)(1 + 1e-4) + 1e-7.
//   cull_scan_kernel (256 lines per workgroup, lane = one line in phase 1)
//     per slab of 2048 sorted triangles (staged in LDS):
//     phase 1: every lane tests ITS line against each group sphere (wave-uniform sphere, SGPR
//              operands) with a conservative test and keeps one mask bit per group;
//     queue:   the (line, group) pairs of all 256 lanes are compacted into an LDS queue
//              (block prefix sum over popcounts), so that
//     phase 2: lanes pull pairs round-robin -- every lane does the same number of exact
//              evaluations however unevenly the pairs are distributed over the lines -- and
//              run the scan's exact point-0 test (same dist_sq arithmetic, bit-identical) on
//              the group's 16 triangles; points 1, 2 (global PTRI record) only where point 0
//              passes.
//
// Culling bound (labels can never be lost).  For a line with |dir|^2 <= 1 + 1e-6 and
// (|x0| + max|P|)^2 <= 100 ("safe", the NaN bound of rrl_scan.hip) let
// delta(P)^2 = |a|^2 - (a.dir)^2, a = P - x0, in exact arithmetic.  delta is a seminorm of a
// (|dir| <= 1) and hence 1-Lipschitz in P; the 1e-6 excess of |dir|^2 adds at most 1e-6 |a|^2.
// The reference value x0_ref = fl((dAC - proj) + 2e-4) satisfies
// |x0_ref - (delta(P0)^2 + 2e-4)| <= 30u |a|^2 <= 1.8e-4 (u = 2^-24), so a hit
// (x0_ref < thr2 <= thr^2 (1 + 2u)) implies delta(P0) < thr and therefore
// delta(c) < thr_max + rho for the centre c of the triangle's group.  Phase 1 evaluates
// d2 = |a_c|^2 - (a_c.dir)^2 with FMAs (error <= 10u |a_c|^2) and keeps the group when
// d2 - 4e-6 |a_c|^2 <= R2: the slack covers the evaluation error, the |dir|^2 excess and the
// rounding of rho and R2 with a factor > 2 to spare.  Unsafe lines are not culled at all:
// tiles of 512 lines containing one are left to the strict loop of scan_kernel.
#include "rrl_ws.h"

#define GRP 16           // triangles per group
#define SLAB_TRIS 2048   // triangles staged in LDS per pass of the cull scan (32 KiB)
#define SLAB_GROUPS (SLAB_TRIS / GRP)
#define SLAB_WORDS (SLAB_GROUPS / 32)
#define QCAP 6144        // (line, group) pairs per queue round (12 KiB of u16)
#define SORT_CAP 16384   // largest cloud the sort kernel handles (64 KiB of LDS for thr)

__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x30000ffu;
    v = (v | (v << 8)) & 0x300f00fu;
    v = (v | (v << 4)) & 0x30c30c3u;
    v = (v | (v << 2)) & 0x9249249u;
    return v;
}

// slot of sorted triangle s inside its group's 256-byte row: rotating by the group index keeps
// lanes that read different groups at the same step on different LDS banks
__device__ __forceinline__ int p0s_slot(int s) { return (s & ~15) | ((s + (s >> 4)) & 15); }

#define SORT_CELLS 4096  // 16^3 grid cells in Morton order

// Counting sort by grid cell (one 1024-lane workgroup per cloud and sample): cells of a 16^3
// grid over the P0 bounding box, visited in Morton order; the order inside a cell is arbitrary
// (it only shapes the groups, never the result).  Three barriers instead of a 78-stage bitonic
// network.
__global__ __launch_bounds__(1024) void tri_sort_kernel(
    const float *__restrict__ ptri1, const float *__restrict__ ptri2, float4 *__restrict__ p0s1,
    float4 *__restrict__ p0s2, int32_t *__restrict__ idx1, int32_t *__restrict__ idx2,
    float4 *__restrict__ grp1, float4 *__restrict__ grp2, int B, int N, int M) {
    extern __shared__ __attribute__((aligned(16))) float thr_s[];  // thr by sorted position
    __shared__ unsigned hist[SORT_CELLS];
    __shared__ float red[16][8];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.x >= (unsigned)B ? 1 : 0, b = blockIdx.x - cloud * B;
    const int n = cloud ? M : N;
    const float *ptri = (cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE;
    const int ng = (n + GRP - 1) / GRP;
    float4 *p0s = (cloud ? p0s2 : p0s1) + (size_t)b * ng * GRP;
    int32_t *idx = (cloud ? idx2 : idx1) + (size_t)b * ng * GRP;
    float4 *grp = (cloud ? grp2 : grp1) + (size_t)b * ng;

    // ---- AABB of the P0s
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int f = tid; f < n; f += 1024) {
        const float *p = ptri + PTRI_STRIDE * (size_t)f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], p[c]); mx[c] = fmaxf(mx[c], p[c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_down(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_down(mx[c], o));
        }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave][c] = mn[c]; red[wave][3 + c] = mx[c]; }
    }
    for (int i = tid; i < SORT_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    if (tid < 6) {
        float r = red[0][tid];
        for (int w = 1; w < 16; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        red[0][tid] = r;
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = red[0][c];
        float ext = red[0][3 + c] - mn[c];
        scale[c] = ext > 0.0f && ext < 3.0e38f ? 15.999f / ext : 0.0f;
    }
    auto cell_of = [&](const float *p) -> unsigned {
        unsigned q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (p[c] - mn[c]) * scale[c];
            q[c] = v >= 15.0f ? 15u : (v > 0.0f ? (unsigned)v : 0u);
        }
        return spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);  // 12 bits
    };

    // ---- histogram over cells, exclusive scan, scatter
    for (int f = tid; f < n; f += 1024) atomicAdd(&hist[cell_of(ptri + PTRI_STRIDE * (size_t)f)], 1u);
    __syncthreads();
    {
        unsigned h[4], tsum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h[k] = hist[4 * tid + k]; tsum += h[k]; }
        unsigned inc = tsum;
        for (int o = 1; o < 64; o <<= 1) {
            unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        unsigned run = base + inc - tsum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hist[4 * tid + k] = run; run += h[k]; }
    }
    __syncthreads();
    for (int f = tid; f < n; f += 1024) {
        const float *p = ptri + PTRI_STRIDE * (size_t)f;
        const int s = (int)atomicAdd(&hist[cell_of(p)], 1u);
        p0s[p0s_slot(s)] = make_float4(p[0], p[1], p[2], p[9]);
        idx[s] = f;
        thr_s[s] = p[10];
    }
    for (int s = n + tid; s < ng * GRP; s += 1024) {  // pad: thr2 = 0 never passes
        p0s[p0s_slot(s)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        idx[s] = 0;
        thr_s[s] = 0.0f;
    }
    __syncthreads();  // the block's own global stores are visible to it after the barrier

    // ---- group spheres (16 consecutive lanes = one group)
    for (int s = tid; s < ng * GRP; s += 1024) {
        const bool valid = s < n;
        const float4 r4 = p0s[p0s_slot(s)];
        const float c[3] = {r4.x, r4.y, r4.z};
        float lo[3], hi[3], tm = thr_s[s];
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = valid ? c[d] : INFINITY; hi[d] = valid ? c[d] : -INFINITY; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 16));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 16));
            }
            tm = fmaxf(tm, __shfl_xor(tm, o, 16));
        }
        float ctr[3], d2 = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ctr[d] = 0.5f * lo[d] + 0.5f * hi[d];
            float e = c[d] - ctr[d];
            d2 += e * e;
        }
        if (!valid) d2 = 0.0f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 16));
        if ((s & 15) == 0) {
            float rho = sqrtf(d2) * 1.00001f + 1e-7f;
            float R = rho + tm;
            float R2 = R * R * 1.0001f + 1e-7f;
            if (!(R2 < 3.0e38f)) R2 = INFINITY;  // non-finite data: keep the group
            grp[s >> 4] = make_float4(ctr[0], ctr[1], ctr[2], R2);
        }
    }
}

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

__global__ __launch_bounds__(256) void cull_scan_kernel(
    const float *__restrict__ ptri1, const float *__restrict__ ptri2, const float4 *__restrict__ p0s1,
    const float4 *__restrict__ p0s2, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    const float4 *__restrict__ grp1, const float4 *__restrict__ grp2, const float *__restrict__ line,
    int32_t *__restrict__ count1, int32_t *__restrict__ hit1, int32_t *__restrict__ count2,
    int32_t *__restrict__ hit2, const uint32_t *__restrict__ pmax, int B, int N, int M, int L) {
    __shared__ __attribute__((aligned(16))) float4 slab[SLAB_TRIS];       // 32 KiB
    __shared__ __attribute__((aligned(16))) float4 lines_lds[256][2];     //  8 KiB
    __shared__ unsigned short queue[QCAP];                                // 12 KiB
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, cloud = z >= B ? 1 : 0, b = z - cloud * B;
    const int n = cloud ? M : N;
    const int ng = (n + GRP - 1) / GRP;
    const float4 *p0s = (cloud ? p0s2 : p0s1) + (size_t)b * ng * GRP;
    const int32_t *idx = (cloud ? idx2 : idx1) + (size_t)b * ng * GRP;
    const float4 *grp = (cloud ? grp2 : grp1) + (size_t)b * ng;
    const float *ptri = (cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE;
    const float *ln = line + (size_t)b * L * 6;
    int32_t *cnt = (cloud ? count2 : count1) + (size_t)b * L;
    int32_t *hit = (cloud ? hit2 : hit1) + (size_t)b * L * RRL_MAX_HITS;

    // this lane's line, and the safety of the whole 512-line tile (this block and its sibling):
    // unsafe tiles belong to scan_kernel's strict loop
    const int l0 = blockIdx.x * 256;
    const int l = l0 + tid;
    const int ls = (blockIdx.x ^ 1) * 256 + tid;
    const float pm = __uint_as_float(pmax[cloud * B + b]);
    float v[6], vs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        v[c] = l < L ? ln[6 * (size_t)l + c] : 0.0f;
        vs[c] = ls < L ? ln[6 * (size_t)ls + c] : 0.0f;
    }
    const bool safe = rrl_line_safe(v, pm) && rrl_line_safe(vs, pm);
    if (!__syncthreads_and(safe)) return;
    lines_lds[tid][0] = make_float4(v[0], v[1], v[2], v[3]);
    lines_lds[tid][1] = make_float4(v[4], v[5], 0.0f, 0.0f);
    const float ux = v[0], uy = v[1], uz = v[2], ox = v[3], oy = v[4], oz = v[5];
    const bool live = l < L;

    for (int g0 = 0; g0 < ng; g0 += SLAB_GROUPS) {
        const int g1 = min(ng, g0 + SLAB_GROUPS);
        __syncthreads();  // previous slab fully consumed (and lines_lds visible)
        for (int i = tid; i < (g1 - g0) * GRP; i += 256) slab[i] = p0s[(size_t)g0 * GRP + i];

        // ---- phase 1: conservative sphere test of every group of the slab
        uint32_t mw[SLAB_WORDS];
#pragma unroll
        for (int w = 0; w < SLAB_WORDS; ++w) {
            uint32_t m = 0;
            const int gb = g0 + w * 32;
            if (gb < g1 && live) {
                kptr gp = (kptr)(uintptr_t)(grp + gb);
                const int cntg = min(32, g1 - gb);
#pragma unroll 8
                for (int j = 0; j < cntg; ++j) {
                    const float cx = gp[4 * j], cy = gp[4 * j + 1], cz = gp[4 * j + 2], R2 = gp[4 * j + 3];
                    float ax = cx - ox, ay = cy - oy, az = cz - oz;
                    float dot = fmaf(az, uz, fmaf(ay, uy, ax * ux));
                    float q = fmaf(az, az, fmaf(ay, ay, ax * ax));
                    float d2 = fmaf(-dot, dot, q);
                    d2 = fmaf(-4e-6f, q, d2);
                    m |= (d2 <= R2 ? 1u : 0u) << j;
                }
            }
            mw[w] = m;
        }

        // ---- queue rounds: compact the block's (line, group) pairs, then balanced phase 2
        for (;;) {
            int mine = 0;
#pragma unroll
            for (int w = 0; w < SLAB_WORDS; ++w) mine += __popc(mw[w]);
            int inc = mine;  // inclusive scan over the wave
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            __syncthreads();  // queue / s_wave of the previous round are no longer read
            if (lane == 63) s_wave[wave] = inc;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w < wave) base += s_wave[w];
                total += s_wave[w];
            }
            if (total == 0) break;  // uniform
            int pos = base + inc - mine;  // exclusive offset of this lane's first pair
#pragma unroll
            for (int w = 0; w < SLAB_WORDS; ++w) {
                uint32_t m = mw[w];
                while (m && pos < QCAP) {
                    const int j = __ffs(m) - 1;
                    m &= m - 1;
                    queue[pos++] = (unsigned short)((tid << 8) | (w * 32 + j));
                }
                mw[w] = m;  // what did not fit waits for the next round
            }
            __syncthreads();
            const int nq = min(total, QCAP);
            // ---- phase 2: exact lazy evaluation, one (line, group) pair per lane and step
            for (int e = tid; e < nq; e += 256) {
                const unsigned ent = queue[e];
                const int ll = ent >> 8, gl = ent & 255;
                const int g = g0 + gl;
                const float4 la = lines_lds[ll][0], lb = lines_lds[ll][1];
                const float4 *row = slab + gl * GRP;
                uint32_t passbits = 0;
#pragma unroll
                for (int t = 0; t < GRP; ++t) {
                    const float4 rec = row[(t + g) & 15];
                    const float x = dist_sq<float>(rec.x, rec.y, rec.z, la.x, la.y, la.z, la.w, lb.x, lb.y);
                    passbits |= (__float_as_uint(x) < __float_as_uint(rec.w) ? 1u : 0u) << t;
                }
                while (passbits) {  // rare: point 0 of sorted triangle g*16+t is within thr
                    const int t = __ffs(passbits) - 1;
                    passbits &= passbits - 1;
                    const int f = idx[(size_t)g * GRP + t];
                    const float *q = ptri + PTRI_STRIDE * (size_t)f;
                    const uint32_t thr2 = __float_as_uint(q[9]);
                    const float x1 = dist_sq<float>(q[3], q[4], q[5], la.x, la.y, la.z, la.w, lb.x, lb.y);
                    const float x2 = dist_sq<float>(q[6], q[7], q[8], la.x, la.y, la.z, la.w, lb.x, lb.y);
                    if (max(__float_as_uint(x1), __float_as_uint(x2)) < thr2) {
                        const int gl_line = l0 + ll;
                        int pos2 = atomicAdd(&cnt[gl_line], 1);
                        if (pos2 < RRL_MAX_HITS) hit[(size_t)gl_line * RRL_MAX_HITS + pos2] = f;
                    }
                }
            }
            if (total <= QCAP) break;  // uniform: nothing left over
        }
    }
}

// Launchers used by rrl_tri_prepare / rrl_line_tri_scan (rrl_scan.hip)
int rrl_launch_tri_sort(void *ws, const WsLayout &w, int B, int N, int M, hipStream_t s) {
    const int nmax = N > M ? N : M;
    const size_t lds = sizeof(float) * (size_t)((nmax + GRP - 1) / GRP * GRP);
    hipLaunchKernelGGL(tri_sort_kernel, dim3((unsigned)(2 * B)), dim3(1024), lds, s, w.f32(ws, RRL_WS_PTRI1),
                       w.f32(ws, RRL_WS_PTRI2), (float4 *)w.f32(ws, RRL_WS_P0S1),
                       (float4 *)w.f32(ws, RRL_WS_P0S2), w.i32(ws, RRL_WS_IDX1), w.i32(ws, RRL_WS_IDX2),
                       (float4 *)w.f32(ws, RRL_WS_GRP1), (float4 *)w.f32(ws, RRL_WS_GRP2), B, N, M);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_launch_cull_scan(const float *line, void *ws, const WsLayout &w, int B, int N, int M, int L,
                         hipStream_t s) {
    hipLaunchKernelGGL(cull_scan_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)(2 * B)), dim3(256), 0,
                       s, w.f32(ws, RRL_WS_PTRI1), w.f32(ws, RRL_WS_PTRI2),
                       (const float4 *)w.f32(ws, RRL_WS_P0S1), (const float4 *)w.f32(ws, RRL_WS_P0S2),
                       w.i32(ws, RRL_WS_IDX1), w.i32(ws, RRL_WS_IDX2), (const float4 *)w.f32(ws, RRL_WS_GRP1),
                       (const float4 *)w.f32(ws, RRL_WS_GRP2), line, w.i32(ws, RRL_WS_COUNT1),
                       w.i32(ws, RRL_WS_HIT1), w.i32(ws, RRL_WS_COUNT2), w.i32(ws, RRL_WS_HIT2),
                       (const uint32_t *)w.i32(ws, RRL_WS_PMAX), B, N, M, L);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_sort_capacity(void) { return SORT_CAP; }
