// rrl_cull.hip -- the build kernel (transform + records + cell sort + group spheres) and K1 with
// sphere culling (scan mode RRL_SCAN_CULL).  Compiled with -fno-slp-vectorize: packed fp32
// issues at half rate on gfx950, so SLP-packing the all-VGPR exact test only adds register
// shuffling (the packed code below is explicit).
//
// The full scan evaluates every (line, triangle) pair although only ~6e-4 of them can pass even
// the first point's test.  Here:
//   tri_build_kernel (one 1024-lane workgroup per cloud and sample)
//     orders the triangles by the 16^3 grid cell of P0, cells in Hilbert-curve order (counting sort
//     in LDS), and writes, in that order, 16-byte (P0, thr2) records (P0S), their original
//     indices (IDX) and, for every group of 16 consecutive triangles, a bounding sphere of the
//     P0s: centre c, rho = max |P0 - c| and the conservative squared radius
//     R2 = ((rho + max thr)^2)(1 + 1e-4) + 1e-7.
//   cull_scan_kernel (128 lines per workgroup, two per lane; its 8 wavefronts hold the same lines
//   and split the group range; every wavefront has private LDS queues, so there is no workgroup
//   synchronisation at all)
//     phase 1: every lane tests ITS two lines (packed fp32) against each group sphere
//              (wave-uniform sphere, scalar loads) with a conservative test; each passing
//              (line, group) pair is written straight to its slot of the batch's entry list
//              (ballot + mbcnt rank: group-major, no per-lane bit masks to unpack later) and the
//              group's 16 records are DMA-copied (global_load ... lds) into the wave's LDS rows;
//     phase 2: when 128 pairs or 12 groups have gathered, the lanes take one pair each per pass
//              -- every lane does the same number of exact evaluations however unevenly the
//              pairs are spread over the lines -- and run the scan's exact point-0 test (same
//              dist_sq arithmetic, bit-identical) on the group's 16 records from LDS.  Point-0
//              passes (~1 in 130 tests) are parked and their points 1, 2 (two dependent global
//              loads) resolved densely afterwards.
//   Measured (B=8, N=M=4096, L=10000): phase 1 alone 30 us, both 65 us; per-wave LDS, the batch
//   geometry (EB, BGRP) and wavefronts per workgroup were swept on the GPU (tools/knob_sweep.sh).
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
// rounding of rho and R2 with a factor > 2 to spare.  Unsafe lines are not culled at all: a
// 64-line workgroup containing one evaluates all its pairs with the strict loop.
#include "rrl_ws.h"

#define GRP 16           // triangles per group
#define SORT_CAP 65536   // largest cloud of the sorted / culled layout (16-bit sorted positions in the scan)

// Position of the 16^3 grid cell (q0, q1, q2) along a 3-D Hilbert curve (12 bits).  Consecutive
// cells of the curve are always face neighbours, so a group of 16 consecutive sorted triangles
// never spans a jump of the curve the way Morton order does: on the bench clouds a line reaches
// 10.2 group spheres per cloud instead of 17.1 (same cells, same sort).  Axes -> transposed index
// by the standard inversion / exchange sweep from the top bit down, Gray decode, then bit
// interleave.  Any permutation of the cells gives the same labels; only the group shapes change.
__device__ __forceinline__ unsigned hilbert_cell(unsigned q0, unsigned q1, unsigned q2) {
    unsigned x[3] = {q0, q1, q2};
#pragma unroll
    for (unsigned q = 8u; q > 1u; q >>= 1) {
        const unsigned p = q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned set = 0u - ((x[i] / q) & 1u);  // all ones when bit q of x[i] is set
            const unsigned t = (x[0] ^ x[i]) & p & ~set;   // clear: exchange the low bits with x[0]
            x[0] ^= (p & set) | t;                         // set: invert the low bits of x[0]
            x[i] ^= t;
        }
    }
    x[1] ^= x[0];
    x[2] ^= x[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned q = 8u; q > 1u; q >>= 1)
        t ^= (q - 1u) & (0u - ((x[2] / q) & 1u));
    unsigned key = 0;
#pragma unroll
    for (int bit = 3; bit >= 0; --bit)
#pragma unroll
        for (int i = 0; i < 3; ++i) key = (key << 1) | (((x[i] ^ t) >> bit) & 1u);
    return key;
}

__device__ __forceinline__ int p0s_slot(int s) { return s; }  // records in sorted order

#define SORT_CELLS 4096  // 16^3 grid cells in Hilbert-curve order

// The build step: everything the scans need from the raw triangles, in two launches.
//   tri_records_kernel (wide: one lane per triangle, 256-lane workgroups over both clouds)
//     * optionally moves the source cloud by its rigid transform (the fused training op) and
//       stores the moved triangles (TRI1) for the later stages and the backward;
//     * thresholds (thr, thr2), the 48-byte prepared records (PTRI) in original order and a
//       compact 16-byte (P0, thr2) record (CREC) for the sort;
//     * per-workgroup partial AABB of the P0s and max |P|^2 (APART);
//     * clears the per-call state of the workspace (and the gradient accumulator G1).
//   tri_sort_kernel (one 1024-lane workgroup per cloud and sample; reads only CREC: one
//   workgroup's memory pipe is the bottleneck here, so it touches 16 bytes per triangle)
//     counting sort by the 16^3 grid cell of P0, cells in Hilbert-curve order (the order inside a cell
//     is arbitrary: it only shapes the groups, never the result), group spheres, max |P|^2.
struct BuildArgs {
    const float *tri1, *tri2;      // raw triangles [B][n][9]; tri1 = source BEFORE the transform
    const float *R, *t;            // per-sample transform of cloud 0, or NULL
    float *tri1_out;               // moved source triangles (TRI1), when R != NULL
    float *ptri1, *ptri2;
    float4 *crec1, *crec2;         // unsorted (P0, thr2)
    float *apart;                  // [clouds][B][nblk][8]: min xyz, max xyz, max |P|^2, pad
    float4 *p0s1, *p0s2;
    int32_t *idx1, *idx2;
    float4 *grp1, *grp2;
    uint32_t *pmax;
    uint4 *zero_base;              // per-call state: [0, zero_vec4)
    size_t zero_vec4;
    uint4 *g1;                     // gradient accumulator to clear (may be NULL)
    size_t g1_vec4;
    uint4 *z2;                     // global cell histogram / cursors of the wide sort (may be NULL)
    size_t z2_vec4;
    int B, N, M, transpose_r, nblk;
};

#define REC_BLK 256

__global__ __launch_bounds__(REC_BLK) void tri_records_kernel(const BuildArgs a) {
    __shared__ float red[REC_BLK / 64][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.z, b = blockIdx.y, B = a.B;
    {   // per-call state and gradient accumulator, spread over all workgroups of the launch
        const size_t nthr = (size_t)gridDim.x * gridDim.y * gridDim.z * REC_BLK;
        const size_t me = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * REC_BLK + tid;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (size_t i = me; i < a.zero_vec4; i += nthr) a.zero_base[i] = z;
        for (size_t i = me; i < a.g1_vec4; i += nthr) a.g1[i] = z;
        for (size_t i = me; i < a.z2_vec4; i += nthr) a.z2[i] = z;
    }
    const int n = cloud ? a.M : a.N;
    const int f = blockIdx.x * REC_BLK + tid;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, p2 = 0.0f;
    if (f < n) {
        const float *raw = (cloud ? a.tri2 : a.tri1) + ((size_t)b * n + f) * 9;
        float c[9], thr, x;
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = raw[i];
        if (cloud == 0 && a.R != nullptr) {
            float m[9], tv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)  // m[i*3+j] multiplies x_i into y_j (rigid_fwd_kernel)
                    m[i * 3 + j] = a.transpose_r ? a.R[b * 9 + j * 3 + i] : a.R[b * 9 + i * 3 + j];
#pragma unroll
            for (int j = 0; j < 3; ++j) tv[j] = a.t[b * 3 + j];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float v0 = c[3 * q], v1 = c[3 * q + 1], v2 = c[3 * q + 2];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    c[3 * q + j] = fmaf(v2, m[6 + j], fmaf(v1, m[3 + j], v0 * m[j])) + tv[j];
            }
            float *moved = a.tri1_out + ((size_t)b * n + f) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) moved[i] = c[i];
        }
        tri_thresholds(c, &thr, &x);  // code/loss.py:94-110
        float4 *row = (float4 *)((cloud ? a.ptri2 : a.ptri1) + ((size_t)b * n + f) * PTRI_STRIDE);
        row[0] = make_float4(c[0], c[1], c[2], c[3]);
        row[1] = make_float4(c[4], c[5], c[6], c[7]);
        row[2] = make_float4(c[8], x, thr, __int_as_float(f));
        const int ngp = (n + GRP - 1) / GRP * GRP;
        ((cloud ? a.crec2 : a.crec1) + (size_t)b * ngp)[f] = make_float4(c[0], c[1], c[2], x);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = c[d]; mx[d] = c[d]; }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            p2 = fmaxf(p2, c[3 * q] * c[3 * q] + c[3 * q + 1] * c[3 * q + 1] + c[3 * q + 2] * c[3 * q + 2]);
        if (!(p2 <= 3.0e38f)) p2 = INFINITY;  // NaN/inf coordinates: never "provably safe"
    }
    if (blockIdx.x * REC_BLK >= n) return;  // uniform: this workgroup has no triangle of the cloud
#pragma unroll
    for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    p2 = wave_max(p2);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave][c] = mn[c]; red[wave][3 + c] = mx[c]; }
        red[wave][6] = p2;
    }
    __syncthreads();
    if (tid < 7) {
        float r = red[0][tid];
        for (int w = 1; w < REC_BLK / 64; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        a.apart[(((size_t)cloud * B + b) * a.nblk + blockIdx.x) * 8 + tid] = r;
    }
}

// NPT > 0: every lane keeps its <= NPT records in registers between the passes (n <= 1024 NPT);
// NPT == 0: they are re-read (16 B per triangle).
template <int NPT>
__global__ __launch_bounds__(1024) void tri_sort_kernel(const BuildArgs a) {
    // NPT > 0: the sorted records and their triangle indices are assembled in LDS ([ng*17] float4,
    //          one float4 of padding per group: lanes on different groups hit different banks;
    //          then [ng*16] int) and leave with coalesced stores;
    // NPT == 0: no dynamic LDS, records re-read from P0S
    extern __shared__ __attribute__((aligned(16))) float dyn_s[];
    float4 *srec = (float4 *)dyn_s;
    __shared__ unsigned hist[SORT_CELLS];
    __shared__ float red[16][8];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B;
    const int cloud = blockIdx.x >= (unsigned)B ? 1 : 0, b = blockIdx.x - cloud * B;
    const int n = cloud ? a.M : a.N;
    const int ng = (n + GRP - 1) / GRP;
    const float4 *crec = (cloud ? a.crec2 : a.crec1) + (size_t)b * ng * GRP;
    float4 *p0s = (cloud ? a.p0s2 : a.p0s1) + (size_t)b * ng * GRP;
    int32_t *idx = (cloud ? a.idx2 : a.idx1) + (size_t)b * ng * GRP;
    float4 *grp = (cloud ? a.grp2 : a.grp1) + (size_t)b * ng;
    int *sidx = (int *)(srec + (size_t)ng * 17);
    auto pad = [](int s) { return s + (s >> 4); };

    // ---- AABB of the P0s and max |P|^2 from the per-workgroup partials of tri_records_kernel
    constexpr int NR = NPT > 0 ? NPT : 1;
    float4 rec[NR];
    if (NPT > 0) {  // issue the record loads first: they overlap the reduction
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (tid + 1024 * k < n) rec[k] = crec[tid + 1024 * k];
    }
    {
        const int nb = (n + REC_BLK - 1) / REC_BLK;
        const float *ap = a.apart + ((size_t)cloud * B + b) * a.nblk * 8;
        float v[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = c < 3 ? INFINITY : (c < 6 ? -INFINITY : 0.0f);
        for (int j = tid; j < nb; j += 1024)
#pragma unroll
            for (int c = 0; c < 7; ++c) v[c] = c < 3 ? fminf(v[c], ap[j * 8 + c]) : fmaxf(v[c], ap[j * 8 + c]);
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = c < 3 ? wave_min(v[c]) : wave_max(v[c]);
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < 7; ++c) red[wave][c] = v[c];
    }
    for (int i = tid; i < SORT_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    if (tid < 7) {
        float r = red[0][tid];
        for (int w = 1; w < 16; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        red[0][tid] = r;
    }
    __syncthreads();
    if (tid == 0) a.pmax[cloud * B + b] = __float_as_uint(red[0][6]);
    float mn[3], scale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = red[0][c];
        float ext = red[0][3 + c] - mn[c];
        scale[c] = ext > 0.0f && ext < 3.0e38f ? 15.999f / ext : 0.0f;
    }
    auto cell_of = [&](const float4 r) -> unsigned {
        const float p[3] = {r.x, r.y, r.z};
        unsigned q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (p[c] - mn[c]) * scale[c];
            q[c] = v >= 15.0f ? 15u : (v > 0.0f ? (unsigned)v : 0u);
        }
        return hilbert_cell(q[0], q[1], q[2]);  // 12 bits
    };

    // ---- histogram over cells, exclusive scan, scatter
    unsigned cell[NR];
    if (NPT > 0) {
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (tid + 1024 * k < n) { cell[k] = cell_of(rec[k]); atomicAdd(&hist[cell[k]], 1u); }
    } else {
        for (int f = tid; f < n; f += 1024) atomicAdd(&hist[cell_of(crec[f])], 1u);
    }
    __syncthreads();
    {
        unsigned h[4], tsum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h[k] = hist[4 * tid + k]; tsum += h[k]; }
        const unsigned inc = (unsigned)wave_incl_scan((int)tsum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        unsigned run = base + inc - tsum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hist[4 * tid + k] = run; run += h[k]; }
    }
    __syncthreads();
    // thr <= sqrtf(thr2) (1 + 2^-22): thr2 is the smallest float whose rounded root reaches thr
    auto thr_bound = [](float thr2) { return sqrtf(thr2) * 1.000001f; };
    if (NPT > 0) {
        // scatter into LDS only; the global arrays are written afterwards, in order
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int f = tid + 1024 * k;
            if (f < n) {
                const int s = (int)atomicAdd(&hist[cell[k]], 1u);
                srec[pad(s)] = rec[k];
                sidx[s] = f;
            }
        }
        for (int s = n + tid; s < ng * GRP; s += 1024) {  // pad: thr2 = 0 never passes
            srec[pad(s)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sidx[s] = 0;
        }
        __syncthreads();
        for (int s = tid; s < ng * GRP; s += 1024) {  // coalesced copy-out
            p0s[s] = srec[pad(s)];
            idx[s] = sidx[s];
        }
        // ---- group spheres: ONE lane per group walks its 16 records (16x less work than 16 lanes
        //      reducing each other's values with DPP butterflies; this phase took 5.3 us of the
        //      kernel's 14.4 on its single CU)
        for (int g = tid; g < ng; g += 1024) {
            const float4 *r = srec + (size_t)g * 17;
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, tm = 0.0f;
            float px[GRP], py[GRP], pz[GRP];
#pragma unroll
            for (int t = 0; t < GRP; ++t) {
                const float4 v = r[t];
                px[t] = v.x; py[t] = v.y; pz[t] = v.z;
                if (g * GRP + t < n) {
                    lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
                    lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
                    lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
                    tm = fmaxf(tm, thr_bound(v.w));
                }
            }
            const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1],
                        cz = 0.5f * lo[2] + 0.5f * hi[2];
            float d2 = 0.0f;
#pragma unroll
            for (int t = 0; t < GRP; ++t) {
                const float ex = px[t] - cx, ey = py[t] - cy, ez = pz[t] - cz;
                const float e2 = ex * ex + ey * ey + ez * ez;
                if (g * GRP + t < n) d2 = fmaxf(d2, e2);
            }
            const float rho = sqrtf(d2) * 1.00001f + 1e-7f;
            const float R = rho + tm;
            float R2 = R * R * 1.0001f + 1e-7f;
            if (!(R2 < 3.0e38f)) R2 = INFINITY;  // non-finite data: keep the group
            grp[g] = make_float4(cx, cy, cz, R2);
        }
        return;
    }
    for (int f = tid; f < n; f += 1024) {
        const float4 r = crec[f];
        const int s = (int)atomicAdd(&hist[cell_of(r)], 1u);
        p0s[s] = r;
        idx[s] = f;
    }
    for (int s = n + tid; s < ng * GRP; s += 1024) {  // pad: thr2 = 0 never passes
        p0s[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        idx[s] = 0;
    }
    __syncthreads();  // the block's own global stores are visible to it after the barrier

    // ---- group spheres: one lane per group, records re-read from P0S (256 contiguous bytes each)
    for (int g = tid; g < ng; g += 1024) {
        const float4 *r = p0s + (size_t)g * GRP;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, tm = 0.0f;
        float px[GRP], py[GRP], pz[GRP];
#pragma unroll
        for (int t = 0; t < GRP; ++t) {
            const float4 v = r[t];
            px[t] = v.x; py[t] = v.y; pz[t] = v.z;
            if (g * GRP + t < n) {
                lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
                lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
                lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
                tm = fmaxf(tm, thr_bound(v.w));
            }
        }
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1],
                    cz = 0.5f * lo[2] + 0.5f * hi[2];
        float d2 = 0.0f;
#pragma unroll
        for (int t = 0; t < GRP; ++t) {
            const float ex = px[t] - cx, ey = py[t] - cy, ez = pz[t] - cz;
            const float e2 = ex * ex + ey * ey + ez * ez;
            if (g * GRP + t < n) d2 = fmaxf(d2, e2);
        }
        const float rho = sqrtf(d2) * 1.00001f + 1e-7f;
        const float R = rho + tm;
        float R2 = R * R * 1.0001f + 1e-7f;
        if (!(R2 < 3.0e38f)) R2 = INFINITY;  // non-finite data: keep the group
        grp[g] = make_float4(cx, cy, cz, R2);
    }
}

// ---------------------------------------------------------------------------------------
// Large clouds (n > 4096): the same sort in three WIDE launches -- one workgroup per cloud is
// bound by a single CU (48 us for 16384 triangles), and a launch boundary costs ~3 us:
//   big_hist_kernel     cell of every triangle -> global histogram (atomics), max |P|^2
//   big_scatter_kernel  every workgroup scans the 4096 bins itself (16 KiB), then places its
//                       triangles at cell base + a global per-cell cursor (atomic)
//   big_sphere_kernel   one lane per group of 16 sorted records
// HISTG = [2B][2][4096] ints (counts, cursors), cleared by tri_records_kernel.
// ---------------------------------------------------------------------------------------
struct CellGrid {
    float mn[3], scale[3], p2;
};

// AABB / max |P|^2 of one cloud from the per-workgroup partials; every lane gets the result.
// red: LDS [4][8]; blockDim = 256.
__device__ __forceinline__ CellGrid load_grid(const BuildArgs &a, int cloud, int b, int n, float (*red)[8]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + REC_BLK - 1) / REC_BLK;
    const float *ap = a.apart + ((size_t)cloud * a.B + b) * a.nblk * 8;
    float v[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) v[c] = c < 3 ? INFINITY : (c < 6 ? -INFINITY : 0.0f);
    for (int j = tid; j < nb; j += 256)
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = c < 3 ? fminf(v[c], ap[j * 8 + c]) : fmaxf(v[c], ap[j * 8 + c]);
#pragma unroll
    for (int c = 0; c < 7; ++c) v[c] = c < 3 ? wave_min(v[c]) : wave_max(v[c]);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 7; ++c) red[wave][c] = v[c];
    __syncthreads();
    CellGrid g;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g.mn[c] = fminf(fminf(red[0][c], red[1][c]), fminf(red[2][c], red[3][c]));
        const float hi = fmaxf(fmaxf(red[0][3 + c], red[1][3 + c]), fmaxf(red[2][3 + c], red[3][3 + c]));
        const float ext = hi - g.mn[c];
        g.scale[c] = ext > 0.0f && ext < 3.0e38f ? 15.999f / ext : 0.0f;
    }
    g.p2 = fmaxf(fmaxf(red[0][6], red[1][6]), fmaxf(red[2][6], red[3][6]));
    return g;
}

__device__ __forceinline__ unsigned grid_cell(const CellGrid &g, const float4 r) {
    const float p[3] = {r.x, r.y, r.z};
    unsigned q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (p[c] - g.mn[c]) * g.scale[c];
        q[c] = v >= 15.0f ? 15u : (v > 0.0f ? (unsigned)v : 0u);
    }
    return hilbert_cell(q[0], q[1], q[2]);  // 12 bits
}

__global__ __launch_bounds__(256) void big_hist_kernel(const BuildArgs a, unsigned *__restrict__ histg) {
    __shared__ float red[4][8];
    const int cloud = blockIdx.z, b = blockIdx.y;
    const int n = cloud ? a.M : a.N;
    if ((int)blockIdx.x * 256 >= n) return;
    const int ng = (n + GRP - 1) / GRP;
    const float4 *crec = (cloud ? a.crec2 : a.crec1) + (size_t)b * ng * GRP;
    const CellGrid g = load_grid(a, cloud, b, n, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.pmax[cloud * a.B + b] = __float_as_uint(g.p2);
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < n) atomicAdd(&histg[((size_t)(cloud * a.B + b) * 2) * SORT_CELLS + grid_cell(g, crec[f])], 1u);
}

__global__ __launch_bounds__(256) void big_scatter_kernel(const BuildArgs a, unsigned *__restrict__ histg) {
    __shared__ float red[4][8];
    __shared__ unsigned base[SORT_CELLS];
    __shared__ unsigned wsum[4];
    const int cloud = blockIdx.z, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = cloud ? a.M : a.N;
    if ((int)blockIdx.x * 256 >= n) return;
    const int ng = (n + GRP - 1) / GRP;
    const float4 *crec = (cloud ? a.crec2 : a.crec1) + (size_t)b * ng * GRP;
    float4 *p0s = (cloud ? a.p0s2 : a.p0s1) + (size_t)b * ng * GRP;
    int32_t *idx = (cloud ? a.idx2 : a.idx1) + (size_t)b * ng * GRP;
    unsigned *cnt = histg + ((size_t)(cloud * a.B + b) * 2) * SORT_CELLS, *cur = cnt + SORT_CELLS;
    const int f = blockIdx.x * 256 + tid;
    const float4 r = f < n ? crec[f] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // in flight during the scan
    {   // exclusive scan of the 4096 counts (16 per lane)
        unsigned h[16], tsum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { h[k] = cnt[16 * tid + k]; tsum += h[k]; }
        const unsigned inc = (unsigned)wave_incl_scan((int)tsum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned run = inc - tsum;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int k = 0; k < 16; ++k) { base[16 * tid + k] = run; run += h[k]; }
    }
    const CellGrid g = load_grid(a, cloud, b, n, red);  // its barrier also publishes base[]
    if (f < n) {
        const unsigned c = grid_cell(g, r);
        const unsigned s = base[c] + atomicAdd(&cur[c], 1u);
        p0s[s] = r;
        idx[s] = f;
    }
    if (blockIdx.x == 0)  // pad: thr2 = 0 never passes
        for (int s = n + tid; s < ng * GRP; s += 256) { p0s[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); idx[s] = 0; }
}

__global__ __launch_bounds__(256) void big_sphere_kernel(const BuildArgs a) {
    const int cloud = blockIdx.z, b = blockIdx.y;
    const int n = cloud ? a.M : a.N;
    const int ng = (n + GRP - 1) / GRP;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    const float4 *r = (cloud ? a.p0s2 : a.p0s1) + ((size_t)b * ng + g) * GRP;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, tm = 0.0f;
    float px[GRP], py[GRP], pz[GRP];
#pragma unroll
    for (int t = 0; t < GRP; ++t) {
        const float4 v = r[t];
        px[t] = v.x; py[t] = v.y; pz[t] = v.z;
        if (g * GRP + t < n) {
            lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
            lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
            lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
            tm = fmaxf(tm, sqrtf(v.w) * 1.000001f);  // thr <= sqrtf(thr2)
        }
    }
    const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
    float d2 = 0.0f;
#pragma unroll
    for (int t = 0; t < GRP; ++t) {
        const float ex = px[t] - cx, ey = py[t] - cy, ez = pz[t] - cz;
        const float e2 = ex * ex + ey * ey + ez * ez;
        if (g * GRP + t < n) d2 = fmaxf(d2, e2);
    }
    const float rho = sqrtf(d2) * 1.00001f + 1e-7f;
    const float R = rho + tm;
    float R2 = R * R * 1.0001f + 1e-7f;
    if (!(R2 < 3.0e38f)) R2 = INFINITY;  // non-finite data: keep the group
    ((cloud ? a.grp2 : a.grp1) + (size_t)b * ng)[g] = make_float4(cx, cy, cz, R2);
}

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

#ifndef BGRP
#define BGRP 12    // groups per batch (<= 16: 4-bit slot in an entry): their 16 x 256-byte rows are staged in the wave's LDS
#endif
#ifndef WCCAP
#define WCCAP 256  // parked point-0 candidates per wave (1 KiB)
#endif
#ifndef WPB
#define WPB 8      // wavefronts per workgroup: they split the group range of the block's lines
#endif
#define ROWS 17    // float4 per staged row: 16 records + 16 bytes of padding (bank spread)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

struct WaveCtx {
    const float4 *lines;          // this wave's 64 lines in LDS: [64][2]
    const float4 *p0s;            // sorted (P0, thr2) records of the cloud
    const int32_t *idx;           // sorted position -> original triangle index
    const float *ptri;            // prepared triangles (original order)
    int32_t *cnt, *hit;           // per-line hit count / slots of the cloud
    int lbase;                    // first line of this wave
    float4 *rows;                 // LDS [BGRP][16] staged records of the current batch
    unsigned short *ent;          // LDS [64] entries of the current batch: line << 4 | batch slot
    int *bgrp;                    // LDS [BGRP] group index of each batch slot
    unsigned *cands;              // LDS [WCCAP]
};

// cand = line_in_wave << 16 | sorted triangle position: evaluate points 1 and 2
__device__ __forceinline__ void resolve_candidate(const WaveCtx &c, unsigned cand) {
    const int ll = cand >> 16, spos = cand & 0xffff;
    const float4 la = c.lines[2 * ll], lb = c.lines[2 * ll + 1];
    const int f = c.idx[spos];
    const float *q = c.ptri + PTRI_STRIDE * (size_t)f;
    const uint32_t thr2 = __float_as_uint(q[9]);
    const float x1 = dist_sq<float>(q[3], q[4], q[5], la.x, la.y, la.z, la.w, lb.x, lb.y);
    const float x2 = dist_sq<float>(q[6], q[7], q[8], la.x, la.y, la.z, la.w, lb.x, lb.y);
    if (max(__float_as_uint(x1), __float_as_uint(x2)) < thr2) {
        const int l = c.lbase + ll;
        int pos = atomicAdd(&c.cnt[l], 1);
        if (pos < RRL_MAX_HITS) c.hit[(size_t)l * RRL_MAX_HITS + pos] = f;
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS is processed in order per wave; this only stops the compiler from reordering across it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Phase 2 for one batch of a wave: <= EB (line, group) pairs over <= BGRP groups, listed
// group-major in c.ent (phase 1 writes every pair at its final position).  The groups' records
// were copied, coalesced, into the wave's LDS rows (row stride padded so lanes on different
// groups hit different banks); every lane runs the exact point-0 test of one pair on the 16
// triangles.  Point-0 passes (~1 in 130 tests) are parked; their points 1, 2 need two dependent
// global loads and are resolved densely when enough have gathered.
#ifndef EB
#define EB 128  // entries per batch (two passes of 64 lanes)
#endif

__device__ __forceinline__ void run_batch(const WaveCtx &c, int nent, int &ncand, int lane) {
    // the rows of this batch were requested (global -> LDS DMA) as their groups were appended in
    // phase 1; wait for the stragglers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_fence();
#ifdef CULL_SKIP_PHASE2  // timing experiment only: batches are formed but not evaluated
    return;
#endif
    for (int e0 = 0; e0 < nent; e0 += 64) {  // one pass unless a group is hit by > 64 of the lines
        uint32_t passbits = 0;
        int ll = 0, g = 0;
        if (e0 + lane < nent) {
            const unsigned e = c.ent[e0 + lane];
            ll = e >> 4;
            const int k = e & 15;
            g = c.bgrp[k];
            const float4 la = c.lines[2 * ll], lb = c.lines[2 * ll + 1];
            const float4 *row = c.rows + k * ROWS;
#pragma unroll
            for (int t = 0; t < GRP; ++t) {
                const float4 rec = row[t];
                const float x = dist_sq<float>(rec.x, rec.y, rec.z, la.x, la.y, la.z, la.w, lb.x, lb.y);
                passbits |= (__float_as_uint(x) < __float_as_uint(rec.w) ? 1u : 0u) << t;
            }
        }
        while (__any(passbits != 0)) {
            const bool has = passbits != 0;
            const unsigned long long m = __ballot(has);
            const int t = has ? __ffs(passbits) - 1 : 0;
            passbits &= passbits - 1;
            const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
            const unsigned cand = ((unsigned)ll << 16) | (unsigned)(g * GRP + t);
            if (has) {
                if (pos < WCCAP) c.cands[pos] = cand;
                else resolve_candidate(c, cand);
            }
            ncand += __popcll(m);
        }
        if (ncand > WCCAP - 64) {  // uniform: keep room for the next pass
            wave_lds_fence();
            const int nc = min(ncand, WCCAP);
            for (int i = lane; i < nc; i += 64) resolve_candidate(c, c.cands[i]);
            ncand = 0;
        }
    }
}

// LPB lines per workgroup (two per lane: packed fp32 in phase 1, and the wave-uniform work of the
// group loop -- scalar loads, ballots, batch bookkeeping -- is shared by 128 lines).  The four
// wavefronts hold the SAME lines and split the GROUP range between them, which quadruples the
// number of independent (latency-bound) wavefronts; each has private LDS queues, so there is no
// workgroup synchronisation at all.
#define LPB 128

__global__ __launch_bounds__(64 * WPB) void cull_scan_kernel(
    const float *__restrict__ ptri1, const float *__restrict__ ptri2, const float4 *__restrict__ p0s1,
    const float4 *__restrict__ p0s2, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    const float4 *__restrict__ grp1, const float4 *__restrict__ grp2, const float *__restrict__ line,
    int32_t *__restrict__ count1, int32_t *__restrict__ hit1, int32_t *__restrict__ count2,
    int32_t *__restrict__ hit2, int32_t *__restrict__ status, const uint32_t *__restrict__ pmax, int B,
    int N, int M, int L) {
    __shared__ __attribute__((aligned(16))) float4 lines_lds[LPB][2];      // 4 KiB, same in all 4 waves
    __shared__ __attribute__((aligned(16))) float4 rows_lds[WPB][BGRP * ROWS];  // 17 KiB
    __shared__ unsigned cands_lds[WPB][WCCAP];                             // 4 KiB
    __shared__ unsigned short ent_lds[WPB][LPB];
    __shared__ int bgrp_lds[WPB][BGRP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler
    // XCD-aware mapping: workgroups go to the 8 XCDs round-robin by linear id, and x is the fast
    // index -- with (cloud, sample) on x, all workgroups of one cloud land on the same XCD (when
    // 2B is a multiple of 8), so each XCD's L2 holds 1/8 of the records instead of a copy of all
    const int z = blockIdx.x, cloud = z >= B ? 1 : 0, b = z - cloud * B;
    const int lblk = blockIdx.y;
    const int n = cloud ? M : N;
    const int ng = (n + GRP - 1) / GRP;
    const float4 *grp = (cloud ? grp2 : grp1) + (size_t)b * ng;
    const float *ln = line + (size_t)b * L * 6;

    const int l0 = lblk * LPB + lane, l1 = l0 + 64;
    const bool live0 = l0 < L, live1 = l1 < L;
    float v0[6], v1[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        v0[c] = live0 ? ln[6 * (size_t)l0 + c] : 0.0f;
        v1[c] = live1 ? ln[6 * (size_t)l1 + c] : 0.0f;
    }
    // every wavefront stores the same lines: identical values, so no barrier is needed
    lines_lds[lane][0] = make_float4(v0[0], v0[1], v0[2], v0[3]);
    lines_lds[lane][1] = make_float4(v0[4], v0[5], 0.0f, 0.0f);
    lines_lds[64 + lane][0] = make_float4(v1[0], v1[1], v1[2], v1[3]);
    lines_lds[64 + lane][1] = make_float4(v1[4], v1[5], 0.0f, 0.0f);

    // Culling (and the lazy evaluation of points 1, 2) is only exact for lines that satisfy the
    // NaN-impossibility bound.  All four wavefronts hold the same lines, so the vote is wave-local
    // and uniform over the workgroup: a block with an offending line evaluates ALL its
    // (line, triangle) pairs strictly instead -- the reference's semantics, NaN included.
    const float pm = __uint_as_float(pmax[cloud * B + b]);
    if (!__all(rrl_line_safe(v0, pm) && rrl_line_safe(v1, pm))) {
        kptr tp = (kptr)(uintptr_t)((cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE);
        int32_t *cnt = (cloud ? count2 : count1) + (size_t)b * L;
        int32_t *hit = (cloud ? hit2 : hit1) + (size_t)b * L * RRL_MAX_HITS;
        // this workgroup's slice of the triangles (gridDim.z slices), split over its wavefronts
        const int nsl = WPB * (int)gridDim.z, sl = (int)blockIdx.z * WPB + wave;
        const int tq = (n + nsl - 1) / nsl, t0 = min(n, sl * tq), t1 = min(n, t0 + tq);
        uint32_t nanacc = 0;
        tp += (size_t)t0 * PTRI_STRIDE;
        for (int t = t0; t < t1; ++t, tp += PTRI_STRIDE) {
            const uint32_t thr2 = __float_as_uint(tp[9]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *v = h ? v1 : v0;
                const int l = h ? l1 : l0;
                const uint32_t x0 = __float_as_uint(dist_sq<float>(tp[0], tp[1], tp[2], v[0], v[1], v[2], v[3], v[4], v[5]));
                const uint32_t x1 = __float_as_uint(dist_sq<float>(tp[3], tp[4], tp[5], v[0], v[1], v[2], v[3], v[4], v[5]));
                const uint32_t x2 = __float_as_uint(dist_sq<float>(tp[6], tp[7], tp[8], v[0], v[1], v[2], v[3], v[4], v[5]));
                const uint32_t mm = max(max(x0, x1), x2);  // negative or NaN: sign bit set -> huge
                if (l < L) {
                    nanacc = max(nanacc, mm);
                    if (mm < thr2) {
                        const int pos = atomicAdd(&cnt[l], 1);
                        if (pos < RRL_MAX_HITS) hit[(size_t)l * RRL_MAX_HITS + pos] = __float_as_int(tp[11]);
                    }
                }
            }
        }
        if (nanacc >= 0x80000000u) atomicOr(&status[0], 1);
        return;
    }

    WaveCtx ctx;
    ctx.lines = &lines_lds[0][0];
    ctx.p0s = (cloud ? p0s2 : p0s1) + (size_t)b * ng * GRP;
    ctx.idx = (cloud ? idx2 : idx1) + (size_t)b * ng * GRP;
    ctx.ptri = (cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE;
    ctx.cnt = (cloud ? count2 : count1) + (size_t)b * L;
    ctx.hit = (cloud ? hit2 : hit1) + (size_t)b * L * RRL_MAX_HITS;
    ctx.lbase = lblk * LPB;
    ctx.rows = rows_lds[wave];
    ctx.ent = ent_lds[wave];
    ctx.bgrp = bgrp_lds[wave];
    ctx.cands = cands_lds[wave];

    // ---- phase 1: conservative sphere test of every group against the lane's two lines (packed
    //      fp32); passing (line, group) pairs are collected by ballot/popcount into batches that
    //      phase 2 consumes at once
    const v2f ux = {v0[0], v1[0]}, uy = {v0[1], v1[1]}, uz = {v0[2], v1[2]};
    const v2f ox = {v0[3], v1[3]}, oy = {v0[4], v1[4]}, oz = {v0[5], v1[5]};
    int nent = 0, ngrp = 0, ncand = 0;  // wave-uniform
    kptr gp = (kptr)(uintptr_t)grp;
    // this workgroup's slice of the groups (gridDim.z slices: few lines against a big cloud would
    // otherwise leave most CUs idle), split over its wavefronts
    const int nsl = WPB * (int)gridDim.z, sl = (int)blockIdx.z * WPB + wave;
    const int gq = (ng + nsl - 1) / nsl;
    const int gbeg = min(ng, sl * gq), gend = min(ng, gbeg + gq);
    for (int g = gbeg; g < gend; ++g) {
        const float cx = gp[4 * g], cy = gp[4 * g + 1], cz = gp[4 * g + 2], R2 = gp[4 * g + 3];
        const v2f ax = cx - ox, ay = cy - oy, az = cz - oz;
        const v2f dot = __builtin_elementwise_fma(az, uz, __builtin_elementwise_fma(ay, uy, ax * ux));
        const v2f q = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
        v2f d2 = __builtin_elementwise_fma(-dot, dot, q);
        d2 = __builtin_elementwise_fma((v2f){-4e-6f, -4e-6f}, q, d2);
        const bool pass0 = live0 && d2.x <= R2, pass1 = live1 && d2.y <= R2;
        const unsigned long long m0 = __ballot(pass0), m1 = __ballot(pass1);
        if (m0 | m1) {
            const int c0 = __popcll(m0), c = c0 + __popcll(m1);
            if (nent + c > EB || ngrp == BGRP) {  // uniform: the batch is full
                run_batch(ctx, nent, ncand, lane);
                nent = ngrp = 0;
            }
            // every passing (line, group) pair goes straight to its slot of the batch's entry
            // list: group-major, lines in lane order (rank among the passing lanes by mbcnt)
            if (pass0) ctx.ent[nent + __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] =
                (unsigned short)((lane << 4) | ngrp);
            if (pass1) ctx.ent[nent + c0 + __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] =
                (unsigned short)(((64 + lane) << 4) | ngrp);
            if (lane == 0) ctx.bgrp[ngrp] = g;
            // asynchronous copy of the group's 16 records into LDS row `ngrp` (lanes 0..15, 16 B
            // each; the DMA writes wave-uniform base + lane * 16)
            if (lane < 16)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(ctx.p0s + (size_t)g * GRP + lane),
                                                 (lds_void_t *)(ctx.rows + ngrp * ROWS), 16, 0, 0);
            nent += c;
            ++ngrp;
        }
    }
    if (nent) run_batch(ctx, nent, ncand, lane);
    wave_lds_fence();
    const int nc = min(ncand, WCCAP);
    for (int i = lane; i < nc; i += 64) resolve_candidate(ctx, ctx.cands[i]);
}

// Launchers used by rrl_tri_prepare / rrl_line_tri_scan (rrl_scan.hip)
int rrl_launch_tri_build(const float *tri1, const float *tri2, void *ws, const WsLayout &w, int B,
                         int N, int M, int clouds, const RrlXform *xf, hipStream_t s) {
    const int nmax = clouds == 2 && M > N ? M : N;
    const size_t ngmax = (size_t)(nmax + GRP - 1) / GRP;
    const size_t lds = nmax <= 4096 ? ngmax * (17 * sizeof(float4) + GRP * sizeof(int)) : 16;
    BuildArgs a;
    a.tri1 = xf ? xf->src : tri1;
    a.tri2 = tri2;
    a.R = xf ? xf->R : nullptr;
    a.t = xf ? xf->t : nullptr;
    a.tri1_out = xf ? w.f32(ws, RRL_WS_TRI1) : nullptr;
    a.ptri1 = w.f32(ws, RRL_WS_PTRI1);
    a.ptri2 = w.f32(ws, RRL_WS_PTRI2);
    a.crec1 = (float4 *)w.f32(ws, RRL_WS_CREC1);
    a.crec2 = (float4 *)w.f32(ws, RRL_WS_CREC2);
    a.apart = w.f32(ws, RRL_WS_APART);
    a.p0s1 = (float4 *)w.f32(ws, RRL_WS_P0S1);
    a.p0s2 = (float4 *)w.f32(ws, RRL_WS_P0S2);
    a.idx1 = w.i32(ws, RRL_WS_IDX1);
    a.idx2 = w.i32(ws, RRL_WS_IDX2);
    a.grp1 = (float4 *)w.f32(ws, RRL_WS_GRP1);
    a.grp2 = (float4 *)w.f32(ws, RRL_WS_GRP2);
    a.pmax = (uint32_t *)w.i32(ws, RRL_WS_PMAX);
    a.zero_base = (uint4 *)((char *)ws + w.off[RRL_WS_STATUS]);
    a.zero_vec4 = w.zero_bytes / 16;
    a.g1 = xf && xf->zero_g1 ? (uint4 *)((char *)ws + w.off[RRL_WS_GACC]) : nullptr;  // small: 12 B + 16 floats
    a.g1_vec4 = a.g1 ? (w.off[RRL_WS_KJC] - w.off[RRL_WS_GACC]) / 16 : 0;
    a.z2 = nmax > 4096 ? (uint4 *)((char *)ws + w.off[RRL_WS_HISTG]) : nullptr;
    a.z2_vec4 = a.z2 ? (size_t)2 * B * 2 * SORT_CELLS * sizeof(unsigned) / 16 : 0;
    a.B = B; a.N = N; a.M = M;
    a.transpose_r = xf ? xf->transpose_r : 0;
    const int nall = N > M ? N : M;  // APART is laid out for the larger cloud
    a.nblk = (nall + REC_BLK - 1) / REC_BLK;
    hipLaunchKernelGGL(tri_records_kernel, dim3((unsigned)((nmax + REC_BLK - 1) / REC_BLK), (unsigned)B, (unsigned)clouds),
                       dim3(REC_BLK), 0, s, a);
    if (nmax <= 4096) {
        hipLaunchKernelGGL(tri_sort_kernel<4>, dim3((unsigned)(clouds * B)), dim3(1024), lds, s, a);
    } else {  // wide three-launch sort (HISTG was cleared by tri_records_kernel)
        unsigned *histg = (unsigned *)w.i32(ws, RRL_WS_HISTG);
        const dim3 gt((unsigned)((nmax + 255) / 256), (unsigned)B, (unsigned)clouds);
        hipLaunchKernelGGL(big_hist_kernel, gt, dim3(256), 0, s, a, histg);
        hipLaunchKernelGGL(big_scatter_kernel, gt, dim3(256), 0, s, a, histg);
        const dim3 gs((unsigned)(((nmax + GRP - 1) / GRP + 255) / 256), (unsigned)B, (unsigned)clouds);
        hipLaunchKernelGGL(big_sphere_kernel, gs, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_launch_cull_scan(const float *line, void *ws, const WsLayout &w, int B, int N, int M, int L,
                         int clouds, hipStream_t s) {
    // group slices: enough workgroups for ~4 per CU, at least 4 groups per wavefront
    const int tiles = (L + LPB - 1) / LPB, nmax = clouds == 2 && M > N ? M : N;
    const int ngmax = (nmax + GRP - 1) / GRP;
    int gsplit = (1024 + tiles * clouds * B - 1) / (tiles * clouds * B);
    const int cap = ngmax / (4 * WPB);
    if (gsplit > cap) gsplit = cap;
    if (gsplit < 1) gsplit = 1;
    hipLaunchKernelGGL(cull_scan_kernel, dim3((unsigned)(clouds * B), (unsigned)tiles, (unsigned)gsplit), dim3(64 * WPB), 0,
                       s, w.f32(ws, RRL_WS_PTRI1), w.f32(ws, RRL_WS_PTRI2),
                       (const float4 *)w.f32(ws, RRL_WS_P0S1), (const float4 *)w.f32(ws, RRL_WS_P0S2),
                       w.i32(ws, RRL_WS_IDX1), w.i32(ws, RRL_WS_IDX2), (const float4 *)w.f32(ws, RRL_WS_GRP1),
                       (const float4 *)w.f32(ws, RRL_WS_GRP2), line, w.i32(ws, RRL_WS_COUNT1),
                       w.i32(ws, RRL_WS_HIT1), w.i32(ws, RRL_WS_COUNT2), w.i32(ws, RRL_WS_HIT2),
                       w.i32(ws, RRL_WS_STATUS), (const uint32_t *)w.i32(ws, RRL_WS_PMAX), B, N, M, L);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_sort_capacity(void) { return SORT_CAP; }
