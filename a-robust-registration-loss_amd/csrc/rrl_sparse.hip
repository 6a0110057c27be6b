// rrl_sparse.hip -- everything after the dense scan, plus the fused forward/backward entries.
//   K2 line_pair_dist   code/loss.py:115-167   (one lane per line, <= 4x4 block in registers)
//   K3+K4 loss_reduce   code/loss.py:223-230   (radix-select median + Welsch min/mean, one
//                                               workgroup per sample, fixed-point bucket sums)
//   K5 backward         autograd of code/loss.py:170-232
// About 9 % of the lines are selected; these kernels touch O(L) data and are latency bound
// (a few microseconds each) next to the O(L*(N+M)) scan.
#include <stdlib.h>
#include <string.h>

#include "rrl_ws.h"

#define FIX_SHIFT 40  // bucket sums in 2^-40 fixed point: order-independent, bit-deterministic

// sqrt(dist_sq) of the three points of triangle f and the detached weights of
// code/loss.py:92: w_k = d_k / ((d0 + d1) + d2).  Same arithmetic as the scan, so the
// distances are bit-identical to the ones that decided the label.
#ifdef RRL_STAMPS  // experiments only (RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp): 100 MHz time stamps of workgroup 0's lane 0
__device__ unsigned long long g_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_stamps[i] = wall_clock64(); } while (0)
extern "C" int rrl_debug_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_wstamps[12 * 2048];  // [stamp][workgroup]: per-workgroup stamps of the tail kernel
#define STAMPW(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 2048u) g_wstamps[(i) * 2048 + wg_] = wall_clock64(); } } while (0)
extern "C" int rrl_debug_wstamps(unsigned long long *out, int clear) {
    if (clear) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_wstamps)) != hipSuccess) return -1; return hipMemset(p_, 0, sizeof(unsigned long long) * 12 * 2048) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * 12 * 2048) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_pstamps[8 * 2048];  // ... of the per-line stage
#define STAMPP(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 2048u) g_pstamps[(i) * 2048 + wg_] = wall_clock64(); } } while (0)
extern "C" int rrl_debug_pstamps(unsigned long long *out, int clear) {
    if (clear) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_pstamps)) != hipSuccess) return -1; return hipMemset(p_, 0, sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1;
}
#define STAMPC(i) STAMPP(i)  // (the sampler's count pass, rrl_sampler.h: stamps 6, 7 of the same table)
#else
#define STAMP(i)
#define STAMPW(i)
#define STAMPP(i)
#endif
#include "rrl_sampler.h"  // the sampler's count pass as a device function: pair_count_kernel carries it

// LDS-only workgroup barrier: this wavefront's LDS traffic is complete, its vector-memory loads AND STORES stay in flight
// (__syncthreads() waits for both: a barrier behind a store costs the store's acknowledgement, ~0.5 us)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void hit_weights(const float *p, const float *ln, float *w) {
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
__device__ __forceinline__ void inter_point(const float *p, const float *w, float *q) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = w[0] * p[c];
        s = s + w[1] * p[3 + c];
        s = s + w[2] * p[6 + c];
        q[c] = s / 3.0f;
    }
}

// the 9 coordinates of triangle f from its 48-byte prepared record: three 16-byte loads instead
// of nine 4-byte gathers from the 36-byte input rows (a wavefront-level gather costs ~64 cycles
// of the CU's address path per instruction, whatever its width)
__device__ __forceinline__ void tri_coords(const float *__restrict__ ptri, int stride, int f, float *c) {
    if (stride != PTRI_STRIDE) {  // raw 36-byte rows (what the per-line stage reads since round 4)
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] = ptri[9 * (size_t)f + i];
        return;
    }
    const float4 *row = (const float4 *)(ptri + PTRI_STRIDE * (size_t)f);
    const float4 r0 = row[0], r1 = row[1], r2 = row[2];
    c[0] = r0.x; c[1] = r0.y; c[2] = r0.z; c[3] = r0.w;
    c[4] = r1.x; c[5] = r1.y; c[6] = r1.z; c[7] = r1.w;
    c[8] = r2.x;
}

__device__ __forceinline__ void sort4(int *h, int n) {  // ascending, n <= 4
#pragma unroll
    for (int i = 1; i < RRL_MAX_HITS; ++i)
#pragma unroll
        for (int j = RRL_MAX_HITS - 1; j >= i; --j)
            if (j < n && h[j] < h[j - 1]) { int t = h[j]; h[j] = h[j - 1]; h[j - 1] = t; }
}

// Scatter of a wavefront's gradient rows (round 5).  Every LIVE lane owns nine products that go to nine CONSECUTIVE floats
// of its triangle's gradient row [9].  One lane per row meant nine instructions of 64 scattered 4-byte atomics -- 64 cache
// lines per instruction through the CU's address path: +4.5 us on the C2 tail kernel against the (dR, dt) variant, 53 us for
// loss_bwd_kernel at B = 64.  Here the rows meet in the wavefront's LDS strip, compacted by rank among the live lanes, and
// lane p of a round adds element (p % 9) of row (p / 9): one instruction covers seven whole rows (8 .. 14 cache lines),
// ceil(9 nlive / 64) instructions per wavefront instead of nine.  The products are computed by the owning lane exactly as
// before; only WHICH lane issues an atomic changes (float atomics: the sums agree to their rounding order, as before).
// rowkey = triangle index | (cloud 2 ? 1u << 31 : 0); g1b / g2b: the sample's gradient rows [n][9]; strip: 64 x 10 words.
#define SCAT_STRIDE 10
// fx1b / fx2b != NULL (deterministic mode, include/rrl.h rrl_set_deterministic): the element goes to the sample's 64-bit
// fixed-point accumulators instead -- llrint(value * inv_unit), inv_unit a power of two: integer atomics commute, so the sums
// do not depend on the order of arrival; a non-finite value raises *nonfinite (the conversion then writes NaN rows).
__device__ __forceinline__ void wave_scatter_rows(bool live, const float (&v)[9], unsigned rowkey, float *__restrict__ g1b,
                                                  float *__restrict__ g2b, unsigned *strip, int lane,
                                                  unsigned long long *fx1b = nullptr, unsigned long long *fx2b = nullptr,
                                                  double inv_unit = 0.0, int32_t *nonfinite = nullptr) {
    const unsigned long long mask = __ballot(live);
    const int nlive = __popcll(mask);
    if (nlive == 0) return;  // uniform
    if (live) {
        unsigned *row = strip + __popcll(mask & ((1ull << lane) - 1ull)) * SCAT_STRIDE;
        row[0] = rowkey;
#pragma unroll
        for (int q = 0; q < 9; ++q) row[1 + q] = __float_as_uint(v[q]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wavefront: its LDS traffic is in order
    const int total = nlive * 9;
    for (int p = lane; p < total; p += 64) {
        const int rk = p / 9, e = p - rk * 9;
        const unsigned key = strip[rk * SCAT_STRIDE];
        const float val = __uint_as_float(strip[rk * SCAT_STRIDE + 1 + e]);
        if (fx1b) {  // uniform
            if (!(fabsf(val) < INFINITY)) { atomicOr(&nonfinite[key >> 31], 1); continue; }
            const long long q = __double2ll_rn((double)val * inv_unit);
            if (q) atomicAdd(((key >> 31) ? fx2b : fx1b) + (size_t)(key & 0x7fffffffu) * 9 + e, (unsigned long long)q);
            continue;
        }
        float *dst = ((key >> 31) ? g2b : g1b) + (size_t)(key & 0x7fffffffu) * 9 + e;
        atomicAdd(dst, val);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the strip is free for the wavefront's next pass
}

// Phase 2 of line_pair_dist_kernel: EIGHT lanes per selected line, one per (cloud, hit slot) --
// a selected line has up to 4 + 4 hits and one lane doing them in turn was the kernel's long pole
// (8 us of 13).  Each lane gathers its triangle (three 16-byte loads of the prepared record, or
// the raw row with stride 9), recomputes the hit distances with the scan's arithmetic, the
// weights and the intersection point, and stores them; the eight lanes then exchange their
// points through LDS (same wavefront: no barrier) and each fills two entries of the k x j block
// of squared distances -- by line for the backward kernels and as the canonical 4 x 4 tile
// (+inf outside the block) at the line's compact slot for the reduce kernel.
// What a lane of the per-line stage knows about its (selected line, cloud, hit slot) after the FIRST pass (ranks 0 .. 127) --
// kept in registers for the single-tile kernels, whose backward then reads nothing of it back (pair_reduce_scatter_kernel).
struct PairKeep {
    int total;   // the tile's selected lines (uniform)
    int k, j;    // the line's hit counts (0, 0: this lane's rank holds no line)
    int f;       // this lane's triangle (hit slot a = lane & 3 of cloud (lane >> 2) & 1, ascending order), when a < count
    float w[3];  // its weights
    float4 q;    // its intersection point
    float4 *sq;  // LDS: the 8 intersection points of the line (cloud 1: 0..3, cloud 2: 4..7)
    // in: LDS the caller provides for the first pass's compact rows (what the reduce reads back: the (k | j << 4) byte and
    // the canonical 4 x 4 D tile of ranks 0 .. 127), or NULL
    uint8_t *kjc_lds;  // [128]
    float *dc_lds;     // [128][16]
};

__device__ __forceinline__ void pair_hit(const float *__restrict__ tri1, const float *__restrict__ tri2,
                                         const float *__restrict__ line, const int32_t *__restrict__ hit1,
                                         const int32_t *__restrict__ hit2, int32_t *__restrict__ hs1,
                                         int32_t *__restrict__ hs2, float *__restrict__ w1,
                                         float *__restrict__ w2, float4 *__restrict__ Q1,
                                         float4 *__restrict__ Q2, float *__restrict__ D,
                                         float *__restrict__ dc_slot, float4 *s_q /* LDS [8] of this line */,
                                         unsigned *s_mh /* LDS [2048] or NULL */,
                                         int b, int N, int M, size_t gl, int k, int j, int sub, int st1,
                                         int st2, float *t2 /* out: this lane's two tile entries (+inf: outside the block) */,
                                         int bi, size_t gli /* multi-pose: the instance's problem, its line's row there */,
                                         PairKeep *keep = nullptr) {
    const int cloud = sub >> 2, a = sub & 3;
    const int cnt = cloud ? j : k;
    float q[3] = {0.0f, 0.0f, 0.0f};
    if (a < cnt) {
        float ln[6];
        {
            const float2 *lp = (const float2 *)(line + gli * 6);  // 24-byte rows: 8-byte aligned
            const float2 a0 = lp[0], a1 = lp[1], a2 = lp[2];
            ln[0] = a0.x; ln[1] = a0.y; ln[2] = a1.x; ln[3] = a1.y; ln[4] = a2.x; ln[5] = a2.y;
        }
        int h[RRL_MAX_HITS];
        {
            const int4 r = cloud ? ((const int4 *)hit2)[gli] : ((const int4 *)hit1)[gl];  // (cloud 2 was scanned by its problem's first instance)
            const int rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int t = 0; t < RRL_MAX_HITS; ++t) h[t] = t < cnt ? rr[t] : 0x7fffffff;
        }
        sort4(h, cnt);  // ascending triangle index == nonzero() order (code/loss.py:125-131)
        const int f = a == 0 ? h[0] : (a == 1 ? h[1] : (a == 2 ? h[2] : h[3]));
        const float *tb = cloud ? tri2 + (size_t)bi * M * st2 : tri1 + (size_t)b * N * st1;
        float w[3], c[9];
        tri_coords(tb, cloud ? st2 : st1, f, c);
        hit_weights(c, ln, w);
        inter_point(c, w, q);
        (cloud ? hs2 : hs1)[gl * RRL_MAX_HITS + a] = f;
        (cloud ? Q2 : Q1)[gl * RRL_MAX_HITS + a] = make_float4(q[0], q[1], q[2], 0.0f);
        float *wd = (cloud ? w2 : w1) + (gl * RRL_MAX_HITS + a) * 3;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) wd[cc] = w[cc];
        if (keep) {
            keep->f = f;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) keep->w[cc] = w[cc];
            keep->q = make_float4(q[0], q[1], q[2], 0.0f);
        }
    }
    s_q[sub] = make_float4(q[0], q[1], q[2], 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the 8 lanes share a wavefront: LDS is in order
    // D[a][b] = sum_c (q1 - q2)^2, code/loss.py:38-52: entries 2 sub and 2 sub + 1 of the 4 x 4 tile
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ee = 2 * sub + e, ra = ee >> 2, rb = ee & 3;
        t2[e] = INFINITY;
        if (ra < k && rb < j) {
            const float4 p1 = s_q[ra], p2 = s_q[4 + rb];
            float dx = p1.x - p2.x, dy = p1.y - p2.y, dz = p1.z - p2.z;
            float sq = dx * dx;
            sq = sq + dy * dy;
            sq = sq + dz * dz;
            D[gl * 16 + ra * j + rb] = sq;
            t2[e] = sq;
            // the median's first radix pass (bits 30..20 of the bit pattern; D >= 0), tallied where the value is born
            if (s_mh) atomicAdd(&s_mh[(__float_as_uint(sq) >> 20) & 2047u], 1u);
        }
    }
    ((float2 *)dc_slot)[sub] = make_float2(t2[0], t2[1]);
}

// XCD-aware placement of the per-sample stages (round 5).  Workgroups go to the 8 XCDs round-robin by linear id (observed;
// a matter of speed only), each XCD has its own L2, and everything a sample's per-line stage, reduce and backward read was
// written by workgroups of that sample: the culled scan runs the (cloud, sample) pair on the fast grid index, i.e. sample b
// (both clouds, when B % 8 == 0) on XCD b % 8.  With B % 8 == 0 the stages below decode their place in the grid so that
// sample b's workgroups run on XCD b % 8 too: (x, y) of a grid (gx fast, B samples) for linear id `lin`.
__device__ __forceinline__ void xcd_sample_of(int lin, int gx, int &x, int &b) {
    const int slot = lin >> 3;
    x = slot % gx;
    b = (lin & 7) + 8 * (slot / gx);
}
static int xcd_align_on() {  // RRL_XCD_ALIGN=0 turns it off (experiments)
    static int v = -1;
    if (v < 0) { const char *e = getenv("RRL_XCD_ALIGN"); v = e && e[0] == '0' ? 0 : 1; }
    return v;
}

struct PairArgs {
    const float *tri1, *tri2, *line;  // the triangles of both clouds (raw 36-byte rows: st1 = st2 = 9), the lines
    const int32_t *count1, *hit1, *count2, *hit2;
    uint8_t *kj;
    int32_t *sel_out, *nsel, *hs1, *hs2;
    float *w1, *w2;
    float4 *Q1, *Q2;
    float *D, *dc;
    uint8_t *kjc;
    uint32_t *lidc;          // line | kj << 24 at the compact slot (or NULL)
    float *vlist;            // [B][ntile][16384] dense list of the tile's valid D values (with mhist; or NULL)
    int32_t *vlcnt;
    int32_t *blkcnt;
    uint32_t *mhist, *mctl;  // tiled reduce: per-sample histogram of the D values' top 11 bits, bucket counts (or NULL)
    int B, N, M, L, s_m, s_n, e_m, e_n, st1, st2;
    int Bt;  // multi-pose evaluation (rrl_opts.problems): tri2, line and cloud 2's scan (count2, hit2) of instance b are those
             // of problem b % Bt; 0: every instance has its own
    int xcd_align;  // line_pair_dist_kernel: sample b's workgroups on XCD b % 8 (xcd_sample_of; B % 8 == 0)
    int32_t *zc1, *zc2;  // chained steps (include/rrl.h RRL_F_CHAIN): COUNT1 / COUNT2 again, writable -- every lane zeroes its
                         // line's two counts behind its own read, so that the NEXT step's scan finds them cleared; or NULL
};

// One tile of 1024 lines of sample b by a 1024-lane workgroup.  Phase 1: every lane classifies its line
// and the selected ones (~9 %) are compacted through LDS, so that phase 2 -- the gather-heavy part -- runs
// on dense wavefronts, eight lanes per line; the compacted line ids also go to SEL[b] for the backward.
__device__ __forceinline__ void pair_body(const PairArgs &a, int b, int tile, int ntile, PairKeep *keep = nullptr) {
    __shared__ int s_list[1024];
    __shared__ int s_wave[16];
    __shared__ int s_total;
    __shared__ float4 s_q[128][8];  // intersection points of the lines of one pass
    __shared__ unsigned s_mh[2048];  // this tile's share of MHIST
    __shared__ unsigned s_bc[16];    // ... and of the bucket counts
    __shared__ unsigned s_nv;        // ... and the length of its value list
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L;
    const int bi = (a.Bt > 0 && b >= a.Bt) ? b % a.Bt : b;  // the instance's problem (multi-pose)
    int base_reg = 0;
    if (wave == 0) STAMPP(0);
    const bool tally = a.mhist != nullptr;  // uniform
    if (tally) {  // (the barriers of phase 1 publish the clearing)
        s_mh[tid] = 0u;
        s_mh[tid + 1024] = 0u;
        if (tid < 16) s_bc[tid] = 0u;
        if (tid == 0) s_nv = 0u;
    }
    {
        const int l = tile * 1024 + tid;
        bool sel = false;
        unsigned kjb = 0;
        if (l < L) {
            const size_t gl = (size_t)b * L + l;
            const int k = a.count1[gl], j = a.count2[(size_t)bi * L + l];
            if (a.zc1) { a.zc1[gl] = 0; a.zc2[gl] = 0; }  // (chained steps; never multi-pose: bi == b)
            sel = k >= a.s_m && k < a.e_m && j >= a.s_n && j < a.e_n;
            kjb = sel ? (unsigned)(k | (j << 4)) : 0u;
            a.kj[gl] = (uint8_t)kjb;
        }
        const unsigned long long mask = __ballot(sel);
        if (lane == 0) s_wave[wave] = __popcll(mask);
        if (wave == 0) STAMPP(1);
        // (LDS-only barriers in this stage: what they publish is LDS -- the KJ store above, and later the stage's stores, are
        //  read by other launches, or by this workgroup behind a fence of its own in the single-tile kernels)
        lds_barrier();
        if (sel && tally) atomicAdd(&s_bc[((kjb & 15u) - 1u) * 4u + ((kjb >> 4) - 1u)], 1u);
        // every wavefront scans the 16 counts itself -- one LDS read, a DPP prefix -- for its own base and the total (round 5:
        // lane 0 of the workgroup used to walk them, one LDS round trip after the other, between two barriers)
        const int cw = lane < 16 ? s_wave[lane] : 0;
        const int iw = wave_incl_scan(cw);
        const int wbase = __builtin_amdgcn_readlane(iw - cw, wave), acc = __builtin_amdgcn_readlane(iw, 15);
        if (tid == 0) {
            s_total = acc;
            // the slot range in SEL[b] is only needed for the last store of the kernel: the
            // atomic's round trip overlaps the gathers below
            base_reg = acc ? atomicAdd(&a.nsel[b], acc) : 0;
        }
        // the line id (L < 2^24) and its (k, j) byte travel together: phase 2 needs no second look at the counts
        if (sel) s_list[wbase + __popcll(mask & ((1ull << lane) - 1ull))] = (int)((unsigned)l | (kjb << 24));
        lds_barrier();
    }
    // Compact copies for the reduce kernel live at slot = 1024 * tile + rank: no global
    // counter is needed to place them (BLKCNT[b][tile] tells the consumer how many each workgroup
    // wrote), so nothing here waits for an atomic.  SEL[b] (dense list of the selected line ids,
    // for the backward kernels) is written at the end by wavefront 0 alone: lane 0 holds the base
    // returned by the nsel atomic, whose round trip has long been hidden by the gathers.
    const size_t Lp = (size_t)ntile * 1024;
    const int total = s_total;
    STAMP(1);
    if (wave == 0) STAMPP(2);
    if (keep) {
        keep->total = total; keep->k = keep->j = 0; keep->f = 0; keep->sq = s_q[tid >> 3];
        keep->w[0] = keep->w[1] = keep->w[2] = 0.0f; keep->q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (tid == 0) a.blkcnt[(size_t)b * ntile + tile] = total;
    float *vl = tally && a.vlist ? a.vlist + ((size_t)b * ntile + tile) * 16384 : nullptr;
    for (int r0 = 0; r0 < total; r0 += 128) {  // 128 selected lines per pass, 8 lanes each
        const int rank = r0 + (tid >> 3), sub = tid & 7;
        float t2[2] = {INFINITY, INFINITY};
        unsigned nvl = 0;  // this lane's entries inside the line's k x j block
        if (rank < total) {
            const unsigned e = (unsigned)s_list[rank];
            const int l = (int)(e & 0xffffffu), k = (int)((e >> 24) & 15u), j = (int)(e >> 28);
            const size_t gl = (size_t)b * L + l;
            const size_t slot = (size_t)b * Lp + (size_t)tile * 1024 + rank;
            if (sub == 0) {
                a.kjc[slot] = (uint8_t)(k | (j << 4));
                if (a.lidc) a.lidc[slot] = e;
            }
            pair_hit(a.tri1, a.tri2, a.line, a.hit1, a.hit2, a.hs1, a.hs2, a.w1, a.w2, a.Q1, a.Q2, a.D, a.dc + slot * 16,
                     s_q[tid >> 3], tally ? s_mh : nullptr, b, a.N, a.M, gl, k, j, sub, a.st1, a.st2, t2, bi, (size_t)bi * L + l,
                     r0 == 0 ? keep : nullptr);
            if (keep && r0 == 0) {
                keep->k = k; keep->j = j;
                if (keep->dc_lds) {
                    ((float2 *)(keep->dc_lds + rank * 16))[sub] = make_float2(t2[0], t2[1]);
                    if (sub == 0) keep->kjc_lds[rank] = (uint8_t)(k | (j << 4));
                }
            }
            nvl = (((2 * sub) >> 2) < k && ((2 * sub) & 3) < j ? 1u : 0u) | (((2 * sub + 1) >> 2) < k && ((2 * sub + 1) & 3) < j ? 2u : 0u);
        }
        if (vl) {  // (all lanes: uniform) the valid entries join the tile's dense value list: one LDS cursor atomic per wavefront
            const unsigned mine = (nvl & 1u) + (nvl >> 1);
            const unsigned incl = (unsigned)wave_incl_scan((int)mine);
            unsigned wbase = 0;
            if (lane == 63 && incl) wbase = atomicAdd(&s_nv, incl);
            wbase = (unsigned)__builtin_amdgcn_readlane((int)wbase, 63);
            unsigned at = wbase + incl - mine;
            if (nvl & 1u) vl[at++] = t2[0];
            if (nvl & 2u) vl[at] = t2[1];
        }
    }
    STAMP(2);
    if (wave == 0) STAMPP(3);
    if (tally) {  // flush the tile's tallies: <= one device atomic per populated bin and workgroup
        lds_barrier();
        uint32_t *mh = a.mhist + (size_t)b * 2048;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned v = s_mh[tid + 1024 * q];
            if (v) atomicAdd(&mh[tid + 1024 * q], v);
        }
        if (tid < 16 && s_bc[tid]) atomicAdd(&a.mctl[(size_t)b * 64 + tid], s_bc[tid]);
        if (vl) {  // the list's length; padded with -1 (no D value is negative) to whole 16-byte groups
            const unsigned nv = s_nv;
            if (tid == 0) a.vlcnt[(size_t)b * ntile + tile] = (int)nv;
            if (tid < ((4u - (nv & 3u)) & 3u)) vl[nv + tid] = -1.0f;
        } else if (tid == 0 && a.vlcnt) {
            a.vlcnt[(size_t)b * ntile + tile] = -1;  // no list this time: a tail kernel run on this state reports it (NaN loss)
        }
    }
    if (wave == 0) STAMPP(4);
    if (wave == 0) {
        const int base = __builtin_amdgcn_readfirstlane(base_reg);
        for (int i = lane; i < total; i += 64) a.sel_out[(size_t)b * L + base + i] = s_list[i] & 0xffffff;
    }
    if (wave == 0) STAMPP(5);
}

__global__ __launch_bounds__(1024) void line_pair_dist_kernel(const PairArgs a) {
    int tile = blockIdx.x, b = blockIdx.y;
    if (a.xcd_align) xcd_sample_of(tile + (int)gridDim.x * b, (int)gridDim.x, tile, b);  // (uniform)
    pair_body(a, b, tile, (int)gridDim.x);
}

// tri1 / tri2: the triangles as the loss sees them -- the caller's rows, or TRI1 (the moved source of the fused op) --, raw
// 36-byte rows indexed by triangle.  (The prepared 48-byte records, PTRI, belong to the scans alone since round 4: the
// prepared build keeps them at their SORTED positions, and a target's records may live in another workspace.)
// The per-line stage AND the count pass of the next epoch's line sampler in ONE launch (round 4b; rrl_ws.h RrlCountRider,
// rrl_demo_epoch): both run 1024-lane workgroups, the count pass needs nothing the scan produced, and launches of one
// stream never overlap on this stack.  Workgroups [0, tiles x rounds) count (the long ones start first), the others run
// the per-line stage.  Same bodies, same results as the two launches.
struct CountKArgs {
    const unsigned long long *rng_state;
    const float *r, *centers, *aabb2, *rows;
    unsigned long long *accept;
    int n_rows, n, rounds, prefilter, gx, gy;
};
__global__ __launch_bounds__(1024) void pair_count_kernel(const PairArgs a, const CountKArgs c, int pair_gx) {
    __shared__ SampleCountLds clds;
    const int ncount = c.gx * c.gy, lin = (int)blockIdx.x;
    if (lin < ncount) {  // uniform per workgroup
        sample_count_body(clds, nullptr, c.rng_state, c.r, c.centers, nullptr, c.aabb2, c.accept, 1, c.n, c.rounds, c.prefilter, 0,
                          lin % c.gx, lin / c.gx, 0, c.gx, c.rows, c.n_rows);
        return;
    }
    const int l2 = lin - ncount;
    pair_body(a, l2 / pair_gx, l2 % pair_gx, pair_gx);
}

// tar_ws != NULL: cloud 2's hit counts / hit lists are read where its scan left them -- the workspace of the evaluation the
// target was carried over from (round 4b: they used to be copied into this workspace first, two launches per evaluation).
static PairArgs pair_args(const float *tri1, const float *tri2, const float *line, void *ws, const WsLayout &w, int B, int N,
                          int M, int L, int s_m, int s_n, int e_m, int e_n, bool tally = true, const void *tar_ws = nullptr,
                          int Bt = 0) {
    PairArgs a;
    a.mhist = tally ? w.u32(ws, RRL_WS_MHIST) : nullptr;
    a.mctl = tally ? w.u32(ws, RRL_WS_MCTL) : nullptr;
    a.tri1 = tri1;
    a.tri2 = tri2;
    a.line = line;
    a.count1 = w.i32(ws, RRL_WS_COUNT1); a.hit1 = w.i32(ws, RRL_WS_HIT1);
    a.count2 = tar_ws ? w.i32(tar_ws, RRL_WS_COUNT2) : w.i32(ws, RRL_WS_COUNT2);
    a.hit2 = tar_ws ? w.i32(tar_ws, RRL_WS_HIT2) : w.i32(ws, RRL_WS_HIT2);
    a.kj = w.u8(ws, RRL_WS_KJ);
    a.sel_out = w.i32(ws, RRL_WS_SEL); a.nsel = w.i32(ws, RRL_WS_NSEL);
    a.hs1 = w.i32(ws, RRL_WS_HS1); a.hs2 = w.i32(ws, RRL_WS_HS2);
    a.w1 = w.f32(ws, RRL_WS_W1); a.w2 = w.f32(ws, RRL_WS_W2);
    a.Q1 = (float4 *)w.f32(ws, RRL_WS_Q1); a.Q2 = (float4 *)w.f32(ws, RRL_WS_Q2);
    a.D = w.f32(ws, RRL_WS_D); a.dc = w.f32(ws, RRL_WS_VALS);
    a.kjc = w.u8(ws, RRL_WS_KJC); a.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    a.lidc = w.u32(ws, RRL_WS_LIDC);
    a.vlist = w.f32(ws, RRL_WS_VLIST); a.vlcnt = w.i32(ws, RRL_WS_VLCNT);
    a.B = B; a.N = N; a.M = M; a.L = L;
    a.s_m = s_m; a.s_n = s_n; a.e_m = e_m; a.e_n = e_n;
    a.st1 = 9; a.st2 = 9;
    a.Bt = Bt;  // multi-pose (RrlCall::problems)
    a.xcd_align = B % 8 == 0 && xcd_align_on();
    a.zc1 = a.zc2 = nullptr;
    return a;
}

static int reduce_kind(int mode, int B, int nblk, int pool, bool with_bwd);

// with_bwd: the reduce that follows will carry the direct backward (rrl_registration_step) -- it decides, with the shape,
// whether the tail kernel runs and wants the dense value lists
static int line_pair_dist_impl(const float *tri1, const float *tri2, const float *line, void *ws, size_t ws_bytes, int B,
                               int N, int M, int L, int s_m, int s_n, int e_m, int e_n, int pool,
                               const RrlCall &o, void *stream, bool with_bwd = false) {
    if (!tri1 || !tri2 || !line || !ws || B < 0 || N < 0 || M < 0 || L < 0 || L >= (1 << 24)) return RRL_E_ARG;  // 24-bit line ids in LDS
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0 || L == 0) return 0;
    PairArgs pa = pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, true, o.tar_ws);
    pa.Bt = o.problems;
    if (o.leave_clean && !o.problems && !o.tar_ws) { pa.zc1 = w.i32(ws, RRL_WS_COUNT1); pa.zc2 = w.i32(ws, RRL_WS_COUNT2); }
    if (reduce_kind(o.reduce_mode, B, (L + 1023) / 1024, pool, with_bwd) != 2) pa.vlist = nullptr;  // only the tail kernel reads VLIST
    if (RrlCountRider *cr = o.count_rider) {  // the next epoch's count pass rides along (pair_count_kernel)
        const int ctiles = (cr->n + 1023) / 1024;
        if (B == 1 && cr->rounds > 0 && cr->n > 0 && (long)ctiles * cr->rounds < 512 && cr->rows && cr->n_rows > 0) {
            const CountKArgs c = {cr->rng_state, cr->r, cr->centers, cr->aabb2, cr->rows, cr->accept, cr->n_rows, cr->n,
                                  cr->rounds, rrl_sample_prefilter(), ctiles, cr->rounds};
            const int pgx = (L + 1023) / 1024;
            hipLaunchKernelGGL(pair_count_kernel, dim3((unsigned)(ctiles * cr->rounds + pgx * B)), dim3(1024), 0,
                               (hipStream_t)stream, pa, c, pgx);
            RRL_LAUNCH_CHECK();
            cr->done = 1;
            return 0;
        }
    }
    hipLaunchKernelGGL(line_pair_dist_kernel, dim3((unsigned)((L + 1023) / 1024), (unsigned)B), dim3(1024), 0,
                       (hipStream_t)stream, pa);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_line_pair_dist_ex(const float *tri1, const float *tri2, const float *line,
                                     void *ws, size_t ws_bytes, int B, int N, int M, int L, int s_m,
                                     int s_n, int e_m, int e_n, int pool, const rrl_opts *opts, void *stream) {
    return line_pair_dist_impl(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m, e_n, pool, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line,
                                  void *ws, size_t ws_bytes, int B, int N, int M, int L, int s_m,
                                  int s_n, int e_m, int e_n, int pool, void *stream) {
    return rrl_line_pair_dist_ex(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// K3+K4: one 1024-lane workgroup per sample
// ---------------------------------------------------------------------------------------
// Welsch1(x, c) = 1 - exp(-(x / c) / 2), code/loss.py:20-21
__device__ __forceinline__ float welsch(float d, float med) {
    return 1.0f - expf(-(d / med) / 2.0f);
}

// The k x j block of D values is stored row-major with stride j; bring it into a 4 x 4
// register tile (static indices only; entries outside the block are +inf).
__device__ __forceinline__ void load_block(const float *__restrict__ Dl, int k, int j, float *Dm) {
#pragma unroll
    for (int a = 0; a < RRL_MAX_HITS; ++a)
#pragma unroll
        for (int b = 0; b < RRL_MAX_HITS; ++b) Dm[a * 4 + b] = (a < k && b < j) ? Dl[a * j + b] : INFINITY;
}

// Row/column minima of the Welsch-weighted tile with first-occurrence argmin (torch.min,
// SURVEY.md Q11).  All indices static after unrolling (no scratch).  Welsch1(inf) = 1 - exp(-inf)
// = 1 would tie with saturated entries, so padding is forced back to +inf.
__device__ __forceinline__ void welsch_block(const float *Dm, float med, float *rowmin, float *colmin,
                                             int *arg_b, int *arg_a) {
    float Wl[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Wl[q] = Dm[q] < INFINITY ? welsch(Dm[q], med) : INFINITY;
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

// Dense index i of a sample's selected lines -> compact slot.  s_pref[x] = selected lines of the
// pair kernel's workgroups 0 .. x-1 (exclusive prefix of BLKCNT, nblk + 1 entries in LDS).
__device__ __forceinline__ size_t slot_of(const int *s_pref, int nblk, int i) {
    int lo = 0, hi = nblk;  // largest x with s_pref[x] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_pref[mid] <= i) lo = mid; else hi = mid;
    }
    return (size_t)lo * 1024 + (size_t)(i - s_pref[lo]);
}

// exclusive prefix of one sample's BLKCNT row into s_pref[0 .. nblk]; returns the total (all lanes)
__device__ __forceinline__ int load_prefix(const int32_t *__restrict__ cnt, int nblk, int *s_pref, int tid) {
    __syncthreads();  // s_pref may still be in use for the previous sample
    if (nblk <= 64) {  // the usual case (L <= 65536): one wavefront scans
        if (tid < 64) {
            const int c = tid < nblk ? cnt[tid] : 0;
            const int incl = wave_incl_scan(c);
            if (tid < nblk) s_pref[tid] = incl - c;
            if (tid == 63) s_pref[nblk] = incl;
        }
    } else if (tid == 0) {
        int acc = 0;
        for (int x = 0; x < nblk; ++x) { s_pref[x] = acc; acc += cnt[x]; }
        s_pref[nblk] = acc;
    }
    __syncthreads();
    return s_pref[nblk];
}

struct ReduceLds {
    int *s_pref;
    unsigned *s_hist, *s_wtot, *s_prefix, *s_rank, *s_nvals;
    unsigned *s_cand;   // [0] population of the bin chosen by pass 0, [1] gather cursor (zero between uses)
    unsigned *s_whist;  // [128] histogram of the wave-private passes
    unsigned long long *s_sum;
    int *s_cnt;
    int *s_bad;  // a Welsch term was not finite (median 0: identical clouds): the loss is NaN like the reference's
};

// Median + Welsch sums of one reduce workgroup with the first RT selected lines of every lane kept
// in registers (RT = 1 covers ns <= 1024, the usual case; RT = 3 up to 3072; beyond that the
// tiles are re-read).  Returns the median and n (the number of D values) through refs.
template <int RT>
__device__ __forceinline__ void reduce_core(const uint8_t *__restrict__ kjc, const float *__restrict__ dc,
                                            const int32_t *__restrict__ blkcnt, const ReduceLds &L_, int ns_m,
                                            int B, int nblk, int bm, int b0, int b1, int tid, float &med_o,
                                            unsigned &n_o) {
    int *s_pref = L_.s_pref;
    unsigned *s_hist = L_.s_hist, *s_wtot = L_.s_wtot, *s_prefix = L_.s_prefix, *s_rank = L_.s_rank;
    unsigned long long *s_sum = L_.s_sum;
    int *s_cnt = L_.s_cnt;
    unsigned &s_nvals = *L_.s_nvals;
    const size_t Lp = (size_t)nblk * 1024;
    float tile[RT][16];
    unsigned c0[RT];
    unsigned myvals = 0;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        c0[r] = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[r][q] = INFINITY;
        const int i = tid + 1024 * r;
        if (i < ns_m) {
            const size_t slot = (size_t)bm * Lp + slot_of(s_pref, nblk, i);
            c0[r] = kjc[slot];
            const float4 *row = (const float4 *)(dc + slot * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = row[q];
                tile[r][4 * q] = v.x; tile[r][4 * q + 1] = v.y; tile[r][4 * q + 2] = v.z; tile[r][4 * q + 3] = v.w;
            }
            myvals += (c0[r] & 15u) * (c0[r] >> 4);
        }
    }
    for (int i = tid + 1024 * RT; i < ns_m; i += 1024) {
        const unsigned c = kjc[(size_t)bm * Lp + slot_of(s_pref, nblk, i)];
        myvals += (c & 15u) * (c >> 4);
    }
    {
        const unsigned tot = (unsigned)wave_sum_i((int)myvals);
        if ((tid & 63) == 0 && tot) atomicAdd(&s_nvals, tot);
    }
    __syncthreads();
    const unsigned n = s_nvals;
    if (tid == 0) s_rank[0] = n ? (n - 1) / 2 : 0;
    __syncthreads();
    STAMP(9);

    // ---- lower median = element of rank (n-1)/2 (torch.median): MSB-first radix select on the
    //      bit patterns (D >= 0: unsigned order == float order).  Pass 0 (bits 30..20) is
    //      workgroup-wide: LDS histogram, block-wide exclusive scan (DPP wave scans + wave
    //      totals), pick the bin that holds the rank; the histogram is cleared as it is read.
    //      The values of that bin (~n/20 of them) are then gathered into LDS and ONE wavefront
    //      finishes the remaining 20 bits in three wave-private passes (7 + 7 + 6 bits, 128-bin
    //      histogram, no workgroup barrier): 5 barriers in all instead of 9.  More than 2048
    //      values in the bin (near-identical D values): the workgroup-wide passes 1, 2 as before.
    unsigned prefix = 0;
    auto wg_pass = [&](int pass) {
        const int sh = pass == 0 ? 20 : (pass == 1 ? 9 : 0);
        const int width = pass == 2 ? 9 : 11;
        const unsigned dmask = (1u << width) - 1u;
        const int hi = sh + width;  // bits >= hi must equal the prefix (hi = 31 on the first pass)
        auto tally = [&](unsigned x) {  // +inf (padding of a tile) never agrees with a prefix of finite data
            if (x != 0x7f800000u && ((x ^ prefix) >> hi) == 0u) atomicAdd(&s_hist[(x >> sh) & dmask], 1u);
        };
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) tally(__float_as_uint(tile[r][q]));
        for (int i = tid + 1024 * RT; i < ns_m; i += 1024) {
            const float *row = dc + ((size_t)bm * Lp + slot_of(s_pref, nblk, i)) * 16;
            for (int q = 0; q < 16; ++q) tally(__float_as_uint(row[q]));
        }
        __syncthreads();
        const unsigned h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
        s_hist[2 * tid] = 0;
        s_hist[2 * tid + 1] = 0;
        const unsigned incl = (unsigned)wave_incl_scan((int)(h0 + h1));
        if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wtot[w];
        const unsigned excl = base + incl - (h0 + h1), r = s_rank[pass];
        if (r >= excl && r < excl + h0 + h1) {  // exactly one lane
            const unsigned second = r >= excl + h0 ? 1u : 0u;
            s_prefix[pass] = prefix | ((2u * tid + second) << sh);
            s_rank[pass + 1] = r - excl - (second ? h0 : 0u);
            if (pass == 0) L_.s_cand[0] = second ? h1 : h0;  // population of the chosen bin
        }
        __syncthreads();
        prefix = s_prefix[pass];
    };
    if (n > 0 && n <= 128u) {
        // ---- a tiny sample (<= 128 values: C5's 512 lines select ~13): the values meet in LDS (s_hist is all zero and free
        //      until the next evaluation) and every one counts the smaller ones itself -- the element of rank (n-1)/2 by
        //      definition, ties broken by position (bit patterns of non-negative floats order like the values); two barriers instead of the radix select's five and its 2048-bin scans
        //      (3.1 us of the single-tile kernel's 12.0 at C5, tools/stamps.py).
        unsigned at = myvals ? atomicAdd(&L_.s_cand[1], myvals) : 0u;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q)  // the k x j block of the canonical tile (myvals entries: the cursor's share)
                if ((unsigned)(q >> 2) < (c0[r] & 15u) && (unsigned)(q & 3) < (c0[r] >> 4) && at < 128u) s_hist[at++] = __float_as_uint(tile[r][q]);
        if ((unsigned)tid >= n && (unsigned)tid < n + 3u) s_hist[tid] = 0xffffffffu;  // pad to whole 16-byte groups: never counted
        __syncthreads();
        if ((unsigned)tid < n) {
            const unsigned x = s_hist[tid];
            unsigned c = 0;
            for (unsigned u = 0; u < n; u += 4) {  // (uniform addresses: one LDS read serves the wavefront)
                const uint4 y = *(const uint4 *)&s_hist[u];
                c += (y.x < x || (y.x == x && u < (unsigned)tid)) ? 1u : 0u;
                c += (y.y < x || (y.y == x && u + 1 < (unsigned)tid)) ? 1u : 0u;
                c += (y.z < x || (y.z == x && u + 2 < (unsigned)tid)) ? 1u : 0u;
                c += (y.w < x || (y.w == x && u + 3 < (unsigned)tid)) ? 1u : 0u;
            }
            if (c == s_rank[0]) s_prefix[2] = x;
        }
        __syncthreads();
        prefix = s_prefix[2];
        if ((unsigned)tid < 132u) s_hist[tid] = 0;
        if (tid == 0) L_.s_cand[1] = 0;
    } else if (n > 0) {
        wg_pass(0);
        STAMP(10);
        const unsigned ncand = L_.s_cand[0];
        if (ncand <= 2048u) {
            // gather the bin's values into s_hist (all zero now, free until the next evaluation)
            auto in_bin = [&](unsigned x) { return x != 0x7f800000u && ((x ^ prefix) >> 20) == 0u; };
            unsigned mine = 0;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) mine += in_bin(__float_as_uint(tile[r][q])) ? 1u : 0u;
            for (int i = tid + 1024 * RT; i < ns_m; i += 1024) {
                const float *row = dc + ((size_t)bm * Lp + slot_of(s_pref, nblk, i)) * 16;
                for (int q = 0; q < 16; ++q) mine += in_bin(__float_as_uint(row[q])) ? 1u : 0u;
            }
            const unsigned incl = (unsigned)wave_incl_scan((int)mine);
            unsigned wbase = 0;
            if ((tid & 63) == 63 && incl) wbase = atomicAdd(&L_.s_cand[1], incl);  // one atomic per wavefront
            wbase = (unsigned)__builtin_amdgcn_readlane((int)wbase, 63);
            unsigned at = wbase + incl - mine;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned x = __float_as_uint(tile[r][q]);
                    if (in_bin(x)) s_hist[at++] = x;
                }
            for (int i = tid + 1024 * RT; i < ns_m; i += 1024) {
                const float *row = dc + ((size_t)bm * Lp + slot_of(s_pref, nblk, i)) * 16;
                for (int q = 0; q < 16; ++q) {
                    const unsigned x = __float_as_uint(row[q]);
                    if (in_bin(x)) s_hist[at++] = x;
                }
            }
            __syncthreads();
            if (tid < 64) {  // wave 0 alone: LDS operations of one wavefront execute in order
                unsigned pre = prefix, rk = s_rank[1];
                unsigned *wh = L_.s_whist;
#pragma unroll
                for (int wp = 0; wp < 3; ++wp) {
                    const int sh = wp == 0 ? 13 : (wp == 1 ? 6 : 0);
                    const int width = wp == 2 ? 6 : 7;
                    const unsigned dmask = (1u << width) - 1u;
                    const int hi = sh + width;
                    wh[2 * tid] = 0;
                    wh[2 * tid + 1] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (unsigned i = tid; i < ncand; i += 64) {
                        const unsigned x = s_hist[i];
                        if (((x ^ pre) >> hi) == 0u) atomicAdd(&wh[(x >> sh) & dmask], 1u);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const unsigned h0 = wh[2 * tid], h1 = wh[2 * tid + 1];
                    const unsigned inc2 = (unsigned)wave_incl_scan((int)(h0 + h1));
                    const unsigned excl = inc2 - (h0 + h1);
                    const bool here = rk >= excl && rk < excl + h0 + h1;  // exactly one lane
                    const unsigned second = rk >= excl + h0 ? 1u : 0u;
                    const unsigned long long who = __ballot(here);
                    const int src = __ffsll((long long)who) - 1;
                    pre |= (unsigned)__builtin_amdgcn_readlane((int)((2u * tid + second) << sh), src);
                    rk = (unsigned)__builtin_amdgcn_readlane((int)(rk - excl - (second ? h0 : 0u)), src);
                }
                if (tid == 0) s_prefix[2] = pre;
            }
            __syncthreads();
            prefix = s_prefix[2];
            // the gathered values leave s_hist non-zero: clear it for the next use (pool mode, later samples)
            for (unsigned i = tid; i < ncand; i += 1024) s_hist[i] = 0;
            if (tid == 0) L_.s_cand[1] = 0;
        } else {
            wg_pass(1);
            wg_pass(2);
        }
    }
    const float med = n ? __uint_as_float(prefix) : 0.0f;
    STAMP(11);

    // ---- Welsch + symmetric min per selected line, bucket sums in LDS (fixed point)
    // The forward needs the VALUES of the row / column minima only, and Welsch1 is non-decreasing
    // in D: min_b Welsch(D[a][b]) = Welsch(min_b D[a][b]) -- k + j exponentials instead of 16
    // (which entry attains the minimum matters to the backward alone: welsch_block there).
    auto accumulate = [&](const float *Dl, int k, int j) {
        float row = 0.0f, col = 0.0f;
#pragma unroll
        for (int a = 0; a < RRL_MAX_HITS; ++a)
            if (a < k) row += welsch(fminf(fminf(Dl[a * 4], Dl[a * 4 + 1]), fminf(Dl[a * 4 + 2], Dl[a * 4 + 3])), med);
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb)
            if (bb < j) col += welsch(fminf(fminf(Dl[bb], Dl[4 + bb]), fminf(Dl[8 + bb], Dl[12 + bb])), med);
        // Wl in [0,1], <= 4 terms: 2^-40 fixed point keeps ~2^-38 relative resolution.  A NaN term (0 / 0 with
        // median 0, code/loss.py:20-21 gives NaN there too) cannot be carried by the fixed-point sums: flag it
        if (!(row <= 4.0f) || !(col <= 4.0f)) { atomicOr(L_.s_bad, 1); row = col = 0.0f; }
        const int bi = (k - 1) * 4 + (j - 1);
        atomicAdd(&s_sum[bi * 2 + 0], (unsigned long long)((double)row * (double)(1ll << FIX_SHIFT) + 0.5));
        atomicAdd(&s_sum[bi * 2 + 1], (unsigned long long)((double)col * (double)(1ll << FIX_SHIFT) + 0.5));
        atomicAdd(&s_cnt[bi], 1);
    };
    for (int b = b0; b < b1; ++b) {
        int ns = ns_m;
        if (b != bm) ns = load_prefix(blkcnt + (size_t)b * nblk, nblk, s_pref, tid);  // pool mode only
        if (b == bm) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (tid + 1024 * r < ns) accumulate(tile[r], (int)(c0[r] & 15u), (int)(c0[r] >> 4));
        }
        for (int i = b == bm ? tid + 1024 * RT : tid; i < ns; i += 1024) {
            {
                const size_t slot = (size_t)b * Lp + slot_of(s_pref, nblk, i);
                const unsigned c = kjc[slot];
                float Dl[16];
                const float4 *row = (const float4 *)(dc + slot * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 r = row[q];
                    Dl[4 * q] = r.x; Dl[4 * q + 1] = r.y; Dl[4 * q + 2] = r.z; Dl[4 * q + 3] = r.w;
                }
                accumulate(Dl, (int)(c & 15u), (int)(c >> 4));
            }
        }
    }
    STAMP(12);
    med_o = med;
    n_o = n;
}

// The loss of a sample from its sixteen (k, j) buckets, by ONE wavefront (all 64 lanes call; lane bi < 16 brings bucket bi =
// (k - 1) 4 + (j - 1): its line count S and fixed-point row / column sums): the double-precision means, the bucket terms
// exp(-|k-j|/2) (mean_row + mean_col) (code/loss.py:215), and their sum in the reference's k-major order over the
// non-empty buckets of the range -- taken lane by lane through v_readlane in bucket order: an empty or out-of-range
// bucket's term is +0 and adding it changes nothing, so the sum has the bits of the loop that skips them.  (Round 5: that
// loop, one lane walking s_cnt[] / s_term[] in LDS -- ~40 dependent LDS round trips -- took 3.2 us of the single-tile
// kernel's 15.9, in-kernel time stamps of tools/stamps.py.)
struct BucketFinal {
    float acc;  // sum of the terms (the loss is acc / C)
    int C, nselected, nvalues;
};
__device__ __forceinline__ BucketFinal bucket_final(unsigned long long srow, unsigned long long scol, int S, int lane, int s_m,
                                                    int s_n, int e_m, int e_n) {
    const int k = (lane & 15) / 4 + 1, j = (lane & 3) + 1;
    const bool in = lane < 16 && S > 0 && k >= s_m && k < e_m && j >= s_n && j < e_n;
    float term = 0.0f;
    if (in) {
        const double sc = 1.0 / (double)(1ll << FIX_SHIFT);
        float mrow = (float)((double)srow * sc / ((double)S * k));
        float mcol = (float)((double)scol * sc / ((double)S * j));
        float wkj = expf(-0.5f * (float)abs(k - j));  // code/loss.py:215
        term = wkj * (mrow + mcol);
    }
    BucketFinal r;
    r.C = __popcll(__ballot(in));
    r.nselected = wave_sum_i(in ? S : 0);
    r.nvalues = wave_sum_i(lane < 16 ? S * k * j : 0);
    float acc = 0.0f;
#pragma unroll
    for (int bi = 0; bi < 16; ++bi)  // k-major, the reference's accumulation order
        acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), bi));
    r.acc = acc;
    return r;
}

struct ReduceArgs {
    const uint8_t *kjc;
    const float *dc;
    const int32_t *blkcnt;
    float *med_out;
    int32_t *bcnt_out;
    int64_t *bsum_out;
    int32_t *info;
    float *loss;
    const int32_t *status;
    int B, nblk, s_m, s_n, e_m, e_n, pool;
};

// K3+K4 of one group g (a sample, or all samples with the last one's median when pool) by a 1024-lane workgroup
// SOLO (the single-tile kernels, round 5): the workgroup has just run the per-line stage of sample g itself --
//   total    its selected lines (no BLKCNT read-back),
//   kjc / dc LDS copies of its compact rows when total <= 128 (no read-back of KJC / VALS and no fence in front of it), else NULL,
//   fwave    the wavefront that turns the sums into the loss (the last one: at <= 120 lines it holds none), while the others
//            return as soon as the bucket counts and the median exist -- all the backward needs: out_med, s_cnt (LDS [16]),
//   out_loss the loss (lane 0 of wavefront fwave).
struct SoloReduce {
    int total;
    const uint8_t *kjc;
    const float *dc;
    int fwave;
    float out_med, out_loss;
    const int *s_cnt;
};

__device__ __forceinline__ void reduce_body(const ReduceArgs &ra, int g, SoloReduce *solo = nullptr) {
    const uint8_t *__restrict__ kjc = ra.kjc;
    const float *__restrict__ dc = ra.dc;
    const int32_t *__restrict__ blkcnt = ra.blkcnt;
    float *__restrict__ med_out = ra.med_out;
    int32_t *__restrict__ bcnt_out = ra.bcnt_out;
    int64_t *__restrict__ bsum_out = ra.bsum_out;
    int32_t *__restrict__ info = ra.info;
    float *__restrict__ loss = ra.loss;
    const int32_t *__restrict__ status = ra.status;
    const int B = ra.B, nblk = ra.nblk, s_m = ra.s_m, s_n = ra.s_n, e_m = ra.e_m, e_n = ra.e_n, pool = ra.pool;
    extern __shared__ int s_pref[];  // nblk + 1
    __shared__ __attribute__((aligned(16))) unsigned s_hist[2048];
    __shared__ unsigned s_wtot[16];
    __shared__ unsigned s_prefix[3], s_rank[4];  // one slot per pass: no barrier between read and rewrite
    __shared__ unsigned s_nvals;
    __shared__ unsigned s_cand[2], s_whist[128];
    __shared__ unsigned long long s_sum[32];
    __shared__ int s_cnt[16];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int bm = pool ? B - 1 : g;  // whose values define the median
    const int b0 = pool ? 0 : g, b1 = pool ? B : g + 1;
    const int st0 = status[0];  // (the scan's NaN flag, for the info row: requested now -- at the end it would be one more round trip)
    if (tid < 32) s_sum[tid] = 0ull;
    if (tid < 16) s_cnt[tid] = 0;
    if (tid == 0) { s_nvals = 0; s_cand[0] = 0; s_cand[1] = 0; s_bad = 0; }
    s_hist[tid] = 0;
    s_hist[tid + 1024] = 0;

    int ns_m;
    if (solo) {  // (one tile: the prefix is (0, total))
        ns_m = solo->total;
        if (tid == 0) { s_pref[0] = 0; s_pref[1] = ns_m; }
        __syncthreads();
    } else {
        ns_m = load_prefix(blkcnt + (size_t)bm * nblk, nblk, s_pref, tid);
    }
    STAMP(4);
    const ReduceLds lds = {s_pref, s_hist, s_wtot, s_prefix, s_rank, &s_nvals, s_cand, s_whist, s_sum, s_cnt, &s_bad};
    float med;
    unsigned n;
    if (solo && solo->kjc) reduce_core<1>(solo->kjc, solo->dc, blkcnt, lds, ns_m, 1, 1, 0, 0, 1, tid, med, n);  // (<= 128 rows, from LDS)
    else if (ns_m <= 1024) reduce_core<1>(kjc, dc, blkcnt, lds, ns_m, B, nblk, bm, b0, b1, tid, med, n);
    else reduce_core<3>(kjc, dc, blkcnt, lds, ns_m, B, nblk, bm, b0, b1, tid, med, n);
    __syncthreads();
    STAMP(5);

    // ---- loss = ( sum_{non-empty (k,j), k-major} exp(-|k-j|/2) (mean_row + mean_col) ) / C
    const int ft = tid - (solo ? 64 * solo->fwave : 0);  // lane of the finishing wavefront (wavefront 0 but for SOLO)
    if (solo) { solo->out_med = med; solo->s_cnt = s_cnt; solo->out_loss = 0.0f; }
    if (ft < 0 || ft >= 64) return;
    if (ft < 16) bcnt_out[g * 16 + ft] = s_cnt[ft];
    if (ft < 32) bsum_out[(size_t)g * 32 + ft] = (int64_t)s_sum[ft];
    const BucketFinal f = bucket_final(ft < 16 ? s_sum[ft * 2 + 0] : 0ull, ft < 16 ? s_sum[ft * 2 + 1] : 0ull,
                                       ft < 16 ? s_cnt[ft] : 0, ft, s_m, s_n, e_m, e_n);  // one lane per bucket
    if (ft == 0) {
        const float lv = s_bad ? __builtin_nanf("") : (f.C ? f.acc / (float)f.C : 0.0f);  // code/loss.py:230
        med_out[g] = med;
        loss[g] = lv;
        info[g * 4 + 0] = f.C;
        info[g * 4 + 1] = f.nselected;
        info[g * 4 + 2] = (int)n;
        info[g * 4 + 3] = st0;  // the scan's NaN flag next to the bucket count: one 16-byte read-back decides the call
        if (solo) solo->out_loss = lv;
    }
}

__global__ __launch_bounds__(1024) void loss_reduce_kernel(const ReduceArgs ra) { reduce_body(ra, (int)blockIdx.x); }

// ---------------------------------------------------------------------------------------
// K3+K4, TILED (round 3): one 256-lane workgroup per 1024-line tile of the per-line stage instead of one
// 1024-lane workgroup per sample.  The single workgroup was bound by ONE compute unit: 11.3 us at C2 on 8 of
// 256 CUs, 24.7 us at the demo's shape (2600 selected lines of one sample through one CU).  Here
//   * the median's first radix pass (top 11 bits) arrives as a per-sample histogram that the per-line stage
//     tallied where the D values were born (MHIST) -- complete at the kernel boundary, so EVERY workgroup of a
//     sample finds the median's bin, the rank inside it and n by itself, without talking to anyone;
//   * the values of that bin (~n / 20) are published to a per-sample list (MCAND; one returning cursor atomic
//     per wavefront, write-through stores), the sample's workgroups meet at an arrival counter, and each then
//     finishes the select on the whole list by itself (the wave-private passes of the single-workgroup
//     kernel): one hop, no broadcast of the result;
//   * every workgroup adds the Welsch terms of its own lines to the sample's fixed-point bucket sums (MSUM,
//     64-bit device atomics: order-independent, so the loss keeps its bits) and the LAST one to arrive at a
//     second counter turns them into the loss.
// Cross-workgroup words follow the guide's hand-off rules: relaxed agent-scope atomic stores / loads (sc1:
// write-through, L1-bypassing), every storing wavefront drains (s_waitcnt vmcnt(0)) before its workgroup
// arrives, one lane polls.  The spin needs the sample's workgroups co-resident: the host takes this path only
// while B x tiles <= 1024 workgroups of 256 lanes (4 per CU) and bounds every spin (MCTL[19] + a NaN loss
// instead of a hang).  More than 2048 values in the bin (near-identical D values): tile 0 finishes the
// select alone with two more streaming passes and publishes the median; the others wait for it.
// Same arithmetic and summation rules as reduce_body: bit-identical median and loss.
// ---------------------------------------------------------------------------------------
#define MCTL_CURSOR 16
#define MCTL_TICK1 17
#define MCTL_TICK2 18
#define MCTL_ERR 19
#define MCTL_MEDBITS 20
#define MCTL_MEDRDY 21
#define MCTL_BAD 22
#define MCAND_CAP 2048

__device__ __forceinline__ unsigned ld_agent(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls one word until it reaches `want`; bounded (limit polls of ~1-2 us each: a fraction of a second by
// default), so workgroups that are not co-resident -- a partitioned device, a CU mask, another process or stream holding
// the slots -- become a flag, and the sample is then REPAIRED by its last workgroup (loss_reduce_tiled_kernel), not lost
__device__ __forceinline__ bool spin_reach(const uint32_t *p, unsigned want, unsigned limit) {
    for (unsigned it = 0; it < limit; ++it) {
        if (ld_agent(p) >= want) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// The remaining 20 bits of the radix select by ONE wavefront over vals[0, ncand) in LDS (the values of the bin
// `pre` chosen by the 11-bit pass; rk = rank inside it): three wave-private passes of 7 + 7 + 6 bits, 128-bin
// histogram wh, no workgroup barrier.  Every lane returns the median's bit pattern.
__device__ __forceinline__ unsigned wave_select20(const unsigned *vals, unsigned ncand, unsigned pre, unsigned rk,
                                                  unsigned *wh, int lane) {
#pragma unroll
    for (int wp = 0; wp < 3; ++wp) {
        const int sh = wp == 0 ? 13 : (wp == 1 ? 6 : 0);
        const int width = wp == 2 ? 6 : 7;
        const unsigned dmask = (1u << width) - 1u;
        const int hi = sh + width;
        wh[2 * lane] = 0;
        wh[2 * lane + 1] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (unsigned i = lane; i < ncand; i += 64) {
            const unsigned x = vals[i];
            if (((x ^ pre) >> hi) == 0u) atomicAdd(&wh[(x >> sh) & dmask], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned h0 = wh[2 * lane], h1 = wh[2 * lane + 1];
        const unsigned inc2 = (unsigned)wave_incl_scan((int)(h0 + h1));
        const unsigned excl = inc2 - (h0 + h1);
        const bool here = rk >= excl && rk < excl + h0 + h1;  // exactly one lane
        const unsigned second = rk >= excl + h0 ? 1u : 0u;
        const unsigned long long who = __ballot(here);
        const int src = who ? __ffsll((long long)who) - 1 : 0;
        pre |= (unsigned)__builtin_amdgcn_readlane((int)((2u * lane + second) << sh), src);
        rk = (unsigned)__builtin_amdgcn_readlane((int)(rk - excl - (second ? h0 : 0u)), src);
    }
    return pre;
}

struct TiledArgs {
    const uint8_t *kjc;
    const float *dc;
    const int32_t *blkcnt;
    uint32_t *mhist, *mctl, *mcand;
    unsigned long long *msum;
    float *med_out;
    int32_t *bcnt_out;
    int64_t *bsum_out;
    int32_t *info;
    float *loss;
    int32_t *status;      // [0] the scan's NaN flag (read); [2] += samples repaired after a hand-off time-out
    int B, nblk, s_m, s_n, e_m, e_n;
    unsigned spin_limit;  // polls before a waiting workgroup gives up (rrl_set_spin_limit: tests set 0)
    int xcd_align;        // sample b's workgroups on XCD b % 8 (xcd_sample_of; B % 8 == 0): its in-launch hand-offs stay in one L2
    float *payload;       // != NULL (rrl_loss_step_ex): the sample's last workgroup adds its loss to payload[0 .. 1] (tail_payload)
};

// Hand-offs between the workgroups of this launch (candidate list + TICK1, MEDRDY) are bounded spins.  A workgroup whose
// spin times out (the others were not resident in time) adds NOTHING to the sample's sums, raises MCTL_ERR and still
// draws its TICK2 ticket; the sample's LAST workgroup then sees the flag and recomputes the whole sample by itself --
// median from all tiles' values, Welsch sums over all its lines, the single-workgroup kernel's arithmetic on the same
// multiset -- so the result is bit-identical to the undisturbed one instead of NaN (round 3), at the cost of one
// workgroup's serial pass over ~1000 lines.  STATUS[2] counts such samples.
__device__ __forceinline__ void tiled_payload(float *payload, uint32_t *mctl, float lv);  // (= tail_payload, defined with the tail kernel)
__global__ __launch_bounds__(256) void loss_reduce_tiled_kernel(const TiledArgs a) {
    __shared__ unsigned s_vals[MCAND_CAP];  // the bin's values (usual route) / histogram of the streaming passes
    __shared__ unsigned s_wtot[4];
    __shared__ unsigned s_pick[3];          // bin, rank inside it, its population
    __shared__ unsigned s_whist[128];
    __shared__ unsigned long long s_sum[32];
    __shared__ unsigned s_flag[3];          // [0] spin ok, [1] this workgroup arrived last, [2] non-finite Welsch term
    __shared__ unsigned s_med;
    __shared__ int s_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.nblk;
    int tile = blockIdx.x, b = blockIdx.y;
    if (a.xcd_align) xcd_sample_of(tile + nblk * b, nblk, tile, b);  // (uniform)
    const size_t Lp = (size_t)nblk * 1024;
    uint32_t *ctl = a.mctl + (size_t)b * 64;
    uint32_t *cand = a.mcand + (size_t)b * MCAND_CAP;
    const float *__restrict__ dc = a.dc;
    const uint8_t *__restrict__ kjc = a.kjc;

    // ---- one round of independent loads: the tile's count, its first 256 compact rows (speculative: the slots
    //      exist whether or not they were written), the sample's histogram
    const int cnt = a.blkcnt[(size_t)b * nblk + tile];
    const int st0 = a.status[0];  // (for the info row: requested with round 1)
    const size_t slot0 = (size_t)b * Lp + (size_t)tile * 1024;
    float tl[16];
    unsigned c0;
    {
        const float4 *row = (const float4 *)(dc + (slot0 + tid) * 16);
        const float4 v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
        c0 = kjc[slot0 + tid];
        tl[0] = v0.x; tl[1] = v0.y; tl[2] = v0.z; tl[3] = v0.w; tl[4] = v1.x; tl[5] = v1.y; tl[6] = v1.z; tl[7] = v1.w;
        tl[8] = v2.x; tl[9] = v2.y; tl[10] = v2.z; tl[11] = v2.w; tl[12] = v3.x; tl[13] = v3.y; tl[14] = v3.z; tl[15] = v3.w;
    }
    unsigned hb[8];
    {
        const uint4 *hp = (const uint4 *)(a.mhist + (size_t)b * 2048) + 2 * tid;
        const uint4 h0 = hp[0], h1 = hp[1];
        hb[0] = h0.x; hb[1] = h0.y; hb[2] = h0.z; hb[3] = h0.w; hb[4] = h1.x; hb[5] = h1.y; hb[6] = h1.z; hb[7] = h1.w;
    }
    if (tid < 32) s_sum[tid] = 0ull;
    if (tid < 3) s_flag[tid] = tid == 0 ? 1u : 0u;
    if (tid >= cnt) {  // not a selected line of this tile
        c0 = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) tl[q] = INFINITY;
    }

    // ---- pick the bin of the rank among 2048 counts held 8 per lane (hb): exclusive scan over the workgroup,
    //      the lane whose range holds the rank reports (bin, rank inside, population).  Returns the total.
    auto pick_bin = [&](unsigned rank_or_none, bool have_rank) -> unsigned {
        unsigned tsum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tsum += hb[k];
        const unsigned incl = (unsigned)wave_incl_scan((int)tsum);
        __syncthreads();  // s_wtot / s_pick free again
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += s_wtot[w];
        const unsigned total = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        const unsigned rank = have_rank ? rank_or_none : (total ? (total - 1) / 2 : 0u);  // lower median: sorted[(n - 1) / 2]
        unsigned e = base + incl - tsum;
        if (rank >= e && rank < e + tsum) {  // exactly one lane (none when total == 0)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (rank >= e && rank < e + hb[k]) { s_pick[0] = 8u * tid + k; s_pick[1] = rank - e; s_pick[2] = hb[k]; }
                e += hb[k];
            }
        }
        __syncthreads();
        return total;
    };
    const unsigned n = pick_bin(0u, false);
    if (n == 0) {  // nothing selected in this sample (the same for all its workgroups): loss 0, no bucket
        if (tile == 0) {
            if (tid < 16) a.bcnt_out[b * 16 + tid] = 0;
            if (tid < 32) a.bsum_out[(size_t)b * 32 + tid] = 0;
            if (tid == 0) {
                a.med_out[b] = 0.0f;
                a.loss[b] = 0.0f;
                a.info[b * 4 + 0] = 0; a.info[b * 4 + 1] = 0; a.info[b * 4 + 2] = 0; a.info[b * 4 + 3] = st0;
            }
        }
        return;
    }
    const unsigned bin = s_pick[0], r1 = s_pick[1], pop = s_pick[2];
    unsigned prefix = bin << 20;

    // rows of this tile beyond the 256 held in registers (a tile has more than 256 selected lines only when more
    // than a quarter of its lines are selected): re-read per phase
    auto for_extra_rows = [&](auto &&fn) {
        for (int i = tid + 256; i < cnt; i += 256) {
            const float4 *row = (const float4 *)(dc + (slot0 + i) * 16);
            float D_[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = row[q];
                D_[4 * q] = v.x; D_[4 * q + 1] = v.y; D_[4 * q + 2] = v.z; D_[4 * q + 3] = v.w;
            }
            fn(D_, (unsigned)kjc[slot0 + i]);
        }
    };

    // the remaining 20 bits when the bin is crowded (near-identical D values): this workgroup alone streams over ALL the
    // sample's values twice more (bits 19..9, 8..0); returns the median's bit pattern in every lane
    auto crowded_select = [&]() -> unsigned {
        unsigned pre = bin << 20, rk = r1;
        for (int pass = 1; pass <= 2; ++pass) {
            const int sh = pass == 1 ? 9 : 0, width = pass == 1 ? 11 : 9, hi = sh + width;
            const unsigned dmask = (1u << width) - 1u;
            __syncthreads();
            for (int i = tid; i < MCAND_CAP; i += 256) s_vals[i] = 0u;
            __syncthreads();
            for (int t = 0; t < nblk; ++t) {
                const int ct = a.blkcnt[(size_t)b * nblk + t];
                const float *base = dc + ((size_t)b * Lp + (size_t)t * 1024) * 16;
                for (int i = tid; i < ct * 16; i += 256) {
                    const unsigned x = __float_as_uint(base[i]);
                    if (x != 0x7f800000u && ((x ^ pre) >> hi) == 0u) atomicAdd(&s_vals[(x >> sh) & dmask], 1u);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) hb[k] = s_vals[8 * tid + k];
            pick_bin(rk, true);
            pre |= s_pick[0] << sh;
            rk = s_pick[1];
        }
        return pre;
    };

    if (pop <= MCAND_CAP) {
        // ---- publish this tile's values of the bin, meet, read the whole list, finish the select
        auto in_bin = [&](unsigned x) { return x != 0x7f800000u && (x >> 20) == bin; };
        unsigned mine = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) mine += in_bin(__float_as_uint(tl[q])) ? 1u : 0u;
        for_extra_rows([&](const float *D_, unsigned) {
            for (int q = 0; q < 16; ++q) mine += in_bin(__float_as_uint(D_[q])) ? 1u : 0u;
        });
        const unsigned incl = (unsigned)wave_incl_scan((int)mine);
        unsigned wbase = 0;
        if (lane == 63 && incl) wbase = __hip_atomic_fetch_add(&ctl[MCTL_CURSOR], incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wbase = (unsigned)__builtin_amdgcn_readlane((int)wbase, 63);
        unsigned at = wbase + incl - mine;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned x = __float_as_uint(tl[q]);
            if (in_bin(x)) { if (at < MCAND_CAP) st_agent(&cand[at], x); ++at; }
        }
        for_extra_rows([&](const float *D_, unsigned) {
            for (int q = 0; q < 16; ++q) {
                const unsigned x = __float_as_uint(D_[q]);
                if (in_bin(x)) { if (at < MCAND_CAP) st_agent(&cand[at], x); ++at; }
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wavefront drains before the workgroup arrives
        __syncthreads();
        if (tid == 0) {  // the last to arrive learns it from its own ticket and does not poll at all
            const unsigned prev = __hip_atomic_fetch_add(&ctl[MCTL_TICK1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u < (unsigned)nblk && !spin_reach(&ctl[MCTL_TICK1], (unsigned)nblk, a.spin_limit)) s_flag[0] = 0u;
        }
        __syncthreads();
        for (unsigned i = tid; i < pop; i += 256) s_vals[i] = ld_agent(&cand[i]);
        __syncthreads();
        if (tid < 64) {
            const unsigned m = wave_select20(s_vals, pop, prefix, r1, s_whist, tid);
            if (tid == 0) s_med = m;
        }
        __syncthreads();
        prefix = s_med;
    } else if (tile == 0) {
        // ---- crowded bin: this workgroup alone streams over ALL the sample's values twice more (bits 19..9, 8..0)
        prefix = crowded_select();
        if (tid == 0) {
            st_agent(&ctl[MCTL_MEDBITS], prefix);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_agent(&ctl[MCTL_MEDRDY], 1u);
        }
    } else {
        if (tid == 0) {
            if (!spin_reach(&ctl[MCTL_MEDRDY], 1u, a.spin_limit)) s_flag[0] = 0u;
            s_med = ld_agent(&ctl[MCTL_MEDBITS]);
        }
        __syncthreads();
        prefix = s_med;
    }
    float med = __uint_as_float(prefix);

    // ---- Welsch terms of this tile's lines into the workgroup's fixed-point sums (as reduce_core::accumulate)
    auto accumulate = [&](const float *Dl, int k, int j) {
        float row = 0.0f, col = 0.0f;
#pragma unroll
        for (int q = 0; q < RRL_MAX_HITS; ++q)
            if (q < k) row += welsch(fminf(fminf(Dl[q * 4], Dl[q * 4 + 1]), fminf(Dl[q * 4 + 2], Dl[q * 4 + 3])), med);
#pragma unroll
        for (int q = 0; q < RRL_MAX_HITS; ++q)
            if (q < j) col += welsch(fminf(fminf(Dl[q], Dl[4 + q]), fminf(Dl[8 + q], Dl[12 + q])), med);
        if (!(row <= 4.0f) || !(col <= 4.0f)) { atomicOr(&s_flag[2], 1u); row = col = 0.0f; }
        const int bi = (k - 1) * 4 + (j - 1);
        atomicAdd(&s_sum[bi * 2 + 0], (unsigned long long)((double)row * (double)(1ll << FIX_SHIFT) + 0.5));
        atomicAdd(&s_sum[bi * 2 + 1], (unsigned long long)((double)col * (double)(1ll << FIX_SHIFT) + 0.5));
    };
    const bool handoff_ok = s_flag[0] != 0u;  // uniform (written before the last barrier): false = a spin timed out: this
    if (handoff_ok) {                         // workgroup's median may be wrong -- it adds nothing; the last one repairs
        if (c0) accumulate(tl, (int)(c0 & 15u), (int)(c0 >> 4));
        for_extra_rows([&](const float *D_, unsigned c) { if (c) accumulate(D_, (int)(c & 15u), (int)(c >> 4)); });
    }
    __syncthreads();
    if (tid < 32) {
        const unsigned long long v = s_sum[tid];
        if (v) atomicAdd(&a.msum[(size_t)b * 32 + tid], v);
    }
    if (tid == 32 && (s_flag[2] || !s_flag[0])) {
        if (s_flag[2]) atomicOr(&ctl[MCTL_BAD], 1u);
        if (!s_flag[0]) atomicOr(&ctl[MCTL_ERR], 1u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_flag[1] = __hip_atomic_fetch_add(&ctl[MCTL_TICK2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblk - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_flag[1]) return;

    // ---- the last workgroup of the sample.  A hand-off timed out somewhere (MCTL_ERR): recompute the sample alone.
    const bool repair = ld_agent(&ctl[MCTL_ERR]) != 0u;  // uniform: every producer's flag precedes its ticket
    if (repair) {
        __syncthreads();
        if (tid < 32) s_sum[tid] = 0ull;
        if (tid == 0) { s_flag[2] = 0u; s_flag[0] = 0u; }  // s_flag[0]: cursor of the gathered values
        __syncthreads();
        unsigned pre = bin << 20;
        if (pop <= MCAND_CAP) {  // the bin's values of ALL tiles into LDS (any order), then the wave-private passes
            for (int t = 0; t < nblk; ++t) {
                const int ct = a.blkcnt[(size_t)b * nblk + t];
                const float *base = dc + ((size_t)b * Lp + (size_t)t * 1024) * 16;
                for (int i = tid; i < ct * 16; i += 256) {
                    const unsigned x = __float_as_uint(base[i]);
                    if (x != 0x7f800000u && (x >> 20) == bin) {
                        const unsigned at = atomicAdd(&s_flag[0], 1u);
                        if (at < MCAND_CAP) s_vals[at] = x;
                    }
                }
            }
            __syncthreads();
            if (tid < 64) {
                const unsigned m = wave_select20(s_vals, pop, pre, r1, s_whist, tid);
                if (tid == 0) s_med = m;
            }
            __syncthreads();
            pre = s_med;
        } else {
            pre = crowded_select();
        }
        med = __uint_as_float(pre);
        for (int t = 0; t < nblk; ++t) {  // every selected line of the sample, the same per-line arithmetic
            const int ct = a.blkcnt[(size_t)b * nblk + t];
            const size_t s0 = (size_t)b * Lp + (size_t)t * 1024;
            for (int i = tid; i < ct; i += 256) {
                const float4 *row = (const float4 *)(dc + (s0 + i) * 16);
                float D_[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = row[q];
                    D_[4 * q] = v.x; D_[4 * q + 1] = v.y; D_[4 * q + 2] = v.z; D_[4 * q + 3] = v.w;
                }
                const unsigned c = kjc[s0 + i];
                if (c) accumulate(D_, (int)(c & 15u), (int)(c >> 4));
            }
        }
        __syncthreads();
        if (tid == 0) atomicAdd(&a.status[2], 1);
    }
    // loss = ( sum_{non-empty (k,j), k-major} exp(-|k-j|/2) (mean_row + mean_col) ) / C
    if (tid < 32) {
        const unsigned long long v = repair ? s_sum[tid]
                                            : __hip_atomic_load(&a.msum[(size_t)b * 32 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_sum[tid] = v;
        a.bsum_out[(size_t)b * 32 + tid] = (int64_t)v;
        __hip_atomic_store(&a.msum[(size_t)b * 32 + tid], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a second reduce on this state starts clean
    }
    if (tid >= 32 && tid < 64) {  // the sample's control words in the same round of loads: bucket counts, error / bad flags
        const unsigned v = ld_agent(&ctl[tid - 32]);
        if (tid < 48) {
            s_cnt[tid - 32] = (int)v;
            a.bcnt_out[b * 16 + tid - 32] = (int)v;
        } else if (tid - 32 == MCTL_BAD && !repair) {
            if (v) atomicOr(&s_flag[2], 2u);  // (s_flag[2] bit 0 was this workgroup's own; bit 1: anyone's)
        }
    }
    __syncthreads();
    if (tid >= 64) return;  // wavefront 0: one lane per bucket (bucket_final)
    const BucketFinal f = bucket_final(tid < 16 ? s_sum[tid * 2 + 0] : 0ull, tid < 16 ? s_sum[tid * 2 + 1] : 0ull,
                                       tid < 16 ? s_cnt[tid] : 0, tid, a.s_m, a.s_n, a.e_m, a.e_n);
    if (tid == 0) {
        const float acc = f.acc;
        const int C = f.C, nselected = f.nselected, nvalues = f.nvalues;
        const bool bad = repair ? (s_flag[2] & 1u) != 0u : (s_flag[2] & 2u) != 0u;  // a non-finite Welsch term (median 0)
        const float lv = bad ? __builtin_nanf("") : (C ? acc / (float)C : 0.0f);  // code/loss.py:230
        a.med_out[b] = med;
        a.loss[b] = lv;
        a.info[b * 4 + 0] = C;
        a.info[b * 4 + 1] = nselected;
        a.info[b * 4 + 2] = nvalues;
        a.info[b * 4 + 3] = st0;
        st_agent(&ctl[MCTL_CURSOR], 0u); st_agent(&ctl[MCTL_TICK1], 0u); st_agent(&ctl[MCTL_TICK2], 0u);
        st_agent(&ctl[MCTL_MEDRDY], 0u); st_agent(&ctl[MCTL_BAD], 0u); st_agent(&ctl[MCTL_ERR], 0u);
        if (a.payload && C > 0) tiled_payload(a.payload, a.mctl, lv);  // (order-independent: tail_payload)
    }
}

// ---------------------------------------------------------------------------------------
// K3 + K4 (+ K5') "tail" kernel (round 3): the tiled reduce WITHOUT the exchange, and optionally the direct
// backward of the fused training op in the same launch.
// loss_reduce_tiled_kernel lets the workgroups of a sample exchange the median bin's values (cursor atomic,
// write-through stores, arrival counter, poll, read-back: four to five dependent cross-workgroup round trips).
// Here every live workgroup reads ALL compact D tiles of its sample itself -- they sit in the L2 (57 KB per
// sample at C2, 168 KB at the demo's shape) -- filters the bin's values into its own LDS and selects the median
// alone: the same bits in every workgroup, nobody waits for anybody.  What remains shared is order-independent:
// the fixed-point bucket sums (MSUM) and ONE arrival counter whose last arriver turns them into the loss.
// Geometry: a workgroup = 512 lanes = up to TAIL_LINES (64) selected lines of one 1024-line tile (16 workgroups
// per tile in the grid; those beyond the tile's count return at once, like loss_bwd_rt_kernel's): the per-line
// arithmetic is exp-heavy (~1200 instructions per lane), and more lines per compute unit would queue on its SIMDs
// (1024-lane workgroups with a whole tile each: 5.6 us in this phase instead of ~2).  Lanes 256 .. 511 only help
// to stream and to pick the bin.
// do_bwd: the gradient of the workgroup's lines to (dL/dR, dL/dt) needs the median, the bucket counts (known
// since the per-line stage: MCTL) and dL/dloss (an input) -- not the loss -- so it runs in the same workgroup:
// its chain of dependent loads (compact slot -> line -> Q1 / Q2 / hit / weights -> source triangle) is requested
// at the START of the kernel and is in flight while the median is found (the LDS-only barriers below do not
// drain vector memory).  As separate launches the reduce and the backward cost 11.2 + 8.3 us at C2.
// payload[0 .. 1] (sum of the valid losses, their number) without a last-of-all hand-over: every sample's
// finaliser adds its loss to a 2^-40 fixed-point sum (returning atomic) and offers float(sum so far) to
// payload[0] by an unsigned atomicMax on the bit pattern -- the partial sums are monotone (losses >= 0), so the
// maximum is the complete sum, independent of the order; a NaN loss offers the (larger) NaN pattern.
// Same arithmetic and summation rules as reduce_body / loss_bwd_rt_kernel: bit-identical median and loss;
// (dR, dt) up to the order of the float atomics, as before; payload[0] may differ from the two-call path's
// double-precision sum in its last bit.
// ---------------------------------------------------------------------------------------
#define TAIL_MAX_TILES 32
#define TAIL_LANES 512
#define TAIL_LINES 64  // selected lines per workgroup (four lanes each)
#ifndef TAIL_SUBS
#define TAIL_SUBS 4    // workgroups per tile (grid z): workgroup s takes the tile's chunks s, s + 4, ... of TAIL_LINES lines
#endif
#ifndef TAIL_RPL
#define TAIL_RPL 4     // 16-byte groups of D values per lane and streaming round
#endif
#define MCTL_LSUM 34   // (row of sample 0, 8-byte aligned) uint64: fixed-point sum of the valid samples' losses

struct TailArgs {
    const uint32_t *lidc;
    const float *dc;
    const float *vlist;
    const int32_t *vlcnt;
    const int32_t *blkcnt;
    const uint32_t *mhist;
    uint32_t *mctl;
    unsigned long long *msum;
    float *med_out;
    int32_t *bcnt_out;
    int64_t *bsum_out;
    int32_t *info;
    float *loss;
    const int32_t *status;
    int B, nblk, s_m, s_n, e_m, e_n;
    int do_bwd, N, L, transpose_r;
    const int32_t *hs1;
    const float *w1;
    const float4 *Q1, *Q2;
    const float *grad_loss, *src;
    float *gR, *gt, *payload;
    float *grad_tri1;  // != NULL: the backward SCATTERS dL/dpoints1 [B][N][9] (rrl_loss_step) instead of summing (dR, dt)
    int Bt;            // multi-pose (rrl_opts.problems): src has Bt entries, instance b is a pose of entry b % Bt; 0: its own
    int xcd_align;     // sample b's workgroups on XCD b % 8 (xcd_sample_of; B % 8 == 0)
    // chained steps (include/rrl.h RRL_F_CHAIN): the CHAIN words [B][4], which the sample's last workgroup zeroes on exit
    // (or NULL); chain_flags != 0: this step's scan ran in the fused launch -- its NaN flag and time-outs are CHAIN[b][1],
    // CHAIN[b][3], not STATUS[0]
    uint32_t *chain;
    int chain_flags;
};


// one sample's final loss into payload[0 .. 1] (header); one lane
__device__ __forceinline__ void tail_payload(const TailArgs &a, float lv) {
    atomicAdd(&a.payload[1], 1.0f);
    if (lv != lv) { atomicMax((unsigned *)&a.payload[0], 0x7fc00000u); return; }
    const unsigned long long mine = (unsigned long long)((double)lv * (double)(1ll << FIX_SHIFT) + 0.5);
    const unsigned long long old = __hip_atomic_fetch_add((unsigned long long *)(a.mctl + MCTL_LSUM), mine, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
    const float tot = (float)((double)(old + mine) * (1.0 / (double)(1ll << FIX_SHIFT)));
    atomicMax((unsigned *)&a.payload[0], __float_as_uint(tot));
}

__device__ __forceinline__ void tiled_payload(float *payload, uint32_t *mctl, float lv) {
    TailArgs t;
    t.payload = payload; t.mctl = mctl;
    tail_payload(t, lv);
}

// (>= 4 wavefronts per SIMD = two 512-lane workgroups per CU: beyond 128 VGPRs a grid of more than 256 live workgroups -- B >= 16
//  at ten tiles -- would run in two generations: measured 17 -> 23.6 us at B = 16 when an edit pushed the kernel to 132)
//  SCATTER: the backward goes to points1.grad (rrl_loss_step) instead of (dR, dt) -- a template parameter, so that neither
//  instantiation carries the other's registers (source coordinates and 12 sums / the 9-float gradient row).
// RPL: 16-byte groups of D values per lane and streaming round -- 2 where 2 x (lanes per tile) groups cover a tile's list
// (<= 10 tiles per sample: ~100 groups, a tile of ~100 selected lines holds ~50), else TAIL_RPL; chosen by the host.
template <bool SCATTER, int RPL>
__device__ __forceinline__ void tail_body(const TailArgs &a, int tile_in, int b_in, int sub) {  // (tile, sample, sub): the workgroup's place
    __shared__ unsigned s_vals[MCAND_CAP];  // the bin's values (usual route) / histogram of the streaming passes
    __shared__ unsigned s_wtot[TAIL_LANES / 64];
    __shared__ unsigned s_pick[3];          // bin, rank inside it, its population
    __shared__ unsigned s_whist[128];
    __shared__ unsigned long long s_sum[32];
    __shared__ unsigned s_flag[2];          // [0] this workgroup arrived last, [1] non-finite Welsch term (bit 0 own, bit 1 anyone's)
    __shared__ unsigned s_med, s_ncand;
    __shared__ int s_cnt[16];
    __shared__ int s_pref[TAIL_MAX_TILES + 1];
    __shared__ int s_vpref[TAIL_MAX_TILES + 1];  // prefix of the tiles' value lists, in 16-byte groups
    __shared__ int s_misc[3];                    // live workgroups of the sample, its longest value list, its non-empty buckets
    __shared__ float s_red[4][12];
    __shared__ unsigned s_scat[4][64 * SCAT_STRIDE];  // wave_scatter_rows strips of the four wavefronts that hold lines
    constexpr int NW = TAIL_LANES / 64, BPL = 2048 / TAIL_LANES;  // wavefronts; histogram bins per lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (sub is the SLOW grid index: the workgroups that certainly have lines are dispatched first)
    const int nblk = a.nblk;
    int tile = tile_in, b = b_in;
    if (a.xcd_align) xcd_sample_of(tile + nblk * b, nblk, tile, b);  // (uniform; every sub-grid of nblk x B workgroups is a multiple of 8)
    const size_t Lp = (size_t)nblk * 1024;
    uint32_t *ctl = a.mctl + (size_t)b * 64;
    const float *__restrict__ dc = a.dc;
    const uint32_t *__restrict__ lidc = a.lidc;
    const size_t slot0 = (size_t)b * Lp + (size_t)tile * 1024;
    const bool do_bwd = a.do_bwd != 0;  // uniform
    constexpr bool scatter = SCATTER;  // gradient to the points (rrl_loss_step), not to (R, t)
    if (wave == 0) STAMPW(0);

    // ---- round 1: the sample's tile counts, histogram and bucket counts; the compact tile of "this lane's" line
    //      (lanes 0 .. 255, four per line: lane h of line r adds hit slot h's gradient, lane 0 the line's Welsch terms).
    //      Round 5: EVERYTHING of this round is requested before the tile's own count is looked at -- the count only decides
    //      whether the workgroup has lines at all and which of its line slots are real, never an address (a tile's compact
    //      slots exist whether or not they are filled): the early exit used to cost every live workgroup one dependent
    //      round trip (count -> the line's slot).  Vector loads return in order, so the wait for the count (requested
    //      first) does not wait for the rest.
    const int mycnt = a.blkcnt[(size_t)b * nblk + tile];
    const int bc = tid < nblk ? a.blkcnt[(size_t)b * nblk + tid] : 0;
    const int vraw = tid < nblk ? a.vlcnt[(size_t)b * nblk + tid] : 0;  // (-1: the per-line stage built no list -- a knob
    const int vc = vraw > 0 ? (vraw + 3) >> 2 : 0;                      // changed between the stages: flagged below)
    const int h = tid & 3;
    int r = sub * TAIL_LINES + (tid >> 2);  // compact rank within the tile of this lane's line (lanes < 256); first chunk
    unsigned kl_raw = 0u;
    float4 dr_raw = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    if (tid < 4 * TAIL_LINES) {  // (r <= 255: inside the tile's 1024 slots; a slot beyond the count holds stale data: masked below)
        kl_raw = lidc[slot0 + r];
        dr_raw = ((const float4 *)(dc + (slot0 + r) * 16))[h];
    }
    unsigned hb[BPL];
    {
        const uint4 hq = ((const uint4 *)(a.mhist + (size_t)b * 2048))[tid];
        hb[0] = hq.x; hb[1] = hq.y; hb[2] = hq.z; hb[3] = hq.w;
        static_assert(BPL == 4, "one 16-byte load of the histogram per lane");
    }
    const unsigned bkt = tid < 16 ? ctl[tid] : 0u;
    const float gl_in0 = do_bwd ? a.grad_loss[b] : 0.0f;
    // (the scan's NaN flag, for the info row: at the end it would be one more round trip of the last arriver; a chained
    //  step's fused launch keeps it per sample, next to the count of source workgroups that gave up waiting for their records)
    int st0;
    bool chain_tmo = false;
    if (a.chain_flags) {
        const uint4 cw = *(const uint4 *)(a.chain + 4 * (size_t)b);
        st0 = (int)cw.y;
        chain_tmo = cw.w != 0u;
    } else {
        st0 = a.status[0];
    }
    // ... and the sample's D values (the tiles' dense lists, VLIST), which the median's gather streams: lane (vt, vq) = (tile,
    // slot) reads the 16-byte groups vq + LPT u of tile vt -- an address that needs no count either (a list's 16384 slots
    // exist; a group beyond the list's length holds stale data and is masked by the count once it is here).  Round 5: the
    // gather used to map a dense group index through the prefix of the counts -- its loads could only leave after round 1
    // had come back and the bin was picked: one dependent round trip in front of the median (2.25 us of the kernel's 11.7 at
    // C2, in-kernel time stamps of tools/stamps_tail.py).
    const int LPT = TAIL_LANES / nblk;  // lanes per tile (uniform; nblk <= 32: >= 16)
    const int vt = tid / LPT, vq = tid - vt * LPT;
    const bool vlane = vt < nblk;
    const float4 *__restrict__ vbase = (const float4 *)(a.vlist + ((size_t)b * nblk + (vlane ? vt : 0)) * 16384);
    float4 vpre[RPL];
#pragma unroll
    for (int u = 0; u < RPL; ++u)
        vpre[u] = vlane ? vbase[vq + LPT * u] : make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
    if (sub * TAIL_LINES >= mycnt && !(tile == 0 && sub == 0)) return;  // uniform: no line for this workgroup
    bool mine_on = tid < 4 * TAIL_LINES && r < mycnt;
    unsigned kl = mine_on ? kl_raw : 0u;
    float dr[4];  // row h of the line's canonical D tile (the four lanes of a line hold one row each)
    dr[0] = mine_on ? dr_raw.x : INFINITY; dr[1] = mine_on ? dr_raw.y : INFINITY;
    dr[2] = mine_on ? dr_raw.z : INFINITY; dr[3] = mine_on ? dr_raw.w : INFINITY;
    auto load_line = [&]() {  // (the later chunks of a crowded tile: their slots depend on the count)
        kl = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) dr[q] = INFINITY;
        if (mine_on) {
            kl = lidc[slot0 + r];
            const float4 v = ((const float4 *)(dc + (slot0 + r) * 16))[h];
            dr[0] = v.x; dr[1] = v.y; dr[2] = v.z; dr[3] = v.w;
        }
    };
    const float gl_in = gl_in0;
    if (tid < 32) s_sum[tid] = 0ull;
    if (tid < 2) s_flag[tid] = 0u;
    if (tid == 0) s_ncand = 0u;
    if (tid < 16) s_cnt[tid] = (int)bkt;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (vraw < 0) atomicOr(&s_flag[1], 1u);  // (after the clearing above: same wavefront 0, LDS in order)
    if (wave == 0) {  // exclusive prefix of the tile counts (nblk <= 32) -- and what every wavefront needs of them, once
        const int incl = wave_incl_scan(bc), vincl = wave_incl_scan(vc);
        if (lane < nblk) { s_pref[lane + 1] = incl; s_vpref[lane + 1] = vincl; }
        if (lane == 0) { s_pref[0] = 0; s_vpref[0] = 0; }
        // workgroups of this sample that get past the test above (they all arrive at TICK2); the longest value list (16-byte
        // groups); the non-empty buckets of the range
        const int wl = min((bc + TAIL_LINES - 1) / TAIL_LINES, TAIL_SUBS);
        const int nl = wave_sum_i(lane < nblk ? (lane == 0 && wl == 0 ? 1 : wl) : 0);
        const int vm = (int)wave_max((float)vc);  // (vc <= 4096: exact in fp32)
        const int kk = lane / 4 + 1, jj = (lane & 3) + 1;
        const int cn = __popcll(__ballot(lane < 16 && bkt > 0u && kk >= a.s_m && kk < a.e_m && jj >= a.s_n && jj < a.e_n));
        if (lane == 0) { s_misc[0] = nl; s_misc[1] = vm; s_misc[2] = cn; }
    }
    int k, j;  // 0, 0 for a lane without a line

    // ---- the gradient chain of this lane's line, requested now: hit / Q1 / weights of hit slot h and the line's Q2
    //      points hang off the line index; the source triangle off the hit index (one round later, below) -- none of it
    //      depends on the median
    bool bwd_live;
    float4 q1;
    float qx[4], qy[4], qz[4], wq[3], xs[9];
    int fhit;
    auto request_line = [&]() {
        k = (int)((kl >> 24) & 15u); j = (int)(kl >> 28);
        bwd_live = do_bwd && mine_on && h < k;
        q1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        fhit = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) wq[q] = 0.0f;
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o) { qx[o] = qy[o] = qz[o] = 0.0f; }
        if (bwd_live) {
            const size_t gl = (size_t)b * a.L + (kl & 0xffffffu);
            fhit = a.hs1[gl * 4 + h];
            q1 = a.Q1[gl * 4 + h];
            const float *w = a.w1 + (gl * 4 + h) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) wq[q] = w[q];
#pragma unroll
            for (int o = 0; o < RRL_MAX_HITS; ++o)
                if (o < j) { const float4 t = a.Q2[gl * 4 + o]; qx[o] = t.x; qy[o] = t.y; qz[o] = t.z; }
        }
    };
    auto request_source = [&]() {
#pragma unroll
        for (int q = 0; q < 9; ++q) xs[q] = 0.0f;
        if (bwd_live && !scatter) {  // (the scatter needs no source coordinates)
            const float *x = a.src + ((size_t)((a.Bt > 0 && b >= a.Bt) ? b % a.Bt : b) * a.N + fhit) * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q) xs[q] = x[q];
        }
    };
    request_line();
    lds_barrier();  // s_pref, s_cnt
    if (wave == 0) STAMPW(8);
    const int myvc = vlane ? s_vpref[vt + 1] - s_vpref[vt] : 0;  // 16-byte groups of D values in this lane's tile's list
    const int nlive = s_misc[0], vmax = s_misc[1];  // (uniform)

    // ---- pick the bin of a rank among 2048 counts held BPL per lane: exclusive scan over the workgroup, the lane whose
    //      range holds the rank reports (bin, rank inside, population).  Returns the total.
    auto pick_bin = [&](unsigned rank_in, bool have_rank) -> unsigned {
        unsigned tsum = 0;
#pragma unroll
        for (int q = 0; q < BPL; ++q) tsum += hb[q];
        const unsigned incl = (unsigned)wave_incl_scan((int)tsum);
        if (have_rank) lds_barrier();  // s_wtot / s_pick free again (a later call; the first one finds them unused)
        if (lane == 63) s_wtot[wave] = incl;
        lds_barrier();
        unsigned base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned v = s_wtot[w];
            if (w < wave) base += v;
            total += v;
        }
        const unsigned rank = have_rank ? rank_in : (total ? (total - 1) / 2 : 0u);  // lower median: sorted[(n - 1) / 2]
        unsigned e = base + incl - tsum;
        if (rank >= e && rank < e + tsum) {  // exactly one lane (none when total == 0)
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                if (rank >= e && rank < e + hb[q]) { s_pick[0] = (unsigned)(BPL * tid + q); s_pick[1] = rank - e; s_pick[2] = hb[q]; }
                e += hb[q];
            }
        }
        lds_barrier();
        return total;
    };
    const unsigned n = pick_bin(0u, false);
    if (wave == 0) STAMPW(1);
    if (n == 0) {  // nothing selected in this sample: its only live workgroup is (tile 0, sub 0): loss 0, no bucket
        if (tid < 16) a.bcnt_out[b * 16 + tid] = 0;
        if (tid < 32) a.bsum_out[(size_t)b * 32 + tid] = 0;
        if (tid == 0) {
            a.med_out[b] = 0.0f;
            a.loss[b] = 0.0f;
            a.info[b * 4 + 0] = 0; a.info[b * 4 + 1] = 0; a.info[b * 4 + 2] = 0; a.info[b * 4 + 3] = st0;
            if (chain_tmo) a.loss[b] = __builtin_nanf("");
            if (a.chain) *(uint4 *)(a.chain + 4 * (size_t)b) = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const unsigned bin = s_pick[0], r1 = s_pick[1], pop = s_pick[2];
    unsigned prefix = bin << 20;

    // ---- every D value of the sample from the tiles' dense lists (VLIST: only the valid entries, ~2 per line instead of the
    //      16 slots of a canonical tile; -1 pads), RPL 16-byte groups per lane and round, LPT RPL groups of every tile
    //      per round: fn(groups) for each round; the first round's groups are the ones requested in round 1 (pre = true: the
    //      first sweep) or loaded again (a later sweep of the crowded-bin route); after_issue() runs once, when the first
    //      round of loads is in flight
    auto stream_rows = [&](bool pre, auto &&after_issue, auto &&fn) {
        bool first = true;
        for (int g0 = 0; g0 < vmax; g0 += LPT * RPL) {  // uniform trip count
            float4 v[RPL];
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const int g = g0 + vq + LPT * u;
                v[u] = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
                if (g < myvc) v[u] = pre && g0 == 0 ? vpre[u] : vbase[g];
            }
            if (first) { after_issue(); first = false; }
            fn(v);
        }
        if (first) after_issue();
    };
    auto each16 = [](const float4 *v, auto &&g) {
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            g(__float_as_uint(v[u].x)); g(__float_as_uint(v[u].y));
            g(__float_as_uint(v[u].z)); g(__float_as_uint(v[u].w));
        }
    };

    if (pop <= MCAND_CAP) {
        // the bin's values into this workgroup's own list: count, ONE cursor atomic per wavefront and round, plain stores
        stream_rows(true, [&]() { request_source(); }, [&](const float4 *v) {
            unsigned mine = 0;
            each16(v, [&](unsigned x) { mine += (x >> 20) == bin ? 1u : 0u; });
            const unsigned incl = (unsigned)wave_incl_scan((int)mine);
            unsigned wbase = 0;
            if (lane == 63 && incl) wbase = atomicAdd(&s_ncand, incl);
            wbase = (unsigned)__builtin_amdgcn_readlane((int)wbase, 63);
            unsigned at = wbase + incl - mine;
            each16(v, [&](unsigned x) {
                if ((x >> 20) == bin) { if (at < MCAND_CAP) s_vals[at] = x; ++at; }
            });
        });
        lds_barrier();
        if (wave == 0) STAMPW(9);
        if (tid < 64) {
            const unsigned m = wave_select20(s_vals, pop, prefix, r1, s_whist, tid);
            if (tid == 0) s_med = m;
        }
        if (wave == 0) STAMPW(10);
        lds_barrier();
        prefix = s_med;
    } else {
        // ---- crowded bin (near-identical D values): two more streaming passes (bits 19..9, 8..0), by every workgroup itself
        request_source();
        unsigned rk = r1;
        for (int pass = 1; pass <= 2; ++pass) {
            const int sh = pass == 1 ? 9 : 0, width = pass == 1 ? 11 : 9, hi = sh + width;
            const unsigned dmask = (1u << width) - 1u;
            lds_barrier();
#pragma unroll
            for (int q = 0; q < BPL; ++q) s_vals[tid + TAIL_LANES * q] = 0u;
            lds_barrier();
            const unsigned pre = prefix;
            stream_rows(false, [&]() {}, [&](const float4 *v) {
                each16(v, [&](unsigned x) {
                    if (((x ^ pre) >> hi) == 0u) atomicAdd(&s_vals[(x >> sh) & dmask], 1u);
                });
            });
            lds_barrier();
#pragma unroll
            for (int q = 0; q < BPL; ++q) hb[q] = s_vals[BPL * tid + q];
            pick_bin(rk, true);
            prefix |= s_pick[0] << sh;
            rk = s_pick[1];
        }
    }
    const float med = __uint_as_float(prefix);
    if (wave == 0) STAMPW(2);

    // ---- the workgroup's lines: Welsch terms into the fixed-point sums (lane 0 of a line; as reduce_core::accumulate) and
    //      hit slot h's gradient terms (as loss_bwd_rt_kernel)
    const int C = s_misc[2];
    // ---- the sample-wide part: this workgroup's fixed-point sums -> MSUM, one arrival ticket, and the LAST arriver of the
    //      sample turns the sums into the loss.  ONE wavefront runs it.  Round 5: when all of the workgroup's lines sit in its
    //      first chunk (<= 256 selected lines in the tile: always, at the shapes measured) the wavefront is the workgroup's
    //      LAST one -- it holds no lines, so its vector-memory queue is empty -- and it starts as soon as the Welsch sums are
    //      in LDS, while the wavefronts that hold lines go on with the gradient: the chain sums -> acknowledged -> ticket ->
    //      read-back (5.3 us of a 14.8 us launch at C2: the same kernel returning before it takes 9.5) no longer waits for the
    //      gradient's loads and the scatter's atomics in wavefront 0's queue.
    auto finish = [&]() {
    STAMPW(3);
    if (lane < 32) {
        const unsigned long long v = s_sum[lane];
        if (v) atomicAdd(&a.msum[(size_t)b * 32 + lane], v);
    }
    if (lane == 32 && s_flag[1]) atomicOr(&ctl[MCTL_BAD], 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's sums have arrived before it takes its ticket
    STAMPW(4);
    unsigned last = 0;
    if (lane == 0)
        last = __hip_atomic_fetch_add(&ctl[MCTL_TICK2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nlive - 1) ? 1u : 0u;
    STAMPW(5);
    if (!__builtin_amdgcn_readfirstlane((int)last)) return;

    // ---- the last workgroup of the sample: loss = ( sum_{non-empty (k,j), k-major} exp(-|k-j|/2) (mean_row + mean_col) ) / C
    unsigned long long tot = 0ull;
    unsigned anybad = 0u;
    if (lane < 32) {
        tot = __hip_atomic_load(&a.msum[(size_t)b * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.bsum_out[(size_t)b * 32 + lane] = (int64_t)tot;
        __hip_atomic_store(&a.msum[(size_t)b * 32 + lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a second reduce on this state starts clean
    } else if (lane < 48) {
        a.bcnt_out[b * 16 + lane - 32] = s_cnt[lane - 32];
    } else if (lane == 48) {
        anybad = ld_agent(&ctl[MCTL_BAD]);
    }
    const bool bad = __builtin_amdgcn_readlane((int)anybad, 48) != 0;
    // lane 2 q + c holds sum c (row / column) of bucket q: one lane per bucket takes both (the double-precision means, as
    // reduce_body)
    const unsigned long long trow = __shfl(tot, (2 * lane) & 63), tcol = __shfl(tot, (2 * lane + 1) & 63);
    const BucketFinal f = bucket_final(trow, tcol, lane < 16 ? s_cnt[lane] : 0, lane, a.s_m, a.s_n, a.e_m, a.e_n);
    STAMPW(11);
    if (lane == 0) {
        const float accl = f.acc;
        const int Cn = f.C, nselected = f.nselected, nvalues = f.nvalues;
        const float lv = bad || chain_tmo ? __builtin_nanf("") : (Cn ? accl / (float)Cn : 0.0f);  // code/loss.py:230
        a.med_out[b] = med;
        a.loss[b] = lv;
        a.info[b * 4 + 0] = Cn;
        a.info[b * 4 + 1] = nselected;
        a.info[b * 4 + 2] = nvalues;
        a.info[b * 4 + 3] = st0;
        st_agent(&ctl[MCTL_TICK2], 0u); st_agent(&ctl[MCTL_BAD], 0u);
        if (a.chain) *(uint4 *)(a.chain + 4 * (size_t)b) = make_uint4(0u, 0u, 0u, 0u);  // the next chained step finds them cleared
        if (do_bwd && a.payload && Cn > 0) tail_payload(a, lv);
    }
    STAMPW(6);
    };
    const bool single_chunk = mycnt <= TAIL_SUBS * TAIL_LINES;  // uniform
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    for (int chunk = sub;; chunk += TAIL_SUBS) {  // (a second trip only when the tile has more than 256 selected lines)
    if (chunk != sub) {
        r = chunk * TAIL_LINES + (tid >> 2);
        mine_on = tid < 4 * TAIL_LINES && r < mycnt;
        load_line();
        request_line();
        request_source();
    }
    int arg_b_own = 0, arg_a[4] = {0, 0, 0, 0};
    float er[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // exp(-(D[h][b] / med) / 2): row h of the tile
    if (wave < 4) {  // uniform: the wavefronts that hold lines.  Every lane of them takes part (the four lanes of a line
        // exchange by quad DPP, which reads active lanes only): lanes without a line carry +inf tiles and k = j = 0.
        // The Welsch tile of a line is SHARED by its four lanes: lane h evaluates row h -- <= 4 exponentials and
        // divisions instead of the 16 + 8 every lane of loss_bwd_rt_kernel / reduce_core::accumulate spends -- and
        //   * the line's Welsch terms are those of the row / column minima of D (Welsch1 is evaluated by the same
        //     instructions on the same input: the entry that holds the minimum of D holds Welsch1(min D), bit for bit),
        //   * the gradient's exp(-(D/med)/2) is the very value 1 - Welsch1 was formed from.
        const int jmax = (int)wave_max((float)j);  // bounds the columns worth evaluating (uniform)
        float wr[4];
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb) {
            const float d = dr[bb];
            wr[bb] = INFINITY;
            if (bb < jmax) {
                const float e = expf(-(d / med) / 2.0f);  // == welsch(): 1 - e
                er[bb] = e;
                if (d < INFINITY) wr[bb] = 1.0f - e;
            }
        }
        // row h: first-occurrence argmin of the Welsch values (the gradient's routing, welsch_block), and the Welsch value
        // of the row's smallest D (the line's row term)
        float bestw = wr[0];
#pragma unroll
        for (int bb = 1; bb < RRL_MAX_HITS; ++bb)
            if (wr[bb] < bestw) { bestw = wr[bb]; arg_b_own = bb; }
        const float dmin = fminf(fminf(dr[0], dr[1]), fminf(dr[2], dr[3]));
        const float rowterm = dr[0] == dmin ? wr[0] : (dr[1] == dmin ? wr[1] : (dr[2] == dmin ? wr[2] : wr[3]));
        // the whole Welsch tile and the whole D tile, from the other three lanes of the line
        float W[16], Dm[16];
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb) {
            W[0 + bb] = quad_bcast<0>(wr[bb]); W[4 + bb] = quad_bcast<1>(wr[bb]);
            W[8 + bb] = quad_bcast<2>(wr[bb]); W[12 + bb] = quad_bcast<3>(wr[bb]);
            Dm[0 + bb] = quad_bcast<0>(dr[bb]); Dm[4 + bb] = quad_bcast<1>(dr[bb]);
            Dm[8 + bb] = quad_bcast<2>(dr[bb]); Dm[12 + bb] = quad_bcast<3>(dr[bb]);
        }
        float colterm[4];
#pragma unroll
        for (int bb = 0; bb < RRL_MAX_HITS; ++bb) {
            float best = W[bb];
            int m = 0;
#pragma unroll
            for (int aa = 1; aa < RRL_MAX_HITS; ++aa)
                if (W[aa * 4 + bb] < best) { best = W[aa * 4 + bb]; m = aa; }
            arg_a[bb] = m;
            const float cmin = fminf(fminf(Dm[bb], Dm[4 + bb]), fminf(Dm[8 + bb], Dm[12 + bb]));
            colterm[bb] = Dm[bb] == cmin ? W[bb] : (Dm[4 + bb] == cmin ? W[4 + bb] : (Dm[8 + bb] == cmin ? W[8 + bb] : W[12 + bb]));
        }
        const float rt0 = quad_bcast<0>(rowterm), rt1 = quad_bcast<1>(rowterm), rt2 = quad_bcast<2>(rowterm), rt3 = quad_bcast<3>(rowterm);
        if (mine_on && h == 0) {  // the line's Welsch terms, summed in reduce_core::accumulate's order
            float row = 0.0f, col = 0.0f;
            if (0 < k) row += rt0;
            if (1 < k) row += rt1;
            if (2 < k) row += rt2;
            if (3 < k) row += rt3;
#pragma unroll
            for (int q = 0; q < RRL_MAX_HITS; ++q)
                if (q < j) col += colterm[q];
            if (!(row <= 4.0f) || !(col <= 4.0f)) { atomicOr(&s_flag[1], 1u); row = col = 0.0f; }
            const int bi = (k - 1) * 4 + (j - 1);
            atomicAdd(&s_sum[bi * 2 + 0], (unsigned long long)((double)row * (double)(1ll << FIX_SHIFT) + 0.5));
            atomicAdd(&s_sum[bi * 2 + 1], (unsigned long long)((double)col * (double)(1ll << FIX_SHIFT) + 0.5));
        }
    }
    if (single_chunk) {  // (uniform) every wavefront's Welsch sums are in LDS: the finisher wavefront takes them from here
        lds_barrier();
        if (wave == NW - 1) finish();  // (then it falls through the rest like the other wavefronts without lines)
    }
    float sv[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // this lane's gradient row (scatter)
    bool sc_live = false;
    if (bwd_live && C > 0) {
        const int S = s_cnt[(k - 1) * 4 + (j - 1)];
        const float wkj = expf(-0.5f * (float)abs(k - j));
        const float scale = gl_in * wkj / (float)C;
        const float inv_row = 1.0f / ((float)S * (float)k), inv_col = 1.0f / ((float)S * (float)j);
        float gq[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o) {
            if (o >= j) continue;
            float sw = 0.0f;
            if (arg_b_own == o) sw += inv_row;
            if (arg_a[o] == h) sw += inv_col;
            if (sw == 0.0f) continue;
            const float gD = scale * sw * er[o] / (2.0f * med);  // er[o] = exp(-(D[h][o] / med) / 2)
            gq[0] += 2.0f * (q1.x - qx[o]) * gD;
            gq[1] += 2.0f * (q1.y - qy[o]) * gD;
            gq[2] += 2.0f * (q1.z - qz[o]) * gD;
        }
        if (scatter) {  // dL/dP1[f][kk] += w_kk / 3 * dL/dq1 (loss_bwd_kernel's expression; the records launch cleared the target)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const float wk = wq[kk] / 3.0f;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) sv[3 * kk + cc] = wk * gq[cc];
            }
            sc_live = true;
        } else {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const float wk = wq[kk] / 3.0f;  // q = mean_k(w_k P_k)
            float gv[3];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) gv[cc] = wk * gq[cc];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const float xc = xs[3 * kk + cc];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) acc[cc * 3 + jj] = fmaf(xc, gv[jj], acc[cc * 3 + jj]);
                acc[9 + cc] += gv[cc];
            }
        }
        }
    }
    if (scatter && wave < 4)  // (wave-uniform) the rows of this wavefront's lines, transposed through its LDS strip
        wave_scatter_rows(sc_live, sv, (unsigned)fhit, a.grad_tri1 + (size_t)b * a.N * 9, nullptr, s_scat[wave], lane);
    if ((chunk + TAIL_SUBS) * TAIL_LINES >= mycnt) break;  // uniform
    }
    if (do_bwd && !scatter && wave < 4) {  // (the lines sit in the first four wavefronts)
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 12; ++q) s_red[wave][q] = acc[q];
    }
    __syncthreads();
#ifdef TAIL_EXP_NOFINAL  // timing experiment only (results invalid): what do the fixed-point sums -> ticket -> read-back cost?
    return;
#endif
    // ---- from here on wavefront 1 adds the workgroup's gradient sums and wavefront 0 does everything else by itself (its
    //      lanes see each other's LDS writes in program order: no workgroup barrier any more); the rest is done
    if (wave == 1 && do_bwd && !scatter && mycnt > 0 && lane < 12) {
        const int q = lane;
        const float v = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
        int o = q;  // m-index (i, j) -> memory order of R
        if (q < 9 && a.transpose_r) o = (q % 3) * 3 + q / 3;
        if (q < 9) atomicAdd(&a.gR[b * 9 + o], v); else atomicAdd(&a.gt[b * 3 + (q - 9)], v);
        if (a.payload) atomicAdd(&a.payload[2 + o], v);
    }
    if (wave == 0) STAMPW(7);
    if (wave != 0 || single_chunk) return;  // (single_chunk: the finisher wavefront took this part on long ago)
    finish();
}

template <bool SCATTER, int RPL>
__global__ __launch_bounds__(TAIL_LANES) __attribute__((amdgpu_waves_per_eu(4, 8))) void loss_tail_kernel(const TailArgs a) {
    tail_body<SCATTER, RPL>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ... with the NEXT epoch's sampler write pass riding along (rrl_demo_epoch: RrlWriteRider; bwd_write_kernel does the same for
// the direct backward's own launch): the first gx * gy workgroups of a 1-D grid run one (tile of 1024 candidates, round) of the
// write pass each, the rest are the tail kernel's (tile fastest, sub slowest, as in its own grid).  The write pass touches the
// sampler's buffers and the line buffer only -- nothing the tail kernel reads.
struct WriteKArgs {
    unsigned long long *rng_state;
    const float *r, *centers;
    const unsigned long long *accept;
    float *lines;
    int32_t *filled;
    int n, rounds, gx, gy;
};
template <bool SCATTER>
__global__ __launch_bounds__(TAIL_LANES) __attribute__((amdgpu_waves_per_eu(4, 8))) void tail_write_kernel(const TailArgs a,
                                                                                                          const WriteKArgs c) {
    extern __shared__ int s_tc_dyn[];  // the write pass's tile counts [rounds][tiles]
    const int nwrite = c.gx * c.gy, lin = (int)blockIdx.x;
    if (lin < nwrite) {  // uniform per workgroup
        sample_write_body<TAIL_LANES>(s_tc_dyn, nullptr, c.rng_state, c.r, c.centers, nullptr, nullptr, c.accept, c.lines, c.filled, 1,
                                      c.n, c.rounds, lin % c.gx, lin / c.gx, 0, c.gx, (unsigned)nwrite);
        return;
    }
    const int l2 = lin - nwrite, per = a.nblk * a.B;
    tail_body<SCATTER, TAIL_RPL>(a, l2 % a.nblk, (l2 / a.nblk) % a.B, l2 / per);
}

// K2 + K3 + K4 in ONE launch when a sample has a single tile of lines (L <= 1024) and the samples are not pooled:
// the workgroup that ran the per-line stage of sample b owns everything the reduce of sample b reads, so it
// simply carries on (a launch and the reduce's first load round less: small-L shapes such as C5 are nothing
// but launch latency).  Same bodies, same results.
__global__ __launch_bounds__(1024) void pair_reduce_kernel(const PairArgs pa, const ReduceArgs ra) {
    __shared__ __attribute__((aligned(16))) float s_dct[128 * 16];  // the first pass's compact D tiles and (k | j << 4) bytes:
    __shared__ uint8_t s_kjct[128];                                 // at <= 128 selected lines the reduce reads nothing back
    PairKeep kp;
    kp.kjc_lds = s_kjct; kp.dc_lds = s_dct;
    pair_body(pa, (int)blockIdx.x, 0, 1, &kp);
    const bool small = kp.total <= 128;  // (uniform)
    if (!small) __threadfence_block();   // this workgroup's KJC / VALS / BLKCNT stores are complete ...
    __syncthreads();                     // ... before any of its lanes reads them back (small: the LDS copies are)
    SoloReduce so;
    so.total = kp.total; so.kjc = small ? s_kjct : nullptr; so.dc = small ? s_dct : nullptr; so.fwave = 0;
    reduce_body(ra, (int)blockIdx.x, &so);
}

// The DEFAULT reduce mode (include/rrl.h rrl_set_reduce_mode; a call's rrl_opts.reduce_mode overrides it): 0 auto, 1 single,
// 2 tiled (the tail kernel wherever legal), 3 xchg (the exchange kernel wherever legal); reduce_kind() below turns a mode
// and a shape into the kernel.  Env RRL_REDUCE=single|tiled|xchg.
static int g_reduce_mode = -1;  // -1: read RRL_REDUCE once
extern "C" int rrl_set_reduce_mode(int mode) {
    if (mode < 0 || mode > 3) return RRL_E_ARG;
    g_reduce_mode = mode;
    return 0;
}
static int default_reduce_mode() {
    if (g_reduce_mode < 0) {
        const char *e = getenv("RRL_REDUCE");
        g_reduce_mode = !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 't' ? 2 : (e[0] == 'x' ? 3 : 0)));
    }
    return g_reduce_mode;
}
static bool default_deterministic();
// include/rrl.h rrl_opts -> the options of one call (csrc/rrl_ws.h RrlCall).  Fields the caller's struct does not
// reach (struct_bytes), -1 and NULL mean the process-wide default.
RrlCall rrl_resolve_opts(const rrl_opts *p) {
    RrlCall o;
    o.clear_ptr = nullptr;
    o.clear_bytes = 0;
    o.rider = nullptr;  // (set below from rrl_opts.chamfer)
    o.count_rider = nullptr;
    o.write_rider = nullptr;
    o.tar_ws = nullptr;
    rrl_opts v;
    memset(&v, 0, sizeof v);
    v.reduce_mode = v.deterministic = v.sort_parts = v.scan_variant = -1;
    if (p && p->struct_bytes >= 8) memcpy(&v, p, (size_t)p->struct_bytes < sizeof v ? (size_t)p->struct_bytes : sizeof v);
    o.flags = v.flags;
    o.reduce_mode = v.reduce_mode >= 0 && v.reduce_mode <= 3 ? v.reduce_mode : default_reduce_mode();
    o.deterministic = v.deterministic >= 0 ? (v.deterministic ? 1 : 0) : (default_deterministic() ? 1 : 0);
    o.sort_parts = v.sort_parts >= 0 && v.sort_parts <= 16 ? v.sort_parts : rrl_default_sort_parts();
    const int sv = v.scan_variant;
    o.scan_variant = (sv == 0 || sv == 1 || sv == 2 || sv == 4 || sv == 8) ? sv : rrl_default_scan_variant();
    o.order1 = v.order1;
    o.order2 = v.order2;
    if (v.scan_counters) { o.counters = (unsigned long long *)v.scan_counters; o.counter_rows = v.scan_counter_rows; }
    else rrl_default_scan_counters(&o.counters, &o.counter_rows);
    o.rider = v.chamfer;  // (done is the caller's to clear; the scan's launcher sets it when the walk rides along)
    o.payload = v.payload;
    o.payload_in_reduce = 0;
    o.problems = v.problems > 0 ? v.problems : 0;
    o.chain_left = v.chain_left;
    o.leave_clean = o.fused_build = 0;
    o.xf = nullptr;
    o.tri1_in = nullptr;
    return o;
}
// Which reduce kernel: 0 one workgroup per sample, 1 tiled with the candidate exchange (loss_reduce_tiled_kernel), 2 the
// tail kernel (no exchange: every workgroup streams its sample's dense value lists; one 512-lane workgroup or two
// per compute unit, so it serves the small, latency-bound grids: B x tiles <= 256, <= 32 tiles per sample).
// mode 0 (auto): the tail kernel where the direct backward rides along (with_bwd: rrl_registration_step -- measured
// -1.9 .. -3.4 us per step at C2 / L = 4096 / C4, round 5b: -6 us at the demo's 20 tiles; as a reduce alone it is within
// +-1 % of the exchange kernel), else the exchange kernel for >= 2 tiles while the grid
// is co-resident, else the single workgroup; 1: single; 2 ("tiled"): the tail kernel wherever it is legal (also forward
// only, also one tile: tests), exchange beyond; 3 ("xchg"): the exchange kernel wherever it is legal.
// Test hook: polls a waiting workgroup of the exchange reduce makes before it gives up (default 2^18, ~0.3 s); 0 makes
// every hand-off "time out", so the repair path runs on every sample (tests/test_gpu_stress.py).  Env RRL_SPIN_LIMIT.
static long g_spin_limit = -1;
extern "C" int rrl_set_spin_limit(long long polls) {
    if (polls < 0 || polls > 0xffffffffll) return RRL_E_ARG;
    g_spin_limit = (long)polls;
    return 0;
}
static unsigned spin_limit() {
    if (g_spin_limit < 0) {
        const char *e = getenv("RRL_SPIN_LIMIT");
        g_spin_limit = e ? atol(e) : (1l << 18);
        if (g_spin_limit < 0) g_spin_limit = 1l << 18;
    }
    return (unsigned)g_spin_limit;
}
// Workgroups of loss_reduce_tiled_kernel that are co-resident on the CURRENT device when it has the device to itself:
// compute units (as the runtime reports them: a CPX partition or a CU mask reports fewer) x the occupancy the
// runtime computes for this kernel.  (Round 3 hard-coded 1024 = 256 CUs x 4.)  Co-residency is a matter of speed only
// since round 4 -- a workgroup that waits in vain is repaired by its sample's last workgroup -- but a grid beyond the
// capacity would make that slow path the usual one, so the exchange kernel is only chosen within it.
static long xchg_capacity() {
    static long cap[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cap[dev] == 0) {
        hipDeviceProp_t p;
        int per_cu = 0;
        long c = 256;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, loss_reduce_tiled_kernel, 256, 0) == hipSuccess && per_cu > 0)
            c = (long)p.multiProcessorCount * per_cu;
        if (const char *e = getenv("RRL_XCHG_CAPACITY")) c = atol(e);  // experiments / tests
        cap[dev] = c < 1 ? 1 : (c > 4096 ? 4096 : c);
    }
    return cap[dev];
}
// (sample, tile) pairs up to which the tail kernel serves a call (experiments: RRL_TAIL_MAX_WG)
static long tail_max_wg() {
    static long v = -1;
    if (v < 0) {
        const char *e = getenv("RRL_TAIL_MAX_WG");
        v = e ? atol(e) : 256;  // measured at 10 tiles per sample (round 5): B = 16 78.5 -> 77.7, B = 24 98.0 -> 96.7 us per step with the
        if (v < 1) v = 256;     // tail kernel, B = 32 121 -> 128, B = 64 200 -> 213 (the exchange reduce + a backward launch win there)
    }
    return v;
}
static int reduce_kind(int mode, int B, int nblk, int pool, bool with_bwd) {
    if (pool || mode == 1) return 0;
    const bool xchg_ok = (long)B * nblk <= xchg_capacity();
    if (mode == 3) return xchg_ok && nblk >= 1 ? 1 : 0;
    const bool tail_ok = nblk <= TAIL_MAX_TILES && (long)B * nblk <= tail_max_wg();
    if (mode == 2 && tail_ok && nblk >= 1) return 2;
    // (round 5: up to TAIL_MAX_TILES line tiles, not 16 -- the demo's 20: tail 11.1 us against tiled reduce 10.6 + backward 5.8 / 7.2)
    if (mode == 0 && tail_ok && with_bwd && nblk >= 2) return 2;
    return xchg_ok && nblk >= 2 ? 1 : 0;
}
// the direct backward that may ride in the tail kernel's launch (rrl_registration_step)
struct TailBwd {
    const float *grad_loss, *src;
    float *gR, *gt, *payload;
    int transpose_r;
    float *grad_tri1;  // scatter target (rrl_loss_step) instead of (gR, gt)
};

static ReduceArgs reduce_args(void *ws, const WsLayout &w, float *loss, int B, int L, int s_m, int s_n, int e_m, int e_n,
                              int pool) {
    ReduceArgs r;
    r.kjc = w.u8(ws, RRL_WS_KJC); r.dc = w.f32(ws, RRL_WS_VALS); r.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    r.med_out = w.f32(ws, RRL_WS_MED); r.bcnt_out = w.i32(ws, RRL_WS_BCNT); r.bsum_out = w.i64(ws, RRL_WS_BSUM);
    r.info = w.i32(ws, RRL_WS_INFO); r.loss = loss; r.status = w.i32(ws, RRL_WS_STATUS);
    r.B = B; r.nblk = (L + 1023) / 1024;
    r.s_m = s_m; r.s_n = s_n; r.e_m = e_m; r.e_n = e_n; r.pool = pool;
    return r;
}

// tb != NULL: the caller wants the direct backward too; *bwd_done tells whether this launch carried it
static int loss_reduce_impl(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m, int s_n, int e_m,
                            int e_n, int pool, const TailBwd *tb, bool *bwd_done, const RrlCall &o, void *stream) {
    if (bwd_done) *bwd_done = false;
    if (!ws || !loss || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0) return 0;
    const int nblk = (L + 1023) / 1024;
    const int kind = reduce_kind(o.reduce_mode, B, nblk, pool, tb != nullptr);
    if (kind == 2) {
        TailArgs t;
        t.lidc = w.u32(ws, RRL_WS_LIDC); t.dc = w.f32(ws, RRL_WS_VALS); t.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
        t.vlist = w.f32(ws, RRL_WS_VLIST); t.vlcnt = w.i32(ws, RRL_WS_VLCNT);
        t.mhist = w.u32(ws, RRL_WS_MHIST); t.mctl = w.u32(ws, RRL_WS_MCTL);
        t.msum = (unsigned long long *)w.i64(ws, RRL_WS_MSUM);
        t.med_out = w.f32(ws, RRL_WS_MED); t.bcnt_out = w.i32(ws, RRL_WS_BCNT); t.bsum_out = w.i64(ws, RRL_WS_BSUM);
        t.info = w.i32(ws, RRL_WS_INFO); t.loss = loss; t.status = w.i32(ws, RRL_WS_STATUS);
        t.B = B; t.nblk = nblk; t.s_m = s_m; t.s_n = s_n; t.e_m = e_m; t.e_n = e_n;
        t.do_bwd = tb ? 1 : 0; t.N = N; t.L = L; t.transpose_r = tb ? tb->transpose_r : 0;
        t.hs1 = w.i32(ws, RRL_WS_HS1); t.w1 = w.f32(ws, RRL_WS_W1);
        t.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); t.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
        t.grad_loss = tb ? tb->grad_loss : nullptr; t.src = tb ? tb->src : nullptr;
        t.gR = tb ? tb->gR : nullptr; t.gt = tb ? tb->gt : nullptr; t.payload = tb ? tb->payload : nullptr;
        t.grad_tri1 = tb ? tb->grad_tri1 : nullptr;
        t.Bt = o.problems;
        t.xcd_align = B % 8 == 0 && xcd_align_on();
        t.chain = o.leave_clean ? w.u32(ws, RRL_WS_CHAIN) : nullptr;
        t.chain_flags = o.fused_build ? 1 : 0;
        // the next epoch's sampler write pass rides along (tail_write_kernel; rrl_demo_epoch) -- when this launch carries the
        // backward (nothing after it reads the line buffer the pass overwrites) and the ballots of THAT count pass are there
        RrlWriteRider *wr = tb ? o.write_rider : nullptr;
        const int wtiles = wr ? (wr->n + 1023) / 1024 : 0;
        const bool ride = wr && (!o.count_rider || o.count_rider->done) && B == 1 && wr->n > 0 && wr->rounds > 0 &&
                          (long)wtiles * wr->rounds < 512 && sizeof(int32_t) * (size_t)wr->rounds * wtiles <= 32 * 1024;
        if (ride) {
            const WriteKArgs wk = {wr->rng_state, wr->r, wr->centers, wr->accept, wr->lines, wr->filled, wr->n, wr->rounds, wtiles, wr->rounds};
            const dim3 g((unsigned)(wk.gx * wk.gy + nblk * B * TAIL_SUBS));
            const size_t lds = sizeof(int32_t) * (size_t)wk.rounds * wk.gx;
            if (t.grad_tri1) hipLaunchKernelGGL(tail_write_kernel<true>, g, dim3(TAIL_LANES), lds, (hipStream_t)stream, t, wk);
            else hipLaunchKernelGGL(tail_write_kernel<false>, g, dim3(TAIL_LANES), lds, (hipStream_t)stream, t, wk);
            wr->done = 1;
        } else {
            const dim3 g((unsigned)nblk, (unsigned)B, TAIL_SUBS);
            const bool two = TAIL_LANES / nblk >= 48;  // (see tail_body: groups per lane and round)
#define RRL_TAIL(S_, R_) hipLaunchKernelGGL((loss_tail_kernel<S_, R_>), g, dim3(TAIL_LANES), 0, (hipStream_t)stream, t)
            if (t.grad_tri1) { if (two) RRL_TAIL(true, 2); else RRL_TAIL(true, TAIL_RPL); }
            else { if (two) RRL_TAIL(false, 2); else RRL_TAIL(false, TAIL_RPL); }
#undef RRL_TAIL
        }
        RRL_LAUNCH_CHECK();
        if (bwd_done) *bwd_done = tb != nullptr;
        return 0;
    }
    if (kind == 1) {
        TiledArgs t;
        t.kjc = w.u8(ws, RRL_WS_KJC); t.dc = w.f32(ws, RRL_WS_VALS); t.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
        t.mhist = w.u32(ws, RRL_WS_MHIST); t.mctl = w.u32(ws, RRL_WS_MCTL); t.mcand = w.u32(ws, RRL_WS_MCAND);
        t.msum = (unsigned long long *)w.i64(ws, RRL_WS_MSUM);
        t.med_out = w.f32(ws, RRL_WS_MED); t.bcnt_out = w.i32(ws, RRL_WS_BCNT); t.bsum_out = w.i64(ws, RRL_WS_BSUM);
        t.info = w.i32(ws, RRL_WS_INFO); t.loss = loss; t.status = w.i32(ws, RRL_WS_STATUS);
        t.B = B; t.nblk = nblk; t.s_m = s_m; t.s_n = s_n; t.e_m = e_m; t.e_n = e_n;
        t.spin_limit = spin_limit();
        t.xcd_align = B % 8 == 0 && xcd_align_on();
        t.payload = o.payload_in_reduce ? o.payload : nullptr;
        hipLaunchKernelGGL(loss_reduce_tiled_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, t);
        RRL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(loss_reduce_kernel, dim3((unsigned)(pool ? 1 : B)), dim3(1024), sizeof(int) * (size_t)(nblk + 1),
                       (hipStream_t)stream, reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, pool));
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_loss_reduce_ex(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L,
                                  int s_m, int s_n, int e_m, int e_n, int pool, const rrl_opts *opts, void *stream) {
    return loss_reduce_impl(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, nullptr, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_loss_reduce(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L,
                               int s_m, int s_n, int e_m, int e_n, int pool, void *stream) {
    return rrl_loss_reduce_ex(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, stream);
}

// K3 + K4 over CALLER-SUPPLIED rows (the merge step of the line-sharded single-sample mode, rrl_hip/dist.py): every rank
// ran the scan and the per-line stage on ITS share of one sample's lines; the selected lines' canonical D tiles and
// (k | j << 4) bytes of all ranks, gathered into one dense list, are reduced here by the single-workgroup kernel exactly
// as if one per-line stage had produced them (full tiles of 1024 rows: blkcnt is filled by a tiny launch).  The median
// is over the same multiset and the bucket sums are order-independent fixed point, so loss, median, bucket counts and
// sums are bit-identical to the unsharded evaluation.
__global__ void rows_blkcnt_kernel(int32_t *__restrict__ blkcnt, int nblk, int nrows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nblk) blkcnt[t] = min(1024, nrows - 1024 * t);
}

extern "C" int rrl_loss_reduce_rows(const float *rows16, const uint8_t *kj, int nrows, int32_t *blkcnt_scratch, float *loss,
                                    float *med, int32_t *bcnt, int64_t *bsum, int32_t *info, const int32_t *status,
                                    int s_m, int s_n, int e_m, int e_n, void *stream) {
    if (!rows16 || !kj || !blkcnt_scratch || !loss || !med || !bcnt || !bsum || !info || !status || nrows < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    const int nblk = nrows > 0 ? (nrows + 1023) / 1024 : 1;
    hipLaunchKernelGGL(rows_blkcnt_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, blkcnt_scratch,
                       nblk, nrows);
    ReduceArgs r;
    r.kjc = kj; r.dc = rows16; r.blkcnt = blkcnt_scratch;
    r.med_out = med; r.bcnt_out = bcnt; r.bsum_out = bsum; r.info = info; r.loss = loss; r.status = status;
    r.B = 1; r.nblk = nblk; r.s_m = s_m; r.s_n = s_n; r.e_m = e_m; r.e_n = e_n; r.pool = 0;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), sizeof(int) * (size_t)(nblk + 1), (hipStream_t)stream, r);
    RRL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K5 backward.  dL/dD[a][b] = gout * w_kj / C * exp(-D/(2 med)) / (2 med)
//                              * ( [b = argmin_b(a)] / (S k) + [a = argmin_a(b)] / (S j) )
// dL/dq1[a] = sum_b 2 (q1_a - q2_b) dL/dD;  dL/dP1[f_a][kk] += w_kk / 3 * dL/dq1[a];
// weights, median and labels carry no gradient (code/loss.py:112, 224).
// ---------------------------------------------------------------------------------------
// Grid (round 5): (line tiles, samples, BWDS_SUBS) over the COMPACT slots the per-line stage left per 1024-line tile
// (BLKCNT[b][tile] selected lines at slots 1024 tile + rank, LIDC = line | kj << 24): a workgroup takes the tile's lines
// rank 32 (sub + BWDS_SUBS i) ..., eight lanes per line; workgroups beyond the tile's count return after one load.  The
// round-1 grid covered every POSSIBLE slot of the sample's dense list (8 L lanes per sample, ~9 % live, every dead lane
// walking the same chain of loads up to the DPP exchange): 11 us at C2, 53 us at B = 64; and every live lane issued its
// nine atomics itself (wave_scatter_rows above).
#define BWDS_LINES 32  // selected lines per pass of a 256-lane workgroup
#define BWDS_SUBS 4
struct ScatArgs {
    const uint32_t *lidc;
    const int32_t *blkcnt, *hs1, *hs2, *bcnt, *info;
    const float *w1, *w2, *D, *med, *grad_loss;
    const float4 *Q1, *Q2;
    float *g1, *g2;
    int N, M, L;
    unsigned long long *fx;  // deterministic mode: GFIX -- [B][N + M][9] fixed-point accumulators, then int32 [B][2] non-finite flags; or NULL
    int fxbits;              // ... fractional bits below the sample's bound exponent (scat_unit_exp)
    int fxB;                 // ... samples (the flags sit behind the B accumulators)
};
// Deterministic scatter: the exponent e with |any single contribution| < 2^e for a sample with upstream gradient gl, C valid
// buckets and median m.  A contribution is w/3 * sum_{o < 4} 2 (q1 - q2)_c * scale * sw * exp(-D / 2m) / (2m) with w <= 1,
// scale <= |gl| / C, sw <= 2 and |q1 - q2| exp(-D / 2m) / m <= sqrt(D) exp(-D / 2m) / m <= 0.607 / sqrt(m): below
// 1.62 |gl| / (C sqrt(m)); two more binades of slack.  Non-positive / non-finite bound: 0 (such a sample's contributions are
// zero or non-finite).  The SAME expression in the scatter and in the conversion: same bits.
__device__ __forceinline__ int scat_unit_exp(float gl, int C, float m) {
    const float bnd = 1.62f * fabsf(gl) / ((float)(C > 0 ? C : 1) * sqrtf(m));
    if (!(bnd > 0.0f) || !(bnd < INFINITY)) return 0;
    return ilogbf(bnd) + 3;
}
__device__ __forceinline__ int scat_fx_bits(int L) {  // <= L contributions per element: 62 - ceil(log2 L) fractional bits
    int lg = 1;
    while ((1 << lg) < L && lg < 30) ++lg;
    return 62 - lg;
}

// The arithmetic of the scatter backward for one lane = (selected line, side, hit slot h), 8 lanes per line ((side, h) =
// lane bits 2 and 0..1); every lane of the wavefront calls (DPP exchange, LDS transpose).  valid: the lane has a line;
// live: ... and a hit whose gradient row is wanted; d[o]: row h (cloud 1) / column h (cloud 2) of the line's D tile (+inf
// outside the block); S: the (k, j) bucket's line count; mine / f / wq: this hit's intersection point, triangle and weights;
// other_of(o): intersection point o of the OTHER cloud (asked for only where an entry carries gradient).
template <class OtherF>
__device__ __forceinline__ void bwd_scatter_math(bool live, int k, int j, int side, int h, const float (&d)[4], int S,
                                                 float4 mine, int f, const float (&wq)[3], int C, float m, float gl_in,
                                                 OtherF other_of, float *__restrict__ g1b, float *__restrict__ g2b,
                                                 unsigned *strip, int lane, unsigned long long *fx1b = nullptr,
                                                 unsigned long long *fx2b = nullptr, double inv_unit = 0.0, int32_t *nonfinite = nullptr) {
    const int ocnt = side ? k : j;
    const int omax = (int)wave_max((float)(k > j ? k : j));  // uniform
    float er[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wr[4];
#pragma unroll
    for (int o = 0; o < RRL_MAX_HITS; ++o) {
        wr[o] = INFINITY;
        if (o < omax) {
            const float ex = expf(-(d[o] / m) / 2.0f);  // == welsch(): 1 - e
            er[o] = ex;
            if (d[o] < INFINITY) wr[o] = 1.0f - ex;
        }
    }
    int own = 0;  // first-occurrence argmin of this row / column
    {
        float bestw = wr[0];
#pragma unroll
        for (int o = 1; o < RRL_MAX_HITS; ++o)
            if (wr[o] < bestw) { bestw = wr[o]; own = o; }
    }
    // the other quad's minima: lane (side, .) reads lane (1 - side, .) of its line, then slot o of that quad
    const int up = __builtin_amdgcn_update_dpp(0, own, 0x104, 0xf, 0xf, true);  // row_shl:4: from lane + 4
    const int dn = __builtin_amdgcn_update_dpp(0, own, 0x114, 0xf, 0xf, true);  // row_shr:4: from lane - 4
    const float oth = __int_as_float(side ? dn : up);
    const int oarg[4] = {__float_as_int(quad_bcast<0>(oth)), __float_as_int(quad_bcast<1>(oth)),
                         __float_as_int(quad_bcast<2>(oth)), __float_as_int(quad_bcast<3>(oth))};
    float sv[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        const float wkj = expf(-0.5f * (float)abs(k - j));
        const float scale = gl_in * wkj / (float)C;
        const float inv_row = 1.0f / ((float)S * (float)k), inv_col = 1.0f / ((float)S * (float)j);
        float gq[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o) {
            if (o >= ocnt) continue;
            // entry (a, bb) = (h, o) for cloud 1, (o, h) for cloud 2: it is the row minimum when arg_b[a] == bb and the column
            // minimum when arg_a[bb] == a
            float sw = 0.0f;
            if (side) {
                if (oarg[o] == h) sw += inv_row;  // arg_b[o] == h
                if (own == o) sw += inv_col;      // arg_a[h] == o
            } else {
                if (own == o) sw += inv_row;      // arg_b[h] == o
                if (oarg[o] == h) sw += inv_col;  // arg_a[o] == h
            }
            if (sw == 0.0f) continue;
            // dWl/dD = exp(-D/(2 med)) / (2 med);  dD/dq1 = 2 (q1 - q2) = -dD/dq2
            const float gD = scale * sw * er[o] / (2.0f * m);
            const float4 other = other_of(o);
            gq[0] += 2.0f * (mine.x - other.x) * gD;
            gq[1] += 2.0f * (mine.y - other.y) * gD;
            gq[2] += 2.0f * (mine.z - other.z) * gD;
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const float wk = wq[kk] / 3.0f;  // q = mean_k(w_k P_k)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) sv[3 * kk + cc] = wk * gq[cc];
        }
    }
    wave_scatter_rows(live, sv, (unsigned)f | (side ? 0x80000000u : 0u), g1b, g2b, strip, lane, fx1b, fx2b, inv_unit, nonfinite);
}

// One pass of the scatter backward from the workspace: this lane = (the tile's selected line of rank r, side, hit slot h);
// slot0: the tile's first compact slot, cnt its selected lines, C / m / gl_in the sample's valid count, median and upstream
// gradient.  Every lane of the wavefront calls.
__device__ __forceinline__ void bwd_scatter_pass(const ScatArgs &a, int b, size_t slot0, int cnt, int r, int g, int C, float m,
                                                 float gl_in, unsigned *strip, int lane) {
    const float *__restrict__ D = a.D;
    const int side = (lane >> 2) & 1, h = lane & 3;
    const bool valid = r < cnt && C > 0;
    const unsigned e = valid ? a.lidc[slot0 + r] : 0u;  // line | k << 24 | j << 28: no second look at the counts
    const int k = (int)((e >> 24) & 15u), j = (int)(e >> 28);
    const size_t gl = (size_t)b * a.L + (e & 0xffffffu);
    const int mycnt = side ? j : k, ocnt = side ? k : j;  // this lane's hit slots, the other cloud's
    const bool live = valid && h < mycnt && !(side && !a.g2);
    float d[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    if (valid && h < mycnt) {
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o)
            if (o < ocnt) d[o] = D[gl * 16 + (side ? o * j + h : h * j + o)];  // row h (cloud 1) / column h (cloud 2)
    }
    // (requested now, used after the exchange: the loads of the gradient's own chain overlap the Welsch arithmetic)
    int f = 0, S = 1;
    float4 mine = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float wq[3] = {0.0f, 0.0f, 0.0f};
    if (live) {
        S = a.bcnt[g * 16 + (k - 1) * 4 + (j - 1)];
        mine = (side ? a.Q2 : a.Q1)[gl * 4 + h];
        f = (side ? a.hs2 : a.hs1)[gl * 4 + h];
        const float *w = (side ? a.w2 : a.w1) + (gl * 4 + h) * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) wq[q] = w[q];
    }
    const float4 *__restrict__ Qo = side ? a.Q1 : a.Q2;
    if (a.fx) {  // uniform: deterministic mode -- fixed-point accumulators, unit 2^(scat_unit_exp - fxbits)
        unsigned long long *fx1b = a.fx + (size_t)b * (a.N + a.M) * 9, *fx2b = fx1b + (size_t)a.N * 9;
        int32_t *flags = (int32_t *)(a.fx + (size_t)a.fxB * (a.N + a.M) * 9) + 2 * b;
        bwd_scatter_math(live, k, j, side, h, d, S, mine, f, wq, C, m, gl_in, [&](int o) { return Qo[gl * 4 + o]; }, nullptr, nullptr,
                         strip, lane, fx1b, fx2b, ldexp(1.0, a.fxbits - scat_unit_exp(gl_in, C, m)), flags);
        return;
    }
    bwd_scatter_math(live, k, j, side, h, d, S, mine, f, wq, C, m, gl_in, [&](int o) { return Qo[gl * 4 + o]; },
                     a.g1 + (size_t)b * a.N * 9, a.g2 ? a.g2 + (size_t)b * a.M * 9 : nullptr, strip, lane);
}

// Deterministic mode: the fixed-point accumulators -> fp32 gradients.  grid (ceil(max(N, M) * 9 / 256), B, clouds)
__global__ __launch_bounds__(256) void scatter_fix_to_float_kernel(const ScatArgs a, int B, int pool) {
    const int b = blockIdx.y, side = blockIdx.z, g = pool ? 0 : b;
    const int n = side ? a.M : a.N;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * 9) return;
    const int C = a.info[g * 4];
    const float m = a.med[g], gl_in = a.grad_loss[g];
    const double unit = ldexp(1.0, scat_unit_exp(gl_in, C, m) - a.fxbits);
    const long long q = (long long)(a.fx + (size_t)b * (a.N + a.M) * 9 + (side ? (size_t)a.N * 9 : 0))[i];
    const int32_t bad = ((const int32_t *)(a.fx + (size_t)B * (a.N + a.M) * 9))[2 * b + side];
    (side ? a.g2 + (size_t)b * a.M * 9 : a.g1 + (size_t)b * a.N * 9)[i] = bad ? __builtin_nanf("") : (float)((double)q * unit);
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const ScatArgs a, int B, int pool, int xcd_align) {
    __shared__ unsigned s_scat[4][64 * SCAT_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = blockIdx.z, ntile = gridDim.x;
    int tile = blockIdx.x, b = blockIdx.y;
    if (xcd_align) xcd_sample_of(tile + ntile * b, ntile, tile, b);  // (uniform) sample b's workgroups on XCD b % 8
    const int g = pool ? 0 : b;
    const int cnt = a.blkcnt[(size_t)b * ntile + tile];
    if (sub * BWDS_LINES >= cnt) return;  // uniform: no line for this workgroup
    const int C = a.info[g * 4];
    const float m = a.med[g], gl_in = a.grad_loss[g];
    const size_t slot0 = ((size_t)b * ntile + tile) * 1024;
    for (int r0 = sub * BWDS_LINES; r0 < cnt; r0 += BWDS_LINES * BWDS_SUBS)  // uniform; a second trip only beyond 128 lines
        bwd_scatter_pass(a, b, slot0, cnt, r0 + (tid >> 3), g, C, m, gl_in, s_scat[wave], lane);
}

// K2 + K3 + K4 + K5 in ONE launch for a single tile of lines (L <= 1024; C5: N = M = 16384, L = 512) -- SURVEY 8(d)'s step
// (rrl_loss_step_ex: backward to points1.grad) at the shapes pair_reduce_kernel serves: the workgroup of sample b ran the
// per-line stage and the reduce of sample b, so it carries on with the scatter backward of the
// sample's selected lines (loss_bwd_kernel's arithmetic, 8 lines per wavefront and pass).  Same bodies, same
// results; payload[0 .. 1] as in the tail kernel.  grad_tri1 is zero on entry (the records launch cleared it).
// At most 128 selected lines (one pass of the per-line stage: every (line, cloud, hit) lane still HOLDS its triangle, weights and
// intersection point, and the line's eight points sit in the stage's LDS) nothing is read back: the reduce takes the compact
// rows from an LDS copy, the backward re-evaluates its D row / column from the points with the stage's own expression
// (bit-identical to the stored tile), takes the median from the reduce's registers and the bucket counts from its LDS -- and
// starts as soon as those exist, while the workgroup's last wavefront turns the sums into the loss (SoloReduce).  More lines:
// through the workspace, as above.
__global__ __launch_bounds__(1024) void pair_reduce_scatter_kernel(const PairArgs pa, const ReduceArgs ra, const ScatArgs a,
                                                                    float *__restrict__ payload, uint32_t *__restrict__ mctl) {
    __shared__ unsigned s_scat[16][64 * SCAT_STRIDE];
    __shared__ __attribute__((aligned(16))) float s_dct[128 * 16];  // the first pass's compact D tiles ...
    __shared__ uint8_t s_kjct[128];                                 // ... and (k | j << 4) bytes, for the reduce
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float gl_in = a.grad_loss[b];  // (requested now)
    PairKeep kp;
    kp.kjc_lds = s_kjct; kp.dc_lds = s_dct;
    STAMP(0);
    pair_body(pa, b, 0, 1, &kp);
    const int cnt = kp.total;       // (uniform)
    const bool small = cnt <= 128;  // one pass of the per-line stage: the reduce and the backward read nothing back
    if (!small) __threadfence_block();  // this workgroup's KJC / VALS / BLKCNT / LIDC stores are complete ...
    __syncthreads();                    // ... before any of its lanes reads them back (small: the LDS copies are)
    STAMP(3);
    SoloReduce so;
    so.total = cnt; so.kjc = small ? s_kjct : nullptr; so.dc = small ? s_dct : nullptr; so.fwave = 15;
    reduce_body(ra, b, &so);  // (every wavefront but the last returns once the median and the bucket counts exist)
    STAMP(6);
    // the non-empty buckets of the range (code/loss.py:230's C), by every wavefront from the counts in LDS
    const int sc = lane < 16 ? so.s_cnt[lane] : 0, kk = (lane & 15) / 4 + 1, jj = (lane & 3) + 1;
    const int C = __popcll(__ballot(lane < 16 && sc > 0 && kk >= ra.s_m && kk < ra.e_m && jj >= ra.s_n && jj < ra.e_n));
    const float m = so.out_med;
    if (tid == 64 * 15 && payload && C > 0) {  // (order-independent: fixed-point sum of the valid samples' losses)
        TailArgs t;
        t.payload = payload; t.mctl = mctl;
        tail_payload(t, so.out_loss);
    }
    if (small) {
        if ((tid & ~63) >> 3 >= cnt) return;  // wave-uniform: none of this wavefront's eight ranks holds a line
        const int side = (lane >> 2) & 1, h = lane & 3, k = kp.k, j = kp.j;
        const int mycnt = side ? j : k, ocnt = side ? k : j;
        const bool valid = (tid >> 3) < cnt && C > 0;
        const bool live = valid && h < mycnt && !side;  // (grad_tri2 never rides here)
        float d[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        if (valid && h < mycnt) {
#pragma unroll
            for (int o = 0; o < RRL_MAX_HITS; ++o)
                if (o < ocnt) {  // pair_hit's expression for entry (ra, rb) = (h, o) / (o, h): cloud 1's point minus cloud 2's
                    const float4 p1 = kp.sq[side ? o : h], p2 = kp.sq[4 + (side ? h : o)];
                    const float dx = p1.x - p2.x, dy = p1.y - p2.y, dz = p1.z - p2.z;
                    float sq = dx * dx;
                    sq = sq + dy * dy;
                    sq = sq + dz * dz;
                    d[o] = sq;
                }
        }
        const int S = live ? so.s_cnt[(k - 1) * 4 + (j - 1)] : 1;
        const float4 *sq = kp.sq;
        bwd_scatter_math(live, k, j, side, h, d, S, kp.q, kp.f, kp.w, C, m, gl_in, [&](int o) { return sq[side ? o : 4 + o]; },
                         a.g1 + (size_t)b * a.N * 9, nullptr, s_scat[wave], lane);
        STAMP(8);
        return;
    }
    __threadfence_block();  // (more than one pass: from the workspace -- the finishing wavefront's BCNT is read back)
    __syncthreads();
    for (int r0 = 0; r0 < cnt; r0 += 128)  // uniform
        bwd_scatter_pass(a, b, (size_t)b * 1024, cnt, r0 + (tid >> 3), b, C, m, gl_in, s_scat[wave], lane);
}

// ---------------------------------------------------------------------------------------
// K5' backward of the fused training op when only dL/dR and dL/dt are wanted (the usual case:
// the source cloud is data).  The gradient of a moved point y = x m + t contributes x (x) g to
// dL/dm and g to dL/dt, so every (selected line, hit) lane adds its three points' terms straight
// into 12 per-lane sums: no scatter into a per-triangle gradient, no pass over the N points
// afterwards.  Every live workgroup adds its 12 sums to dL/dR, dL/dt of its sample and to the
// 14-float shard payload with float atomics (26 fire-and-forget adds per workgroup); the outputs
// must be ZERO on entry (the forward clears the workspace field GACC for this).  ONE launch.
// Handing the partials to a "last workgroup" for a fixed-order sum (write-through stores, ticket,
// agent-scope loads: four dependent cross-XCD round trips) cost 5 us more, a second tiny launch
// for it 2 us more (measured); the price of the atomics is run-to-run rounding noise in the
// gradient (the loss itself stays bit-deterministic).
// ---------------------------------------------------------------------------------------
#define BWD_LINES 64  // selected lines per 256-lane workgroup (4 hit slots each)

__device__ __forceinline__ int bwd_live_blocks(int ns) { return ns > 0 ? (ns + BWD_LINES - 1) / BWD_LINES : 1; }

// The (dL/dR, dL/dt) terms of ONE selected line's hit slot h (li: its index within sample b, or -1) added to acc[12]
// (9 sums of x (x) g in m-index order, 3 of g).  ALL lanes of the wavefront must call it: the four lanes of a line share
// its Welsch tile by quad DPP.
// The arithmetic of the direct (dR, dt) backward for one lane = (selected line, hit slot h of cloud 1), four lanes per line (a
// DPP quad); every lane of the wavefront calls, lanes without a line or hit carry +inf rows and k = j = 0.  dr: row h of the
// line's D tile; S: the (k, j) bucket's line count; mine / wq / xs: this hit's intersection point, weights and SOURCE triangle
// (unmoved); qx / qy / qz: cloud 2's intersection points; acc: 9 sums for dL/dm (m-index order) + 3 for dL/dt.
__device__ __forceinline__ void bwd_rt_math(bool live, int k, int j, int h, const float (&dr)[4], int S, float4 mine,
                                            const float (&qx)[4], const float (&qy)[4], const float (&qz)[4],
                                            const float (&wq)[3], const float (&xs)[9], int C, float m, float gl_in, float *acc) {
    const int jmax = (int)wave_max((float)j);  // uniform
    float er[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wr[4];
#pragma unroll
    for (int o = 0; o < RRL_MAX_HITS; ++o) {
        wr[o] = INFINITY;
        if (o < jmax) {
            const float e = expf(-(dr[o] / m) / 2.0f);  // == welsch(): 1 - e
            er[o] = e;
            if (dr[o] < INFINITY) wr[o] = 1.0f - e;
        }
    }
    int arg_b_own = 0, arg_a[4];
    {
        float bestw = wr[0];
#pragma unroll
        for (int o = 1; o < RRL_MAX_HITS; ++o)
            if (wr[o] < bestw) { bestw = wr[o]; arg_b_own = o; }
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o) {
            const float w0 = quad_bcast<0>(wr[o]), w1_ = quad_bcast<1>(wr[o]), w2_ = quad_bcast<2>(wr[o]), w3 = quad_bcast<3>(wr[o]);
            float best = w0;
            int mm = 0;
            if (w1_ < best) { best = w1_; mm = 1; }
            if (w2_ < best) { best = w2_; mm = 2; }
            if (w3 < best) { best = w3; mm = 3; }
            arg_a[o] = mm;
        }
    }
    if (live) {
        const float wkj = expf(-0.5f * (float)abs(k - j));
        const float scale = gl_in * wkj / (float)C;
        const float inv_row = 1.0f / ((float)S * (float)k), inv_col = 1.0f / ((float)S * (float)j);
        float gq[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o) {
            if (o >= j) continue;
            float sw = 0.0f;
            if (arg_b_own == o) sw += inv_row;
            if (arg_a[o] == h) sw += inv_col;
            if (sw == 0.0f) continue;
            // same expressions as loss_bwd_kernel
            const float gD = scale * sw * er[o] / (2.0f * m);
            gq[0] += 2.0f * (mine.x - qx[o]) * gD;
            gq[1] += 2.0f * (mine.y - qy[o]) * gD;
            gq[2] += 2.0f * (mine.z - qz[o]) * gD;
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const float wk = wq[kk] / 3.0f;  // q = mean_k(w_k P_k)
            float gv[3];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) gv[cc] = wk * gq[cc];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const float xc = xs[3 * kk + cc];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) acc[cc * 3 + jj] = fmaf(xc, gv[jj], acc[cc * 3 + jj]);
                acc[9 + cc] += gv[cc];
            }
        }
    }
}

__device__ __forceinline__ void bwd_rt_line(int li, int h, int b, int L, int N, const uint8_t *__restrict__ kj,
                                            const int32_t *__restrict__ hs1, const float *__restrict__ w1,
                                            const float4 *__restrict__ Q1, const float4 *__restrict__ Q2,
                                            const float *__restrict__ D, const float *__restrict__ med,
                                            const int32_t *__restrict__ bcnt, const int32_t *__restrict__ info,
                                            const float *__restrict__ grad_loss, const float *__restrict__ src, float *acc,
                                            int bs /* the source entry of sample b (multi-pose: b % Bt) */) {
    const int C = info[b * 4];
    // The four lanes of a line SHARE its Welsch tile (round 3): lane h evaluates row h -- <= 4 exponentials and divisions
    // where every lane used to evaluate all 16 entries -- and the rows travel by quad DPP; all lanes take part (DPP
    // reads active lanes only), lanes without a line or hit slot carry +inf rows and k = j = 0.  Same expressions, same
    // first-occurrence minima, same results as welsch_block on the whole tile.
    const bool valid = li >= 0 && C > 0;
    const size_t gl = (size_t)b * L + (valid ? li : 0);
    const unsigned c = valid ? kj[gl] : 0u;
    const int k = c & 15, j = c >> 4;
    const bool live = valid && h < k;
    const float m = med[b];
    float dr[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    float4 mine = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float qx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qy[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float wq[3] = {0.0f, 0.0f, 0.0f}, xs[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int S = 1;
    if (live) {
#pragma unroll
        for (int o = 0; o < RRL_MAX_HITS; ++o)
            if (o < j) {
                dr[o] = D[gl * 16 + h * j + o];
                const float4 t = Q2[gl * 4 + o];
                qx[o] = t.x; qy[o] = t.y; qz[o] = t.z;
            }
        S = bcnt[b * 16 + (k - 1) * 4 + (j - 1)];
        mine = Q1[gl * 4 + h];
        const int f = hs1[gl * 4 + h];
        const float *w = w1 + (gl * 4 + h) * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) wq[q] = w[q];
        const float *x = src + ((size_t)bs * N + f) * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q) xs[q] = x[q];
    }
    bwd_rt_math(live, k, j, h, dr, S, mine, qx, qy, qz, wq, xs, C, m, live ? grad_loss[b] : 0.0f, acc);
}

// DET = true (rrl_set_deterministic): instead of the atomics every workgroup stores its 12 sums to
// part[b][workgroup][12] and loss_bwd_rt_finalize_kernel adds them in index order; the assignment of
// lines to workgroups is fixed too (see below) -- bit-reproducible from run to run, one more (tiny) launch.
// (bx, by: the workgroup's place in the backward's grid, gx its first extent -- blockIdx / gridDim in loss_bwd_rt_kernel,
//  decoded from a linear index in bwd_write_kernel)
template <bool DET>
__device__ __forceinline__ void loss_bwd_rt_body(
    const uint8_t *__restrict__ kj, const int32_t *__restrict__ sel, const int32_t *__restrict__ nsel,
    const int32_t *__restrict__ hs1, const float *__restrict__ w1, const float4 *__restrict__ Q1,
    const float4 *__restrict__ Q2, const float *__restrict__ D, const float *__restrict__ med,
    const int32_t *__restrict__ bcnt, const int32_t *__restrict__ info,
    const float *__restrict__ grad_loss, const float *__restrict__ src, float *__restrict__ gR,
    float *__restrict__ gt, float *__restrict__ payload, const float *__restrict__ loss, int B, int N,
    int L, int transpose_r, float *__restrict__ part, const int bx, const int by, const int gx, const int Bt = 0) {
    __shared__ float red[4][12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = by;
    const int h = tid & 3;
    int li = -1;  // this lane's selected line (index within the sample), or none
    if constexpr (DET) {
        // SEL lists the selected lines in the order the pair kernel's workgroups happened to reserve their
        // slots (an atomic): grouping lines into workgroups by SEL position would change the rounding of
        // the partial sums from run to run.  Here workgroup (tile, sub) takes the selected lines of its
        // 1024-line tile with rank 64 sub .. 64 sub + 63 in a FIXED order (round r, then thread), found
        // from the KJ bytes of the tile.
        __shared__ int s_line[BWD_LINES];
        __shared__ int s_wc[4][4];
        const int tile = bx >> 4, sub = bx & 15;
        bool sl[4];
        unsigned long long bm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = tile * 1024 + r * 256 + tid;
            sl[r] = l < L && kj[(size_t)b * L + l] != 0;
            bm[r] = __ballot(sl[r]);
            if (lane == 0) s_wc[r][wave] = __popcll(bm[r]);
        }
        if (tid < BWD_LINES) s_line[tid] = -1;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mine = before;
            for (int w = 0; w < 4; ++w) {
                if (w < wave) mine += s_wc[r][w];
                before += s_wc[r][w];
            }
            const int rel = mine + __popcll(bm[r] & ((1ull << lane) - 1ull)) - BWD_LINES * sub;
            if (sl[r] && rel >= 0 && rel < BWD_LINES) s_line[rel] = tile * 1024 + r * 256 + tid;
        }
        __syncthreads();
        li = s_line[tid >> 2];
    } else {
        const int ns = nsel[b];
        if (bx >= bwd_live_blocks(ns)) return;  // uniform: nothing selected in this slice
        const int i = bx * BWD_LINES + (tid >> 2);
        if (i < ns) li = sel[(size_t)b * L + i];
    }
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    bwd_rt_line(li, h, b, L, N, kj, hs1, w1, Q1, Q2, D, med, bcnt, info, grad_loss, src, acc, (Bt > 0 && b >= Bt) ? b % Bt : b);
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 12; ++q) red[wave][q] = acc[q];
    __syncthreads();
    if (tid < 12) {
        const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        int o = tid;  // m-index (i, j) -> memory order of R
        if (tid < 9 && transpose_r) o = (tid % 3) * 3 + tid / 3;
        if constexpr (DET) {
            part[((size_t)b * gx + bx) * 12 + o] = v;
        } else {
            if (tid < 9) atomicAdd(&gR[b * 9 + o], v); else atomicAdd(&gt[b * 3 + (tid - 9)], v);
            if (payload) atomicAdd(&payload[2 + o], v);
        }
    }
    if (!DET && payload && bx == 0 && b == 0 && tid >= 64 && tid < 66) {  // [sum of valid losses, #valid]
        double sp = 0.0;
        for (int k = 0; k < B; ++k) sp += info[k * 4] > 0 ? (tid == 64 ? (double)loss[k] : 1.0) : 0.0;
        payload[tid - 64] = (float)sp;
    }
}

template <bool DET>
__global__ __launch_bounds__(256) void loss_bwd_rt_kernel(
    const uint8_t *__restrict__ kj, const int32_t *__restrict__ sel, const int32_t *__restrict__ nsel,
    const int32_t *__restrict__ hs1, const float *__restrict__ w1, const float4 *__restrict__ Q1,
    const float4 *__restrict__ Q2, const float *__restrict__ D, const float *__restrict__ med,
    const int32_t *__restrict__ bcnt, const int32_t *__restrict__ info,
    const float *__restrict__ grad_loss, const float *__restrict__ src, float *__restrict__ gR,
    float *__restrict__ gt, float *__restrict__ payload, const float *__restrict__ loss, int B, int N,
    int L, int transpose_r, float *__restrict__ part, int Bt, int xcd_align) {
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_align) xcd_sample_of(bx + (int)gridDim.x * by, (int)gridDim.x, bx, by);  // (uniform) sample `by` on XCD by % 8
    loss_bwd_rt_body<DET>(kj, sel, nsel, hs1, w1, Q1, Q2, D, med, bcnt, info, grad_loss, src, gR, gt, payload, loss, B, N, L,
                          transpose_r, part, bx, by, (int)gridDim.x, Bt);
}

// The direct backward AND the write pass of the next epoch's line sampler in ONE launch (round 4b; rrl_ws.h RrlWriteRider,
// rrl_demo_epoch): 256-lane workgroups both; workgroups [0, tiles x rounds) write, the others run the backward.
struct BwdKArgs {
    const uint8_t *kj;
    const int32_t *sel, *nsel, *hs1;
    const float *w1;
    const float4 *Q1, *Q2;
    const float *D, *med;
    const int32_t *bcnt, *info;
    const float *grad_loss, *src;
    float *gR, *gt, *payload;
    const float *loss;
    int B, N, L, transpose_r;
    float *part;
    int gx;
};
template <bool DET>
__global__ __launch_bounds__(256) void bwd_write_kernel(const BwdKArgs a, const WriteKArgs c) {
    extern __shared__ int s_tc_dyn[];  // the write pass's tile counts [rounds][tiles]
    const int nwrite = c.gx * c.gy, lin = (int)blockIdx.x;
    if (lin < nwrite) {  // uniform per workgroup
        sample_write_body<256>(s_tc_dyn, nullptr, c.rng_state, c.r, c.centers, nullptr, nullptr, c.accept, c.lines, c.filled, 1, c.n,
                               c.rounds, lin % c.gx, lin / c.gx, 0, c.gx, (unsigned)nwrite);
        return;
    }
    const int l2 = lin - nwrite;
    loss_bwd_rt_body<DET>(a.kj, a.sel, a.nsel, a.hs1, a.w1, a.Q1, a.Q2, a.D, a.med, a.bcnt, a.info, a.grad_loss, a.src, a.gR, a.gt,
                          a.payload, a.loss, a.B, a.N, a.L, a.transpose_r, a.part, l2 % a.gx, l2 / a.gx, a.gx);
}

// Fixed-order tail of the deterministic direct backward: one lane per (sample, entry) adds the live
// workgroups' partials in index order (double accumulator), then 14 lanes build the payload over the
// samples in index order.  One workgroup; B * 12 <= a few hundred sums of <= L / 64 terms.
__global__ __launch_bounds__(256) void loss_bwd_rt_finalize_kernel(const float *__restrict__ part,
                                                                   const int32_t *__restrict__ info,
                                                                   const float *__restrict__ loss, float *__restrict__ gR,
                                                                   float *__restrict__ gt, float *__restrict__ payload,
                                                                   int B, int nblk) {
    for (int e = threadIdx.x; e < B * 12; e += 256) {
        const int b = e / 12, o = e % 12;
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += (double)part[((size_t)b * nblk + k) * 12 + o];
        if (o < 9) gR[b * 9 + o] = (float)s; else gt[b * 3 + (o - 9)] = (float)s;
    }
    if (!payload) return;
    __syncthreads();  // gR / gt of every sample are written (same workgroup)
    if (threadIdx.x < 14) {
        const int q = threadIdx.x;
        double s = 0.0;
        for (int b = 0; b < B; ++b) {
            if (q == 0) s += info[b * 4] > 0 ? (double)loss[b] : 0.0;
            else if (q == 1) s += info[b * 4] > 0 ? 1.0 : 0.0;
            else if (q < 11) s += (double)gR[b * 9 + (q - 2)];
            else s += (double)gt[b * 3 + (q - 11)];
        }
        payload[q] = (float)s;
    }
}

// K2 + K3 + K4 + K5' in ONE launch for a single tile of lines (L <= 1024; C5: N = M = 16384, L = 512): the workgroup of
// sample b ran the per-line stage and the reduce of sample b, so it holds everything the direct backward of sample b
// reads -- no other workgroup is involved at all.  Same bodies (pair_body, reduce_body, bwd_rt_line), same results;
// (dR, dt) of a sample by ONE workgroup in a fixed order (deterministic here), payload[0 .. 1] as in the tail kernel.
struct SoloBwd {
    const uint8_t *kj;
    const int32_t *sel, *nsel, *hs1, *bcnt, *info;
    const float *w1, *D, *med, *grad_loss, *src, *loss;
    const float4 *Q1, *Q2;
    float *gR, *gt, *payload;
    uint32_t *mctl;
    int B, N, L, transpose_r;
    int Bt;  // multi-pose (rrl_opts.problems)
};

__global__ __launch_bounds__(1024) void pair_reduce_bwd_kernel(const PairArgs pa, const ReduceArgs ra, const SoloBwd a) {
    __shared__ float s_red[16][12];
    __shared__ __attribute__((aligned(16))) float s_dct[128 * 16];  // the first pass's compact D tiles and (k | j << 4) bytes
    __shared__ uint8_t s_kjct[128];                                 // for the reduce (pair_reduce_scatter_kernel)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bs = (a.Bt > 0 && b >= a.Bt) ? b % a.Bt : b;  // the instance's source entry (multi-pose)
    const float gl_in = a.grad_loss[b];  // (requested now)
    PairKeep kp;
    kp.kjc_lds = s_kjct; kp.dc_lds = s_dct;
    pair_body(pa, b, 0, 1, &kp);
    const int ns = kp.total;       // (uniform)
    const bool small = ns <= 128;  // one pass of the per-line stage: the reduce and the backward read nothing of it back
    // small: this lane's SOURCE triangle (lanes of cloud 1 with a hit), requested now -- the only load of the backward
    const int side = (lane >> 2) & 1, h = lane & 3;
    float xs[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (small && !side && h < kp.k) {
        const float *x = a.src + ((size_t)bs * a.N + kp.f) * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q) xs[q] = x[q];
    }
    if (!small) __threadfence_block();  // this workgroup's KJC / VALS / BLKCNT stores are complete ...
    __syncthreads();                    // ... before any of its lanes reads them back (small: the LDS copies are)
    SoloReduce so;
    so.total = ns; so.kjc = small ? s_kjct : nullptr; so.dc = small ? s_dct : nullptr; so.fwave = 15;
    reduce_body(ra, b, &so);
    const int sc = lane < 16 ? so.s_cnt[lane] : 0, kk = (lane & 15) / 4 + 1, jj = (lane & 3) + 1;
    const int C = __popcll(__ballot(lane < 16 && sc > 0 && kk >= ra.s_m && kk < ra.e_m && jj >= ra.s_n && jj < ra.e_n));
    if (tid == 64 * 15 && a.payload && C > 0) {  // payload[0 .. 1] as in the tail kernel (order-independent)
        TailArgs t;
        t.payload = a.payload; t.mctl = a.mctl;
        tail_payload(t, so.out_loss);
    }
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    if (small) {
        if ((tid & ~63) >> 3 < ns) {  // wave-uniform: this wavefront's eight ranks hold lines
            const int k = kp.k, j = kp.j;  // (0, 0 beyond the last line)
            const bool live = (tid >> 3) < ns && C > 0 && !side && h < k;
            float dr[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            float qx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qy[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (live) {
#pragma unroll
                for (int o = 0; o < RRL_MAX_HITS; ++o)
                    if (o < j) {  // pair_hit's expression for entry (h, o): cloud 1's point minus cloud 2's
                        const float4 p1 = kp.sq[h], p2 = kp.sq[4 + o];
                        const float dx = p1.x - p2.x, dy = p1.y - p2.y, dz = p1.z - p2.z;
                        float sq = dx * dx;
                        sq = sq + dy * dy;
                        sq = sq + dz * dz;
                        dr[o] = sq;
                        qx[o] = p2.x; qy[o] = p2.y; qz[o] = p2.z;
                    }
            }
            const int S = live ? so.s_cnt[(k - 1) * 4 + (j - 1)] : 1;
            // (the cloud-2 lanes of a line form a quad of their own: they carry +inf rows and k = j = 0 like lanes without a line)
            bwd_rt_math(live, side ? 0 : k, side ? 0 : j, h, dr, S, kp.q, qx, qy, qz, kp.w, xs, C, so.out_med, gl_in, acc);
        }
    } else {
        __threadfence_block();  // (more than one pass: through the workspace -- MED / BCNT / INFO of the finishing wavefront too)
        __syncthreads();
        for (int i0 = 0; i0 < ns; i0 += 256) {  // uniform: 256 selected lines per pass, four lanes each
            const int i = i0 + (tid >> 2);
            const int li = i < ns ? a.sel[(size_t)b * a.L + i] : -1;
            bwd_rt_line(li, tid & 3, b, a.L, a.N, a.kj, a.hs1, a.w1, a.Q1, a.Q2, a.D, a.med, a.bcnt, a.info, a.grad_loss, a.src, acc, bs);
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 12; ++q) s_red[wave][q] = acc[q];
    __syncthreads();
    if (tid < 12) {
        float v = 0.0f;
        for (int w = 0; w < 16; ++w) v += s_red[w][tid];
        int o = tid;  // m-index (i, j) -> memory order of R
        if (tid < 9 && a.transpose_r) o = (tid % 3) * 3 + tid / 3;
        if (tid < 9) atomicAdd(&a.gR[b * 9 + o], v); else atomicAdd(&a.gt[b * 3 + (tid - 9)], v);  // (zero on entry)
        if (a.payload) atomicAdd(&a.payload[2 + o], v);
    }
}

static int g_deterministic = -1;  // -1: read RRL_DETERMINISTIC once
extern "C" int rrl_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}
static bool default_deterministic() {
    if (g_deterministic < 0) {
        const char *e = getenv("RRL_DETERMINISTIC");
        g_deterministic = (e && e[0] == '1') ? 1 : 0;
    }
    return g_deterministic == 1;
}

static ScatArgs scat_args(const void *ws, const WsLayout &w, const float *grad_loss, float *g1, float *g2, int N, int M, int L) {
    ScatArgs a;
    a.lidc = w.u32(ws, RRL_WS_LIDC); a.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    a.hs1 = w.i32(ws, RRL_WS_HS1); a.hs2 = w.i32(ws, RRL_WS_HS2); a.bcnt = w.i32(ws, RRL_WS_BCNT); a.info = w.i32(ws, RRL_WS_INFO);
    a.w1 = w.f32(ws, RRL_WS_W1); a.w2 = w.f32(ws, RRL_WS_W2); a.D = w.f32(ws, RRL_WS_D); a.med = w.f32(ws, RRL_WS_MED);
    a.grad_loss = grad_loss;
    a.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); a.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
    a.g1 = g1; a.g2 = g2; a.N = N; a.M = M; a.L = L;
    a.fx = nullptr; a.fxbits = 0; a.fxB = 0;
    return a;
}

static int scat_fx_bits_host(int L) {
    int lg = 1;
    while ((1 << lg) < L && lg < 30) ++lg;
    return 62 - lg;
}

// deterministic: include/rrl.h rrl_set_deterministic -- the scatter accumulates in the workspace's fixed-point field (which this
// call clears and therefore WRITES: the one entry that touches the workspace of a finished forward) and one more launch
// converts; grad_tri1 / grad_tri2 are overwritten, not accumulated.
static int loss_backward_impl(const float *tri1, const float *tri2, const void *ws,
                              size_t ws_bytes, const float *grad_loss, float *grad_tri1,
                              float *grad_tri2, int B, int N, int M, int L, int pool, bool zero1,
                              void *stream, bool deterministic = false) {
    if (!tri1 || !tri2 || !ws || !grad_loss || !grad_tri1) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (zero1 && (rc = rrl_fill(grad_tri1, 0u, sizeof(float) * 9 * (size_t)B * N, s))) return rc;
    if (grad_tri2 && (rc = rrl_fill(grad_tri2, 0u, sizeof(float) * 9 * (size_t)B * M, s))) return rc;
    if (B == 0 || L == 0) return 0;
    ScatArgs sa = scat_args(ws, w, grad_loss, grad_tri1, grad_tri2, N, M, L);
    if (deterministic && (N + M) > 0) {
        sa.fx = (unsigned long long *)((char *)const_cast<void *>(ws) + w.off[RRL_WS_GFIX]);
        sa.fxbits = scat_fx_bits_host(L);
        sa.fxB = B;
        if ((rc = rrl_fill(sa.fx, 0u, 8 * (size_t)B * (N + M) * 9 + 8 * (size_t)B, s))) return rc;
    }
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((L + 1023) / 1024), (unsigned)B, BWDS_SUBS), dim3(256), 0, s, sa, B, pool,
                       B % 8 == 0 && xcd_align_on() ? 1 : 0);
    RRL_LAUNCH_CHECK();
    if (sa.fx) {
        const int nmax = grad_tri2 && M > N ? M : N;
        hipLaunchKernelGGL(scatter_fix_to_float_kernel, dim3((unsigned)(((size_t)nmax * 9 + 255) / 256), (unsigned)B, grad_tri2 ? 2u : 1u),
                           dim3(256), 0, s, sa, B, pool);
        RRL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int rrl_loss_backward(const float *tri1, const float *tri2, const void *ws,
                                 size_t ws_bytes, const float *grad_loss, float *grad_tri1,
                                 float *grad_tri2, int B, int N, int M, int L, int pool,
                                 void *stream) {
    return loss_backward_impl(tri1, tri2, ws, ws_bytes, grad_loss, grad_tri1, grad_tri2, B, N, M, L,
                              pool, true, stream, default_deterministic());
}

int rrl_fused_backward(int B, int N, int M);
int rrl_launch_reg_bwd(const float *src, const float *R, float *g1, float *grad_src, float *partial,
                       float *gR, float *gt, float *payload, const float *loss, const int32_t *info,
                       int32_t *done, int B, int N, int transpose_r, hipStream_t s);

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

int rrl_tri_prepare_clouds(const float *tri1, const float *tri2, void *ws, size_t ws_bytes, int B,
                           int N, int M, int L, int clouds, const RrlXform *xf, const float *line, const RrlCall &o,
                           void *stream);
int rrl_line_tri_scan_clouds(const float *line, void *ws, size_t ws_bytes, int B, int N, int M, int L,
                             int mode, int chunk, int clouds, int lmax_ready, const RrlCall &o, void *stream);
int rrl_sort_capacity(void);
int rrl_cull_scan_can_fuse(int B, int N, int M, int L, const RrlCall &o);  // rrl_cull.hip

// target_ws != NULL: a workspace of the same (B, N, M, L) that already went through a forward with
// the SAME tri2 and line (RPM / FMR evaluate several source poses against one target and one
// line set, rpm/Train_RPM.py:204-231): the target's hit counts and hit lists are copied from it
// and only the source cloud is prepared, sorted and scanned.
// xf != NULL: tri1 is the workspace field TRI1, filled by the prepare step from xf->src.
static int loss_forward_impl(const float *tri1, const float *tri2, const float *line, void *ws,
                             size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                             int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                             const void *target_ws, const RrlXform *xf, RrlCall o, void *stream,
                             const TailBwd *tb = nullptr, bool *bwd_done = nullptr) {
    if (bwd_done) *bwd_done = false;
    if (!tri1 || !tri2 || !line || !ws || !loss) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    if (target_ws == ws) return RRL_E_ARG;
    const int clouds = target_ws ? 1 : 2;
    o.tar_ws = target_ws;
    // multi-pose evaluation (rrl_opts.problems = Bt): the B instances are B / Bt poses of Bt problems; the inputs have Bt
    // entries.  Served by the sorted layout of scan mode cull through the fused entries that move the source (xf); anything
    // else is an argument error (the caller evaluates pose after pose then)
    if (o.problems >= B) o.problems = 0;
    if (o.problems > 0 && (B % o.problems != 0 || !xf || pool || target_ws || mode != RRL_SCAN_CULL ||
                           (N > M ? N : M) > rrl_sort_capacity() || N <= 0 || M <= 0))
        return RRL_E_ARG;
    // prepared clouds (include/rrl.h rrl_opts): honoured by the sorted layout of scan mode cull, with the orders of every
    // cloud this call builds; anything else takes the plain path (same results)
    if (o.prepared() && (mode != RRL_SCAN_CULL || (N > M ? N : M) > rrl_sort_capacity() ||
                         (clouds == 2 && !o.order2 && !(o.flags & RRL_F_TARGET_KEPT))))
        o.order1 = o.order2 = nullptr;
    // a kept target: cloud 2's records / tree / partials stay as the previous call on this workspace left them
    const int build_clouds = o.target_kept() ? 1 : clouds;
    // Chained steps (include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED).  The chain lives where the per-line stage + tail kernel
    // serve the call: they are the ones that leave COUNT1 / COUNT2 and the CHAIN words cleared.
    const bool chain_path = B > 0 && L > 1024 && !pool && clouds == 2 && !o.problems && mode == RRL_SCAN_CULL &&
                            (N > M ? N : M) <= rrl_sort_capacity() && N > 0 && M > 0 &&
                            reduce_kind(o.reduce_mode, B, (L + 1023) / 1024, pool, tb != nullptr) == 2;
    o.leave_clean = (o.flags & RRL_F_CHAIN) && chain_path ? 1 : 0;
    if (o.chain_left) *o.chain_left = o.leave_clean;
    // ... and a step that FINDS them cleared runs source records + target scan + source scan as ONE launch
    o.fused_build = (o.flags & RRL_F_CHAINED) && chain_path && o.target_kept() && !o.count_rider && !o.write_rider &&
                    rrl_cull_scan_can_fuse(B, N, M, L, o) ? 1 : 0;
    o.xf = xf;
    o.tri1_in = tri1;
    int rc;
    RrlRange step("rrl forward");
    if (!o.fused_build) {
        RrlRange r("K1' records + sort + tree");
        if ((rc = rrl_tri_prepare_clouds(tri1, tri2, ws, ws_bytes, B, N, M, L, build_clouds, xf, line, o, stream))) return rc;
    }
    // (a carried-over target: the per-line stage reads cloud 2's hit counts and lists in `target_ws` itself -- pair_args)
    {
        RrlRange r("K1 line<->triangle scan");
        // (the records kernel reduced the lines' maxima whenever it ran: the sorted path)
        const int lmax_ready = (N > M ? N : M) <= rrl_sort_capacity() && B > 0 && (clouds == 2 && M > N ? M : N) > 0 && L > 0;
        if ((rc = rrl_line_tri_scan_clouds(line, ws, ws_bytes, B, N, M, L, mode, chunk, clouds, lmax_ready, o, stream))) return rc;
    }
    if (L >= 1 && L <= 1024 && !pool && B > 0 && o.reduce_mode < 2) {  // one tile of lines per sample: K2 + K3 + K4 in one launch
        RrlRange r("K2 + K3 + K4 (single tile)");
        WsLayout w(B, N, M, L);
        if (ws_bytes < w.total) return RRL_E_WS;
        if (tb && tb->grad_tri1) {  // ... and the scatter backward to points1.grad too (rrl_loss_step_ex)
            hipLaunchKernelGGL(pair_reduce_scatter_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                               pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                               reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0),
                               scat_args(ws, w, tb->grad_loss, tb->grad_tri1, nullptr, N, M, L), tb->payload, w.u32(ws, RRL_WS_MCTL));
            RRL_LAUNCH_CHECK();
            if (bwd_done) *bwd_done = true;
            return 0;
        }
        if (tb && !tb->grad_tri1) {  // ... and the direct backward too (rrl_registration_step)
            SoloBwd sb;
            sb.kj = w.u8(ws, RRL_WS_KJ); sb.sel = w.i32(ws, RRL_WS_SEL); sb.nsel = w.i32(ws, RRL_WS_NSEL);
            sb.hs1 = w.i32(ws, RRL_WS_HS1); sb.bcnt = w.i32(ws, RRL_WS_BCNT); sb.info = w.i32(ws, RRL_WS_INFO);
            sb.w1 = w.f32(ws, RRL_WS_W1); sb.D = w.f32(ws, RRL_WS_D); sb.med = w.f32(ws, RRL_WS_MED);
            sb.grad_loss = tb->grad_loss; sb.src = tb->src; sb.loss = loss;
            sb.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); sb.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
            sb.gR = tb->gR; sb.gt = tb->gt; sb.payload = tb->payload; sb.mctl = w.u32(ws, RRL_WS_MCTL);
            sb.B = B; sb.N = N; sb.L = L; sb.transpose_r = tb->transpose_r; sb.Bt = o.problems;
            hipLaunchKernelGGL(pair_reduce_bwd_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                               pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                               reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0), sb);
            RRL_LAUNCH_CHECK();
            if (bwd_done) *bwd_done = true;
            return 0;
        }
        hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                           pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                           reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0));
        RRL_LAUNCH_CHECK();
        return 0;
    }
    {
        RrlRange r("K2 per-line distances");
        if ((rc = line_pair_dist_impl(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m,
                                      e_n, pool, o, stream, tb != nullptr)))
            return rc;
    }
    RrlRange r("K3+K4 median + Welsch reduce");
    return loss_reduce_impl(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, tb, bwd_done, o, stream);
}

extern "C" int rrl_loss_forward_ex(const float *tri1, const float *tri2, const float *line,
                                   void *ws, size_t ws_bytes, float *loss, int B, int N, int M,
                                   int L, int s_m, int s_n, int e_m, int e_n, int pool, int mode,
                                   int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    return loss_forward_impl(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                             pool, mode, chunk, target_ws, nullptr, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_loss_forward_cached(const float *tri1, const float *tri2, const float *line,
                                       void *ws, size_t ws_bytes, float *loss, int B, int N, int M,
                                       int L, int s_m, int s_n, int e_m, int e_n, int pool, int mode,
                                       int chunk, const void *target_ws, void *stream) {
    return rrl_loss_forward_ex(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                               pool, mode, chunk, target_ws, nullptr, stream);
}

// The drop-in call (code/loss.py:170-232 as the reference's callers use it: one sample, a Python-level
// decision on the result) in ONE entry: the forward, then INFO[0 .. 4 G) (nbuckets, nselected, nvalues, NaN
// flag per group) copied to host_info, then a wait for the stream -- the only entry of the library that
// synchronises, because the reference's return value (a tensor, or None when no bucket is populated, or an
// exit on NaN) is a host-side decision by contract.  host_info: 4 G int32 in host memory (pinned memory makes
// the copy asynchronous up to the wait).
extern "C" int rrl_loss_forward_info(const float *tri1, const float *tri2, const float *line, void *ws,
                                     size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                                     int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                                     const void *target_ws, int32_t *host_info, void *stream) {
    if (!host_info) return RRL_E_ARG;
    int rc = loss_forward_impl(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                               pool, mode, chunk, target_ws, nullptr, rrl_resolve_opts(nullptr), stream);
    if (rc) return rc;
    const int G = pool ? 1 : B;
    if (G <= 0) return 0;
    WsLayout w(B, N, M, L);
    hipError_t e = hipMemcpyAsync(host_info, w.i32(ws, RRL_WS_INFO), sizeof(int32_t) * 4 * (size_t)G,
                                  hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rrl_loss_forward(const float *tri1, const float *tri2, const float *line, void *ws,
                                size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                                int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                                void *stream) {
    return rrl_loss_forward_cached(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m,
                                   e_n, pool, mode, chunk, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// fused training op: rigid transform of the source + loss, and its backward to (dR, dt)
// ---------------------------------------------------------------------------------------
extern "C" int rrl_registration_forward_ex(const float *src, const float *R, const float *t,
                                           const float *tri2, const float *line, void *ws,
                                           size_t ws_bytes, float *loss, int B, int N, int M,
                                           int L, int transpose_r, int s_m, int s_n, int e_m,
                                           int e_n, int mode, int chunk, const void *target_ws,
                                           const rrl_opts *opts, void *stream) {
    if (!src || !R || !t || !tri2 || !line || !ws || !loss) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    // the transform runs inside the prepare step
    const RrlXform xf = {src, R, t, transpose_r, 1};  // 1: clear GACC for the backward's atomics
    return loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m,
                             s_n, e_m, e_n, 0, mode, chunk, target_ws, &xf, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_registration_forward_cached(const float *src, const float *R, const float *t,
                                               const float *tri2, const float *line, void *ws,
                                               size_t ws_bytes, float *loss, int B, int N, int M,
                                               int L, int transpose_r, int s_m, int s_n, int e_m,
                                               int e_n, int mode, int chunk, const void *target_ws,
                                               void *stream) {
    return rrl_registration_forward_ex(src, R, t, tri2, line, ws, ws_bytes, loss, B, N, M, L, transpose_r, s_m, s_n,
                                       e_m, e_n, mode, chunk, target_ws, nullptr, stream);
}

extern "C" int rrl_registration_forward(const float *src, const float *R, const float *t,
                                        const float *tri2, const float *line, void *ws,
                                        size_t ws_bytes, float *loss, int B, int N, int M, int L,
                                        int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                        int chunk, void *stream) {
    return rrl_registration_forward_cached(src, R, t, tri2, line, ws, ws_bytes, loss, B, N, M, L,
                                           transpose_r, s_m, s_n, e_m, e_n, mode, chunk, nullptr,
                                           stream);
}

static int registration_backward_impl(const float *src, const float *R, const float *tri2,
                                      void *ws, size_t ws_bytes, const float *loss,
                                      const float *grad_loss, float *grad_src, float *gR, float *gt,
                                      float *payload, int B, int N, int M, int L, int transpose_r,
                                      const RrlCall &o, void *stream);

// Forward + direct backward of the fused training op in ONE call (dL/dloss is an input, so nothing has to come back
// to the host in between): when the tail kernel serves the shape, the backward rides in its launch (5 launches per
// step instead of 6, and the reduce's and the backward's chains of dependent loads overlap); otherwise exactly
// rrl_registration_forward_cached followed by rrl_registration_backward.  gR, gt (and payload) should be the
// workspace's GACC field, which the forward's first launch clears; other buffers are cleared here first.
extern "C" int rrl_registration_step_ex(const float *src, const float *R, const float *t, const float *tri2,
                                        const float *line, void *ws, size_t ws_bytes, float *loss,
                                        const float *grad_loss, float *gR, float *gt, float *payload, int B, int N,
                                        int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                        int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    return rrl_registration_step_call(src, R, t, tri2, line, ws, ws_bytes, loss, grad_loss, gR, gt, payload, B, N, M, L,
                                      transpose_r, s_m, s_n, e_m, e_n, mode, chunk, target_ws, rrl_resolve_opts(opts), stream);
}
int rrl_registration_step_call(const float *src, const float *R, const float *t, const float *tri2, const float *line,
                               void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *gR, float *gt,
                               float *payload, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n,
                               int mode, int chunk, const void *target_ws, const RrlCall &o, void *stream) {
    if (!src || !R || !t || !tri2 || !line || !ws || !loss || !grad_loss || !gR || !gt) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    const int nblk = (L + 1023) / 1024;
    bool done = false;
    int rc;
    const bool solo = L <= 1024 && o.reduce_mode < 2;  // one tile of lines: per-line stage + reduce + backward by one workgroup per sample
    if (B > 0 && L > 0 && !o.deterministic && (solo || reduce_kind(o.reduce_mode, B, nblk, 0, true) == 2)) {
        float *gacc = w.f32(ws, RRL_WS_GACC);
        hipStream_t s = (hipStream_t)stream;
        if (gR != gacc || gt != gacc + 9 * (size_t)B || (payload && payload != gacc + 12 * (size_t)B)) {
            if ((rc = rrl_fill(gR, 0u, sizeof(float) * 9 * (size_t)B, s))) return rc;
            if ((rc = rrl_fill(gt, 0u, sizeof(float) * 3 * (size_t)B, s))) return rc;
            if (payload && (rc = rrl_fill(payload, 0u, sizeof(float) * 14, s))) return rc;
        }
        const RrlXform xf = {src, R, t, transpose_r, 1};
        const TailBwd tb = {grad_loss, src, gR, gt, payload, transpose_r, nullptr};
        rc = loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0,
                               mode, chunk, target_ws, &xf, o, stream, &tb, &done);
        if (rc || done) return rc;
    } else {
        const RrlXform xf = {src, R, t, transpose_r, 1};
        rc = loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0,
                               mode, chunk, target_ws, &xf, o, stream);
        if (rc) return rc;
    }
    return registration_backward_impl(src, R, tri2, ws, ws_bytes, loss, grad_loss, nullptr, gR, gt, payload, B, N, M, L,
                                      transpose_r, o, stream);
}
extern "C" int rrl_registration_step(const float *src, const float *R, const float *t, const float *tri2,
                                     const float *line, void *ws, size_t ws_bytes, float *loss,
                                     const float *grad_loss, float *gR, float *gt, float *payload, int B, int N,
                                     int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                     int chunk, const void *target_ws, void *stream) {
    return rrl_registration_step_ex(src, R, t, tri2, line, ws, ws_bytes, loss, grad_loss, gR, gt, payload, B, N, M, L,
                                    transpose_r, s_m, s_n, e_m, e_n, mode, chunk, target_ws, nullptr, stream);
}

// SURVEY 8(d)'s own definition -- T-apply + S + P + median + Welsch reduce + backward to points1.grad -- in ONE call
// (include/rrl.h rrl_loss_step_ex): the forward of rrl_loss_forward_ex / rrl_registration_forward_ex with the scatter
// backward of rrl_loss_backward riding in the reduce's launch where the tail kernel serves the shape (the records launch
// clears grad_tri1 with the per-call state: 4 launches with prepared orders); elsewhere, and when grad_tri2 is wanted,
// forward + loss_bwd_kernel.  Bit-identical loss; gradients to the rounding of the float atomics.
extern "C" int rrl_loss_step_ex(const float *tri1, const float *R, const float *t, const float *tri2, const float *line,
                                void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *grad_tri1,
                                float *grad_tri2, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m,
                                int e_n, int mode, int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    if (!tri1 || !tri2 || !line || !ws || !loss || !grad_loss || !grad_tri1 || ((R == nullptr) != (t == nullptr))) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    RrlCall o = rrl_resolve_opts(opts);
    o.clear_ptr = grad_tri1;
    o.clear_bytes = sizeof(float) * 9 * (size_t)B * N;
    const int nblk = (L + 1023) / 1024;
    // the shard payload of the step (rrl_opts.payload): in the workspace's GACC field the records launch clears it with the
    // rest of the accumulator (the fused op's convention); any other buffer is cleared here first
    float *payload = o.payload;
    const bool pay_in_ws = payload && R && payload == w.f32(ws, RRL_WS_GACC) + 12 * (size_t)B;
    if (payload && !pay_in_ws) {
        int rcf = rrl_fill(payload, 0u, sizeof(float) * 14, (hipStream_t)stream);
        if (rcf) return rcf;
    }
    const RrlXform xf = {tri1, R, t, transpose_r, pay_in_ws ? 1 : 0};
    const float *p1 = R ? w.f32(ws, RRL_WS_TRI1) : tri1;  // points1: the moved source, or the caller's triangles as given
    // (the tail kernel, or -- one tile of lines per sample -- the single-tile kernel: loss_forward_impl's own conditions)
    // (deterministic: the fixed-point scatter of loss_bwd_kernel + its conversion launch, never the riding float atomics)
    const bool ride = B > 0 && L > 0 && !grad_tri2 && !o.deterministic &&
                      (L > 1024 ? reduce_kind(o.reduce_mode, B, nblk, 0, true) == 2 : o.reduce_mode < 2);
    // no riding backward, but the exchange reduce serves the call: its last arrivers add the payload (no payload launch)
    const bool pay_in_reduce = payload && !ride && B > 0 && L > 1024 && reduce_kind(o.reduce_mode, B, nblk, 0, false) == 1;
    o.payload_in_reduce = pay_in_reduce ? 1 : 0;
    const TailBwd tb = {grad_loss, nullptr, nullptr, nullptr, payload, 0, grad_tri1};
    bool done = false;
    int rc = loss_forward_impl(p1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0, mode, chunk, target_ws,
                               R ? &xf : nullptr, o, stream, ride ? &tb : nullptr, &done);
    if (rc || done) return rc;
    // (grad_tri1 was cleared by the build step's first launch -- or by its fill on the unsorted path; an empty batch /
    //  cloud launches nothing: clear here)
    if (B == 0 || (N == 0 && M == 0)) return rrl_fill(grad_tri1, 0u, o.clear_bytes, (hipStream_t)stream);
    rc = loss_backward_impl(p1, tri2, ws, ws_bytes, grad_loss, grad_tri1, grad_tri2, B, N, M, L, 0, false, stream, o.deterministic != 0);
    if (rc || !payload || L <= 0 || pay_in_reduce) return rc;
    return rrl_shard_payload(loss, ws, ws_bytes, nullptr, nullptr, payload, B, N, M, L, stream);
}

extern "C" int rrl_registration_backward_ex(const float *src, const float *R, const float *tri2,
                                            void *ws, size_t ws_bytes, const float *loss,
                                            const float *grad_loss, float *grad_src, float *gR, float *gt,
                                            float *payload, int B, int N, int M, int L, int transpose_r,
                                            const rrl_opts *opts, void *stream) {
    return registration_backward_impl(src, R, tri2, ws, ws_bytes, loss, grad_loss, grad_src, gR, gt, payload, B, N, M, L,
                                      transpose_r, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_registration_backward(const float *src, const float *R, const float *tri2,
                                         void *ws, size_t ws_bytes, const float *loss,
                                         const float *grad_loss, float *grad_src, float *gR, float *gt,
                                         float *payload, int B, int N, int M, int L, int transpose_r,
                                         void *stream) {
    return rrl_registration_backward_ex(src, R, tri2, ws, ws_bytes, loss, grad_loss, grad_src, gR, gt, payload, B, N, M, L,
                                        transpose_r, nullptr, stream);
}

static int registration_backward_impl(const float *src, const float *R, const float *tri2,
                                      void *ws, size_t ws_bytes, const float *loss,
                                      const float *grad_loss, float *grad_src, float *gR, float *gt,
                                      float *payload, int B, int N, int M, int L, int transpose_r,
                                      const RrlCall &o, void *stream) {
    if (!src || !R || !tri2 || !ws || !grad_loss || !gR || !gt) return RRL_E_ARG;
    if (payload && !loss) return RRL_E_ARG;
    if (o.problems > 0 && o.problems < B && grad_src) return RRL_E_ARG;  // multi-pose: the direct backward only (dL/dsrc would sum over the poses)
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    float *g1 = w.f32(ws, RRL_WS_G1);
    hipStream_t s = (hipStream_t)stream;
    RrlRange step("K5 rrl backward");
    if (!grad_src && B > 0 && L > 0) {
        // only dL/dR, dL/dt (+ payload): ONE launch, straight from the selected lines; the outputs
        // are accumulated with atomics -- clear them unless they are the workspace's GACC field,
        // which the forward left zeroed
        float *gacc = w.f32(ws, RRL_WS_GACC);
        const int nblk = o.deterministic ? 16 * ((L + 1023) / 1024) : (L + BWD_LINES - 1) / BWD_LINES;
#define RRL_BWD_RT(DET, PART)                                                                                  \
        hipLaunchKernelGGL(loss_bwd_rt_kernel<DET>, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, s,            \
                           w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL), w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), \
                           w.f32(ws, RRL_WS_W1), (const float4 *)w.f32(ws, RRL_WS_Q1),                             \
                           (const float4 *)w.f32(ws, RRL_WS_Q2), w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_MED),       \
                           w.i32(ws, RRL_WS_BCNT), w.i32(ws, RRL_WS_INFO), grad_loss, src, gR, gt, payload, loss,  \
                           B, N, L, transpose_r, PART, o.problems, B % 8 == 0 && xcd_align_on() ? 1 : 0)
        // the next epoch's sampler write pass rides along (bwd_write_kernel; rrl_demo_epoch)
        RrlWriteRider *wr = o.write_rider;
        const int wtiles = wr ? (wr->n + 1023) / 1024 : 0;
        const bool ride = wr && (!o.count_rider || o.count_rider->done) &&  // (the ballots of THAT count pass: it must have ridden)
                          B == 1 && wr->n > 0 && wr->rounds > 0 && (long)wtiles * wr->rounds < 512 &&
                          sizeof(int32_t) * (size_t)wr->rounds * wtiles <= 48 * 1024;
        const WriteKArgs wk = ride ? WriteKArgs{wr->rng_state, wr->r, wr->centers, wr->accept, wr->lines, wr->filled, wr->n,
                                                wr->rounds, wtiles, wr->rounds}
                                   : WriteKArgs{};
#define RRL_BWD_WRITE(DET, PART)                                                                               \
        do {                                                                                                   \
            const BwdKArgs ba = {w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL), w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), \
                                 w.f32(ws, RRL_WS_W1), (const float4 *)w.f32(ws, RRL_WS_Q1),                    \
                                 (const float4 *)w.f32(ws, RRL_WS_Q2), w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_MED), \
                                 w.i32(ws, RRL_WS_BCNT), w.i32(ws, RRL_WS_INFO), grad_loss, src, gR, gt, payload, loss, \
                                 B, N, L, transpose_r, PART, nblk};                                             \
            hipLaunchKernelGGL(bwd_write_kernel<DET>, dim3((unsigned)(wk.gx * wk.gy + nblk * B)), dim3(256),     \
                               sizeof(int32_t) * (size_t)wk.rounds * wk.gx, s, ba, wk);                         \
            wr->done = 1;                                                                                      \
        } while (0)
        if (o.deterministic) {
            // partials in VALS (the reduce kernel's input tiles: dead after the forward; B * Lp * 16 floats
            // >= B * 16 ceil(L / 1024) * 12), fixed-order sums by a second launch: nothing to clear
            float *part = w.f32(ws, RRL_WS_VALS);
            if (ride) RRL_BWD_WRITE(true, part);
            else RRL_BWD_RT(true, part);
            hipLaunchKernelGGL(loss_bwd_rt_finalize_kernel, dim3(1), dim3(256), 0, s, part, w.i32(ws, RRL_WS_INFO), loss,
                               gR, gt, payload, B, nblk);
            RRL_LAUNCH_CHECK();
            return 0;
        }
        if (gR != gacc || gt != gacc + 9 * (size_t)B || (payload && payload != gacc + 12 * (size_t)B)) {
            int rc;
            if ((rc = rrl_fill(gR, 0u, sizeof(float) * 9 * (size_t)B, s))) return rc;
            if ((rc = rrl_fill(gt, 0u, sizeof(float) * 3 * (size_t)B, s))) return rc;
            if (payload && (rc = rrl_fill(payload, 0u, sizeof(float) * 14, s))) return rc;
        }
        if (ride) RRL_BWD_WRITE(false, nullptr);
        else RRL_BWD_RT(false, nullptr);
#undef RRL_BWD_RT
#undef RRL_BWD_WRITE
        RRL_LAUNCH_CHECK();
        return 0;
    }
    if (rrl_fused_backward(B, N, M)) {
        // dL/dsrc wanted too: scatter of the line gradients into G1 (cleared first), then rigid
        // backward + payload (reg_bwd_kernel)
        int rc = loss_backward_impl(w.f32(ws, RRL_WS_TRI1), tri2, ws, ws_bytes, grad_loss, g1, nullptr,
                                    B, N, M, L, 0, true, stream);
        if (rc) return rc;
        return rrl_launch_reg_bwd(src, R, g1, grad_src, w.f32(ws, RRL_WS_RPART), gR, gt, payload, loss,
                                  w.i32(ws, RRL_WS_INFO), w.i32(ws, RRL_WS_STATUS) + 3, B, N, transpose_r,
                                  s);
    }
    int rc = rrl_loss_backward(w.f32(ws, RRL_WS_TRI1), tri2, ws, ws_bytes, grad_loss, g1, nullptr, B,
                               N, M, L, 0, stream);
    if (rc) return rc;
    rc = rrl_rigid_apply_bwd(src, R, g1, grad_src, gR, gt, w.f32(ws, RRL_WS_RPART), B, 3 * N,
                             transpose_r, 0, stream);
    if (rc) return rc;
    if (payload) rc = rrl_shard_payload(loss, ws, ws_bytes, gR, gt, payload, B, N, M, L, stream);
    return rc;
}
