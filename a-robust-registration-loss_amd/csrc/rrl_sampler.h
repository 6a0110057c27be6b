// rrl_sampler.h -- device code of the line sampler (K8: code/loss.py:265-432; rrl_geom.hip explains the two passes): the
// box / face tests, the candidate construction and the COUNT pass as a device function, shared by rrl_geom.hip (the
// sampler's own launches) and by the launch that carries the next epoch's count pass beside the per-line stage of the
// demo epoch (pair_count_kernel, rrl_sparse.hip: launches of one stream never overlap on this stack).
#pragma once
#include "rrl_ws.h"

// torch.cross and torch.norm / F.normalize as PyTorch's CPU kernels evaluate them (their vectorised
// loops are compiled with FP contraction): cross_i = fma(a_j, b_k, -(a_k b_j)), the second product
// rounded first; |v| = sqrt(fma(z, z, fma(y, y, x x))).  With exactly these the sub-area test below
// reproduces the reference's accept decisions on tests/golden/sampler.npz bit for bit (unfused
// products: 17 of 400 differ -- the test is a knife-edge equality).  The library is compiled with
// -ffp-contract=off, so only these explicit fmaf calls fuse.
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ float norm3f(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }

// box corner k of the reference's table (code/loss.py:325-351): corner 0 = max, 7 = min
__device__ __forceinline__ void corner(const float *bb /*min3,max3*/, int k, float *o) {
    o[0] = (k & 4) ? bb[0] : bb[3];
    o[1] = (k & 2) ? bb[1] : bb[4];
    o[2] = (k & 1) ? bb[2] : bb[5];
}

static __constant__ int BOX_FACES[12][3] = {{2, 0, 6}, {0, 4, 6}, {5, 4, 0}, {5, 0, 1}, {6, 4, 5}, {5, 7, 6},
                                     {3, 0, 2}, {1, 0, 3}, {3, 2, 6}, {6, 7, 3}, {5, 1, 3}, {3, 7, 5}};

// Does the line cross >= 1 of the 12 box triangles by the reference's sub-area test
// (code/loss.py:265-316)?  hit = all three sub-areas > 0 and their sum <= the triangle area.
// Everything that depends on the box alone -- corners, unit normal, area of each face: a cross
// product, a correctly rounded sqrt and three divisions per face -- is computed ONCE per workgroup
// into an LDS table by 24 lanes (same expressions, same rounding as evaluating it per candidate,
// which made the sampler's two kernels 35 us each at 10 x 20000 candidates).
#define FACE_FLOATS 16  // A[3] B[3] C[3] nh[3] S pad[3]

__device__ __forceinline__ void face_entry(const float *bb, int f, float *o) {
    float A[3], Bq[3], C[3];
    corner(bb, BOX_FACES[f][0], A);
    corner(bb, BOX_FACES[f][1], Bq);
    corner(bb, BOX_FACES[f][2], C);
    float e1[3] = {Bq[0] - A[0], Bq[1] - A[1], Bq[2] - A[2]};
    float e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    float nr[3];
    cross3(e1, e2, nr);
    float S = norm3f(nr[0], nr[1], nr[2]);
    float den = fmaxf(S, 1e-12f);  // F.normalize eps
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = A[c]; o[3 + c] = Bq[c]; o[6 + c] = C[c]; o[9 + c] = nr[c] / den; }
    o[12] = S;
}

// one face: tab = its FACE_FLOATS entry in LDS
__device__ __forceinline__ bool face_hit(const float *tab, const float *ln) {
    const float4 t0 = ((const float4 *)tab)[0], t1 = ((const float4 *)tab)[1], t2 = ((const float4 *)tab)[2],
                 t3 = ((const float4 *)tab)[3];
    const float A[3] = {t0.x, t0.y, t0.z}, Bq[3] = {t0.w, t1.x, t1.y}, C[3] = {t1.z, t1.w, t2.x};
    const float nh[3] = {t2.y, t2.z, t2.w}, S = t3.x;
    float num = nh[0] * (A[0] - ln[3]);
    num = num + nh[1] * (A[1] - ln[4]);
    num = num + nh[2] * (A[2] - ln[5]);
    float dn = nh[0] * ln[0];
    dn = dn + nh[1] * ln[1];
    dn = dn + nh[2] * ln[2];
    float tt = num / (dn + 1e-12f);
    float I[3] = {tt * ln[0] + ln[3], tt * ln[1] + ln[4], tt * ln[2] + ln[5]};
    float ia[3] = {I[0] - A[0], I[1] - A[1], I[2] - A[2]};
    float ib[3] = {I[0] - Bq[0], I[1] - Bq[1], I[2] - Bq[2]};
    float ic[3] = {I[0] - C[0], I[1] - C[1], I[2] - C[2]};
    float c0[3], c1[3], c2[3];
    cross3(ib, ic, c0);
    cross3(ic, ia, c1);
    cross3(ia, ib, c2);
    float ba = norm3f(c0[0], c0[1], c0[2]), bb2 = norm3f(c1[0], c1[1], c1[2]), bc = norm3f(c2[0], c2[1], c2[2]);
    return (ba > 0.0f) && (bb2 > 0.0f) && (bc > 0.0f) && (((ba + bb2) + bc) <= S);
}

// Conservative pre-test: false ONLY when the line certainly stays outside the box inflated by
// pad_a = 1e-3 extent_a + 1e-5 max extent per axis (slab method on the infinite line).  Such a line
// can never pass the reference's test: the point I where it meets a face's plane lies on the line
// (to rounding), hence outside the face rectangle by >= pad along an in-plane axis, so the three
// sub-areas exceed the triangle's area by >= 0.5 * edge * pad >= 5e-6 size^2 relative to a rounding
// error of ~1e-7 size^2 in the cross products (far-away I: the sub-areas dwarf S outright; I from a
// near-zero denominator is inf/NaN and fails every comparison).  With the demo's radius (the full
// box diagonal) ~85 % of the candidates end here, for ~40 instructions instead of ~4000.
// inv[a] = 1 / ln[a] (slab_inv: one division per axis, shared by the two boxes of a candidate): the slab parameters are
// products with it -- 2 ulp from the quotients, against pads of 1e-3 of the extent and a 1e-5 relative tolerance below.
__device__ __forceinline__ void slab_inv(const float *ln, float *inv) {
#pragma unroll
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / ln[a];  // (inf for a zero component: that axis takes the parallel branch)
}
__device__ __forceinline__ bool slab_maybe(const float *bb, const float *ln, const float *inv) {
    const float ext[3] = {bb[3] - bb[0], bb[4] - bb[1], bb[5] - bb[2]};
    const float big = fmaxf(fmaxf(ext[0], ext[1]), ext[2]);
    if (!(big < 3.0e38f)) return true;  // non-finite box: decide by the full test
    float tlo = -INFINITY, thi = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pad = 1e-3f * ext[a] + 1e-5f * big + 1e-30f;
        const float lo = bb[a] - pad, hi = bb[3 + a] + pad, o = ln[3 + a], u = ln[a];
        if (fabsf(u) < 1e-12f) {
            if (o < lo || o > hi) return false;
        } else {
            float t1 = (lo - o) * inv[a], t2 = (hi - o) * inv[a];
            if (t1 > t2) { const float t = t1; t1 = t2; t2 = t; }
            tlo = fmaxf(tlo, t1);
            thi = fminf(thi, t2);
        }
    }
    return !(tlo > thi + 1e-5f * (fabsf(tlo) + fabsf(thi)));  // NaN: keep
}

struct SampleGeom {
    float rad, ctr[3], bb1[6], bb2[6];
    bool filter;
};

__device__ __forceinline__ SampleGeom sample_geom(const float *r, const float *centers, const float *aabb1,
                                                  const float *aabb2, int b) {
    SampleGeom g;
    g.rad = r[b];
#pragma unroll
    for (int c = 0; c < 3; ++c) g.ctr[c] = centers[b * 3 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        g.bb1[c] = aabb1 ? aabb1[b * 6 + c] : 0.0f;
        g.bb2[c] = aabb2 ? aabb2[b * 6 + c] : 0.0f;
    }
    g.filter = aabb1 != nullptr && aabb2 != nullptr;  // NULL boxes: keep every candidate
    return g;
}

// The library's own generator for the sampler's uniforms (round 3; opt-in like every GPU-drawn stream: the reference
// draws from torch's CPU generator).  Philox4x32-10, counter-based: the four uniforms of candidate i of round rd of
// sample b in call number `call` are one block keyed by the seed -- no state to advance per draw, the count and the
// write pass regenerate identical candidates, and a captured step needs no host-side generator bookkeeping (torch's
// GPU generator costs two fill kernels per graph replay to move its offset: ~9 us of the demo's 131 us epoch).
// state[0] = seed, state[1] = call counter (advanced by the LAST workgroup of the write pass), state[2] = its ticket.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned *out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// candidate i of round rd: code/loss.py:394-411.  rands != NULL: the caller's uniforms [rd][4][b][i]; else rng_state.
__device__ __forceinline__ void sample_line(const SampleGeom &g, const float *__restrict__ rands,
                                            const unsigned long long *__restrict__ rng_state, int B, int n,
                                            int b, int rd, int i, float *ln) {
    const float pi32 = 3.14159274101257324f;  // torch.pi of code/loss.py:9
    float u[4];
    if (rands) {
        const float *rr = rands + ((size_t)rd * 4 * B + b) * n;  // [rd][s][b][i]
        const size_t sstride = (size_t)B * n;
        u[0] = rr[i]; u[1] = rr[sstride + i]; u[2] = rr[2 * sstride + i]; u[3] = rr[3 * sstride + i];
    } else {
        const unsigned long long seed = rng_state[0], call = rng_state[1];
        unsigned x[4];
        philox4x32_10((unsigned)i, (unsigned)rd | ((unsigned)b << 16), (unsigned)call, (unsigned)(call >> 32), (unsigned)seed,
                      (unsigned)(seed >> 32), x);
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = (float)(x[q] >> 8) * (1.0f / 16777216.0f);  // 24 bits: [0, 1) like torch.rand
    }
    float al1 = (u[0] * 2.0f) * pi32, v1 = u[1] * 2.0f - 1.0f;
    float al2 = (u[2] * 2.0f) * pi32, v2 = u[3] * 2.0f - 1.0f;
    float s1 = sqrtf(1.0f - v1 * v1), s2 = sqrtf(1.0f - v2 * v2);
    float sn1, cs1, sn2, cs2;  // (sincosf: ONE argument reduction for the pair -- the same bits as sinf / cosf, which each
    sincosf(al1, &sn1, &cs1);  //  evaluate both polynomials after their own reduction and pick one)
    sincosf(al2, &sn2, &cs2);
    float q1[3] = {(g.rad * s1) * cs1, (g.rad * sn1) * s1, g.rad * v1};
    float q2[3] = {(g.rad * s2) * cs2, (g.rad * sn2) * s2, g.rad * v2};
    float d[3] = {q2[0] - q1[0], q2[1] - q1[1], q2[2] - q1[2]};
    float den = fmaxf(norm3f(d[0], d[1], d[2]), 1e-12f);  // F.normalize
#pragma unroll
    for (int c = 0; c < 3; ++c) { ln[c] = d[c] / den; ln[3 + c] = q1[c] + g.ctr[c]; }
}

// One 1024-lane workgroup per (tile of 1024 candidates, round, sample):
//   1. every lane builds its candidate line and runs the conservative slab pre-test on both boxes;
//      the survivors (~15 % at the demo's radius) are compacted into LDS;
//   2. the exact test runs as (survivor, face) TASKS, one per lane per pass -- 24 independent faces
//      per survivor instead of one lane walking a chain of 24 x ~165 dependent instructions (that
//      chain made this kernel 37 us at 10 x 20000 candidates whatever the launch geometry);
//   3. accepted = some face of box 1 AND some face of box 2 (code/loss.py:415-432); one 64-bit
//      ballot per wavefront is stored for the write pass.
// prefilter = 0 sends every candidate through step 2 (tests: identical ballots).
// (aabb1_rows != NULL: box 1 is the AABB of these n_rows partial rows [.][8] = (min xyz, max xyz, ..) -- the loss step's
//  APART rows of the moved source: what rrl_se3_adam_step reduces into the next epoch's box1, taken here by a count pass
//  that rides in the step's per-line launch BEFORE the pose launch has written box1)
struct SampleCountLds {
    __attribute__((aligned(16))) float faces[24][FACE_FLOATS];
    float lines_c[1024][6];
    unsigned short surv[1024];
    unsigned hits[1024];
    unsigned char flag[1024];
    int wave_cnt[16];
};
#ifndef STAMPC
#define STAMPC(i)
#endif
__device__ __forceinline__ void sample_count_body(
    SampleCountLds &lds_, const float *__restrict__ rands, const unsigned long long *__restrict__ rng_state,
    const float *__restrict__ r, const float *__restrict__ centers,
    const float *__restrict__ aabb1, const float *__restrict__ aabb2,
    unsigned long long *__restrict__ accept, int B, int n, int rounds, int prefilter, int rd0,
    const int bx, const int by, const int bz, const int gx, const float *__restrict__ aabb1_rows = nullptr, int n_rows = 0) {
    float (&faces)[24][FACE_FLOATS] = lds_.faces;
    float (&lines_c)[1024][6] = lds_.lines_c;
    unsigned short (&surv)[1024] = lds_.surv;
    unsigned (&hits)[1024] = lds_.hits;
    unsigned char (&flag)[1024] = lds_.flag;
    int (&wave_cnt)[16] = lds_.wave_cnt;
    const int tile = bx, rd = rd0 + by, b = bz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) STAMPC(0);
    if (rd0 > 0) {
        // Rounds rd0.. run as a second launch: the reference skips every round once more than n
        // candidates were accepted (code/loss.py:368-369), and the earlier launch's ballots say
        // exactly how many were (sum over rounds < rd0 > n  <=>  round rd0 and all later ones are
        // skipped).  With the trainers' radius two rounds fill the buffer: 3 of 10 rounds are evaluated.
        int c = 0;
        const unsigned long long *aq = accept + (size_t)b * rounds * gx * 16;
        for (int q = tid; q < rd0 * gx * 16; q += 1024) c += __popcll(aq[q]);
        c = wave_sum_i(c);
        if (lane == 0) wave_cnt[wave] = c;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < 16; ++w) before += wave_cnt[w];
        __syncthreads();
        if (before > n) {  // uniform
            if (lane == 0) accept[(((size_t)b * rounds + rd) * gx + tile) * 16 + wave] = 0ull;
            return;
        }
    }
    SampleGeom g = sample_geom(r, centers, aabb1_rows ? aabb2 : aabb1, aabb2, b);
    if (aabb1_rows != nullptr) {  // uniform (one pair: b = 0)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float v = c < 3 ? INFINITY : -INFINITY;
            for (int q = 0; q < n_rows; ++q) {
                const float x = aabb1_rows[(size_t)q * 8 + c];
                v = c < 3 ? fminf(v, x) : fmaxf(v, x);
            }
            g.bb1[c] = v;
        }
    }
    const int i = tile * 1024 + tid;
    bool ok = i < n;
    if (g.filter) {  // uniform
        if (tid < 24) face_entry(tid < 12 ? g.bb1 : g.bb2, tid % 12, faces[tid]);
        float ln[6];
        bool pre = false;
        if (ok) {
            sample_line(g, rands, rng_state, B, n, b, rd, i, ln);
            float inv[3];
            slab_inv(ln, inv);
            pre = !prefilter || (slab_maybe(g.bb1, ln, inv) && slab_maybe(g.bb2, ln, inv));
        }
        const unsigned long long m = __ballot(pre);
        if (wave == 0) STAMPC(1);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        hits[tid] = 0;
        flag[tid] = 0;
        __syncthreads();
        // (every wavefront scans the 16 counts itself: one LDS read + a DPP prefix, not 16 reads per lane)
        const int cw = lane < 16 ? wave_cnt[lane] : 0, iw = wave_incl_scan(cw);
        const int woff = __builtin_amdgcn_readlane(iw - cw, wave), total = __builtin_amdgcn_readlane(iw, 15);
        if (pre) {
            const int k = woff + __popcll(m & ((1ull << lane) - 1ull));
            surv[k] = (unsigned short)tid;
#pragma unroll
            for (int c = 0; c < 6; ++c) lines_c[k][c] = ln[c];
        }
        __syncthreads();
        if (wave == 0) STAMPC(2);
        for (int t = tid; t < total * 24; t += 1024) {
            const int k = t / 24, f = t % 24;
            if (face_hit(faces[f], lines_c[k])) atomicOr(&hits[k], f < 12 ? 1u : 2u);
        }
        __syncthreads();
        if (wave == 0) STAMPC(3);
        if (tid < total && hits[tid] == 3u) flag[surv[tid]] = 1;
        __syncthreads();
        ok = flag[tid] != 0;
    }
    const unsigned long long mask = __ballot(ok);
    // ballot of wave w of tile t sits at [b][rd][t][w]
    if (lane == 0) accept[(((size_t)b * rounds + rd) * gx + tile) * 16 + wave] = mask;
    if (wave == 0) STAMPC(4);
}

// The WRITE pass of one (tile of 1024 candidates, round, sample) by a workgroup of LANES lanes (1024: sample_write_kernel;
// 256: the rider of the direct backward's launch, bwd_write_kernel in rrl_sparse.hip, four passes over the tile's sixteen
// ballots).  s_tc: LDS, rounds x ntiles ints.  nwg: the write pass's workgroups in this launch (the generator's ticket).
template <int LANES>
__device__ __forceinline__ void sample_write_body(
    int *s_tc, const float *__restrict__ rands, unsigned long long *__restrict__ rng_state, const float *__restrict__ r,
    const float *__restrict__ centers, const float *__restrict__ aabb1, const float *__restrict__ aabb2,
    const unsigned long long *__restrict__ accept, float *__restrict__ lines, int32_t *__restrict__ filled,
    int B, int n, int rounds, const int bx, const int by, const int bz, const int ntiles, const unsigned nwg) {
    constexpr int NW = LANES / 64, PASSES = 1024 / LANES;
    __shared__ int s_total, s_w[NW];
    const int tile = bx, rd = by, b = bz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = rounds * ntiles;
    for (int q = tid; q < E; q += LANES) {
        const unsigned long long *aq = accept + ((size_t)b * E + q) * 16;
        int c = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) c += __popcll(aq[w]);
        s_tc[q] = c;
    }
    const unsigned long long *am = accept + (((size_t)b * rounds + rd) * ntiles + tile) * 16;
    __syncthreads();
    // Exclusive prefix of the tile counts in (round, tile) order by the whole workgroup (one lane walking the 200 entries
    // of the demo's call through LDS took 5 of this kernel's 12.6 us).  The reference skips a round once more than n
    // candidates were accepted BEFORE it (code/loss.py:368-369); the counts only grow, so every later round is skipped
    // too, and the slots of the rounds that are not skipped are the plain prefix: base = prefix at (rd, tile), skipped =
    // prefix at the round's first tile > n, total = prefix at the first skipped round (else the grand total).
    int carry = 0;
    for (int base0 = 0; base0 < E; base0 += LANES) {  // uniform
        const int q = base0 + tid;
        const int v = q < E ? s_tc[q] : 0;
        const int incl = wave_incl_scan(v);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int off = carry, all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { off += w < wave ? s_w[w] : 0; all += s_w[w]; }
        if (q < E) s_tc[q] = off + incl - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) s_total = carry;
    __syncthreads();
    for (int q = tid; q < rounds; q += LANES)
        if (s_tc[q * ntiles] > n) atomicMin(&s_total, s_tc[q * ntiles]);
    __syncthreads();
    const int s_base = s_tc[rd * ntiles + tile];
    const bool s_skip = s_tc[rd * ntiles] > n;
    const int total = s_total;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int w16 = p * NW + wave;  // this wavefront's ballot of the tile
        const unsigned long long mask = am[w16];
        int woff = 0;
        for (int w = 0; w < w16; ++w) woff += __popcll(am[w]);
        const int i = tile * 1024 + 64 * w16 + lane;
        const bool ok = (mask >> lane) & 1ull;
        if (ok && !s_skip) {
            const int slot = s_base + woff + __popcll(mask & ((1ull << lane) - 1ull));
            if (slot < n) {
                const SampleGeom g = sample_geom(r, centers, aabb1, aabb2, b);
                float ln[6];
                sample_line(g, rands, rng_state, B, n, b, rd, i, ln);
                float *dst = lines + ((size_t)b * n + slot) * 6;
#pragma unroll
                for (int c = 0; c < 6; ++c) dst[c] = ln[c];
            }
        }
        // unfilled rows stay zero: the workgroups of round 0 clear the part of their tile beyond the total
        if (rd == 0 && i < n && i >= total) {
            float *dst = lines + ((size_t)b * n + i) * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) dst[c] = 0.0f;
        }
    }
    if (rd == 0 && tile == 0 && tid == 0) filled[b] = total;
    if (rng_state != nullptr) {  // the next call draws the next block of the stream: advanced by whoever finishes last
        __syncthreads();         // (every lane of this workgroup has drawn its candidate)
        if (tid == 0) {
            unsigned *ticket = (unsigned *)(rng_state + 2);
            if (atomicAdd(ticket, 1u) == nwg - 1u) {
                rng_state[1] += 1ull;
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
