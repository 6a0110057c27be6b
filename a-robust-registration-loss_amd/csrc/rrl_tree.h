// rrl_tree.h -- the spatial order and the sphere tree shared by the culled line scan (rrl_cull.hip)
// and the Chamfer nearest-neighbour kernel (rrl_chamfer.hip): 16^3 grid cells along a Hilbert
// curve, supergroups of 64 sorted records with a 13-node sphere tree each.
#pragma once
#include "rrl_ws.h"

#define GRP 16           // triangles per group
#define SGG 4            // groups per supergroup
#define SGT (GRP * SGG)  // triangles per supergroup
#define NODE 13          // float4 per supergroup in the sphere tree: [0] supergroup, [1..4] groups, [5..12] halves
#define SORT_CAP 65536   // largest cloud of the sorted / culled layout (16-bit sorted positions in the scan)

// Position of the 16^3 grid cell (q0, q1, q2) along a 3-D Hilbert curve (12 bits).  Consecutive
// cells of the curve are always face neighbours, so a group of 16 consecutive sorted triangles
// never spans a jump of the curve the way Morton order does: on the bench clouds a line reaches
// 10.2 group spheres per cloud instead of 17.1 (same cells, same sort).  Axes -> transposed index
// by the standard inversion / exchange sweep from the top bit down, Gray decode, then bit
// interleave.  Any permutation of the cells gives the same labels; only the group shapes change.
__host__ __device__ constexpr unsigned hilbert_cell(unsigned q0, unsigned q1, unsigned q2) {
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

// The same map as a table indexed by q0 | q1 << 4 | q2 << 8 (8 KiB, built at compile time): the
// single-workgroup sort is bound by ONE CU, where ~90 integer ops per triangle cost 1.4 us.
struct HilbertLut {
    unsigned short v[4096];
};
constexpr HilbertLut make_hilbert_lut() {
    HilbertLut t{};
    for (unsigned q2 = 0; q2 < 16; ++q2)
        for (unsigned q1 = 0; q1 < 16; ++q1)
            for (unsigned q0 = 0; q0 < 16; ++q0) t.v[q0 | (q1 << 4) | (q2 << 8)] = (unsigned short)hilbert_cell(q0, q1, q2);
    return t;
}
static __device__ const HilbertLut HILBERT_LUT = make_hilbert_lut();

// ---- sphere tree ------------------------------------------------------------------------
// Node = (centre, Rs): Rs = (rho + max thr) (1 + 5e-5) + 1e-7 with rho = max |P0 - c| rounded up --
// a conservative RADIUS; the scan compares against (Rs + se)^2 with its per-wavefront slack se
// (see "Culling bound").  Empty node: Rs = NaN, which fails every comparison.
__device__ __forceinline__ float4 finish_sphere(float cx, float cy, float cz, float d2, float tm, bool any) {
    if (!any) return make_float4(0.0f, 0.0f, 0.0f, __builtin_nanf(""));  // empty node: never passes
    const float rho = sqrtf(d2) * 1.00001f + 1e-7f;
    float Rs = (rho + tm) * 1.00005f + 1e-7f;
    if (!(Rs < 1.0e18f)) Rs = 1.0e18f;  // non-finite / huge data: keep the node ((Rs + se)^2 stays finite or +inf)
    return make_float4(cx, cy, cz, Rs);
}

#define QUAD_MIN(v) do { v = fminf(v, RRL_DPP_F(v, 0xB1)); v = fminf(v, RRL_DPP_F(v, 0x4E)); } while (0)
#define QUAD_MAX(v) do { v = fmaxf(v, RRL_DPP_F(v, 0xB1)); v = fmaxf(v, RRL_DPP_F(v, 0x4E)); } while (0)

#define OCT_MIN(v) do { QUAD_MIN(v); v = fminf(v, RRL_DPP_F(v, 0x141)); } while (0)  // 8 aligned lanes
#define OCT_MAX(v) do { QUAD_MAX(v); v = fmaxf(v, RRL_DPP_F(v, 0x141)); } while (0)

// Tree nodes: ONE lane per half walks its 8 sorted records (rec(s) -> (P0, thr2)); the two lanes of
// a group and the eight aligned lanes of a supergroup combine their boxes and farthest-point
// distances with DPP (quad_perm xor 1, xor 2, row_half_mirror), so every level costs one pass
// over the lane's own 8 records in registers and no barrier.  (One lane per GROUP, walking 16
// records and deriving three spheres, kept only 4 wavefronts of the single-CU sort kernel busy:
// 3.5 us of its 11.)  ALL 8 lanes of a supergroup must call (halves past the end of the cloud
// contribute nothing).  n = number of real records.
template <class Get>
__device__ __forceinline__ void half_tree(Get rec, int hh, int n, float4 *__restrict__ tree) {
    constexpr int H = GRP / 2;
    const int nv = min(max(n - hh * H, 0), H);
    float px[H], py[H], pz[H];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, tm2 = 0.0f;
    // thr <= sqrtf(thr2) (1 + 2^-22): thr2 is the smallest float whose rounded root reaches thr
#pragma unroll
    for (int t = 0; t < H; ++t) {
        if (t < nv) {
            const float4 v = rec(hh * H + t);
            px[t] = v.x; py[t] = v.y; pz[t] = v.z;
            lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
            lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
            lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
            tm2 = fmaxf(tm2, v.w);
        } else {
            px[t] = py[t] = pz[t] = 0.0f;
        }
    }
    // ONE correctly rounded root per lane instead of eight (sqrtf is monotone: max of the roots == root of the max;
    // eight fix-up sequences were ~40 % of the tree phase's instructions)
    float tm = sqrtf(tm2) * 1.000001f;
    auto radius2 = [&](float cx, float cy, float cz) {
        float d2 = 0.0f;
#pragma unroll
        for (int t = 0; t < H; ++t) {
            const float ex = px[t] - cx, ey = py[t] - cy, ez = pz[t] - cz;
            const float e2 = ex * ex + ey * ey + ez * ez;
            if (t < nv) d2 = fmaxf(d2, e2);
        }
        return d2;
    };
    float4 *node = tree + (size_t)(hh / (2 * SGG)) * NODE;
    const int hi_ = hh % (2 * SGG);  // half within the supergroup
    float any = nv > 0 ? 1.0f : 0.0f;
    {   // the half itself
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        node[5 + hi_] = finish_sphere(cx, cy, cz, radius2(cx, cy, cz), tm, nv > 0);
    }
    {   // the group: this lane and its xor-1 neighbour
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], RRL_DPP_F(lo[c], 0xB1)); hi[c] = fmaxf(hi[c], RRL_DPP_F(hi[c], 0xB1)); }
        tm = fmaxf(tm, RRL_DPP_F(tm, 0xB1));
        any = fmaxf(any, RRL_DPP_F(any, 0xB1));
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        float d2 = nv > 0 ? radius2(cx, cy, cz) : 0.0f;
        d2 = fmaxf(d2, RRL_DPP_F(d2, 0xB1));
        if ((hi_ & 1) == 0) node[1 + (hi_ >> 1)] = finish_sphere(cx, cy, cz, d2, tm, any > 0.0f);
    }
    {   // the supergroup: the 8 aligned lanes (pairs already combined: xor 2, then the half mirror)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], RRL_DPP_F(lo[c], 0x4E)); lo[c] = fminf(lo[c], RRL_DPP_F(lo[c], 0x141));
            hi[c] = fmaxf(hi[c], RRL_DPP_F(hi[c], 0x4E)); hi[c] = fmaxf(hi[c], RRL_DPP_F(hi[c], 0x141));
        }
        tm = fmaxf(tm, RRL_DPP_F(tm, 0x4E)); tm = fmaxf(tm, RRL_DPP_F(tm, 0x141));
        any = fmaxf(any, RRL_DPP_F(any, 0x4E)); any = fmaxf(any, RRL_DPP_F(any, 0x141));
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        float d2 = nv > 0 ? radius2(cx, cy, cz) : 0.0f;
        OCT_MAX(d2);
        if (hi_ == 0) node[0] = finish_sphere(cx, cy, cz, d2, tm, any > 0.0f);
    }
}

// ---- the same tree by ONE LANE PER RECORD (prepared clouds: the order is known, a wavefront holds a supergroup) ----------
__device__ __forceinline__ float xrow_min(float v) {  // rows of 16 already reduced: combine lanes 0, 16, 32, 48
    const int b = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ float xrow_max(float v) {
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}

// The 13 tree nodes of ONE supergroup by the wavefront whose lanes hold its 64 sorted records (valid: a real record).
// Expressions and association as in half_tree (rrl_tree.h): centre = mid-point of the node's AABB, rho^2 = max squared
// distance of its records, thr_max = sqrt(max thr2) (1 + 1e-6): min / max are exact, so the nodes are bit-identical.
// put(j, v): node j of the supergroup := v (a plain store, or the write-through one of a records body whose nodes another
// workgroup of the same launch reads: records_sorted_body<true>)
template <class Put>
__device__ __forceinline__ void wave_tree_put(float px, float py, float pz, float thr2, bool valid, int lane, Put put) {
    float lo[3] = {valid ? px : INFINITY, valid ? py : INFINITY, valid ? pz : INFINITY};
    float hi[3] = {valid ? px : -INFINITY, valid ? py : -INFINITY, valid ? pz : -INFINITY};
    float tm2 = valid ? thr2 : 0.0f, any = valid ? 1.0f : 0.0f;
    auto dist2 = [&](float cx, float cy, float cz) {
        const float ex = px - cx, ey = py - cy, ez = pz - cz;
        const float e2 = ex * ex + ey * ey + ez * ez;
        return valid ? e2 : 0.0f;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) { OCT_MIN(lo[c]); OCT_MAX(hi[c]); }
    OCT_MAX(tm2);
    OCT_MAX(any);
    float tm = sqrtf(tm2) * 1.000001f;
    {   // the half of 8
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        float d2 = dist2(cx, cy, cz);
        OCT_MAX(d2);
        if ((lane & 7) == 0) put(5 + (lane >> 3), finish_sphere(cx, cy, cz, d2, tm, any > 0.0f));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], RRL_DPP_F(lo[c], 0x140)); hi[c] = fmaxf(hi[c], RRL_DPP_F(hi[c], 0x140)); }
    tm = fmaxf(tm, RRL_DPP_F(tm, 0x140));
    any = fmaxf(any, RRL_DPP_F(any, 0x140));
    {   // the group of 16 (one DPP row)
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        const float d2 = row16_max(dist2(cx, cy, cz));
        if ((lane & 15) == 0) put(1 + (lane >> 4), finish_sphere(cx, cy, cz, d2, tm, any > 0.0f));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = xrow_min(lo[c]); hi[c] = xrow_max(hi[c]); }
    tm = xrow_max(tm);
    any = xrow_max(any);
    {   // the supergroup
        const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
        const float d2 = wave_max(dist2(cx, cy, cz));
        if (lane == 0) put(0, finish_sphere(cx, cy, cz, d2, tm, any > 0.0f));
    }
}
__device__ __forceinline__ void wave_tree(float px, float py, float pz, float thr2, bool valid, int lane, float4 *__restrict__ node) {
    wave_tree_put(px, py, pz, thr2, valid, lane, [&](int j, float4 v) { node[j] = v; });
}

#define SORT_CELLS 4096  // 16^3 grid cells in Hilbert-curve order
