// rrl_cull.hip -- the build kernels (transform + records + cell sort + sphere tree) and K1 with
// hierarchical sphere culling (scan mode RRL_SCAN_CULL).  Compiled with -fno-slp-vectorize:
// packed fp32 issues at half rate on gfx950, so SLP-packing the all-VGPR exact test only adds
// register shuffling (the packed code below is explicit).
//
// The full scan evaluates every (line, triangle) pair although only ~6e-4 of them can pass even
// the first point's test.  Here:
//   build (tri_records_kernel + tri_sort_kernel, or the wide big_* kernels beyond 4096 triangles)
//     orders the triangles by the 16^3 grid cell of P0, cells along a Hilbert curve (counting
//     sort), and writes, in that order, 16-byte (P0, thr2) records (P0S), their original indices
//     (IDX) and a three-level SPHERE TREE over consecutive sorted records: per supergroup of 64
//     triangles one node of 13 float4 = its own sphere, the spheres of its 4 groups of 16 and of
//     their 8 halves of 8.  Every sphere bounds the P0s of its records: centre c, rho = max |P0-c|
//     and the conservative squared radius R2 = ((rho + max thr)^2)(1 + 1e-4) + 1e-7.
//   cull_scan_kernel: a workgroup = one slice of <= SPW supergroups of one cloud (records and
//   tree nodes staged in LDS once, ~10 KiB) x WPB wavefronts of 128 lines each.  Every wavefront
//   walks the tree level by level through private LDS queues, with no workgroup synchronisation
//   after the staging barrier:
//     level A  every lane tests ITS two lines (packed fp32) against each supergroup sphere
//              (wave-uniform sphere, scalar loads); passing (line, supergroup) pairs -> queue A;
//     level B  the lanes pop one (line, supergroup) pair each and test its 4 group spheres
//              -> queue B;   level C: (line, group) -> the 2 half spheres -> queue C;
//     level D  the lanes pop one (line, half) pair each and run a conservative point-0 PREFILTER (FMA
//              chain, below) on the half's 8 records.  Survivors (~3 % of the tested records) are
//              parked and resolved densely with the reference's own unfused arithmetic on all three
//              points (dist_sq, bit-identical to the strict scan; two dependent global loads).
//   Below level A every lane works on a different tree node at the same time, and a level only
//   runs when 64 pairs are waiting (or at the end), so the lanes stay full however unevenly the
//   pairs are spread over the lines.  Per line and cloud at the bench shape: 64 + 8.1 x 4 + 10.2 x 2
//   sphere tests and 10.2 x 8 exact tests, against 256 sphere tests + 10.2 x 16 exact tests of the
//   single-level version of this kernel (17.1 x 16 in Morton order).
//
// Culling bound (labels can never be lost) -- valid at ANY finite data scale.
// Line = (d, x0) with s = |d|^2 <= 1 + 1e-6 as evaluated in fp32; a = P - x0; A >= |a| for every
// point of the cloud (A = |x0| + max|P|); u = 2^-24.  Exact-arithmetic quantities:
//   Q(P) = |a|^2 - (a.d)^2        what the reference evaluates (code/loss.py:84-88)
//   p(P) = distance of P to the line through x0 along d/|d| -- a seminorm of a, 1-Lipschitz in P
//   Q = p^2 - eta (a.d/|d|)^2,  eta = |d|^2 - 1   (|d| <= 1: Q itself is a squared seminorm >= p^2)
// The reference's fp32 value x_ref = fl((dAC - proj) + 2e-4) satisfies
//   |x_ref - (Q(P) + 2e-4)| <= 30u |a|^2   (3u dAC, 7.1u proj, 1u difference, 4u rounding of a; ~2x spare)
// A hit needs x_ref(P0) < thr2 <= thr^2 (1 + 2u) =: T, hence
//   Q(P0) < T + g+,            g = 30u A^2 - 2e-4   (g <= 0: the 2e-4 dominates the rounding noise)
//   p(P0)^2 < T + g+ + x,      x = eta+ A^2         (only needed when |d| > 1)
// and, because x_ref >= 2e-4 - 30u A^2 for every pair, a hit with g <= 0 also needs T > -g.  So with
//   se = sqrt(g + x)                          when g > 0
//   se = min( sqrt(x), x / (2 sqrt(-g)) )     when g <= 0    (sqrt(T + x) - sqrt(T) <= x / (2 sqrt T))
// a hit implies  p(P0) < thr (1 + u) + se  (|d| > 1)  or  sqrt(Q(P0)) < thr (1 + u) + se  (|d| <= 1),
// and by the Lipschitz property the same function of the CENTRE c of any sphere that bounds the
// triangle's P0 (its half, its group and its supergroup alike) is < thr_max (1 + u) + rho + se; in both
// cases Q(c) is <= that square.  A node stores Rs >= (rho + thr_max)(1 + 5e-5) and the test keeps it when
//   d2 - 4e-6 |a_c|^2 <= (Rs + se_w)^2,   d2 = fl(Q(c)) by FMAs (error <= 14u |a_c|^2),
// se_w = the largest se of the wavefront's 128 lines (x 1.0001).  At unit scale with normalised
// directions se_w ~ 6e-5 (0.06 % of a typical radius); at the demo's scale (radius 11.7,
// A^2 ~ 300: g ~ 3.4e-4) se_w ~ 0.02, some 5 % of the radii there -- the reference's own labels are
// that noisy at this scale.  Lines with s > 1 + 1e-6 or non-finite data are never culled: a wavefront
// that holds one evaluates all its (line, triangle) pairs of the slice with the strict loop
// (status[1] counts such wavefronts).
//
// Point-0 prefilter (level D).  The kernel is VALU-issue bound and level D was 35 % of its instructions:
// the reference's unfused evaluation of x_ref(P0) costs 16 operations + compare + mask per record.  The
// prefilter evaluates e = fl(Q(P0)) + c with one FMA chain (3 subtractions, 3 + 3 + 1 FMAs; the staged
// record carries c in place of thr2) and takes its SIGN BIT (one v_alignbit per record), 11 operations:
//   |fl(Q) - Q| <= 14.2 u |a|^2 + 4 u |c|   (a rounded once; three-term FMA chains; |d| ~ 1)
//   hit  =>  Q < thr2 - 2e-4 + 30 u |a|^2                                   (as above)
//   c = -(T' + 1e-6 |T'| + s0),  T' = fl(thr2 - 2e-4),  s0 = 3e-6 A^2 + 1e-9  (3e-6 = 50.3 u)
// so a hit always gives e < 0: the 1e-6 |T'| covers the rounding of T' and 4 u |c|, 50.3 u A^2 covers
// 44.2 u |a|^2 (|a| <= A), and the floor covers u 4e-4 when everything sits at the origin.  A^2 and the
// node slack se are evaluated at the SAMPLE's line maxima (LMAX, from the records kernel: cull_cloud_slack),
// identically in every wavefront of the launch.  e < 0 only PARKS the triangle; whether it is a hit is decided by the exact arithmetic
// in resolve_candidate, so a false candidate costs time, never a label.
//
// NaN (negative sqrt argument; the reference prints and exits, code/loss.py:88-91): provably
// impossible when g <= 0 for every line of the wavefront (A^2 < 111; all unit-scale training data).
// Otherwise (g > 0, e.g. the demo's full-diagonal radius) the walk is WIDENED so that every pair that could
// produce one is evaluated exactly (round 3; the detection equals the strict scan's):
//   x_ref(P_k) < 0 for a point k of triangle f  =>  Q(P_k) < g  =>  sqrt(Q(P_k)) < se (|d| <= 1) resp.
//   p(P_k) < se (|d| > 1) with the line's own se = sqrt(g + x)  =>  (1-Lipschitz, |P_k - P0| <= e01_f =
//   max(|P1-P0|, |P2-P0|))  the same function of P0 is < se + e01_f, hence Q(P0) < (se + e01_f)^2, and of the
//   centre of any sphere that bounds the triangle's P0: < rho + se + max e01.
// The records kernel stores del_f >= e01_f - thr_f (clamped at 0, DEL1 / DEL2); a workgroup of a (cloud, sample)
// with g > 0 ("nanwide") gathers del for its 512 staged records (requested while the lines are still in flight),
// widens every staged node radius by the node's max del (rho + e01 <= rho + thr_max + max del <= Rs + max del;
// the values meet in LDS, one more barrier) and the prefilter constant of every record to
// max(thr2 - 2e-4, (se + sqrt(thr2) + del_f)^2) (+ the same evaluation slack): a triangle whose point 0, 1 or 2
// could see a negative argument is then a prefilter candidate, resolve_candidate evaluates its three points with
// the reference's arithmetic and raises STATUS[0].  Unit-scale data never takes this branch (one uniform test
// per workgroup).  At the demo's scale (diagonal 11.7) it costs 13.0 -> 16.4 us at C1 and 32.7 -> 42.2 at C2,
// in three similar parts: the ring thr -> thr + se, the reach del, and the dependent gather + barrier.
#include <stdio.h>
#include <stdlib.h>

#include "rrl_ws.h"

#include "rrl_tree.h"
#include "rrl_chamfer_walk.h"  // the Chamfer walk as a device function: cull_scan_chamfer_kernel carries it


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
    uint4 *z3;                     // per-call state of the tiled reduce (MHIST, MCTL, MSUM; may be NULL)
    size_t z3_vec4;
    uint32_t *z4;                  // a caller-owned buffer to clear (RrlCall::clear_ptr: the scatter target of rrl_loss_step)
    size_t z4_words;
    uint32_t *z5;                  // the CHAIN words (include/rrl.h RRL_WS_CHAIN): cleared by every records launch EXCEPT the
    size_t z5_words;               //   chained step's fused one, whose scan workgroups are using them (NULL there)
    float *del1, *del2;            // NaN reach of every triangle (may be NULL: not stored)
    const float *line;             // the samples' lines [B][L][6] and where their partial maxima go (may be NULL:
    float2 *lmax;                  //   the scan entry computes them itself then)
    int L;
    int nblk_tri;                  // tri_records_kernel: workgroups [0, nblk_tri) of a (sample, cloud) hold triangles
    uint32_t *zwords;              // tri_sort_kernel (Chamfer path): words its first workgroup clears (may be NULL)
    int nzwords;
    int B, N, M, transpose_r, nblk;
    int nchunk;                    // tri_sort_kernel: chunks of 4096 records per cloud (1: the whole cloud)
    int clouds;                    // clouds built by this launch (1: the target is kept)
    int xcd_align;                 // tri_records_sorted_kernel: run the workgroups of (cloud, sample) pair p on XCD p % 8 -- where
                                   // the culled scan's workgroups of that pair run (its grid has the pair on the fast index)
    int Bt;                        // multi-pose evaluation (rrl_opts.problems): the INPUT clouds, orders and lines have Bt
                                   // entries and instance b uses entry b % Bt; 0 / B: every instance has its own
};
// the input entry of instance b (multi-pose: rrl_opts.problems)
__device__ __forceinline__ int input_of(int b, int Bt) { return (Bt > 0 && b >= Bt) ? b % Bt : b; }

#define REC_BLK 256
#define LMAX_CHUNKS 64  // per-sample partial maxima of the lines' |dir|^2 and |x0|^2 (-> the culled scan's slack)

#ifdef RRL_STAMPS  // experiments only (RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp): per-workgroup 100 MHz time stamps of the records launch
__device__ unsigned long long g_rstamps[8 * 2048];
#define STAMPR(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 2048u) g_rstamps[(i) * 2048 + wg_] = wall_clock64(); } } while (0)
extern "C" int rrl_debug_rstamps(unsigned long long *out, int clear) {
    if (clear) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_rstamps)) != hipSuccess) return -1; return hipMemset(p_, 0, sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstamps), sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1;
}
#else
#define STAMPR(i)
#endif

// |dir|^2 and |x0|^2 of a line as BOTH this pass (partial maxima) and the scan (per-line admission) evaluate them:
// the same fp32 expressions, so a line the scan admits is covered by the maxima.
__device__ __forceinline__ void line_norms(float dx, float dy, float dz, float ox, float oy, float oz, float &s, float &o2) {
    s = dx * dx + dy * dy + dz * dz;
    o2 = ox * ox + oy * oy + oz * oz;
}
// A line can be culled when its direction is (at most) unit length as evaluated and its offset is moderate; any
// NaN fails both comparisons.  Lines that fail send their wavefront through the strict loop.
__device__ __forceinline__ bool line_cullable(float s, float o2) { return s <= 1.000001f && o2 <= 1.0e11f; }

// Chunks ch0, ch0 + stride, ... (< LMAX_CHUNKS) of sample b's lines by one 256-lane workgroup: lmax[b][ch] =
// (max |dir|^2, max |x0|^2) over the chunk's cullable lines (0, 0 for none).  The culled scan derives its slacks
// from the maxima over the 64 chunks, identically in every wavefront -- no exchange inside that kernel.
__device__ __forceinline__ void line_max_chunks(const float *__restrict__ line, int L, float2 *__restrict__ lmax, int b,
                                                int ch0, int stride, float (*red2)[2], int bl /* the lines' entry */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CL = (L + LMAX_CHUNKS - 1) / LMAX_CHUNKS;
    for (int ch = ch0; ch < LMAX_CHUNKS; ch += stride) {  // uniform
        float sm = 0.0f, om = 0.0f;
        const int lend = min(L, (ch + 1) * CL);
        for (int l = ch * CL + tid; l < lend; l += REC_BLK) {
            const float2 *p = (const float2 *)(line + ((size_t)bl * L + l) * 6);  // 24-byte rows: 8-byte aligned
            const float2 q0 = p[0], q1 = p[1], q2 = p[2];                       // dir.xy | dir.z x0.x | x0.yz
            float s, o2;
            line_norms(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, s, o2);
            if (line_cullable(s, o2)) { sm = fmaxf(sm, s); om = fmaxf(om, o2); }
        }
        sm = wave_max(sm);
        om = wave_max(om);
        __syncthreads();  // red2 free again
        if (lane == 0) { red2[wave][0] = sm; red2[wave][1] = om; }
        __syncthreads();
        if (tid == 0) {
            float a0 = red2[0][0], a1 = red2[0][1];
            for (int w = 1; w < REC_BLK / 64; ++w) { a0 = fmaxf(a0, red2[w][0]); a1 = fmaxf(a1, red2[w][1]); }
            lmax[(size_t)b * LMAX_CHUNKS + ch] = make_float2(a0, a1);
        }
    }
}

// the same pass on its own, for callers that prepared the triangles without the lines (rrl_tri_prepare + rrl_line_tri_scan)
__global__ __launch_bounds__(REC_BLK) void line_max_kernel(const float *__restrict__ line, int L, float2 *__restrict__ lmax, int Bt) {
    __shared__ float red2[REC_BLK / 64][2];
    line_max_chunks(line, L, lmax, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, red2, input_of((int)blockIdx.y, Bt));
}

// One triangle of the build step (both records kernels): the raw row, moved by the sample's rigid transform when it
// belongs to the source of the fused op (-> TRI1), its thresholds (code/loss.py:94-110), NaN reach (DEL) and 48-byte
// prepared record (PTRI, original order).  c: the (moved) coordinates, x: thr2, p2: max |P|^2 of its three points
// (+inf for non-finite coordinates).
// prow: the row of PTRI the record goes to -- f (original order: the cold build) or the triangle's SORTED position (prepared
// build: the culled scan then resolves a candidate from its position alone, without the IDX hop; the record carries f).
// PUB (the chained step's build + scan launch): the record and its NaN reach are read by other workgroups of the SAME launch
// -- write-through (sc1) stores, rrl_common.h; TRI1 is read by later launches only: plain.
template <bool PUB = false>
__device__ __forceinline__ void tri_record_row(const BuildArgs &a, int cloud, int b, int n, int f, int prow, float (&c)[9],
                                               float &x, float &p2) {
    const float *raw = (cloud ? a.tri2 : a.tri1) + ((size_t)input_of(b, a.Bt) * n + f) * 9;
    float thr, e01;
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
        if constexpr (!PUB) {  // (PUB: the caller stores TRI1 -- nine scattered 4-byte stores -- BEHIND its ticket: rec_late_stores)
            float *moved = a.tri1_out + ((size_t)b * n + f) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) moved[i] = c[i];
        }
    }
    tri_thresholds(c, &thr, &x, &e01);  // code/loss.py:94-110
    // NaN reach (culled scan, "NaN detection" in the header): points 1, 2 lie within e01 of point 0, and the
    // tree nodes carry thr: del >= e01 - thr in exact arithmetic (e01, thr as rounded here: <= 3u off)
    if (float *del = cloud ? a.del2 : a.del1) {  // (at the record's row, like PTRI)
        const float dv = fmaxf(e01 * 1.000002f - thr, 0.0f) * 1.000001f;
        if constexpr (PUB) st4_sc1(&del[(size_t)b * n + prow], dv);
        else del[(size_t)b * n + prow] = dv;
    }
    if constexpr (PUB) {
        const __amdgpu_buffer_rsrc_t rs = rrl_rsrc((cloud ? a.ptri2 : a.ptri1) + (size_t)b * n * PTRI_STRIDE, (size_t)n * PTRI_STRIDE * 4);
        const unsigned o = (unsigned)prow * (PTRI_STRIDE * 4);
        st16_sc1(rs, o, make_float4(c[0], c[1], c[2], c[3]));
        st16_sc1(rs, o + 16, make_float4(c[4], c[5], c[6], c[7]));
        st16_sc1(rs, o + 32, make_float4(c[8], x, thr, __int_as_float(f)));
    } else {
    float4 *row = (float4 *)((cloud ? a.ptri2 : a.ptri1) + ((size_t)b * n + prow) * PTRI_STRIDE);
    row[0] = make_float4(c[0], c[1], c[2], c[3]);
    row[1] = make_float4(c[4], c[5], c[6], c[7]);
    row[2] = make_float4(c[8], x, thr, __int_as_float(f));
    }
    p2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        p2 = fmaxf(p2, c[3 * q] * c[3 * q] + c[3 * q + 1] * c[3 * q + 1] + c[3 * q + 2] * c[3 * q + 2]);
    if (!(p2 <= 3.0e38f)) p2 = INFINITY;  // NaN/inf coordinates: never "provably safe"
}

// Place of a workgroup in a records launch (1-D grid, round 5): the TRIANGLE workgroups -- nblk_tri per (cloud, sample) pair,
// the launch's critical path: order -> raw row -> record -> stores -- take the lowest ids and so leave the dispatcher first; the
// workgroups that reduce the lines' maxima (LMAX_CHUNKS per sample, one chunk each, a quarter of the time) follow.  The grid
// used to interleave them per sample (16 + 64 at C2): the last sample's triangle workgroups entered 2 us after the first
// (in-kernel time stamps, tools/stamps_records.py) behind 448 line workgroups.  xcd_align: pair p = cloud * B + b on XCD p % 8
// -- workgroups go to the XCDs round-robin by linear id --, where its sort and scan workgroups run.
struct RecPlace {
    int cloud, b, bxr;  // triangle workgroup bxr of (cloud, sample b); or
    int lch;            // >= 0: the line-maxima workgroup of chunk lch of sample b
};
__device__ __forceinline__ RecPlace rec_place(const BuildArgs &a, int clouds, int lin) {
    const int T = a.nblk_tri * a.B * clouds;
    RecPlace r;
    r.lch = -1;
    if (lin >= T) {  // uniform
        const int j = lin - T;
        r.cloud = 0; r.b = j / LMAX_CHUNKS; r.bxr = 0; r.lch = j - r.b * LMAX_CHUNKS;
        return r;
    }
    int p;
    if (a.xcd_align) {
        const int slot = lin >> 3;
        p = (lin & 7) + 8 * (slot / a.nblk_tri);
        r.bxr = slot % a.nblk_tri;
    } else {
        p = lin / a.nblk_tri;
        r.bxr = lin - p * a.nblk_tri;
    }
    r.cloud = p / a.B; r.b = p - r.cloud * a.B;
    return r;
}

// the clearing of the per-call state (and of the small accumulators), spread over all workgroups of a records launch
__device__ __forceinline__ void build_clear_state(const BuildArgs &a, size_t me, size_t nthr) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = me; i < a.zero_vec4; i += nthr) a.zero_base[i] = z;
    for (size_t i = me; i < a.g1_vec4; i += nthr) a.g1[i] = z;
    for (size_t i = me; i < a.z2_vec4; i += nthr) a.z2[i] = z;
    for (size_t i = me; i < a.z3_vec4; i += nthr) a.z3[i] = z;
    for (size_t i = me; i < a.z4_words; i += nthr) a.z4[i] = 0u;  // (4-byte stores: any alignment)
    for (size_t i = me; i < a.z5_words; i += nthr) a.z5[i] = 0u;
}
__device__ __forceinline__ void build_clear_state(const BuildArgs &a) {  // ... of a records launch of REC_BLK-lane workgroups
    build_clear_state(a, (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * REC_BLK + threadIdx.x,
                      (size_t)gridDim.x * gridDim.y * gridDim.z * REC_BLK);
}

__global__ __launch_bounds__(REC_BLK) void tri_records_kernel(const BuildArgs a) {
    __shared__ float red[REC_BLK / 64][8];
    __shared__ float red2[REC_BLK / 64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B;
    const RecPlace pl = rec_place(a, a.clouds, (int)blockIdx.x);  // (uniform)
    const int cloud = pl.cloud, b = pl.b, bxr = pl.bxr;
    build_clear_state(a);  // per-call state and gradient accumulator
    const int n = cloud ? a.M : a.N;
    if (pl.lch >= 0) {  // uniform: the launch's LMAX_CHUNKS extra workgroups per sample reduce its lines
        if (a.lmax != nullptr)  // (one chunk each: they run beside the triangle workgroups)
            line_max_chunks(a.line, a.L, a.lmax, b, pl.lch, LMAX_CHUNKS, red2, input_of(b, a.Bt));
        return;
    }
    const int f = bxr * REC_BLK + tid;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, p2 = 0.0f;
    if (f < n) {
        float c[9], x;
        tri_record_row(a, cloud, b, n, f, f, c, x, p2);
        const int ngp = (n + GRP - 1) / GRP * GRP;
        ((cloud ? a.crec2 : a.crec1) + (size_t)b * ngp)[f] = make_float4(c[0], c[1], c[2], x);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = c[d]; mx[d] = c[d]; }
    }
    if (bxr * REC_BLK >= n) return;  // uniform: this workgroup has no triangle of the cloud
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
        a.apart[(((size_t)cloud * B + b) * a.nblk + bxr) * 8 + tid] = r;
    } else if (tid == 7) {
        a.apart[(((size_t)cloud * B + b) * a.nblk + bxr) * 8 + 7] = 0.0f;  // PTRI layout of this cloud: original order
    }
}

// ---------------------------------------------------------------------------------------
// Prepared clouds (round 4): the records launch of a cloud whose spatial ORDER is already known.
// A rigid motion preserves the spatial order of a cloud, and every caller of the reference moves the same source,
// step after step, against a target that never moves (code/test_demo_optimized_Lie_Algebra.py:57-62,
// rpm/Train_RPM.py:207-231; the reference itself suggests a tree, code/loss.py:260-262).  So the cell sort runs once
// per cloud (rrl_cloud_order) and this kernel replaces tri_records_kernel + tri_sort_kernel in every later step:
//   lane = one SORTED position s of the cloud; f = order[s]; the triangle's raw row is gathered, moved by the sample's
//   pose (source of the fused op), the moved row goes to its ORIGINAL-order row (TRI1: the later stages index it by
//   triangle), the 48-byte record (PTRI: it carries f), the NaN reach, the 16-byte (P0, thr2) record and f to position
//   s (PTRI, DEL, P0S, IDX -- coalesced stores; the scan resolves a candidate from its position alone, one dependent
//   load less than through IDX; slot 7 of the cloud's first APART row tells the scan which layout PTRI has), and the
//   wavefront -- which holds exactly one supergroup of 64 sorted records -- REFITS the supergroup's 13
//   sphere-tree nodes to the moved points with DPP reductions over 8 / 16 / 64 lanes (wave_tree): the same nodes,
//   bit for bit, that tri_sort_kernel derives from the same sorted records.
// Any permutation gives identical labels, hit lists and loss (the tree is a conservative filter; the reference's
// arithmetic decides in resolve_candidate): an order taken in another pose, or a stale one, only shapes the nodes.
// PMAX (max |P|^2 per cloud and sample) is not reduced here -- that would take a hand-over between workgroups of
// this launch --: the scan reduces the <= n / 256 partial rows (APART) in its prologue, next to the line maxima.
// ---------------------------------------------------------------------------------------
// The triangle part of the prepared records launch for one workgroup of blockDim.x = 256 or 512 lanes at place `pl`:
// sorted positions [bxr * blockDim.x, + blockDim.x) of (cloud, sample).  red: LDS [blockDim.x / 64][8].  (One body for
// tri_records_sorted_kernel and for the chained step's build + scan launch, cull_scan_build_kernel, whose leading workgroups
// run it with the scan's 512 lanes.)
// PUB: everything the culled scan reads of this cloud -- PTRI, DEL, P0S, the tree nodes, the partial rows -- leaves by
// write-through (sc1) stores: other workgroups of the same launch read it (cdna_hip_programming.md Guideline 16 R1).
// What a PUBLISHING records body leaves for its caller to store once the ticket is out -- nothing in the launch reads it: the
// moved row (TRI1, nine scattered 4-byte stores per lane: the per-line stage of the NEXT launch reads it) and the index
// entry (IDX: not read when the rows sit at their sorted positions).  Off the hand-off's critical chain.
struct RecLate {
    float c[9];
    int f, s;
    bool valid, in_range;
};
__device__ __forceinline__ void rec_late_stores(const BuildArgs &a, const RecPlace &pl, const RecLate &l) {
    const int n = pl.cloud ? a.M : a.N;
    const int npad = (n + SGT - 1) / SGT * SGT;
    if (l.in_range) (pl.cloud ? a.idx2 : a.idx1)[(size_t)pl.b * npad + l.s] = l.f;
    if (l.valid && pl.cloud == 0 && a.R != nullptr) {
        float *moved = a.tri1_out + ((size_t)pl.b * n + l.f) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) moved[i] = l.c[i];
    }
}
template <bool PUB = false>
__device__ __forceinline__ void records_sorted_body(const BuildArgs &a, const int32_t *__restrict__ order1,
                                                    const int32_t *__restrict__ order2, const RecPlace &pl, float (*red)[8],
                                                    RecLate *late = nullptr) {
    if (late) { late->valid = late->in_range = false; late->f = late->s = 0; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B;
    const int cloud = pl.cloud, b = pl.b, bxr = pl.bxr;
    const int n = cloud ? a.M : a.N;
    const int npad = (n + SGT - 1) / SGT * SGT;
    if (bxr * (int)blockDim.x >= npad) return;  // uniform: the smaller cloud has fewer workgroups
    const int s = bxr * (int)blockDim.x + tid;
    const bool valid = s < n;  // real records occupy the sorted positions [0, n)
    float c[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, x = 0.0f, p2 = 0.0f;
    int f = 0;
    if (valid) {
        f = (cloud ? order2 : order1)[(size_t)input_of(b, a.Bt) * npad + s];
        f = min(max(f, 0), n - 1);  // memory safety only: the order must be a permutation of [0, n)
        tri_record_row<PUB>(a, cloud, b, n, f, s, c, x, p2);
    }
    if (threadIdx.x < 64 && __float_as_int(c[0] + c[4] + c[8] + x) != 0x12345678) STAMPR(2);
    if (s - lane < npad) {  // wave-uniform: this wavefront holds a supergroup
        const float4 rec = valid ? make_float4(c[0], c[1], c[2], x) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (!PUB) (cloud ? a.idx2 : a.idx1)[(size_t)b * npad + s] = f;
        if (late) {
            late->in_range = true; late->valid = valid; late->f = f; late->s = s;
#pragma unroll
            for (int i = 0; i < 9; ++i) late->c[i] = c[i];
        }
        float4 *nodes = (cloud ? a.grp2 : a.grp1) + (size_t)b * (npad / SGT) * NODE;
        if constexpr (PUB) {
            st16_sc1(rrl_rsrc((cloud ? a.p0s2 : a.p0s1) + (size_t)b * npad, (size_t)npad * 16), (unsigned)s * 16u, rec);
            const __amdgpu_buffer_rsrc_t ns = rrl_rsrc(nodes, (size_t)(npad / SGT) * NODE * 16);
            const unsigned nb0 = (unsigned)((s - lane) / SGT) * (NODE * 16u);
            wave_tree_put(c[0], c[1], c[2], x, valid, lane, [&](int j, float4 v) { st16_sc1(ns, nb0 + 16u * (unsigned)j, v); });
        } else {
            (cloud ? a.p0s2 : a.p0s1)[(size_t)b * npad + s] = rec;
            wave_tree(c[0], c[1], c[2], x, valid, lane, nodes + (size_t)((s - lane) / SGT) * NODE);
        }
    }
    if (threadIdx.x < 64) STAMPR(3);
    // partial AABB of the P0s and max |P|^2 per REC_BLK = 256 sorted positions (APART rows), as tri_records_kernel leaves them
    float mn[3], mx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn[d] = wave_min(valid ? c[d] : INFINITY); mx[d] = wave_max(valid ? c[d] : -INFINITY); }
    p2 = wave_max(p2);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[wave][d] = mn[d]; red[wave][3 + d] = mx[d]; }
        red[wave][6] = p2;
    }
    __syncthreads();
    if (threadIdx.x < 64) STAMPR(4);
    const int q = tid & (REC_BLK - 1), w0 = (tid / REC_BLK) * (REC_BLK / 64);  // slot of the row; first wavefront of its 256 lanes
    const int row = s / REC_BLK;
    if (q < 8 && row * REC_BLK < npad) {
        float r = 1.0f;  // slot 7: PTRI layout of this cloud = sorted positions
        if (q < 7) {
            r = red[w0][q];
            for (int w = 1; w < REC_BLK / 64; ++w) r = q < 3 ? fminf(r, red[w0 + w][q]) : fmaxf(r, red[w0 + w][q]);
        }
        float *dst = &a.apart[(((size_t)cloud * B + b) * a.nblk + row) * 8 + q];
        if constexpr (PUB) st4_sc1(dst, r); else *dst = r;
    }
}

__global__ __launch_bounds__(REC_BLK) void tri_records_sorted_kernel(const BuildArgs a, const int32_t *__restrict__ order1,
                                                                      const int32_t *__restrict__ order2) {
    __shared__ float red[REC_BLK / 64][8];
    __shared__ float red2[REC_BLK / 64][2];
    const RecPlace pl = rec_place(a, a.clouds, (int)blockIdx.x);  // (uniform)
    if (threadIdx.x < 64) STAMPR(0);
    build_clear_state(a);
    if (threadIdx.x < 64) STAMPR(1);
    if (pl.lch >= 0) {  // uniform: the line maxima, beside the triangle workgroups (tri_records_kernel)
        if (a.lmax != nullptr)
            line_max_chunks(a.line, a.L, a.lmax, pl.b, pl.lch, LMAX_CHUNKS, red2, input_of(pl.b, a.Bt));
        if (threadIdx.x < 64) STAMPR(5);
        return;
    }
    records_sorted_body(a, order1, order2, pl, red);
}

// PMAX from the partial rows, for callers of the prepared build that do not run the culled scan next (rrl_tri_prepare_ex
// on its own; the scan does this reduction in its prologue): one wavefront per (cloud, sample).
__global__ __launch_bounds__(64) void pmax_from_partials_kernel(const float *__restrict__ apart, uint32_t *__restrict__ pmax,
                                                                int B, int N, int M, int nblk) {
    const int cb = blockIdx.x, cloud = cb >= B ? 1 : 0;
    const int nb = ((cloud ? M : N) + REC_BLK - 1) / REC_BLK;
    float v = 0.0f;
    for (int j = threadIdx.x; j < nb; j += 64) v = fmaxf(v, apart[((size_t)cb * nblk + j) * 8 + 6]);
    v = wave_max(v);
    if (threadIdx.x == 0) pmax[cb] = __float_as_uint(v);
}

// Every lane keeps its <= NPT records in registers between the passes (n <= 1024 NPT).
// RAW = true (the Chamfer path, clouds <= 4096 points): tri1 / tri2 are POINT clouds [B][n][3]; the
// kernel builds its (x, y, z, original index) records, the AABB and the NaN flag (slot 7 of the APART
// row, read by chamfer_tree_kernel) itself -- no records launch in front of it.  12 bytes per point
// through the single CU's memory pipe: the same traffic as the 16-byte CREC records.
template <int NPT, bool RAW>
__global__ __launch_bounds__(1024) void tri_sort_kernel(const BuildArgs a) {
    // the sorted records and their triangle indices are assembled in LDS ([ngp*17] float4, one
    // float4 of padding per group: lanes on different groups hit different banks; then [ngp*16]
    // int) and leave with coalesced stores; ngp = groups padded to whole supergroups
    extern __shared__ __attribute__((aligned(16))) float dyn_s[];
    float4 *srec = (float4 *)dyn_s;
    __shared__ unsigned hist[SORT_CELLS];
    __shared__ __attribute__((aligned(16))) unsigned short hlut[SORT_CELLS];
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B;
    // CHUNKED clouds (a.nchunk > 1, clouds of more than 4096 records): a workgroup sorts ONE chunk of 4096
    // consecutive records (by original index) on its own grid and builds its 64 supergroups; the sorted cloud
    // is the concatenation of its chunks (all but the last are full, so the real records still occupy the
    // sorted positions [0, n)).  Any grouping of the records gives the same labels (rrl_launch_tri_build).
    // PARTS (round 3): gridDim.y workgroups share one cloud (or chunk) WITHOUT talking to each other: every one
    // reads all records, builds the whole cell histogram and its scan (that is the cheap half of the kernel: the
    // loads and the LDS histogram), and then scatters, copies out and builds the tree nodes only for ITS range of
    // supergroups [S0, S1) = sorted positions [64 S0, 64 S1) -- the expensive half, now 1 / parts of it per CU.
    // The sorted position of a record must be the same in every workgroup that could own it: inside a cell the
    // order is arbitrary (an LDS cursor), except in the (at most two) cells that straddle this part's first /
    // last position, where the rank is the record's rank by original index (ballots + a 64-entry prefix).
    const int nch = RAW ? 1 : (a.nchunk > 1 ? a.nchunk : 1);
    const int part = (int)blockIdx.y, nparts = (int)gridDim.y;
    const int chunk = (int)blockIdx.x % nch, cb = (int)blockIdx.x / nch;
    const int cloud = cb >= B ? 1 : 0, b = cb - cloud * B;
    const int nfull = cloud ? a.M : a.N;                 // records of the whole cloud
    const int base0 = chunk * (1024 * NPT);              // first record of the chunk (multiple of 64)
    if (base0 >= nfull) return;                          // uniform: the smaller cloud has fewer chunks
    const int n = min(1024 * NPT, nfull - base0);        // records of this chunk
    const int ngf = (nfull + GRP - 1) / GRP, nsgf = (nfull + SGT - 1) / SGT;
    const int nsg = (n + SGT - 1) / SGT;
    const float4 *crec = (cloud ? a.crec2 : a.crec1) + (size_t)b * ngf * GRP + base0;
    float4 *p0s = (cloud ? a.p0s2 : a.p0s1) + (size_t)b * nsgf * SGT + base0;
    int32_t *idx = (cloud ? a.idx2 : a.idx1) + (size_t)b * nsgf * SGT + base0;
    float4 *tree = (cloud ? a.grp2 : a.grp1) + ((size_t)b * nsgf + base0 / SGT) * NODE;
    const int S0 = (int)((long)part * nsg / nparts), S1 = (int)((long)(part + 1) * nsg / nparts);  // this part's supergroups
    const int p0 = S0 * SGT, p1 = S1 * SGT;                                                        // ... and sorted positions
    const int ngpm = (nsg + nparts - 1) / nparts * SGG;  // groups of the largest part: the LDS layout of every part
    int *sidx = (int *)(srec + (size_t)ngpm * 17);
    auto pad = [](int s) { return s + (s >> 4); };
    __shared__ int s_cb[2];          // the cells that straddle p0 / p1 (-1: the boundary falls between two cells)
    __shared__ unsigned s_bw[2][64]; // their records per (pass k, wavefront): exclusive prefix in index order
    if (tid < 64) STAMPR(0);
    if (tid < 2) s_cb[tid] = -1;
    if (a.zwords != nullptr && blockIdx.x == 0 && blockIdx.y == 0)  // the Chamfer walk's arrival counters (next launch)
        for (int i = tid; i < a.nzwords; i += 1024) a.zwords[i] = 0u;

    // ---- AABB of the P0s and max |P|^2 from the per-workgroup partials of tri_records_kernel
    float4 rec[NPT];
    float bb[7];
    if constexpr (RAW) {
        __shared__ __attribute__((aligned(16))) float s_bb[16][8];
        const float *pts = (cloud ? a.tri2 : a.tri1) + (size_t)b * n * 3;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, p2 = 0.0f;
        bool bad = false;
        // the 12-byte rows are fetched as one flat, coalesced stream of 16-byte loads into the (still
        // unused) LDS region of the sorted records and picked up per point from there: 3-dword loads with
        // a stride of 12 bytes cost the single CU's memory pipe 16.1 us for the whole kernel against 13.8 us
        // for a records launch + this kernel reading 16-byte records
        float *flat = dyn_s;
        {
            const int nf = 3 * n, nv = nf >> 2;
            const bool al = (((uintptr_t)pts) & 15) == 0;  // uniform; sample offsets of 12 n bytes may break it
            if (al) {
                for (int i = tid; i < nv; i += 1024) ((float4 *)flat)[i] = ((const float4 *)pts)[i];
                for (int i = 4 * nv + tid; i < nf; i += 1024) flat[i] = pts[i];
            } else {
                for (int i = tid; i < nf; i += 1024) flat[i] = pts[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int f = tid + 1024 * k;
            if (f < n) {
                const float c0 = flat[3 * f], c1 = flat[3 * f + 1], c2 = flat[3 * f + 2];
                rec[k] = make_float4(c0, c1, c2, __int_as_float(f));
                mn[0] = fminf(mn[0], c0); mx[0] = fmaxf(mx[0], c0);
                mn[1] = fminf(mn[1], c1); mx[1] = fmaxf(mx[1], c1);
                mn[2] = fminf(mn[2], c2); mx[2] = fmaxf(mx[2], c2);
                float q = c0 * c0 + c1 * c1 + c2 * c2;
                if (!(q <= 3.0e38f)) q = INFINITY;
                p2 = fmaxf(p2, q);
                bad |= (c0 != c0) || (c1 != c1) || (c2 != c2);
            }
        }
        ((uint2 *)hlut)[tid] = ((const uint2 *)HILBERT_LUT.v)[tid];
        const float anybad = __any(bad) ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
        p2 = wave_max(p2);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_bb[wave][c] = mn[c]; s_bb[wave][3 + c] = mx[c]; }
            s_bb[wave][6] = p2;
            s_bb[wave][7] = anybad;
        }
        __syncthreads();
        // every row of 16 lanes reduces the 16 wavefront rows itself (two 16-byte LDS reads + DPP), like the
        // partial rows of the records kernel below
        float any7;
        {
            const float4 p0 = ((const float4 *)s_bb[lane & 15])[0], p1 = ((const float4 *)s_bb[lane & 15])[1];
            const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const float r = c < 3 ? row16_min(v[c]) : row16_max(v[c]);
                bb[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r)));
            }
            any7 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(row16_max(v[7]))));
        }
        // the NN kernel ORs slot 7 over the ceil(n / 256) partial rows of the cloud: row 0 carries the flag
        const int nb = (n + REC_BLK - 1) / REC_BLK;
        if (part == 0 && tid < nb) a.apart[(((size_t)cloud * B + b) * a.nblk + tid) * 8 + 7] = tid == 0 ? any7 : 0.0f;
    } else {
#pragma unroll
        for (int k = 0; k < NPT; ++k)  // issue the record loads first: they overlap the reduction
            if (tid + 1024 * k < n) rec[k] = crec[tid + 1024 * k];
        ((uint2 *)hlut)[tid] = ((const uint2 *)HILBERT_LUT.v)[tid];
        // every row of 16 lanes reduces the <= 16 per-workgroup partials itself (DPP): no LDS round
        // trip and no barrier for the AABB
        const int nb = (n + REC_BLK - 1) / REC_BLK;  // <= 16 for n <= 4096
        const float *ap = a.apart + (((size_t)cloud * B + b) * a.nblk + base0 / REC_BLK) * 8;
        // lanes 0..15 of each wavefront load one 32-byte partial row each (two 16-byte loads: the
        // kernel is bound by its single CU's memory pipe), reduce across the row, broadcast
        float4 p0 = make_float4(INFINITY, INFINITY, INFINITY, -INFINITY), p1 = make_float4(-INFINITY, -INFINITY, 0.0f, 0.0f);
        if (lane < 16 && lane < nb) { p0 = ((const float4 *)ap)[2 * lane]; p1 = ((const float4 *)ap)[2 * lane + 1]; }
        const float v[7] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z};
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const float r = c < 3 ? row16_min(v[c]) : row16_max(v[c]);
            bb[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r)));
        }
    }
    for (int i = tid; i < SORT_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    if (tid < 64) STAMPR(1);
#if defined(SORT_STOP) && SORT_STOP == 1  // timing experiments only
    return;
#endif
    if (tid == 0 && part == 0) {
        if (nch > 1) atomicMax(&a.pmax[cloud * B + b], __float_as_uint(bb[6]));  // non-negative floats; PMAX is cleared per call
        else a.pmax[cloud * B + b] = __float_as_uint(bb[6]);
    }
    float mn[3], scale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = bb[c];
        float ext = bb[3 + c] - mn[c];
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
        return hlut[q[0] | (q[1] << 4) | (q[2] << 8)];  // 12 bits, == hilbert_cell(q0, q1, q2)
    };

    // ---- histogram over cells, exclusive scan, scatter
    unsigned cell[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k)
        if (tid + 1024 * k < n) { cell[k] = cell_of(rec[k]); atomicAdd(&hist[cell[k]], 1u); }
    __syncthreads();
    if (tid < 64) STAMPR(2);
#if defined(SORT_STOP) && SORT_STOP == 2  // timing experiments only
    return;
#endif
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
        for (int k = 0; k < 4; ++k) {
            hist[4 * tid + k] = run;
            // a cell with records on both sides of this part's first / last position
            if ((int)run < p0 && p0 < (int)(run + h[k])) s_cb[0] = 4 * tid + k;
            if ((int)run < p1 && p1 < (int)(run + h[k])) s_cb[1] = 4 * tid + k;
            run += h[k];
        }
    }
    __syncthreads();
    if (tid < 64) STAMPR(3);
#if defined(SORT_STOP) && SORT_STOP == 3  // timing experiments only
    return;
#endif
    // ---- ranks by original index (f = tid + 1024 k: pass-major, then wavefront, then lane) inside the straddling cells
    const int cb0 = nparts > 1 ? s_cb[0] : -1, cb1 = nparts > 1 ? s_cb[1] : -1;
    unsigned long long bm[2][NPT];
    unsigned bbase[2] = {0u, 0u};
    if (cb0 >= 0 || cb1 >= 0) {  // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = j ? cb1 : cb0;
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                bm[j][k] = __ballot(c >= 0 && tid + 1024 * k < n && (int)cell[k] == c);
                if (lane == 0 && 16 * k + wave < 64) s_bw[j][16 * k + wave] = (unsigned)__popcll(bm[j][k]);
            }
            if (c >= 0) bbase[j] = hist[c];  // no cursor runs in these cells: the value stays the cell's first position
        }
        __syncthreads();
        if (tid < 64) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned v = tid < 16 * NPT ? s_bw[j][tid] : 0u;
                s_bw[j][tid] = (unsigned)wave_incl_scan((int)v) - v;
            }
        }
        __syncthreads();
    }
    static_assert(NPT <= 4, "s_bw holds 16 wavefronts x NPT passes");
    // scatter into LDS only (this part's range); the global arrays are written afterwards, in order
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int f = tid + 1024 * k;
        if (f < n) {
            const int c = (int)cell[k];
            int s;
            if (c == cb0) s = (int)(bbase[0] + s_bw[0][16 * k + wave]) + __popcll(bm[0][k] & ((1ull << lane) - 1ull));
            else if (c == cb1) s = (int)(bbase[1] + s_bw[1][16 * k + wave]) + __popcll(bm[1][k] & ((1ull << lane) - 1ull));
            else s = (int)atomicAdd(&hist[c], 1u);
            if (s >= p0 && s < p1) {
                srec[pad(s - p0)] = rec[k];
                sidx[s - p0] = base0 + f;
            }
        }
    }
    for (int s = max(n, p0) + tid; s < p1; s += 1024) {  // pad (the last part): thr2 = 0 never passes
        srec[pad(s - p0)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        sidx[s - p0] = 0;
    }
    __syncthreads();
    if (tid < 64) STAMPR(4);
#if defined(SORT_STOP) && SORT_STOP == 4  // timing experiments only
    return;
#endif
    for (int s = p0 + tid; s < p1; s += 1024) {  // coalesced copy-out
        p0s[s] = srec[pad(s - p0)];
        idx[s] = sidx[s - p0];
    }
    if (tid < 64) STAMPR(5);
#if defined(SORT_STOP) && SORT_STOP == 5
    return;
#endif
    // ---- sphere tree: one lane per half of 8 records
    for (int hh = 2 * SGG * S0 + tid; hh < 2 * SGG * S1; hh += 1024)
        half_tree([&](int s_) { return srec[pad(s_ - p0)]; }, hh, n, tree);
    if (tid < 64) STAMPR(6);
}

// ---------------------------------------------------------------------------------------
// Large clouds (n > 4096), WHOLE-CLOUD order: the same sort in three WIDE launches -- one workgroup per
// cloud is bound by a single CU (48 us for 16384 triangles), and a launch boundary costs ~3 us.  The loss
// build uses the chunked single-launch sort instead (rrl_launch_tri_build); these kernels serve the Chamfer
// path (nearest neighbours want whole-cloud groups) and RRL_SORT_WIDE=1:
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
    const int npad = (n + SGT - 1) / SGT * SGT;
    float4 *p0s = (cloud ? a.p0s2 : a.p0s1) + (size_t)b * npad;
    int32_t *idx = (cloud ? a.idx2 : a.idx1) + (size_t)b * npad;
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
        for (int s = n + tid; s < npad; s += 256) { p0s[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); idx[s] = 0; }
}

__global__ __launch_bounds__(256) void big_sphere_kernel(const BuildArgs a) {
    const int cloud = blockIdx.z, b = blockIdx.y;
    const int n = cloud ? a.M : a.N;
    const int nsg = (n + SGT - 1) / SGT;
    const int hh = blockIdx.x * 256 + threadIdx.x;  // 256 = 32 whole supergroups
    if (hh >= nsg * 2 * SGG) return;
    const float4 *r = (cloud ? a.p0s2 : a.p0s1) + (size_t)b * nsg * SGT;
    half_tree([&](int s_) { return r[s_]; }, hh, n, (cloud ? a.grp2 : a.grp1) + (size_t)b * nsg * NODE);
}

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load
typedef const int __attribute__((address_space(4))) * kiptr;

static BuildArgs make_build_args(const float *tri1, const float *tri2, void *ws, const WsLayout &w, int B, int N, int M,
                                 int clouds, const RrlXform *xf, const float *line, int L, const RrlCall &o, bool chunked);
// polls (~1 us each) before a source-cloud workgroup of the chained step's launch gives up waiting for its records (RRL_CHAIN_SPIN)
static unsigned rrl_chain_spin_limit(void) {
    static long v = -1;
    if (v < 0) { const char *e = getenv("RRL_CHAIN_SPIN"); v = e ? atol(e) : (1l << 18); if (v < 0 || v > 0x7fffffffl) v = 1l << 18; }
    return (unsigned)v;
}

// The two geometry variants of the culled scan (rrl_cull_scan.inc): same source, two sets of compile-time knobs.
namespace scan8 {
#include "rrl_cull_scan.inc"
}
#undef SPW
#undef SG_BITS
#undef HF_BITS
#undef WPB
#undef LPW
#undef ROWS
#undef WCCAP
#undef QA_CAP
#undef QC_CAP
#undef CULL_REGLINES
// (experiments: the fat variant's knobs have names of their own -- -DSCAN16_QA_CAP=... -- the plain names configure scan8)
#ifndef SCAN16_QA_CAP
#define SCAN16_QA_CAP 384
#endif
#ifndef SCAN16_QC_CAP
#define SCAN16_QC_CAP 256
#endif
#ifndef SCAN16_WCCAP
#define SCAN16_WCCAP 128
#endif
namespace scan16 {
#define SPW 16
#define CULL_REGLINES 1
#define QA_CAP SCAN16_QA_CAP
#define QC_CAP SCAN16_QC_CAP
#define WCCAP SCAN16_WCCAP
#include "rrl_cull_scan.inc"
}
static_assert(scan8::kWPB == scan16::kWPB && scan8::kLPW == scan16::kLPW, "one line tiling for both variants");

// The grid of a culled scan: wavefronts per workgroup, supergroups per slice, the variant, line tiles, slices.
struct CullGeom {
    int waves, spw, tiles, slices;
    bool fat, may_ride;
};
static CullGeom cull_geometry(int B, int N, int M, int L, int clouds, const RrlCall &o) {
    // a workgroup = (cloud and sample, tile of <= WPB x 128 lines, slice of spw supergroups).  With
    // few lines or small clouds the slices get thinner, so that the launch still has ~1000
    // workgroups for the 256 CUs (measured with tools/attic/geom_sweep.sh: thinner slices cost little,
    // fewer wavefronts per workgroup cost more -- they are only reduced as a last resort)
    constexpr int WPB_ = scan8::kWPB, LPW_ = scan8::kLPW;
    const int nmax = clouds == 2 && M > N ? M : N;
    const int nsgmax = (nmax + SGT - 1) / SGT;
    const int lw = (L + LPW_ - 1) / LPW_;  // wavefronts' worth of lines
    int waves = lw < WPB_ ? lw : WPB_, spw = scan8::kSPW;
    auto wgs = [&]() { return (long)clouds * B * ((lw + waves - 1) / waves) * ((nsgmax + spw - 1) / spw); };
    // FAT slices (scan16: 16 supergroups per workgroup, lines in registers, 4 workgroups per CU) as soon as the grid stays
    // deep with them -- measured (tools/attic/spw_exp2.sh, profiles/r05_experiments.txt): >= 960 fat workgroups win 6 .. 20 %
    // (B = 12 .. 64 at C2's shape, N = 8192, L = 20000), <= 640 lose 7 .. 55 % (C2 itself, C4, the demo).  RRL_CULL_FAT=0 / 1 forces.
    bool fat = (long)clouds * B * ((lw + waves - 1) / waves) * ((nsgmax + 15) / 16) >= 896 && lw >= WPB_;
    if (const char *e = getenv("RRL_CULL_FAT")) fat = e[0] == '1' ? (nsgmax > 8) : (e[0] == '0' ? false : fat);
    if (fat) spw = scan16::kSPW;
    while (!fat && wgs() < 768 && spw > 1) spw >>= 1;
    // (a riding Chamfer walk needs the scan's full 512-lane workgroups -- and brings workgroups of its own: no thinning then)
    const bool may_ride = o.rider && !o.counters && (clouds == 2 || o.tar_ws) && lw >= WPB_ && B <= 32767 && N > 0 && M > 0;
    while (!fat && !may_ride && wgs() < 256 && waves > 2) waves >>= 1;
#ifdef RRL_EXPERIMENT  // (experimental builds only, RRL_HIPCC_FLAGS=-DRRL_EXPERIMENT -> lib_exp: geometry sweeps)
    if (const char *e = getenv("RRL_CULL_GEOM")) {  // "waves,spw" (spw > 8: the fat variant)
        int w_ = 0, s_ = 0;
        if (sscanf(e, "%d,%d", &w_, &s_) == 2 && w_ >= 1 && w_ <= WPB_ && s_ >= 1 && s_ <= scan16::kSPW) {
            waves = w_ < lw ? w_ : lw; spw = s_; fat = s_ > scan8::kSPW;
        }
    }
#endif
    CullGeom g;
    g.waves = waves; g.spw = spw; g.fat = fat; g.may_ride = may_ride;
    g.tiles = (lw + waves - 1) / waves; g.slices = (nsgmax + spw - 1) / spw;
    return g;
}

// Can the chained step's ONE launch (source records + target scan + source scan: cull_scan_build_kernel) serve this call?
// Both clouds scanned here with full 512-lane workgroups, no rider, no counters; the caller (loss_forward_impl) checks the
// rest (prepared orders, kept target, the per-line stage + tail kernel behind it).
int rrl_cull_scan_can_fuse(int B, int N, int M, int L, const RrlCall &o) {
    if (B <= 0 || N <= 0 || M <= 0 || L <= 0 || (N > M ? N : M) > SORT_CAP) return 0;
    if (o.rider || o.counters || o.problems || o.tar_ws) return 0;
    if (const char *e = getenv("RRL_CHAIN")) if (e[0] == '0') return 0;  // (A/B runs)
    const CullGeom g = cull_geometry(B, N, M, L, 2, o);
    if (g.waves != scan8::kWPB) return 0;
    // Where the ONE launch pays (measured, profiles/r06_experiments.txt 2; us per step chained / plain): C2 45.8 / 49.8, B = 4
    // 40.9 / 42.9, B = 12 .. 32 at C2's shape 59.0 / 62.7 .. 110.8 / 116.5, the demo's shape 36.1 / 38.1, C4's 46.2 / 49.4 --
    // and where it does not: B = 64 (5632 workgroups, ten generations deep: the 512 records workgroups are a small share and
    // the launch they save was overlapped with nothing anyway; the tile-local maxima's barrier and the write-through records
    // cost as much) 197.6 / 195.6, and 2 x 16384 triangles with 4096 lines (576 workgroups, ALL resident at once: the source
    // workgroups wait for the 32 records pieces of their sample, which take longer inside the crowd than as a launch of
    // their own; the launch cannot end before they are published + one source workgroup's lifetime) 38.5 / 35.6.  No source
    // workgroup starts before its sample's records are out (8 .. 10 us into a crowded launch), so the launch pays where
    // that wait hides behind target-cloud work: clouds up to 4096 triangles (<= 8 records pieces per sample) on any grid
    // measured, larger clouds only on grids several generations deep.
    const int spw = g.spw, nsg1 = (N + SGT - 1) / SGT, nsg2 = (M + SGT - 1) / SGT;
    const int nrec_b = (int)(((size_t)nsg1 * SGT + 64 * scan8::kWPB - 1) / (64 * scan8::kWPB));
    const long total = (long)B * (nrec_b + (long)g.tiles * ((nsg1 + spw - 1) / spw + (nsg2 + spw - 1) / spw));
    if (g.fat && total > 4000) return 0;
    if (nrec_b > 8 && total < 2048) return 0;
    return 1;
}

int rrl_launch_cull_scan(const float *line, void *ws, const WsLayout &w, int B, int N, int M, int L,
                         int clouds, int lmax_ready, const RrlCall &o, hipStream_t s) {
    const CullGeom g = cull_geometry(B, N, M, L, clouds, o);
    if (o.fused_build && (clouds != 2 || g.waves != scan8::kWPB)) return RRL_E_ARG;  // (rrl_cull_scan_can_fuse said otherwise)
    if (!lmax_ready && !o.fused_build)  // the triangles were prepared without the lines: their partial maxima first (a tiny launch)
        hipLaunchKernelGGL(line_max_kernel, dim3(LMAX_CHUNKS, (unsigned)B), dim3(REC_BLK), 0, s, line, L,
                           (float2 *)w.f32(ws, RRL_WS_LMAX), o.problems);
    // (A PERSISTENT variant -- as many workgroups as fit on the chip, each keeping one line tile staged and pulling
    // (cloud, slice) items from per-tile work queues, the next slice's records prefetched during the walk -- was built
    // and measured in round 3: exact, but 40.7 us against 30.4 at C2 and 29.0 against 13.8 at the demo's shape.  A slot
    // is held for the SLOWEST of a workgroup's eight wavefronts either way (16.5 us per item against a mean wavefront
    // lifetime of 12.4), so queueing the items removed no waiting, and the item barriers added some;
    // profiles/r03_scan_experiments.txt.)
    return g.fat ? scan16::launch_variant(line, ws, w, B, N, M, L, clouds, o, s, g.waves, g.spw, g.tiles, g.slices, g.may_ride)
                 : scan8::launch_variant(line, ws, w, B, N, M, L, clouds, o, s, g.waves, g.spw, g.tiles, g.slices, g.may_ride);
}
// Executed-work counters (profiling; include/rrl.h rrl_scan_counters): while a buffer is set,
// culled scans launch the COUNT instantiation and add to it.
static unsigned long long *g_cull_counters = nullptr;
static long long g_cull_counter_rows = 0;
extern "C" int rrl_scan_counters(uint64_t *dev_counters, long long rows) {
    g_cull_counters = (unsigned long long *)dev_counters;
    g_cull_counter_rows = dev_counters ? rows : 0;
    return 0;
}

// Workgroups per cloud (or chunk) of tri_sort_kernel.  Measured (round 3, profiles/r03_reduce_sort_parts.txt): 1 / 2 / 4
// parts cost 10.4 / 10.3 / 10.2 us at C2 and 8.4 / 8.0 / 9.0-9.6 at the demo's shape -- the phases that split (scatter,
// copy-out, tree) are one pass per lane either way, i.e. latency, not throughput, and the repeated loads cost what
// the split saves.  So the default stays ONE workgroup; rrl_set_sort_parts / RRL_SORT_PARTS select more (tests, sweeps).
static int g_sort_parts = -1;  // 0 automatic, k >= 1 forced; -1: read RRL_SORT_PARTS once
extern "C" int rrl_set_sort_parts(int parts) {
    if (parts < 0 || parts > 16) return RRL_E_ARG;
    g_sort_parts = parts;
    return 0;
}
int rrl_default_sort_parts(void) {
    if (g_sort_parts < 0) {
        const char *e = getenv("RRL_SORT_PARTS");
        int v = e ? atoi(e) : 0;
        if (v < 0 || v > 16) v = 0;
        g_sort_parts = v;
    }
    return g_sort_parts;
}
void rrl_default_scan_counters(unsigned long long **buf, long long *rows) {
    *buf = g_cull_counters;
    *rows = g_cull_counter_rows;
}
static int sort_parts(int nsg, int requested) {  // requested: RrlCall::sort_parts (0 = automatic = one workgroup)
    int k = requested ? requested : 1;
    if (k > nsg) k = nsg;
    return k < 1 ? 1 : k;
}
// dynamic LDS of one sort workgroup: the padded records + indices of the largest part; the RAW variant first stages
// its 3 n floats there
static size_t sort_lds_bytes(int nsg, int parts, int raw_points) {
    const size_t ngpm = (size_t)((nsg + parts - 1) / parts) * SGG;
    size_t lds = ngpm * (17 * sizeof(float4) + GRP * sizeof(int));
    const size_t rawb = (size_t)raw_points * 3 * sizeof(float);
    return lds > rawb ? lds : rawb;
}

// Launchers used by rrl_tri_prepare / rrl_line_tri_scan (rrl_scan.hip)
// The arguments of a records launch (rrl_launch_tri_build; the chained step's fused launch, launch_variant).
static BuildArgs make_build_args(const float *tri1, const float *tri2, void *ws, const WsLayout &w, int B, int N, int M,
                                 int clouds, const RrlXform *xf, const float *line, int L, const RrlCall &o, bool chunked) {
    const int nmax = clouds == 2 && M > N ? M : N;
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
    a.z2 = nmax > 4096 && !chunked ? (uint4 *)((char *)ws + w.off[RRL_WS_HISTG]) : nullptr;
    a.z2_vec4 = a.z2 ? (size_t)2 * B * 2 * SORT_CELLS * sizeof(unsigned) / 16 : 0;
    a.z3 = (uint4 *)((char *)ws + w.state_off);
    a.z3_vec4 = w.state_bytes / 16;
    a.z4 = (uint32_t *)o.clear_ptr;
    a.z4_words = o.clear_ptr ? o.clear_bytes / 4 : 0;
    a.z5 = w.u32(ws, RRL_WS_CHAIN);
    a.z5_words = (size_t)4 * B;
    a.del1 = w.f32(ws, RRL_WS_DEL1);
    a.del2 = w.f32(ws, RRL_WS_DEL2);
    a.zwords = nullptr; a.nzwords = 0;
    a.line = line;
    a.lmax = line && L > 0 ? (float2 *)w.f32(ws, RRL_WS_LMAX) : nullptr;
    a.L = L;
    a.B = B; a.N = N; a.M = M; a.clouds = clouds;
    a.transpose_r = xf ? xf->transpose_r : 0;
    a.Bt = o.problems;
    a.xcd_align = 0;
    const int nall = N > M ? N : M;  // APART is laid out for the larger cloud
    a.nblk = (nall + REC_BLK - 1) / REC_BLK;
    a.nchunk = chunked ? (nmax + 4095) / 4096 : 1;
    a.nblk_tri = (nmax + REC_BLK - 1) / REC_BLK;
    {   // producer and consumer of a cloud's records on the same XCD (round 5: -0.7 us on the records launch, -0.5 us on the
        // scan at C2; RRL_XCD_ALIGN=0 turns it off)
        static int xa = -1;
        if (xa < 0) {
            xa = 1;
#ifdef RRL_EXPERIMENT  // (A/B runs of the placement, experimental builds only)
            if (const char *e = getenv("RRL_XCD_ALIGN")) xa = e[0] == '0' ? 0 : 1;
#endif
        }
        a.xcd_align = xa && (clouds * B) % 8 == 0 ? 1 : 0;
    }
    return a;
}

int rrl_launch_tri_build(const float *tri1, const float *tri2, void *ws, const WsLayout &w, int B,
                         int N, int M, int clouds, const RrlXform *xf, const float *line, int L, const RrlCall &o,
                         hipStream_t s) {
    const int nmax = clouds == 2 && M > N ? M : N;
    const size_t ngpmax = (size_t)(nmax + SGT - 1) / SGT * SGG;  // groups, padded to whole supergroups
    // Clouds of more than 4096 triangles: ONE launch of the single-workgroup sort per chunk of 4096 records
    // (by original index) instead of the wide three-launch sort of the whole cloud (hist, scatter, spheres:
    // ~19 us at N = 16384 against 9).  The chunks are interleaved subsets of the surface, each sorted on its
    // own 16^3 grid; measured faster at every shape tried (C5 54.4 -> 44.1 us, C5 at B = 8 102 -> 62.5,
    // B = 8 / N = 16384 / L = 10000 171 -> 140, N = 65536 / L = 512 77.5 -> 41.8: the whole-cloud grid holds
    // ~27 triangles per cell at N = 16384, in arbitrary order, so its groups are no tighter).
    // RRL_SORT_WIDE=1 keeps the wide sort (experiments / tests; it still serves the Chamfer path).
    const char *wide_env = getenv("RRL_SORT_WIDE");
    const bool chunked = nmax > 4096 && !(wide_env && atoi(wide_env) != 0);
    const size_t ngps = nmax <= 4096 ? ngpmax : (size_t)(4096 / GRP);
    const int parts = sort_parts((int)(ngps / SGG), o.sort_parts);
    const size_t lds = nmax <= 4096 || chunked ? sort_lds_bytes((int)(ngps / SGG), parts, 0) : 16;
    BuildArgs a = make_build_args(tri1, tri2, ws, w, B, N, M, clouds, xf, line, L, o, chunked);
    if (o.prepared()) {  // the order is known: ONE launch (records at their sorted positions + tree refit), no sort
        a.z2 = nullptr; a.z2_vec4 = 0;
        a.nblk_tri = (int)(((size_t)(nmax + SGT - 1) / SGT * SGT + REC_BLK - 1) / REC_BLK);
        hipLaunchKernelGGL(tri_records_sorted_kernel, dim3((unsigned)(a.nblk_tri * B * clouds + (a.lmax ? LMAX_CHUNKS * B : 0))),
                           dim3(REC_BLK), 0, s, a, o.order1, o.order2);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    }
    hipLaunchKernelGGL(tri_records_kernel, dim3((unsigned)(a.nblk_tri * B * clouds + (a.lmax ? LMAX_CHUNKS * B : 0))),
                       dim3(REC_BLK), 0, s, a);
    if (nmax <= 4096 || chunked) {
        hipLaunchKernelGGL((tri_sort_kernel<4, false>), dim3((unsigned)(clouds * B * a.nchunk), (unsigned)parts), dim3(1024), lds, s, a);
    } else {  // wide three-launch sort (HISTG was cleared by tri_records_kernel)
        unsigned *histg = (unsigned *)w.i32(ws, RRL_WS_HISTG);
        const dim3 gt((unsigned)((nmax + 255) / 256), (unsigned)B, (unsigned)clouds);
        hipLaunchKernelGGL(big_hist_kernel, gt, dim3(256), 0, s, a, histg);
        hipLaunchKernelGGL(big_scatter_kernel, gt, dim3(256), 0, s, a, histg);
        const dim3 gs((unsigned)((2 * ngpmax + 255) / 256), (unsigned)B, (unsigned)clouds);
        hipLaunchKernelGGL(big_sphere_kernel, gs, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// PMAX of a prepared build for consumers other than the culled scan (include/rrl.h rrl_tri_prepare_ex)
int rrl_launch_pmax_from_partials(void *ws, const WsLayout &w, int B, int N, int M, int clouds, hipStream_t s) {
    const int nall = N > M ? N : M;
    hipLaunchKernelGGL(pmax_from_partials_kernel, dim3((unsigned)(clouds * B)), dim3(64), 0, s, w.f32(ws, RRL_WS_APART),
                       (uint32_t *)w.i32(ws, RRL_WS_PMAX), B, N, M, (nall + REC_BLK - 1) / REC_BLK);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Sort + sphere tree for callers outside the loss workspace (the Chamfer path, rrl_chamfer.hip): the same
// kernels as above.  Clouds of <= 4096 points given as raw1 / raw2 ([B][n][3]) are sorted straight from
// the points (tri_sort_kernel<4, true>: records, AABB and NaN flag built in the kernel); otherwise the
// (x, y, z, w) records (CREC layout) and per-256-record AABB partials (APART layout, nblk rows per cloud
// and sample) must already exist and histg (cleared by the caller) is used beyond 4096 records.
int rrl_launch_cloud_sort(const float *raw1, const float *raw2, float4 *crec1, float4 *crec2, float *apart, int nblk,
                          float4 *p0s1, float4 *p0s2, int32_t *idx1, int32_t *idx2, float4 *grp1, float4 *grp2,
                          uint32_t *pmax, unsigned *histg, uint32_t *zwords, int nzwords, int B, int N, int M, hipStream_t s) {
    const int nmax = M > N ? M : N;
    if (nmax > SORT_CAP || B <= 0 || nmax <= 0) return RRL_E_ARG;
    const size_t ngpmax = (size_t)(nmax + SGT - 1) / SGT * SGG;
    BuildArgs a = {};
    a.tri1 = raw1; a.tri2 = raw2;  // point clouds [B][n][3]: only read by the RAW sort (nmax <= 4096)
    a.crec1 = crec1; a.crec2 = crec2;
    a.apart = apart; a.nblk = nblk;
    a.p0s1 = p0s1; a.p0s2 = p0s2;
    a.idx1 = idx1; a.idx2 = idx2;
    a.grp1 = grp1; a.grp2 = grp2;
    a.pmax = pmax;
    a.zwords = zwords; a.nzwords = nzwords;
    a.B = B; a.N = N; a.M = M; a.Bt = 0;
    // (the chunked sort of rrl_launch_tri_build was tried here too: a nearest-neighbour walk evaluates twice the
    //  pairs on chunked clouds -- 63.0 -> 64.4 us at N = M = 16384, 188 -> 380 at 65536: whole-cloud order stays)
    if (nmax <= 4096) {
        const int parts = sort_parts((int)(ngpmax / SGG), rrl_default_sort_parts());
        if (raw1 && raw2) hipLaunchKernelGGL((tri_sort_kernel<4, true>), dim3((unsigned)(2 * B), (unsigned)parts), dim3(1024),
                                             sort_lds_bytes((int)(ngpmax / SGG), parts, nmax), s, a);
        else hipLaunchKernelGGL((tri_sort_kernel<4, false>), dim3((unsigned)(2 * B), (unsigned)parts), dim3(1024),
                                sort_lds_bytes((int)(ngpmax / SGG), parts, 0), s, a);
    } else {
        const dim3 gt((unsigned)((nmax + 255) / 256), (unsigned)B, 2u);
        hipLaunchKernelGGL(big_hist_kernel, gt, dim3(256), 0, s, a, histg);
        hipLaunchKernelGGL(big_scatter_kernel, gt, dim3(256), 0, s, a, histg);
        const dim3 gs((unsigned)((2 * ngpmax + 255) / 256), (unsigned)B, 2u);
        hipLaunchKernelGGL(big_sphere_kernel, gs, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_sort_capacity(void) { return SORT_CAP; }
