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
    float *del1, *del2;            // NaN reach of every triangle (may be NULL: not stored)
    const float *line;             // the samples' lines [B][L][6] and where their partial maxima go (may be NULL:
    float2 *lmax;                  //   the scan entry computes them itself then)
    int L;
    int nblk_tri;                  // tri_records_kernel: workgroups [0, nblk_tri) of a (sample, cloud) hold triangles
    uint32_t *zwords;              // tri_sort_kernel (Chamfer path): words its first workgroup clears (may be NULL)
    int nzwords;
    int B, N, M, transpose_r, nblk;
    int nchunk;                    // tri_sort_kernel: chunks of 4096 records per cloud (1: the whole cloud)
    int Bt;                        // multi-pose evaluation (rrl_opts.problems): the INPUT clouds, orders and lines have Bt
                                   // entries and instance b uses entry b % Bt; 0 / B: every instance has its own
};
// the input entry of instance b (multi-pose: rrl_opts.problems)
__device__ __forceinline__ int input_of(int b, int Bt) { return (Bt > 0 && b >= Bt) ? b % Bt : b; }

#define REC_BLK 256
#define LMAX_CHUNKS 64  // per-sample partial maxima of the lines' |dir|^2 and |x0|^2 (-> the culled scan's slack)

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
        float *moved = a.tri1_out + ((size_t)b * n + f) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) moved[i] = c[i];
    }
    tri_thresholds(c, &thr, &x, &e01);  // code/loss.py:94-110
    // NaN reach (culled scan, "NaN detection" in the header): points 1, 2 lie within e01 of point 0, and the
    // tree nodes carry thr: del >= e01 - thr in exact arithmetic (e01, thr as rounded here: <= 3u off)
    if (float *del = cloud ? a.del2 : a.del1)  // (at the record's row, like PTRI)
        del[(size_t)b * n + prow] = fmaxf(e01 * 1.000002f - thr, 0.0f) * 1.000001f;
    float4 *row = (float4 *)((cloud ? a.ptri2 : a.ptri1) + ((size_t)b * n + prow) * PTRI_STRIDE);
    row[0] = make_float4(c[0], c[1], c[2], c[3]);
    row[1] = make_float4(c[4], c[5], c[6], c[7]);
    row[2] = make_float4(c[8], x, thr, __int_as_float(f));
    p2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        p2 = fmaxf(p2, c[3 * q] * c[3 * q] + c[3 * q + 1] * c[3 * q + 1] + c[3 * q + 2] * c[3 * q + 2]);
    if (!(p2 <= 3.0e38f)) p2 = INFINITY;  // NaN/inf coordinates: never "provably safe"
}

// the clearing of the per-call state (and of the small accumulators), spread over all workgroups of a records launch
__device__ __forceinline__ void build_clear_state(const BuildArgs &a) {
    const size_t nthr = (size_t)gridDim.x * gridDim.y * gridDim.z * REC_BLK;
    const size_t me = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * REC_BLK + threadIdx.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = me; i < a.zero_vec4; i += nthr) a.zero_base[i] = z;
    for (size_t i = me; i < a.g1_vec4; i += nthr) a.g1[i] = z;
    for (size_t i = me; i < a.z2_vec4; i += nthr) a.z2[i] = z;
    for (size_t i = me; i < a.z3_vec4; i += nthr) a.z3[i] = z;
    for (size_t i = me; i < a.z4_words; i += nthr) a.z4[i] = 0u;  // (4-byte stores: any alignment)
}

__global__ __launch_bounds__(REC_BLK) void tri_records_kernel(const BuildArgs a) {
    __shared__ float red[REC_BLK / 64][8];
    __shared__ float red2[REC_BLK / 64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.z, b = blockIdx.y, B = a.B;
    build_clear_state(a);  // per-call state and gradient accumulator
    const int n = cloud ? a.M : a.N;
    if ((int)blockIdx.x >= a.nblk_tri) {  // uniform: the launch's LMAX_CHUNKS extra workgroups per sample reduce its lines
        if (cloud == 0 && a.lmax != nullptr)  // (one chunk each: they run beside the triangle workgroups)
            line_max_chunks(a.line, a.L, a.lmax, b, (int)blockIdx.x - a.nblk_tri, LMAX_CHUNKS, red2, input_of(b, a.Bt));
        return;
    }
    const int f = blockIdx.x * REC_BLK + tid;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, p2 = 0.0f;
    if (f < n) {
        float c[9], x;
        tri_record_row(a, cloud, b, n, f, f, c, x, p2);
        const int ngp = (n + GRP - 1) / GRP * GRP;
        ((cloud ? a.crec2 : a.crec1) + (size_t)b * ngp)[f] = make_float4(c[0], c[1], c[2], x);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = c[d]; mx[d] = c[d]; }
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
    } else if (tid == 7) {
        a.apart[(((size_t)cloud * B + b) * a.nblk + blockIdx.x) * 8 + 7] = 0.0f;  // PTRI layout of this cloud: original order
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
__global__ __launch_bounds__(REC_BLK) void tri_records_sorted_kernel(const BuildArgs a, const int32_t *__restrict__ order1,
                                                                      const int32_t *__restrict__ order2) {
    __shared__ float red[REC_BLK / 64][8];
    __shared__ float red2[REC_BLK / 64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.z, b = blockIdx.y, B = a.B;
    build_clear_state(a);
    const int n = cloud ? a.M : a.N;
    if ((int)blockIdx.x >= a.nblk_tri) {  // uniform: the line maxima, beside the triangle workgroups (tri_records_kernel)
        if (cloud == 0 && a.lmax != nullptr)
            line_max_chunks(a.line, a.L, a.lmax, b, (int)blockIdx.x - a.nblk_tri, LMAX_CHUNKS, red2, input_of(b, a.Bt));
        return;
    }
    const int npad = (n + SGT - 1) / SGT * SGT;
    if ((int)blockIdx.x * REC_BLK >= npad) return;  // uniform: the smaller cloud has fewer workgroups
    const int s = blockIdx.x * REC_BLK + tid;
    const bool valid = s < n;  // real records occupy the sorted positions [0, n)
    float c[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, x = 0.0f, p2 = 0.0f;
    int f = 0;
    if (valid) {
        f = (cloud ? order2 : order1)[(size_t)input_of(b, a.Bt) * npad + s];
        f = min(max(f, 0), n - 1);  // memory safety only: the order must be a permutation of [0, n)
        tri_record_row(a, cloud, b, n, f, s, c, x, p2);
    }
    if (s - lane < npad) {  // wave-uniform: this wavefront holds a supergroup
        (cloud ? a.p0s2 : a.p0s1)[(size_t)b * npad + s] = valid ? make_float4(c[0], c[1], c[2], x) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        (cloud ? a.idx2 : a.idx1)[(size_t)b * npad + s] = f;
        wave_tree(c[0], c[1], c[2], x, valid, lane, (cloud ? a.grp2 : a.grp1) + ((size_t)b * (npad / SGT) + (s - lane) / SGT) * NODE);
    }
    // per-workgroup partial AABB of the P0s and max |P|^2, as tri_records_kernel leaves them
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
    if (tid < 7) {
        float r = red[0][tid];
        for (int w = 1; w < REC_BLK / 64; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        a.apart[(((size_t)cloud * B + b) * a.nblk + blockIdx.x) * 8 + tid] = r;
    } else if (tid == 7) {
        a.apart[(((size_t)cloud * B + b) * a.nblk + blockIdx.x) * 8 + 7] = 1.0f;  // PTRI layout of this cloud: sorted positions
    }
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
#if defined(SORT_STOP) && SORT_STOP == 4  // timing experiments only
    return;
#endif
    for (int s = p0 + tid; s < p1; s += 1024) {  // coalesced copy-out
        p0s[s] = srec[pad(s - p0)];
        idx[s] = sidx[s - p0];
    }
#if defined(SORT_STOP) && SORT_STOP == 5
    return;
#endif
    // ---- sphere tree: one lane per half of 8 records
    for (int hh = 2 * SGG * S0 + tid; hh < 2 * SGG * S1; hh += 1024)
        half_tree([&](int s_) { return srec[pad(s_ - p0)]; }, hh, n, tree);
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

#ifndef SPW
#define SPW 8      // supergroups per slice (<= 16: SG_BITS); staged once per workgroup
#endif
#define SG_BITS 4              // queue entries: line << SG_BITS | supergroup of the slice,
#define HF_BITS (SG_BITS + 3)  //                line << HF_BITS | half of the slice (7 + 7 bits: 16-bit entries)
static_assert(SPW <= (1 << SG_BITS), "slot bits of the queue entries");
#ifndef WPB
#define WPB 8      // wavefronts per workgroup: same slice, LPW lines each
#endif
#define LPW 128    // lines per wavefront (two per lane in level A)
#ifndef ROWS
#define ROWS 17    // float4 per staged group row: 16 records + 16 bytes of padding (bank spread)
#endif
#ifndef WCCAP
#define WCCAP 128  // parked point-0 candidates per wave
#endif
// queue capacities.  Level A leaves ALL of a wavefront's (line, supergroup) pairs in queue A at once when they fit
// (~128 of the 1024 possible; otherwise supergroup by supergroup with drains in between); level B pops 64 of them
// and pushes their passing halves (~1.4 of 8 each, at most 512: split then) on top of < 64 left-overs of queue C.
#ifndef QA_CAP
#define QA_CAP 256
#endif
#ifndef QC_CAP
#define QC_CAP 256
#endif

#ifndef CULL_REGLINES
#define CULL_REGLINES 0  // 1 (experiment, round 4; exact on the whole suite, NOT shipped): the wavefront's 128 lines stay in
#endif                   // REGISTERS (two per lane) and travel between lanes by ds_bpermute instead of 24 KB of LDS per
//                          workgroup -> 22.6 KB, 62 VGPRs, 4 workgroups = 32 wavefronts per compute unit instead of 3 / 24.
//                          Measured slower: 28.3 us against 26.8-27.4 at C2, 13.9 against 12.9 at C1, even at B = 64 --
//                          twelve cross-lane reads per pass cost more than the extra residency returns; the launch is
//                          paced by each wavefront's own chain of dependent steps, not by the number of resident ones.
struct WaveCtx {
#if CULL_REGLINES
    float v0[6], v1[6];           // this lane's two lines (dir, x0): line `lane` and line `64 + lane` of the wavefront
#else
    const float2 *lr;             // this wave's lines in LDS as they lie in memory: 3 float2 per line (dir.xy | dir.z x0.x | x0.yz)
#endif
    const float4 *recs;           // LDS: staged records of the slice, [group][ROWS]
    const float4 *nodes;          // LDS: staged tree nodes of the slice, [supergroup][NODE]
    unsigned short *qa, *qc;      // LDS queues: line << SG_BITS | sg, line << HF_BITS | half (slice-local)
    unsigned *cands;              // LDS [WCCAP]
    const int32_t *idx;           // sorted position -> original triangle index
    const float *ptri;            // prepared triangles: rows in original order, or -- psorted -- at their sorted positions
    bool psorted;                 // uniform: the layout of ptri (the prepared build leaves sorted rows)
    int32_t *cnt, *hit;           // per-line hit count / slots of the cloud
    int lbase;                    // first line of this wave
    int pos0;                     // sorted position of the slice's first record
    int na, nc, ncand;            // wave-uniform fill levels
    int lane;
    // executed-work counters of the COUNT instantiation (wave-uniform; see rrl_scan_counters)
    unsigned tb, tc, td, tcand;   // level-B half-sphere tests, halves that passed, point-0 prefilter tests, resolved candidates
    int32_t *status;              // NaN flag of the call
};

// line ll of the wave from its raw 24-byte row: (dir, x0.x) and (x0.y, x0.z)
struct LineRow {
    float4 la;
    float2 lb;
};
__device__ __forceinline__ LineRow line_row(const float2 *lr, int ll) {
    const float2 *p = lr + 3 * ll;
    const float2 a = p[0], b = p[1], c = p[2];
    return {make_float4(a.x, a.y, b.x, b.y), c};
}
#if CULL_REGLINES
// Line ll (0 .. 127) of the wavefront from the registers of the lane that holds it: EVERY lane of the wavefront must
// call (a ds_bpermute reads the registers of active lanes only); lanes without work pass any ll.  Twelve cross-lane
// reads + six selects instead of three 8-byte LDS reads -- and 24 KB of LDS per workgroup less: 22.3 KB instead of 47.2,
// so the wavefront limit (8 per SIMD), not LDS, bounds the residency: 4 workgroups per compute unit instead of 3.
__device__ __forceinline__ LineRow line_get(const WaveCtx &c, int ll) {
    const int src = (ll & 63) << 2;
    const bool hi = ll >= 64;
    float r[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int a = __builtin_amdgcn_ds_bpermute(src, __float_as_int(c.v0[q]));
        const int b = __builtin_amdgcn_ds_bpermute(src, __float_as_int(c.v1[q]));
        r[q] = __int_as_float(hi ? b : a);
    }
    return {make_float4(r[0], r[1], r[2], r[3]), make_float2(r[4], r[5])};
}
#endif

// cand = line_in_wave << 16 | sorted triangle position: a triangle whose point 0 passed the conservative
// prefilter of level D.  The reference's own arithmetic (dist_sq, bit-identical to the strict scan) decides
// on all three points here.
__device__ __forceinline__ void resolve_candidate(const WaveCtx &c, unsigned cand, const LineRow &lrw) {
    const int ll = cand >> 16, spos = cand & 0xffff;
    const float4 la = lrw.la;
    const float2 lb = lrw.lb;
    // original-order rows: one more dependent load for the row index (staging the slice's indices in LDS measured 0.8 us
    // slower); sorted rows (prepared build): the row sits at the candidate's position and carries its triangle index
    const int prow = c.psorted ? spos : c.idx[spos];
    const float4 *q = (const float4 *)(c.ptri + PTRI_STRIDE * (size_t)prow);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2];  // P0 P1.x | P1.yz P2.xy | P2.z thr2 thr index
    const int f = __float_as_int(r2.w);
    const uint32_t thr2 = __float_as_uint(r2.y);
    const uint32_t x0 = __float_as_uint(dist_sq<float>(r0.x, r0.y, r0.z, la.x, la.y, la.z, la.w, lb.x, lb.y));
    const uint32_t x1 = __float_as_uint(dist_sq<float>(r0.w, r1.x, r1.y, la.x, la.y, la.z, la.w, lb.x, lb.y));
    const uint32_t x2 = __float_as_uint(dist_sq<float>(r1.z, r1.w, r2.x, la.x, la.y, la.z, la.w, lb.x, lb.y));
    // a negative sqrt argument is the reference's NaN (code/loss.py:88-91); candidates are rare
    if ((x0 | x1 | x2) >= 0x80000000u) atomicOr(&c.status[0], 1);
    if (max(max(x0, x1), x2) < thr2) {  // sign bit set (negative / NaN) -> huge: never a hit
        const int l = c.lbase + ll;
        int pos = atomicAdd(&c.cnt[l], 1);
        if (pos < RRL_MAX_HITS) c.hit[(size_t)l * RRL_MAX_HITS + pos] = f;
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS is processed in order per wave; this only stops the compiler from reordering across it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__device__ __forceinline__ int lane_rank(unsigned long long m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Slack of the culling bound (header) for one (cloud, sample) from the maxima over its cullable lines: s = max |dir|^2,
// o2 = max |x0|^2, pm = max |P|^2.  Every term is monotone in s and o2, so the values cover each admitted line.
struct CloudSlack {
    bool ok;       // pm is finite and moderate (else: every wavefront of this cloud takes the strict loop)
    bool nanwide;  // g > 0: a negative sqrt argument is not excluded at this scale
    float se;      // slack of the node radii
    float s0;      // slack of the point-0 prefilter
};
// max over the wavefront of NON-NEGATIVE floats (or NaN, which then wins: its bit pattern is the largest) on their bit
// patterns: v_max_u32 takes the DPP operand directly (the float version needs a canonicalising v_max per step), and
// the four row results meet in scalar registers.  Every lane receives the result.
__device__ __forceinline__ float wave_max_nonneg(float v) {
    unsigned u = __float_as_uint(v);
#define RRL_DPP_U(x, ctrl) (unsigned)__builtin_amdgcn_update_dpp((int)(x), (int)(x), ctrl, 0xf, 0xf, false)
    u = max(u, RRL_DPP_U(u, 0xB1)); u = max(u, RRL_DPP_U(u, 0x4E));
    u = max(u, RRL_DPP_U(u, 0x141)); u = max(u, RRL_DPP_U(u, 0x140));
#undef RRL_DPP_U
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)u, 0), b = (unsigned)__builtin_amdgcn_readlane((int)u, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)u, 32), d = (unsigned)__builtin_amdgcn_readlane((int)u, 48);
    return __uint_as_float(max(max(a, b), max(c, d)));
}
// square root / reciprocal for the SLACKS only (never for a label): the hardware approximations (1 ulp), nudged upwards
// so that the result is >= the exact one -- a slack may be too large by 1e-6 of itself, never too small.  (The IEEE
// sqrtf / division sequences were ~55 of the ~250 VALU instructions of every wavefront's prologue.)
__device__ __forceinline__ float sqrt_up(float x) { return __builtin_amdgcn_sqrtf(x) * 1.0000005f; }
__device__ __forceinline__ float rcp_up(float x) { return __builtin_amdgcn_rcpf(x) * 1.0000005f; }

__device__ __forceinline__ CloudSlack cull_cloud_slack(float s, float o2, float pm) {
    CloudSlack r;
    r.ok = pm <= 1.0e11f;  // (false for NaN / inf: non-finite coordinates)
    const float A2 = (o2 + pm + 2.0f * sqrt_up(o2 * pm)) * 1.00001f;  // (|x0| + max|P|)^2, rounded up
    const float eta = fmaxf(s - 1.0f, 0.0f) + 2.5e-7f;  // |d|^2 - 1 incl. the rounding of s (<= 3u s)
    const float x = eta * A2;
    const float g = 1.8e-6f * A2 - 2e-4f;  // 30u = 1.788e-6
    r.nanwide = g > 0.0f;
    // (-0.999 g under the root: a smaller divisor, i.e. a larger quotient; the approximations only add to that)
    const float se = g > 0.0f ? sqrt_up(g + x) : fminf(sqrt_up(x), x * 0.5f * rcp_up(__builtin_amdgcn_sqrtf(-0.998f * g)));
    r.se = se * 1.0001f + 1e-12f;
    r.s0 = 3.0e-6f * A2 + 1.0e-9f;  // 44.2 u A^2 of evaluation error on both sides, rounded up, + an absolute floor
    return r;
}

// conservative sphere test of one line against one STAGED tree node (centre, w = (Rs + se)^2 rounded up, or -inf for
// an empty node; see the culling bound above): the node is kept when  d2 - 4e-6 q <= w,  evaluated as ONE FMA chain
// whose SIGN BIT says "cannot be reached":  m = (a.d)^2 + (w - 0.999996 |a|^2)  (exactly the negation of the form
// d2 - 4e-6 q - w; -0 cannot occur, -inf - ... stays negative)
__device__ __forceinline__ float sphere_margin(const float4 nd, const float4 la, const float2 lb) {
    const float ax = nd.x - la.w, ay = nd.y - lb.x, az = nd.z - lb.y;
    const float dot = fmaf(az, la.z, fmaf(ay, la.y, ax * la.x));
    const float q = fmaf(az, az, fmaf(ay, ay, ax * ax));
    return fmaf(dot, dot, fmaf(-q, 0.999996f, nd.w));
}

template <bool COUNT>
__device__ __forceinline__ void flush_cands(WaveCtx &c) {
#ifdef CULL_NO_RESOLVE
    c.ncand = 0;
    return;
#endif
    wave_lds_fence();
    const int nc = min(c.ncand, WCCAP);
    if constexpr (COUNT) c.tcand += (unsigned)c.ncand;  // entries past WCCAP were resolved inline: counted here too
#if CULL_REGLINES
    for (int i0 = 0; i0 < nc; i0 += 64) {  // uniform trip count: every lane fetches (line_get), the live ones resolve
        const int i = i0 + c.lane;
        const unsigned cand = i < nc ? c.cands[i] : 0u;
        const LineRow lrw = line_get(c, (int)(cand >> 16));
        if (i < nc) resolve_candidate(c, cand, lrw);
    }
#else
    for (int i = c.lane; i < nc; i += 64) { const unsigned cand = c.cands[i]; resolve_candidate(c, cand, line_row(c.lr, (int)(cand >> 16))); }
#endif
    c.ncand = 0;
}

// all = false: only while at least 64 entries wait (full lanes); all = true: drain.

// level D: pops (line, half) pairs, ONE per lane (8 records = 32 registers in flight), and runs the point-0
// PREFILTER on the half's 8 records: e = fl(Q(P0)) + c by FMAs, with the staged c = -(thr2 - 2e-4 + slack)
// (see "Point-0 prefilter" above): 10 operations and one v_alignbit per record instead of the 16 + 3 of
// the reference's unfused arithmetic.  e < 0 (sign bit) parks the triangle as a candidate; the exact test of
// all three points happens there (resolve_candidate), on ~3 % of the records.
template <bool COUNT>
__device__ __forceinline__ void proc_c(WaveCtx &c, bool all) {
#ifdef CULL_STOP_C  // timing experiments only (tools/knob_sweep.sh): the level is formed but not run
    c.nc = 0;
    return;
#endif
    while (c.nc >= 64 || (all && c.nc > 0)) {
        const int take = min(c.nc, 64), base = c.nc - take;
        c.nc = base;
        if constexpr (COUNT) c.td += 8u * (unsigned)take;
        wave_lds_fence();
        uint32_t passbits = 0;
        unsigned lh = 0;  // line << 16 | first sorted position of the half
#if CULL_REGLINES
        const unsigned e = c.lane < take ? c.qc[base + c.lane] : 0u;
        const int ll = e >> HF_BITS, h = e & ((1 << HF_BITS) - 1);
        const LineRow lrw = line_get(c, ll);  // (every lane)
        if (c.lane < take) {
#else
        if (c.lane < take) {
            const unsigned e = c.qc[base + c.lane];
            const int ll = e >> HF_BITS, h = e & ((1 << HF_BITS) - 1);
            const LineRow lrw = line_row(c.lr, ll);
#endif
            const float4 la = lrw.la;
            const float2 lb = lrw.lb;
            const float4 *row = c.recs + (h >> 1) * ROWS + (h & 1) * 8;
#pragma unroll
            for (int t = 7; t >= 0; --t) {  // record t ends up in bit t
                const float4 rec = row[t];
                const float ax = rec.x - la.w, ay = rec.y - lb.x, az = rec.z - lb.y;
                const float dot = fmaf(az, la.z, fmaf(ay, la.y, ax * la.x));
                const float q = fmaf(az, az, fmaf(ay, ay, fmaf(ax, ax, rec.w)));
                const float ev = fmaf(-dot, dot, q);
                passbits = __builtin_amdgcn_alignbit(passbits, __float_as_uint(ev), 31);  // passbits << 1 | sign(ev)
            }
            lh = ((unsigned)ll << 16) | (unsigned)(c.pos0 + h * 8);
        }
        while (__any(passbits != 0)) {
#if CULL_REGLINES
            if (c.ncand > WCCAP - 64) flush_cands<COUNT>(c);  // uniform: room for this round's <= 64 candidates (no inline resolve)
#endif
            const bool has = passbits != 0;
            const unsigned long long m = __ballot(has);
            const int t = has ? __ffs(passbits) - 1 : 0;
            passbits &= passbits - 1;
            const int pos = c.ncand + lane_rank(m);
            const unsigned cand = lh + (unsigned)t;
            if (has) {
#if CULL_REGLINES
                c.cands[pos] = cand;
#else
                if (pos < WCCAP) c.cands[pos] = cand;
                else resolve_candidate(c, cand, line_row(c.lr, (int)(cand >> 16)));
#endif
            }
            c.ncand += __popcll(m);
        }
        if (c.ncand > WCCAP - 64) flush_cands<COUNT>(c);  // uniform: keep room for the next pass
    }
}

// The passing halves of one level-B pass (bit k of `pass`: half 8 sg + k of the lane's entry; e2 = line << HF_BITS |
// 8 sg) go to queue C.  Usual case: every lane writes its own run of entries behind an exclusive prefix of the
// counts (one DPP scan instead of eight ballot / rank rounds).  Too many for the queue (dense hits): half by half
// with drains in between.
template <bool COUNT>
__device__ __forceinline__ void push_halves(WaveCtx &c, unsigned pass, unsigned e2) {
    const int cnt = __popc(pass);
    const int incl = wave_incl_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;  // uniform
    if constexpr (COUNT) c.tc += (unsigned)total;
    if (c.nc + total > QC_CAP) proc_c<COUNT>(c, false);  // leaves < 64
    if (c.nc + total > QC_CAP) {
        for (int k = 0; k < 8; ++k) {  // uniform
            const bool p = (pass >> k) & 1u;
            const unsigned long long m = __ballot(p);
            if (m == 0ull) continue;
            if (c.nc > QC_CAP - 64) proc_c<COUNT>(c, false);
            if (p) c.qc[c.nc + lane_rank(m)] = (unsigned short)(e2 | (unsigned)k);
            c.nc += __popcll(m);
        }
        return;
    }
    int at = c.nc + incl - cnt;
    while (pass) {
        const int k = __ffs(pass) - 1;
        pass &= pass - 1;
        c.qc[at++] = (unsigned short)(e2 | (unsigned)k);
    }
    c.nc += total;
}

// level B: pops (line, supergroup) pairs, one per lane, and tests the supergroup's EIGHT half spheres directly
// (round 3: the group level in between -- 4 group tests, then 2 half tests per passing group, 9 tests on average
// -- cost a whole round of queue passes per wavefront for one test less)
template <bool COUNT>
__device__ __forceinline__ void proc_a(WaveCtx &c, bool all) {
#ifdef CULL_STOP_A
    c.na = 0;
    return;
#endif
    while (c.na >= 64 || (all && c.na > 0)) {
        const int take = min(c.na, 64), base = c.na - take;
        c.na = base;
        if constexpr (COUNT) c.tb += 8u * (unsigned)take;
        wave_lds_fence();
        unsigned fail = 0xffu;  // bit k: half k cannot be reached (idle lanes: none can)
        unsigned e2 = 0;
#if CULL_REGLINES
        const unsigned e = c.lane < take ? c.qa[base + c.lane] : 0u;
        const int ll = e >> SG_BITS, sg = e & ((1 << SG_BITS) - 1);
        const LineRow lrw = line_get(c, ll);  // (every lane)
        if (c.lane < take) {
#else
        if (c.lane < take) {
            const unsigned e = c.qa[base + c.lane];
            const int ll = e >> SG_BITS, sg = e & ((1 << SG_BITS) - 1);
            const LineRow lrw = line_row(c.lr, ll);
#endif
            const float4 la = lrw.la;
            const float2 lb = lrw.lb;
            const float4 *nd = c.nodes + sg * NODE + 5;
            fail = 0u;
#pragma unroll
            for (int k = 7; k >= 0; --k)  // half k ends up in bit k
                fail = __builtin_amdgcn_alignbit(fail, __float_as_uint(sphere_margin(nd[k], la, lb)), 31);
            e2 = ((unsigned)ll << HF_BITS) | (unsigned)(8 * sg);
        }
        push_halves<COUNT>(c, ~fail & 0xffu, e2);
    }
}

// The strict loop of a wavefront that cannot be culled (a line with |dir|^2 > 1 + 1e-6 or non-finite
// data): ALL pairs of its 128 lines with the records at sorted positions [s0, s1), the reference's
// semantics, NaN included.  The lane's two lines arrive packed (.x = line l0, .y = line l1).
__device__ __forceinline__ void strict_slice(const float *ptri, const int32_t *idx, bool psorted, int s0, int s1, v2f ux,
                                                       v2f uy, v2f uz, v2f ox, v2f oy, v2f oz, int l0, int l1, int L,
                                                       int32_t *cnt, int32_t *hit, int32_t *status) {
    kptr tp0 = (kptr)(uintptr_t)ptri;
    kiptr ik = (kiptr)(uintptr_t)idx;
    uint32_t nanacc = 0;
    for (int sp = s0; sp < s1; ++sp) {
        kptr tp = tp0 + (size_t)(psorted ? sp : ik[sp]) * PTRI_STRIDE;
        const uint32_t thr2 = __float_as_uint(tp[9]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float dx = h ? ux.y : ux.x, dy = h ? uy.y : uy.x, dz = h ? uz.y : uz.x;
            const float px = h ? ox.y : ox.x, py = h ? oy.y : oy.x, pz = h ? oz.y : oz.x;
            const int l = h ? l1 : l0;
            const uint32_t x0 = __float_as_uint(dist_sq<float>(tp[0], tp[1], tp[2], dx, dy, dz, px, py, pz));
            const uint32_t x1 = __float_as_uint(dist_sq<float>(tp[3], tp[4], tp[5], dx, dy, dz, px, py, pz));
            const uint32_t x2 = __float_as_uint(dist_sq<float>(tp[6], tp[7], tp[8], dx, dy, dz, px, py, pz));
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
}

// The culled walk of one wavefront over the STAGED slice (node_lds / ctx.recs): level A on the lane's two lines,
// level B, level D, candidate resolution.  stamps (COUNT): wall clock after level A, after level B's drain, after
// level D's drain.
template <bool COUNT>
__device__ __forceinline__ void cull_walk(WaveCtx &ctx, const float4 *node_lds, int nsl, int lane, bool live0, bool live1,
                                          v2f ux, v2f uy, v2f uz, v2f ox, v2f oy, v2f oz, unsigned long long *stamps) {
    // ---- level A: conservative sphere test of every supergroup of the slice against the lane's two lines
    //      (packed fp32; the staged supergroup nodes come as wave-uniform LDS reads).  No queue traffic inside the
    //      loop: the outcome is one bit per (line, supergroup) in two lane-private masks (sign bits, v_alignbit).
    unsigned f0m = 0u, f1m = 0u;  // bit s: supergroup s cannot be reached by line 0 / line 1 of the lane
#pragma unroll
    for (int s = SPW - 1; s >= 0; --s) {  // supergroup s ends up in bit s
        float4 nd = node_lds[(s < nsl ? s : 0) * NODE];
        if (s >= nsl) nd.w = -INFINITY;  // uniform: not part of this slice
        const v2f ax = nd.x - ox, ay = nd.y - oy, az = nd.z - oz;
        const v2f dot = __builtin_elementwise_fma(az, uz, __builtin_elementwise_fma(ay, uy, ax * ux));
        const v2f q = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
        const v2f w2 = {nd.w, nd.w};
        const v2f mg = __builtin_elementwise_fma(dot, dot, __builtin_elementwise_fma(-q, (v2f){0.999996f, 0.999996f}, w2));
        f0m = __builtin_amdgcn_alignbit(f0m, __float_as_uint(mg.x), 31);
        f1m = __builtin_amdgcn_alignbit(f1m, __float_as_uint(mg.y), 31);
    }
    unsigned m0 = live0 ? (~f0m & ((1u << SPW) - 1u)) : 0u, m1 = live1 ? (~f1m & ((1u << SPW) - 1u)) : 0u;
    {
        const int cnt_ = __popc(m0) + __popc(m1);
        const int incl = wave_incl_scan(cnt_);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total <= QA_CAP) {  // the usual case (~128): every lane writes its own run of entries
            int at = incl - cnt_;
            while (m0) {
                const int s = __ffs(m0) - 1;
                m0 &= m0 - 1;
                ctx.qa[at++] = (unsigned short)((lane << SG_BITS) | s);
            }
            while (m1) {
                const int s = __ffs(m1) - 1;
                m1 &= m1 - 1;
                ctx.qa[at++] = (unsigned short)(((64 + lane) << SG_BITS) | s);
            }
            ctx.na = total;
        } else {  // dense hits: supergroup by supergroup, level B in between
            for (int s = 0; s < nsl; ++s) {
                const bool p0 = (m0 >> s) & 1u, p1 = (m1 >> s) & 1u;
                const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
                if ((b0 | b1) == 0ull) continue;
                if (ctx.na > QA_CAP - 128) proc_a<COUNT>(ctx, false);
                const int c0 = __popcll(b0);
                if (p0) ctx.qa[ctx.na + lane_rank(b0)] = (unsigned short)((lane << SG_BITS) | s);
                if (p1) ctx.qa[ctx.na + c0 + lane_rank(b1)] = (unsigned short)(((64 + lane) << SG_BITS) | s);
                ctx.na += c0 + __popcll(b1);
            }
        }
    }
    if constexpr (COUNT) stamps[0] = wall_clock64();
    proc_a<COUNT>(ctx, true);
    if constexpr (COUNT) stamps[1] = wall_clock64();
    proc_c<COUNT>(ctx, true);
    if constexpr (COUNT) stamps[2] = wall_clock64();
    flush_cands<COUNT>(ctx);
}

// COUNT = true: the same kernel with executed-work counters -- launched instead of the plain one while
// rrl_scan_counters() holds a buffer.  Every wavefront WRITES one row of 16 u64 (plain stores: thousands of
// same-address atomics serialise at ~12 ns each and distort the kernel they measure), row index = linear
// workgroup id x wavefronts per workgroup + wavefront; rows past the buffer's capacity are dropped:
//   row[0] level-A sphere tests (line x supergroup)   [1] level-B half-sphere tests (8 per (line, supergroup) pair)
//           [2] halves that passed (line x half)           [3] point-0 prefilter tests (line x record)
//           [4] candidates resolved (points 1, 2)          [5] wavefronts that ran
//           [6] wavefronts that took the strict fallback   [7] (line, triangle) pairs of the fallback
//           [8] start, [9] end of the wavefront on the 100 MHz wall clock (the kernel lasts as long as its
//           slowest wavefront: tools/scan_tail.py prints the spread), [10..14] phase stamps
// The scan's LDS (47 KiB per 8-wavefront workgroup) as ONE object: cull_scan_chamfer_kernel overlays it with the Chamfer
// walk's (rrl_chamfer_walk.h ChamLds).
struct CullLds {
#if !CULL_REGLINES
    __attribute__((aligned(16))) float2 line_lds[WPB][LPW * 3];    // 24 KiB: raw 24-byte line rows
#endif
    __attribute__((aligned(16))) float4 rec_lds[SPW * SGG * ROWS]; //  8.5 KiB
    __attribute__((aligned(16))) float4 node_lds[SPW * NODE];      //  1.6 KiB
    __attribute__((aligned(16))) unsigned short qa_lds[WPB][QA_CAP];
    unsigned short qc_lds[WPB][QC_CAP];
    __attribute__((aligned(16))) unsigned cands_lds[WPB][WCCAP];
};

// One workgroup of the culled scan: (bx, by, bz) = its place in the scan's grid (gx, gy: the grid's first two extents)
// -- blockIdx / gridDim in cull_scan_kernel, decoded from a linear index in cull_scan_chamfer_kernel.
template <bool COUNT>
__device__ __forceinline__ void cull_scan_body(
    CullLds &lds_, const float *__restrict__ ptri1, const float *__restrict__ ptri2, const float4 *__restrict__ p0s1,
    const float4 *__restrict__ p0s2, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    const float4 *__restrict__ tree1, const float4 *__restrict__ tree2, const float *__restrict__ line,
    int32_t *__restrict__ count1, int32_t *__restrict__ hit1, int32_t *__restrict__ count2,
    int32_t *__restrict__ hit2, int32_t *__restrict__ status, uint32_t *pmax,
    const float *__restrict__ del1, const float *__restrict__ del2, const float2 *__restrict__ lmax,
    const float *__restrict__ apart, const float *__restrict__ aflag, int nblk_apart, int B,
    int N, int M, int L, int spw, unsigned long long *__restrict__ counters, long long counter_rows,
    const int bx, const int by, const int bz, const int gx, const int gy, const int Bt) {
#if !CULL_REGLINES
    float2 (&line_lds)[WPB][LPW * 3] = lds_.line_lds;
#endif
    float4 (&rec_lds)[SPW * SGG * ROWS] = lds_.rec_lds;
    float4 (&node_lds)[SPW * NODE] = lds_.node_lds;
    unsigned short (&qa_lds)[WPB][QA_CAP] = lds_.qa_lds;
    unsigned short (&qc_lds)[WPB][QC_CAP] = lds_.qc_lds;
    unsigned (&cands_lds)[WPB][WCCAP] = lds_.cands_lds;
    static_assert(WPB * WCCAP >= SPW * SGT, "the NaN-reach scratch aliases the (still unused) candidate buffers");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler
    const unsigned long long wall0 = COUNT ? wall_clock64() : 0ull;
    unsigned long long *crow = nullptr;
    if constexpr (COUNT) {
        const long long wid = (((long long)bz * gy + by) * gx + bx) * (blockDim.x >> 6) + wave;
        if (wid < counter_rows) crow = counters + 16 * wid;
    }
    // XCD-aware mapping: workgroups go to the 8 XCDs round-robin by linear id, and x is the fast
    // index -- with (cloud, sample) on x, all workgroups of one cloud land on the same XCD (when
    // 2B is a multiple of 8), so each XCD's L2 holds 1/8 of the records instead of a copy of all
    const int z = bx, cloud = z >= B ? 1 : 0, b = z - cloud * B;
    const int n = cloud ? M : N;
    const int nsg = (n + SGT - 1) / SGT;
    const int sg0 = bz * spw;
    if (sg0 >= nsg) return;  // uniform: the smaller cloud has fewer slices
    // multi-pose evaluation (rrl_opts.problems): instance b has the target and lines of problem b % Bt -- the target's scan
    // is the same for every pose, so only the first Bt instances scan cloud 2 (the per-line stage reads it there)
    if (cloud && Bt > 0 && b >= Bt) return;  // uniform
    const int nsl = min(spw, nsg - sg0);
    const float4 *p0s = (cloud ? p0s2 : p0s1) + (size_t)b * nsg * SGT;
    const float4 *tree = (cloud ? tree2 : tree1) + (size_t)b * nsg * NODE;

    // ---- everything the prologue needs is requested up front, in one round of independent loads: the slice's
    //      records and nodes (one per lane in the usual 8-wavefront workgroup), the sample's line maxima, this
    //      wavefront's 128 lines
    constexpr int RPT = (SPW * SGT + 64 * WPB - 1) / (64 * WPB);  // records per lane of a full workgroup
    static_assert(SPW * NODE <= 64 * WPB, "one node per lane");
    const bool one_each = (int)blockDim.x == 64 * WPB;
    const int32_t *idx = (cloud ? idx2 : idx1) + (size_t)b * nsg * SGT;
    float4 rec0[RPT], nd0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    static_assert(LMAX_CHUNKS == 64, "one partial row per lane");
    // (the line maxima and max |P|^2 are requested FIRST: vector loads return in order, so the slack arithmetic below can
    //  start while the records, nodes and lines requested after them are still in flight)
    const float2 lm = lmax[(size_t)b * LMAX_CHUNKS + lane];  // (max |dir|^2, max |x0|^2) over 1/64 of the sample's cullable lines
    // layout of this cloud's PTRI rows, left by the records launch that built it (slot 7 of its first partial row; not
    // needed before the first candidate is resolved)
    const float play = aflag[(size_t)(cloud * B + b) * nblk_apart * 8 + 7];
    // max |P|^2 of the cloud: from the sort kernel (PMAX), or -- prepared clouds, whose build has no single-workgroup stage
    // -- from the records kernel's per-workgroup partial rows, reduced here next to the line maxima (one more independent
    // load of the prologue; one workgroup per cloud and sample leaves PMAX for the later consumers)
    float pmv = 0.0f;
    if (apart != nullptr) {  // uniform
        const int nb = (n + REC_BLK - 1) / REC_BLK;
        const float *ap = apart + (size_t)(cloud * B + b) * nblk_apart * 8 + 6;
        for (int j = lane; j < nb; j += 64) pmv = fmaxf(pmv, ap[(size_t)j * 8]);
    } else {
        pmv = __uint_as_float(pmax[cloud * B + b]);
    }
    if (one_each) {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (tid + 64 * WPB * k < nsl * SGT) rec0[k] = p0s[(size_t)sg0 * SGT + tid + 64 * WPB * k];
        if (tid < nsl * NODE) nd0 = tree[(size_t)sg0 * NODE + tid];
    }

    // this wave's 128 lines: a full, 16-byte aligned tile arrives as three coalesced 16-byte loads per lane straight
    // into its LDS rows; the lanes then pick up their own two lines from there
    const float *ln = line + (size_t)input_of(b, Bt) * L * 6;
    const int lw0 = (by * (int)(blockDim.x >> 6) + wave) * LPW;
    const int l0 = lw0 + lane, l1 = l0 + 64;
    const bool live0 = l0 < L, live1 = l1 < L;
    const bool has_lines = lw0 < L;  // a wave without lines (the last tile of the line set) still stages records
#if CULL_REGLINES
    // each lane's own two lines straight into registers: three 8-byte loads per line (rows are 24 bytes, 8-byte aligned;
    // the three instructions of a line cover the wavefront's 1536 contiguous bytes completely); rows past L: zero lines
    float2 g0[3] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)}, g1[3] = {g0[0], g0[0], g0[0]};
    {
        const float2 *s2 = (const float2 *)ln;
        if (live0) { g0[0] = s2[3 * (size_t)l0]; g0[1] = s2[3 * (size_t)l0 + 1]; g0[2] = s2[3 * (size_t)l0 + 2]; }
        if (live1) { g1[0] = s2[3 * (size_t)l1]; g1[1] = s2[3 * (size_t)l1 + 1]; g1[2] = s2[3 * (size_t)l1 + 2]; }
    }
#else
    float2 *lr = line_lds[wave];
    const float *lsrc = ln + (size_t)lw0 * 6;
    const bool full_tile = has_lines && lw0 + LPW <= L && (((uintptr_t)lsrc) & 15) == 0;  // uniform
    float4 t0, t1, t2;
    if (full_tile) {
        const float4 *s4 = (const float4 *)lsrc;
        t0 = s4[lane]; t1 = s4[64 + lane]; t2 = s4[128 + lane];
    }
#endif

    // ---- slack of this (cloud, sample): the same values in every wavefront and workgroup (no exchange, no barrier)
    float pm = pmv;
    if (apart != nullptr) {  // uniform
        pm = wave_max_nonneg(pmv);
        if (by == 0 && bz == 0 && tid == 0) pmax[cloud * B + b] = __float_as_uint(pm);
    }
    const float smax = wave_max_nonneg(lm.x), o2max = wave_max_nonneg(lm.y);
    const CloudSlack cs = cull_cloud_slack(smax, o2max, pm);
    const float se = cs.se, s0 = cs.s0;
    const bool nanwide = cs.nanwide;  // uniform over the launch's workgroups of this cloud and sample
    // nanwide (header, "NaN"): the NaN reach del of this lane's records, requested while the lines are still in flight
    float dv[RPT];
    const bool psorted = __builtin_amdgcn_readfirstlane(__float_as_int(play)) != 0;  // (1.0f: PTRI / DEL rows at sorted positions)
    if (nanwide && one_each) {
        const float *del = (cloud ? del2 : del1) + (size_t)b * n;
        if (psorted) {  // uniform: the prepared build left DEL at the sorted positions -- no dependent gather through IDX
#pragma unroll
            for (int k = 0; k < RPT; ++k) dv[k] = sg0 * SGT + tid + 64 * WPB * k < n ? del[sg0 * SGT + tid + 64 * WPB * k] : 0.0f;
        } else {
            int idx0[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) idx0[k] = tid + 64 * WPB * k < nsl * SGT ? idx[sg0 * SGT + tid + 64 * WPB * k] : 0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) dv[k] = sg0 * SGT + tid + 64 * WPB * k < n ? del[idx0[k]] : 0.0f;
        }
    }
#if CULL_REGLINES
    const float v0[6] = {g0[0].x, g0[0].y, g0[1].x, g0[1].y, g0[2].x, g0[2].y};
    const float v1[6] = {g1[0].x, g1[0].y, g1[1].x, g1[1].y, g1[2].x, g1[2].y};
#else
    if (full_tile) {
        float4 *d4 = (float4 *)lr;
        d4[lane] = t0; d4[64 + lane] = t1; d4[128 + lane] = t2;
    } else if (has_lines) {  // ragged tail / odd alignment: 8-byte pieces (a row is 24 bytes), zeros past the end
        const float2 *s2 = (const float2 *)lsrc;
        const int nf2 = (L - lw0) * 3;
        for (int i = lane; i < LPW * 3; i += 64) lr[i] = i < nf2 ? s2[i] : make_float2(0.0f, 0.0f);
    }

    wave_lds_fence();
    float v0[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, v1[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (has_lines) {
        const LineRow r0 = line_row(lr, lane), r1 = line_row(lr, 64 + lane);
        v0[0] = r0.la.x; v0[1] = r0.la.y; v0[2] = r0.la.z; v0[3] = r0.la.w; v0[4] = r0.lb.x; v0[5] = r0.lb.y;
        v1[0] = r1.la.x; v1[1] = r1.la.y; v1[2] = r1.la.z; v1[3] = r1.la.w; v1[4] = r1.lb.x; v1[5] = r1.lb.y;
    }
#endif
    // Culling (and the lazy evaluation of points 1, 2) is only exact for lines with |dir|^2 <= 1 + 1e-6 and finite,
    // moderate data.  A wavefront with an offending line evaluates ALL pairs of its lines with the slice's
    // triangles strictly instead -- the reference's semantics, NaN included.
    float sa, oa, sb, ob;
    line_norms(v0[0], v0[1], v0[2], v0[3], v0[4], v0[5], sa, oa);
    line_norms(v1[0], v1[1], v1[2], v1[3], v1[4], v1[5], sb, ob);
    const bool fallback = !cs.ok || !__all(line_cullable(sa, oa) && line_cullable(sb, ob));
    unsigned long long fb_pairs = 0;

    const float *ptri = (cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE;
    int32_t *cnt = (cloud ? count2 : count1) + (size_t)b * L;
    int32_t *hit = (cloud ? hit2 : hit1) + (size_t)b * L * RRL_MAX_HITS;

    // nanwide: the slice's del values meet in LDS (a buffer the walk does not use yet) so that every node lane can
    // take the maximum over ITS records -- one more barrier, only on this path
    float *dl = (float *)&cands_lds[0][0];  // [SPW * SGT]
    if (nanwide) {
        if (one_each) {
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (tid + 64 * WPB * k < nsl * SGT) dl[tid + 64 * WPB * k] = dv[k];
        } else {
            const float *del = (cloud ? del2 : del1) + (size_t)b * n;
            for (int i = tid; i < nsl * SGT; i += blockDim.x) {
                const int sp = sg0 * SGT + i;
                dl[i] = sp < n ? del[psorted ? sp : idx[sp]] : 0.0f;
            }
        }
        __syncthreads();
    }
    // ---- stage the slice: records (padded rows) and tree nodes, slacks folded in
    //   (P0, thr2) -> (P0, c): c = -(thr2 - 2e-4 + slack), slightly widened; pad records never pass
    //   (centre, Rs) -> (centre, (Rs + se)^2 rounded up); an empty node (NaN radius) -> -inf: fails by its sign
    auto stage_rec = [&](int i, float4 r, float d) {
        float tp = r.w - RRL_EPS;
        if (nanwide) {  // also a candidate when point 1 or 2 could see a negative argument: Q(P0) < (se + e01)^2
            const float reach = se + sqrtf(r.w) * 1.000001f + d;  // e01 <= thr + del <= sqrt(thr2) + del
            tp = fmaxf(tp, reach * reach * 1.000002f);
        }
        r.w = sg0 * SGT + i < n ? -(tp + 1.0e-6f * fabsf(tp) + s0) : INFINITY;
        rec_lds[(i >> 4) * ROWS + (i & 15)] = r;
    };
    auto stage_node = [&](int i, float4 nd) {
        float rt = nd.w + se;
        if (nanwide) {  // node j of supergroup sg covers the records [0] all 64, [1..4] 16 each, [5..12] 8 each
            const int sg = i / NODE, j = i - sg * NODE;
            const int o = j == 0 ? 0 : (j < 5 ? (j - 1) * GRP : (j - 5) * (GRP / 2)), c4 = j == 0 ? SGT / 4 : (j < 5 ? GRP / 4 : GRP / 8);
            const float4 *q = (const float4 *)(dl + sg * SGT + o);  // 32-byte aligned
            float m = 0.0f;
            for (int t = 0; t < c4; ++t) {
                const float4 v = q[t];
                m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
            rt += m;
        }
        const float w = rt * rt * 1.0000003f;
        nd.w = w == w ? w : -INFINITY;
        node_lds[i] = nd;
    };
    if (one_each) {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (tid + 64 * WPB * k < nsl * SGT) stage_rec(tid + 64 * WPB * k, rec0[k], nanwide ? dv[k] : 0.0f);
        if (tid < nsl * NODE) stage_node(tid, nd0);
    } else {
        for (int i = tid; i < nsl * SGT; i += blockDim.x) stage_rec(i, p0s[(size_t)sg0 * SGT + i], nanwide ? dl[i] : 0.0f);
        for (int i = tid; i < nsl * NODE; i += blockDim.x) stage_node(i, tree[(size_t)sg0 * NODE + i]);
    }
    __syncthreads();  // the one barrier of the (usual) prologue
#ifdef CULL_STOP_STAGE
    return;
#endif
    if (!has_lines) return;  // uniform per wavefront
    const unsigned long long wall_staged = COUNT ? wall_clock64() : 0ull;
    unsigned long long wall_a = 0ull, wall_pa = 0ull, wall_pc = 0ull;

    WaveCtx ctx;
#if CULL_REGLINES
#pragma unroll
    for (int q = 0; q < 6; ++q) { ctx.v0[q] = v0[q]; ctx.v1[q] = v1[q]; }
#else
    ctx.lr = lr;
#endif
    ctx.recs = rec_lds;
    ctx.nodes = node_lds;
    ctx.qa = qa_lds[wave];
    ctx.qc = qc_lds[wave];
    ctx.cands = cands_lds[wave];
    ctx.idx = idx;
    ctx.ptri = ptri;
    ctx.psorted = psorted;
    ctx.cnt = cnt;
    ctx.hit = hit;
    ctx.lbase = lw0;
    ctx.na = ctx.nc = ctx.ncand = 0;
    ctx.lane = lane;
    ctx.tb = ctx.tc = ctx.td = ctx.tcand = 0;
    ctx.status = status;
    ctx.pos0 = sg0 * SGT;
    const v2f ux = {v0[0], v1[0]}, uy = {v0[1], v1[1]}, uz = {v0[2], v1[2]};
    const v2f ox = {v0[3], v1[3]}, oy = {v0[4], v1[4]}, oz = {v0[5], v1[5]};
    unsigned long long ta = 0;

    if (fallback) {  // rare: kept out of line so that its registers do not count against the culled walk
        const int f0 = sg0 * SGT, f1 = min(n, f0 + nsl * SGT);  // real records sit at sorted positions [0, n)
        strict_slice(ptri, idx, ctx.psorted, f0, f1, ux, uy, uz, ox, oy, oz, l0, l1, L, cnt, hit, status);
        if (lane == 0) atomicAdd(&status[1], 1);  // always on: wavefronts that left the culled path
        fb_pairs = (unsigned long long)(f1 - f0) * (unsigned long long)min(LPW, L - lw0);
    } else {
    ta = (unsigned long long)nsl * (unsigned long long)min(LPW, L - lw0);

    unsigned long long stamps[3] = {0ull, 0ull, 0ull};
    cull_walk<COUNT>(ctx, node_lds, nsl, lane, live0, live1, ux, uy, uz, ox, oy, oz, stamps);
    wall_a = stamps[0]; wall_pa = stamps[1]; wall_pc = stamps[2];
    }
    if constexpr (COUNT) {
        if (lane == 0 && crow && has_lines) {
            crow[0] = ta;
            crow[1] = ctx.tb; crow[2] = ctx.tc; crow[3] = ctx.td; crow[4] = ctx.tcand;
            crow[5] = 1ull;
            crow[6] = fallback ? 1ull : 0ull;
            crow[7] = fb_pairs;
            crow[8] = wall0; crow[9] = wall_clock64();
            // phase stamps of the culled walk: staged (after the staging barrier), level A done, the final drains
            // of levels B and D done (the remainder up to [9] is the last candidate flush)
            crow[10] = wall_staged; crow[11] = wall_a; crow[12] = wall_pa; crow[13] = wall_pa; crow[14] = wall_pc;
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(6, 8))) void cull_scan_kernel(
    const float *__restrict__ ptri1, const float *__restrict__ ptri2, const float4 *__restrict__ p0s1,
    const float4 *__restrict__ p0s2, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
    const float4 *__restrict__ tree1, const float4 *__restrict__ tree2, const float *__restrict__ line,
    int32_t *__restrict__ count1, int32_t *__restrict__ hit1, int32_t *__restrict__ count2,
    int32_t *__restrict__ hit2, int32_t *__restrict__ status, uint32_t *pmax,
    const float *__restrict__ del1, const float *__restrict__ del2, const float2 *__restrict__ lmax,
    const float *__restrict__ apart, const float *__restrict__ aflag, int nblk_apart, int B,
    int N, int M, int L, int spw, unsigned long long *__restrict__ counters, long long counter_rows, int Bt) {
    __shared__ CullLds lds_;
    cull_scan_body<COUNT>(lds_, ptri1, ptri2, p0s1, p0s2, idx1, idx2, tree1, tree2, line, count1, hit1, count2, hit2, status,
                          pmax, del1, del2, lmax, apart, aflag, nblk_apart, B, N, M, L, spw, counters, counter_rows,
                          (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, Bt);
}

// ---------------------------------------------------------------------------------------
// The culled scan AND the Chamfer walk of the same evaluation in ONE launch (round 4b; rrl_ws.h RrlChamRider, rrl_demo_epoch).
// The walk (rrl_chamfer_from_loss: nearest neighbours between the moved source's and the target's first points through
// the sorted records and sphere trees the records launch just left) depends on that launch only -- not on the scan -- but
// consecutive launches of a stream never overlap on this stack (hipExtAnyOrderLaunch is ignored on gfx9, event edges
// between streams cost tens of microseconds: profiles/r04_experiments.txt 4, 12), so the only way to run two independent
// 512-lane kernels side by side is to issue them as one grid: workgroups [0, ncham) walk (the long ones start first),
// the others scan.  Same bodies, same results as the two launches; the LDS of the two is overlaid.  The walk takes
// max |P|^2 of its target from the partial rows (pm1 == NULL): PMAX is being rebuilt by this launch's scan.
// ---------------------------------------------------------------------------------------
struct CullKArgs {
    const float *ptri1, *ptri2;
    const float4 *p0s1, *p0s2;
    const int32_t *idx1, *idx2;
    const float4 *tree1, *tree2;
    const float *line;
    int32_t *count1, *hit1, *count2, *hit2, *status;
    uint32_t *pmax;
    const float *del1, *del2;
    const float2 *lmax;
    const float *apart, *aflag;
    const float *aflag_tar;  // the partial rows of the workspace that holds cloud 2 (== aflag unless the target is carried over)
    int nblk_apart, B, N, M, L, spw, gx, gy, Bt;
};
struct ChamKArgs {
    unsigned long long *best_x, *best_y;
    double *partial, *gpart;
    float *value;
    double denom;
    ChamTick tick;
    int gx, gy;
};
static_assert(NWV == WPB, "the walk and the scan share one launch: the same 512-lane workgroups");

__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 8))) void cull_scan_chamfer_kernel(
    const CullKArgs a, const ChamKArgs c) {
    __shared__ union {
        CullLds scan;
        ChamLds walk;
    } lds_;
    const int ncham = c.gx * c.gy, lin = (int)blockIdx.x;
    if (lin < ncham) {  // uniform per workgroup
        chamfer_tree_body<false, true>(lds_.walk, a.p0s1, a.p0s2, a.tree1, a.tree2, a.aflag, a.nblk_apart, c.best_x, c.best_y,
                                       c.partial, a.B, a.N, a.M, nullptr, 0, a.idx1, a.idx2, nullptr, nullptr, c.tick, c.gpart,
                                       c.value, c.denom, lin % c.gx, lin / c.gx, c.gx, c.gy, a.aflag_tar);
        return;
    }
    const int l2 = lin - ncham, bx = l2 % a.gx, r = l2 / a.gx;
    cull_scan_body<false>(lds_.scan, a.ptri1, a.ptri2, a.p0s1, a.p0s2, a.idx1, a.idx2, a.tree1, a.tree2, a.line, a.count1,
                          a.hit1, a.count2, a.hit2, a.status, a.pmax, a.del1, a.del2, a.lmax, a.apart, a.aflag, a.nblk_apart,
                          a.B, a.N, a.M, a.L, a.spw, nullptr, 0, bx, r % a.gy, r / a.gy, a.gx, a.gy, a.Bt);
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
    a.del1 = w.f32(ws, RRL_WS_DEL1);
    a.del2 = w.f32(ws, RRL_WS_DEL2);
    a.zwords = nullptr; a.nzwords = 0;
    a.line = line;
    a.lmax = line && L > 0 ? (float2 *)w.f32(ws, RRL_WS_LMAX) : nullptr;
    a.L = L;
    a.B = B; a.N = N; a.M = M;
    a.transpose_r = xf ? xf->transpose_r : 0;
    a.Bt = o.problems;
    const int nall = N > M ? N : M;  // APART is laid out for the larger cloud
    a.nblk = (nall + REC_BLK - 1) / REC_BLK;
    a.nchunk = chunked ? (nmax + 4095) / 4096 : 1;
    a.nblk_tri = (nmax + REC_BLK - 1) / REC_BLK;
    if (o.prepared()) {  // the order is known: ONE launch (records at their sorted positions + tree refit), no sort
        a.z2 = nullptr; a.z2_vec4 = 0;
        a.nblk_tri = (int)(((size_t)(nmax + SGT - 1) / SGT * SGT + REC_BLK - 1) / REC_BLK);
        hipLaunchKernelGGL(tri_records_sorted_kernel, dim3((unsigned)(a.nblk_tri + (a.lmax ? LMAX_CHUNKS : 0)), (unsigned)B, (unsigned)clouds),
                           dim3(REC_BLK), 0, s, a, o.order1, o.order2);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    }
    hipLaunchKernelGGL(tri_records_kernel, dim3((unsigned)(a.nblk_tri + (a.lmax ? LMAX_CHUNKS : 0)), (unsigned)B, (unsigned)clouds),
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

int rrl_launch_cull_scan(const float *line, void *ws, const WsLayout &w, int B, int N, int M, int L,
                         int clouds, int lmax_ready, const RrlCall &o, hipStream_t s) {
    // a workgroup = (cloud and sample, tile of <= WPB x 128 lines, slice of spw supergroups).  With
    // few lines or small clouds the slices get thinner, so that the launch still has ~1000
    // workgroups for the 256 CUs (measured with tools/geom_sweep.sh: thinner slices cost little,
    // fewer wavefronts per workgroup cost more -- they are only reduced as a last resort)
    const int nmax = clouds == 2 && M > N ? M : N;
    const int nsgmax = (nmax + SGT - 1) / SGT;
    const int lw = (L + LPW - 1) / LPW;  // wavefronts' worth of lines
    int waves = lw < WPB ? lw : WPB, spw = SPW;
    auto wgs = [&]() { return (long)clouds * B * ((lw + waves - 1) / waves) * ((nsgmax + spw - 1) / spw); };
    while (wgs() < 768 && spw > 1) spw >>= 1;
    // (a riding Chamfer walk needs the scan's full 512-lane workgroups -- and brings workgroups of its own: no thinning then)
    const bool may_ride = o.rider && !o.counters && (clouds == 2 || o.tar_ws) && lw >= WPB && B <= 32767 && N > 0 && M > 0;
    while (!may_ride && wgs() < 256 && waves > 2) waves >>= 1;
    if (const char *e = getenv("RRL_CULL_GEOM")) {  // experiments: "waves,spw"
        int w_ = 0, s_ = 0;
        if (sscanf(e, "%d,%d", &w_, &s_) == 2 && w_ >= 1 && w_ <= WPB && s_ >= 1 && s_ <= SPW) { waves = w_ < lw ? w_ : lw; spw = s_; }
    }
    const int tiles = (lw + waves - 1) / waves, slices = (nsgmax + spw - 1) / spw;
    if (!lmax_ready)  // the triangles were prepared without the lines: their partial maxima first (a tiny launch)
        hipLaunchKernelGGL(line_max_kernel, dim3(LMAX_CHUNKS, (unsigned)B), dim3(REC_BLK), 0, s, line, L,
                           (float2 *)w.f32(ws, RRL_WS_LMAX), o.problems);
    // (A PERSISTENT variant -- as many workgroups as fit on the chip, each keeping one line tile staged and pulling
    // (cloud, slice) items from per-tile work queues, the next slice's records prefetched during the walk -- was built
    // and measured in round 3: exact, but 40.7 us against 30.4 at C2 and 29.0 against 13.8 at the demo's shape.  A slot
    // is held for the SLOWEST of a workgroup's eight wavefronts either way (16.5 us per item against a mean wavefront
    // lifetime of 12.4), so queueing the items removed no waiting, and the item barriers added some;
    // profiles/r03_scan_experiments.txt.)
    const int zslices = slices;
#define RRL_CULL_LAUNCH(COUNT)                                                                              \
    hipLaunchKernelGGL(cull_scan_kernel<COUNT>, dim3((unsigned)(clouds * B), (unsigned)tiles, (unsigned)zslices),   \
                       dim3(64 * waves), 0, s, w.f32(ws, RRL_WS_PTRI1), w.f32(ws, RRL_WS_PTRI2),             \
                       (const float4 *)w.f32(ws, RRL_WS_P0S1), (const float4 *)w.f32(ws, RRL_WS_P0S2),       \
                       w.i32(ws, RRL_WS_IDX1), w.i32(ws, RRL_WS_IDX2), (const float4 *)w.f32(ws, RRL_WS_GRP1), \
                       (const float4 *)w.f32(ws, RRL_WS_GRP2), line, w.i32(ws, RRL_WS_COUNT1),               \
                       w.i32(ws, RRL_WS_HIT1), w.i32(ws, RRL_WS_COUNT2), w.i32(ws, RRL_WS_HIT2),             \
                       w.i32(ws, RRL_WS_STATUS), (uint32_t *)w.i32(ws, RRL_WS_PMAX),                         \
                       w.f32(ws, RRL_WS_DEL1), w.f32(ws, RRL_WS_DEL2), (const float2 *)w.f32(ws, RRL_WS_LMAX),   \
                       apart, w.f32(ws, RRL_WS_APART), nblk_apart, B, N, M, L, spw,                          \
                       o.counters, o.counter_rows, o.problems)
    const float *apart = o.prepared() ? w.f32(ws, RRL_WS_APART) : nullptr;  // prepared build: PMAX comes from the partial rows
    const int nblk_apart = ((N > M ? N : M) + REC_BLK - 1) / REC_BLK;
    if (may_ride && waves == WPB) {
        // the evaluation's Chamfer walk rides along (cull_scan_chamfer_kernel): ONE launch for both.  A carried-over target
        // (clouds == 1: only the source is scanned here) is walked in the workspace that holds its records.
        const void *tws = clouds == 2 ? ws : o.tar_ws;
        const ChamLayout C(B, N, M);
        if (o.rider->ws && o.rider->ws_bytes >= C.total && o.rider->best_x && o.rider->best_y && o.rider->value) {
            CullKArgs a;
            a.ptri1 = w.f32(ws, RRL_WS_PTRI1); a.ptri2 = w.f32(tws, RRL_WS_PTRI2);
            a.p0s1 = (const float4 *)w.f32(ws, RRL_WS_P0S1); a.p0s2 = (const float4 *)w.f32(tws, RRL_WS_P0S2);
            a.idx1 = w.i32(ws, RRL_WS_IDX1); a.idx2 = w.i32(tws, RRL_WS_IDX2);
            a.tree1 = (const float4 *)w.f32(ws, RRL_WS_GRP1); a.tree2 = (const float4 *)w.f32(tws, RRL_WS_GRP2);
            a.line = line;
            a.count1 = w.i32(ws, RRL_WS_COUNT1); a.hit1 = w.i32(ws, RRL_WS_HIT1);
            a.count2 = w.i32(ws, RRL_WS_COUNT2); a.hit2 = w.i32(ws, RRL_WS_HIT2);
            a.status = w.i32(ws, RRL_WS_STATUS); a.pmax = (uint32_t *)w.i32(ws, RRL_WS_PMAX);
            a.del1 = w.f32(ws, RRL_WS_DEL1); a.del2 = w.f32(ws, RRL_WS_DEL2);
            a.lmax = (const float2 *)w.f32(ws, RRL_WS_LMAX);
            a.apart = apart; a.aflag = w.f32(ws, RRL_WS_APART); a.aflag_tar = w.f32(tws, RRL_WS_APART);
            a.nblk_apart = nblk_apart; a.B = B; a.N = N; a.M = M; a.L = L; a.spw = spw;
            a.gx = clouds * B; a.gy = tiles; a.Bt = o.problems;
            ChamKArgs c;
            char *cw = (char *)o.rider->ws;
            c.best_x = (unsigned long long *)o.rider->best_x; c.best_y = (unsigned long long *)o.rider->best_y;
            c.partial = (double *)(cw + C.partial); c.gpart = (double *)(cw + C.gpart);
            c.value = o.rider->value;
            c.denom = (double)B * (double)(N + M);
            uint32_t *mctl = w.u32(ws, RRL_WS_MCTL);  // arrival counters of the walk's mean (rrl_chamfer_from_loss_ex)
            c.tick = ChamTick{mctl + 32, mctl + 30, 64, 1};
            c.gx = 2 * B; c.gy = ((N > M ? N : M) + SGT - 1) / SGT;  // patches of the larger cloud (either direction)
            const unsigned nwg = (unsigned)(c.gx * c.gy) + (unsigned)(a.gx * a.gy * zslices);
            hipLaunchKernelGGL(cull_scan_chamfer_kernel, dim3(nwg), dim3(64 * WPB), 0, s, a, c);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) o.rider->done = 1;
            return e == hipSuccess ? 0 : (int)e;
        }
    }
    if (o.counters) RRL_CULL_LAUNCH(true);
    else RRL_CULL_LAUNCH(false);
#undef RRL_CULL_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int rrl_sort_capacity(void) { return SORT_CAP; }
