// rrl_chamfer.hip -- K7 Chamfer monitor on the spatial structures of the scan
// (code/loss.py:38-52 compute_sqrdis_map_2, :236-252 chamfer_dist).
//
// The brute-force kernel (rrl_geom.hip chamfer_nn_kernel) evaluates all N x M pairs in both
// directions: 2.7e8 pair evaluations at B=8, 4096 x 4096, 79 us + 19 us for the mean, at the issue
// limit of VALU ops with a scalar operand.  Here both clouds are put in grid-cell (Hilbert) order under
// the sphere tree of rrl_tree.h -- the sort kernels of rrl_cull.hip, fed with (x, y, z, index) records
// (built inside the sort kernel for clouds <= 4096 points) -- and a workgroup = one PATCH of 64
// consecutive sorted queries (one supergroup of the query cloud) walks the target's tree, NNW x SPLIT
// wavefronts sharing the patch (wavefront (k, c) owns leaf k = group k of 16 records of every target
// supergroup j with j % SPLIT == c):
//   seed      the target supergroup whose centre is nearest to the patch centre is evaluated first for all
//             64 queries (minima folded into LDS): a good upper bound bd per query;
//   prune 1   lane-parallel over the target supergroups: lane j tests this wavefront's leaf of supergroup
//             j against each of the patch's four query-group spheres with that group's largest bound;
//   stage     the surviving leaves' records go to LDS, CHK leaves per round (coalesced loads);
//   prune 2   per staged leaf every lane tests ITS query against the leaf sphere with its CURRENT bound and
//             pushes a (query, leaf) entry into the wavefront's LDS queue when it needs the leaf;
//   passes    whenever 64 entries wait every lane pops one and evaluates "its" 16 records for "its" query
//             (11 VALU ops per pair, keys reduced as a tree), folding the key into the query's minimum with
//             an LDS atomicMin -- all lanes busy on pairs that are actually needed (3.7 % of the dense pairs
//             on the bench clouds against 10.5 % when a leaf is evaluated for the whole wavefront).
// Exactness: a distance is evaluated with the reference's arithmetic ((dx^2 + dy^2) + dz^2, no FMA);
// every pruning test is |q - c|^2 (1 - 1e-4) > (sqrt(bd) (1 + 1e-5) + R)^2 with the tree's conservative
// radii, so a leaf is skipped only if every point in it is STRICTLY farther than the lane's current
// best; the running minimum is the u64 key (distance bits << 32 | ORIGINAL target index), so ties
// resolve to the smallest index in any visiting order == torch.min's first occurrence.  Keys are
// bit-identical to the brute-force kernel's (tests/test_gpu_parity.py).  NaN: a NaN coordinate in the
// target cloud makes every minimum of that sample NaN, a NaN query its own minimum (torch semantics:
// min propagates NaN) -- decided from per-workgroup flags of the records / sort kernel, at no cost in the
// inner loop.  The mean: per-patch double partials, summed in a fixed order by a tiny second launch.
//
// What bounds it (profiles/r02_chamfer_notes.txt): every wavefront starts within 0.5 us and lives ~10 us
// (a chain of dependent memory / LDS round trips and short serial steps), but the kernel lasts as long as
// its SLOWEST patch -- on misaligned clouds the queries far from the target see most of the tree (mean
// lifetime 11 us, last end 25.7 us with 4 wavefronts per patch).  SPLIT = 2 halves that tail: 26.5 -> 22 us.
// Also measured: a release fence per workgroup for a "last workgroup sums" hand-over +30 us (1024
// write-backs of an XCD's L2), relaxed tickets on one address +5 us (same-address device atomics
// serialise at ~12 ns); patches of one (sample, direction) spread over all XCDs +10 us; evaluating a leaf
// for the whole wavefront with the targets broadcast through LDS, v_readlane or the scalar cache: 25-27 us
// each; instrumenting with same-address atomics distorts every memory latency in the kernel (use the
// per-wavefront rows of rrl_chamfer_counters).
#include "rrl_tree.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

int rrl_launch_cloud_sort(const float *raw1, const float *raw2, float4 *crec1, float4 *crec2, float *apart, int nblk,
                          float4 *p0s1, float4 *p0s2, int32_t *idx1, int32_t *idx2, float4 *grp1, float4 *grp2,
                          uint32_t *pmax, unsigned *histg, uint32_t *zwords, int nzwords, int B, int N, int M, hipStream_t s);
int rrl_sort_capacity(void);

struct ChamLayout {
    size_t crec1, crec2, p0s1, p0s2, idx1, idx2, grp1, grp2, apart, pmax, histg, partial, gpart, ctrl, ctrl_bytes, total;
    int nblk;
    __host__ ChamLayout(int B, int N, int M) {
        const size_t b = (size_t)B, n = (size_t)N, m = (size_t)M;
        const size_t nmax = n > m ? n : m;
        nblk = (int)((nmax + 255) / 256);
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
        crec1 = take(16 * b * ((n + 15) / 16) * 16);
        crec2 = take(16 * b * ((m + 15) / 16) * 16);
        p0s1 = take(16 * b * ((n + 63) / 64) * 64);
        p0s2 = take(16 * b * ((m + 63) / 64) * 64);
        idx1 = take(4 * b * ((n + 63) / 64) * 64);
        idx2 = take(4 * b * ((m + 63) / 64) * 64);
        grp1 = take(16 * b * ((n + 63) / 64) * NODE);
        grp2 = take(16 * b * ((m + 63) / 64) * NODE);
        apart = take(4 * 2 * b * 8 * (size_t)nblk);
        pmax = take(4 * 2 * b);
        partial = take(8 * (2 * b * ((nmax + 63) / 64) + 1));  // one per workgroup of the NN launch
        gpart = take(8 * 2 * b);  // per (sample, direction) sum of its patches' partials
        histg = take(nmax > 4096 ? 4 * 2 * b * 2 * SORT_CELLS : 16);  // cleared by pts_records_kernel
        ctrl_bytes = 4 * (32 + 64 * b);  // arrival counters of the mean (ChamTick): zero before the walk -- inside the
        ctrl = take(ctrl_bytes);         //   range pts_records_kernel clears; the small-cloud sort kernel clears them itself
        total = o;
    }
};

extern "C" size_t rrl_chamfer_workspace_bytes(int B, int N, int M) {
    if (B < 0 || N < 0 || M < 0) return 0;
    return ChamLayout(B, N, M).total;
}

// (x, y, z, original index) records in original order + per-workgroup AABB / max |P|^2 partials (the
// sort kernels' inputs; slot 7 of a partial row = "this workgroup saw a NaN coordinate"), and the
// clearing of the wide sort's histogram.
__global__ __launch_bounds__(256) void pts_records_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          float4 *__restrict__ crec1, float4 *__restrict__ crec2,
                                                          float *__restrict__ apart, uint4 *__restrict__ zero,
                                                          size_t zero_vec4, int B, int N, int M, int nblk) {
    __shared__ float red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.z, b = blockIdx.y;
    {
        const size_t nthr = (size_t)gridDim.x * gridDim.y * gridDim.z * 256;
        const size_t me = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + tid;
        for (size_t i = me; i < zero_vec4; i += nthr) zero[i] = make_uint4(0, 0, 0, 0);
    }
    const int n = cloud ? M : N;
    if ((int)blockIdx.x * 256 >= n) return;  // uniform
    const int f = blockIdx.x * 256 + tid;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, p2 = 0.0f;
    bool bad = false;
    if (f < n) {
        const float *p = (cloud ? y : x) + ((size_t)b * n + f) * 3;
        const float c0 = p[0], c1 = p[1], c2 = p[2];
        const int ng = (n + GRP - 1) / GRP;
        ((cloud ? crec2 : crec1) + (size_t)b * ng * GRP)[f] = make_float4(c0, c1, c2, __int_as_float(f));
        mn[0] = mx[0] = c0; mn[1] = mx[1] = c1; mn[2] = mx[2] = c2;
        p2 = c0 * c0 + c1 * c1 + c2 * c2;
        bad = (c0 != c0) || (c1 != c1) || (c2 != c2);
        if (!(p2 <= 3.0e38f)) p2 = INFINITY;
    }
    const float anybad = __any(bad) ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    p2 = wave_max(p2);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave][c] = mn[c]; red[wave][3 + c] = mx[c]; }
        red[wave][6] = p2;
        red[wave][7] = anybad;  // a NaN coordinate among this workgroup's points (read by the NN kernel)
    }
    __syncthreads();
    if (tid < 8) {
        float r = red[0][tid];
        for (int w = 1; w < 4; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        apart[(((size_t)cloud * B + b) * nblk + blockIdx.x) * 8 + tid] = r;
    }
}

// Prepared point clouds (round 4; include/rrl.h rrl_chamfer_tree_fwd_ex, order_x / order_y from rrl_cloud_order on the
// same clouds in any rigid pose): the sort leaves the call.  One lane per SORTED position gathers its point, writes the
// (x, y, z, original index) record there and the wavefront -- one supergroup -- refits its 13 tree nodes (wave_tree,
// rrl_tree.h: the nodes tri_sort_kernel<4, true> derives from the same sorted records); slot 7 of the workgroup's
// partial row = "saw a NaN coordinate" as pts_records_kernel leaves it; the walk's arrival counters are cleared here.
__global__ __launch_bounds__(256) void pts_records_sorted_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const int32_t *__restrict__ order1, const int32_t *__restrict__ order2,
                                                                 float4 *__restrict__ p0s1, float4 *__restrict__ p0s2,
                                                                 int32_t *__restrict__ idx1, int32_t *__restrict__ idx2,
                                                                 float4 *__restrict__ grp1, float4 *__restrict__ grp2,
                                                                 float *__restrict__ apart, uint32_t *__restrict__ zwords, int nzwords,
                                                                 int B, int N, int M, int nblk) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cloud = blockIdx.z, b = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = tid; i < nzwords; i += 256) zwords[i] = 0u;
    const int n = cloud ? M : N;
    const int npad = (n + SGT - 1) / SGT * SGT;
    if ((int)blockIdx.x * 256 >= npad) return;  // uniform
    const int s_ = blockIdx.x * 256 + tid;
    const bool valid = s_ < n;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    int f = 0;
    if (valid) {
        f = (cloud ? order2 : order1)[(size_t)b * npad + s_];
        f = min(max(f, 0), n - 1);  // memory safety only: the order must be a permutation of [0, n)
        const float *p = (cloud ? y : x) + ((size_t)b * n + f) * 3;
        c0 = p[0]; c1 = p[1]; c2 = p[2];
    }
    if (s_ - lane < npad) {  // wave-uniform: this wavefront holds a supergroup
        (cloud ? p0s2 : p0s1)[(size_t)b * npad + s_] = valid ? make_float4(c0, c1, c2, __int_as_float(f)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        (cloud ? idx2 : idx1)[(size_t)b * npad + s_] = f;
        wave_tree(c0, c1, c2, __int_as_float(f), valid, lane, (cloud ? grp2 : grp1) + ((size_t)b * (npad / SGT) + (s_ - lane) / SGT) * NODE);
    }
    const bool bad = valid && ((c0 != c0) || (c1 != c1) || (c2 != c2));
    const float anybad = __any(bad) ? 1.0f : 0.0f;
    if (lane == 0) red[wave] = anybad;
    __syncthreads();
    if (tid == 0 && (int)blockIdx.x < nblk)
        apart[(((size_t)cloud * B + b) * nblk + blockIdx.x) * 8 + 7] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

#define LB_SCALE 0.9999f  // squared distances to node centres are shrunk: evaluation error of the bound (~1e-6)

struct NNWave {
    const float4 *T;      // sorted target records of the sample
    const float4 *tree;   // target tree
    int nt;               // real target records
    float qx, qy, qz;
    bool valid;
    unsigned long long best;  // (distance bits << 32) | original target index
};

__device__ __forceinline__ unsigned long long kmin(unsigned long long a, unsigned long long b) { return b < a ? b : a; }

#ifndef NNW
#define NNW 4  // wavefronts per patch: 4 (leaf = group of 16) or 8 (leaf = half of 8)
#endif
#ifndef SPLIT
#define SPLIT 2  // the candidate supergroups of a patch are dealt to SPLIT sets of NNW wavefronts (j % SPLIT): the kernel's
                 // duration is the time of its SLOWEST patch (misaligned clouds: queries far from the target see
                 // most of the tree), and both the per-lane tests and the entry passes of a patch split this way
#endif
#define NWV (NNW * SPLIT)  // wavefronts per workgroup
#define LEAF (SGT / NNW)
static_assert(NNW == 8 || NNW == 4, "a wavefront owns a half (8 records) or a group (16) of every supergroup");
#define LEAF_NODE(k) (NNW == 8 ? 5 + (k) : 1 + (k))
#ifndef TB
#define TB 1  // per-lane leaf tests per loop iteration (2, 4: no faster -- the kernel is VALU-issue bound, see notes)
#endif

// ---- per-lane leaf evaluation (round 2, v9) -------------------------------------------------------
// Evaluating a leaf for the whole wavefront whenever ANY of its 64 queries needs it wastes ~2/3 of the pair
// evaluations (a leaf is typically needed by ~14 of the 64 lanes).  Instead, like the culled scan's levels:
// a wavefront stages its candidate leaves in LDS (CHK at a time, coalesced loads, one memory latency), every
// lane tests ITS query against each staged leaf and pushes (query, leaf slot) ENTRIES into the wavefront's LDS
// queue (ballot + rank); whenever 64 entries wait, every lane pops one and evaluates "its" 16 records
// for "its" query -- all lanes busy on pairs that are actually needed -- and folds the key into the
// query's minimum with an LDS atomicMin on the u64 key (several lanes may hold the same query).
#ifndef CHK
#define CHK 12                 // candidate leaves staged per round and wavefront (12: 32 KiB of LDS per 8-wavefront workgroup -> 4 per CU)
#endif
#define LROW (LEAF + 1)        // float4 per staged leaf row: +1 of padding spreads the rows over the banks
#define QCAP (64 + 63)         // entries: < 64 left-overs + one push round

struct NNShared {
    float4 *q;                   // [64] the patch's queries (xyz, original index)
    unsigned long long *best;    // [64] u64 keys, LDS atomicMin
    float4 *rec;                 // this wavefront's staged leaves [CHK][LROW]
    int *pos;                    // this wavefront's staged leaf positions [CHK] (sorted position of record 0)
    unsigned *queue;             // this wavefront's entries: query << 8 | slot
};

// pops up to 64 entries [base, base + take) and evaluates them, one per lane
__device__ __forceinline__ void eval_entries(const NNShared &sh, int base, int take, int nt, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < take) {
        const unsigned e = sh.queue[base + lane];
        const int q = (int)(e >> 8), slot = (int)(e & 255u);
        const float4 qv = sh.q[q];
        const int cnt = min(LEAF, nt - sh.pos[slot]);
        const float4 *row = sh.rec + slot * LROW;
        unsigned long long key[LEAF];
#pragma unroll
        for (int t = 0; t < LEAF; ++t) {
            const float4 r = row[t];
            // code/loss.py:51: sum((x - y)**2, -1); (a0 + a1) + a2, no FMA
            const float dx = qv.x - r.x, dy = qv.y - r.y, dz = qv.z - r.z;
            float s = dx * dx;
            s = s + dy * dy;
            s = s + dz * dz;
            key[t] = t < cnt ? (((unsigned long long)__float_as_uint(s) << 32) | (unsigned)__float_as_int(r.w)) : ~0ull;
        }
#pragma unroll
        for (int o = LEAF / 2; o > 0; o >>= 1)
#pragma unroll
            for (int t = 0; t < o; ++t) key[t] = kmin(key[t], key[t + o]);
        atomicMin(&sh.best[q], key[0]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Arrival counters of the in-launch mean (chamfer_tree_kernel's ending): one per (sample, direction) group at
// group + b * stride_b + dir * stride_d, one for the groups.  All zero before the launch; they rewind themselves.
struct ChamTick {
    uint32_t *top, *group;
    int stride_b, stride_d;
};

// COUNT: executed-work counters (rrl_chamfer_counters): [0] patch-level leaf tests (lane-parallel),
// [1] per-lane leaf sphere tests (wave x leaf), [2] (query, leaf) entries evaluated, [3] (query, target)
// pairs evaluated, [4] wavefronts.
// IDX = true (rrl_chamfer_from_loss): the records are the LOSS workspace's sorted (P0, thr2) records; the
// original indices come from its IDX arrays and "a NaN coordinate in the cloud" from its PMAX entries
// (pm1 = &PMAX[0] of the source's workspace, pm2 = &PMAX[B] of the target's).
template <bool COUNT, bool IDX>
__global__ __launch_bounds__(64 * NWV) void chamfer_tree_kernel(
    const float4 *__restrict__ p0s1, const float4 *__restrict__ p0s2, const float4 *__restrict__ grp1,
    const float4 *__restrict__ grp2, const float *__restrict__ apart, int nblk,
    unsigned long long *__restrict__ best_x, unsigned long long *__restrict__ best_y,
    double *__restrict__ partial, int B, int N, int M, unsigned long long *__restrict__ counters, long long counter_rows,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2, const uint32_t *__restrict__ pm1,
    const uint32_t *__restrict__ pm2, const ChamTick tk_, double *__restrict__ gpart, float *__restrict__ value, double denom) {
    __shared__ unsigned long long s_best[64];
    __shared__ __attribute__((aligned(16))) float4 s_q[64];
    __shared__ __attribute__((aligned(16))) float4 s_rec[NWV][CHK * LROW];
    __shared__ int s_pos[NWV][CHK];
    __shared__ unsigned s_queue[NWV][QCAP + 1];
    __shared__ double red[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wavefront of the workgroup
    const int wave = wv % NNW, cls = wv / NNW;                  // its leaf within a supergroup, its supergroup class
    // XCD-aware: workgroups go to the 8 XCDs round-robin by linear id; with (sample, direction) on the fast
    // index every XCD's L2 holds the records and trees of 2 B / 8 of the (cloud pair, direction) combinations only
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int nq = dir ? M : N, nt = dir ? N : M;
    const int nsgq = (nq + SGT - 1) / SGT, nsgt = (nt + SGT - 1) / SGT;
    const int sgq = (int)blockIdx.y;
    double mine = 0.0;
    unsigned c_sg = 0, c_gt = 0, c_ge = 0, c_pairs = 0;
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_stage = 0, t_test = 0, t_pass = 0;  // COUNT: phase clocks
#define RRL_NOW() (COUNT ? (long long)__builtin_readcyclecounter() : 0ll)
    tk[0] = RRL_NOW();
    const long long wall0 = COUNT ? (long long)wall_clock64() : 0ll;
    if (sgq < nsgq) {  // workgroup-uniform
        const float4 *Q = (dir ? p0s2 : p0s1) + (size_t)b * nsgq * SGT;
        const float4 *treeQ = (dir ? grp2 : grp1) + (size_t)b * nsgq * NODE;
        NNWave w;
        w.T = (dir ? p0s1 : p0s2) + (size_t)b * nsgt * SGT;
        w.tree = (dir ? grp1 : grp2) + (size_t)b * nsgt * NODE;
        w.nt = nt;
        const int qi = sgq * SGT + lane;
        w.valid = qi < nq;
        const float4 qr = Q[qi];  // pad records exist up to the supergroup boundary
        const int32_t *idxT = IDX ? (dir ? idx1 : idx2) + (size_t)b * nsgt * SGT : nullptr;
        const int qorig = IDX ? ((dir ? idx2 : idx1) + (size_t)b * nsgq * SGT)[qi] : __float_as_int(qr.w);
        // a staged target record carries its original index in .w
        auto target = [&](int pos) {
            float4 r = w.T[pos];
            if constexpr (IDX) r.w = __int_as_float(idxT[pos]);
            return r;
        };
        w.qx = qr.x; w.qy = qr.y; w.qz = qr.z;
        w.best = ~0ull;
        if (wv == 0) { s_q[lane] = qr; s_best[lane] = ~0ull; }
        const NNShared sh = {s_q, s_best, s_rec[wv], s_pos[wv], s_queue[wv]};
        kptr qn = (kptr)(uintptr_t)(treeQ + (size_t)sgq * NODE);
        const float cqx = qn[0], cqy = qn[1], cqz = qn[2];
        float qg[SGG][4];  // the patch's four query-group spheres (wave-uniform)
#pragma unroll
        for (int g = 0; g < SGG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) qg[g][c] = qn[4 * (1 + g) + c];
        // does the TARGET cloud hold a NaN coordinate (slot 7 of its AABB partial rows)?
        bool tnan = false;
        if (wv == 0) {
            if constexpr (IDX) {  // max |P|^2 of the target cloud: +inf when a coordinate is NaN (or overflows)
                tnan = !(__uint_as_float((dir ? pm1 : pm2)[b]) <= 3.0e38f);
            } else {
                const int ct = dir ? 0 : 1, nb = (nt + 255) / 256;
                const float *ap = apart + ((size_t)ct * B + b) * nblk * 8;
                for (int j = lane; j < nb; j += 64) tnan |= ap[j * 8 + 7] != 0.0f;
                tnan = __any(tnan);
            }
        }

        tk[1] = RRL_NOW();
        // ---- seed: the target supergroup nearest to the patch centre; wavefront k evaluates its leaf k for
        //      all 64 queries (every query needs a bound; the records arrive through the scalar cache)
        float dmin = INFINITY;
        int jmin = 0;
        for (int j = lane; j < nsgt; j += 64) {
            const float4 c = w.tree[(size_t)j * NODE];
            const float dx = cqx - c.x, dy = cqy - c.y, dz = cqz - c.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < dmin) { dmin = d2; jmin = j; }
        }
        const float wmin = wave_min(dmin);
        const unsigned long long who = __ballot(dmin == wmin);
        const int seed = who ? __builtin_amdgcn_readlane(jmin, __ffsll((long long)who) - 1) : 0;  // all NaN: 0
        tk[2] = RRL_NOW();
        __syncthreads();  // s_q / s_best initialised by wavefront 0
        {   // every query needs a bound: 64 entries (lane, slot 0) on this wavefront's leaf of the seed supergroup
            const int cnt = min(LEAF, nt - seed * SGT - wave * LEAF);
            if (cnt > 0 && cls == 0) {
                if (lane == 0) sh.pos[0] = seed * SGT + wave * LEAF;
                if (lane < LEAF) sh.rec[lane] = target(seed * SGT + wave * LEAF + lane);
                sh.queue[lane] = (unsigned)lane << 8;
                eval_entries(sh, 0, nq - sgq * SGT < 64 ? nq - sgq * SGT : 64, nt, lane);
                if constexpr (COUNT) { c_ge += 64u; c_pairs += 64u * (unsigned)cnt; }
            }
        }
        tk[3] = RRL_NOW();
        __syncthreads();
        tk[4] = RRL_NOW();

        // ---- the other supergroups
        int nq_e = 0;  // entries waiting in this wavefront's queue (uniform)
        for (int j0 = 0; j0 < nsgt; j0 += 64) {
            const int j = j0 + lane;
            // prune 1, lane-parallel over the target supergroups: this wavefront's leaf of supergroup j against
            // each of the patch's four query GROUPS (16 consecutive lanes = one group of the query tree) with
            // that group's largest bound -- |cg - cj| <= sqrt(bd_g) + Rg + Rj, squared
            float sb = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned)(s_best[lane] >> 32))) * 1.00001f;
            float sbg[SGG];
            {
                const float rowmax = row16_max(w.valid ? sb : 0.0f);  // fmaxf drops a NaN: see `blind`
                const int rb = __float_as_int(rowmax);
#pragma unroll
                for (int g = 0; g < SGG; ++g) sbg[g] = __int_as_float(__builtin_amdgcn_readlane(rb, 16 * g));
            }
            const bool blind = __any(w.valid && sb != sb);  // some query has no bound yet: no patch-level pruning
            bool cand = false;
            float4 gn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (j < nsgt && j != seed && (j % SPLIT) == cls && nt - j * SGT - wave * LEAF > 0) {  // (an empty leaf is no candidate)
                gn = w.tree[(size_t)j * NODE + LEAF_NODE(wave)];
                cand = blind;
#pragma unroll
                for (int g = 0; g < SGG; ++g) {
                    const float dx = qg[g][0] - gn.x, dy = qg[g][1] - gn.y, dz = qg[g][2] - gn.z;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sbg[g] + qg[g][3] + gn.w;
                    cand = cand || !(d2 * LB_SCALE > t * t);  // NaN radius (empty query group): kept
                }
            }
            unsigned long long m = __ballot(cand);
            if constexpr (COUNT) c_sg += (unsigned)min(64, nsgt - j0);
            while (m) {
                const long long ts0 = RRL_NOW();
                // ---- stage the next <= CHK candidate leaves of this wavefront (one round of coalesced loads)
                const int nc = min(CHK, __popcll(m));
                {
                    const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (cand && ((m >> lane) & 1ull) && rk < nc) sh.pos[rk] = (j0 + lane) * SGT + wave * LEAF;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (int i = lane; i < nc * LEAF; i += 64) {
                    const int slot = i / LEAF, t = i % LEAF;
                    sh.rec[slot * LROW + t] = target(sh.pos[slot] + t);  // rows exist up to the supergroup boundary
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if constexpr (COUNT) { (void)sh.rec[lane].x; }
                const long long ts1 = RRL_NOW();
                t_stage += ts1 - ts0;
                // ---- per staged leaf: prune 2 per lane, entries for the lanes that need it
                for (int c = 0; c < nc; ++c) {
                    const int sl = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    // the leaf node was fetched by lane sl above: broadcast it (no dependent load per candidate)
                    const float gx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.x), sl));
                    const float gy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.y), sl));
                    const float gz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.z), sl));
                    const float gr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.w), sl));
                    if constexpr (COUNT) ++c_gt;
                    // |q - c| <= sqrt(bd) + R, squared, with the query's CURRENT bound (all wavefronts fold into it);
                    // NaN anywhere: visit
                    sb = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned)(s_best[lane] >> 32))) * 1.00001f;
                    const float dx = w.qx - gx, dy = w.qy - gy, dz = w.qz - gz;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sb + gr;
                    const bool need = w.valid && !(d2 * LB_SCALE > t * t);
                    const unsigned long long nm = __ballot(need);
                    if (need) {
                        const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
                        sh.queue[nq_e + rk] = ((unsigned)lane << 8) | (unsigned)c;
                    }
                    nq_e += __popcll(nm);
                    if (nq_e >= 64) {  // uniform: a full pass
                        nq_e -= 64;
                        const long long tp0 = RRL_NOW();
                        eval_entries(sh, nq_e, 64, nt, lane);
                        t_pass += RRL_NOW() - tp0;
                        if constexpr (COUNT) { c_ge += 64u; c_pairs += 64u * (unsigned)LEAF; }
                    }
                }
                // the staged rows are overwritten by the next round: drain the queue
                const long long tp0 = RRL_NOW();
                if (nq_e > 0) {
                    eval_entries(sh, 0, nq_e, nt, lane);
                    if constexpr (COUNT) { c_ge += (unsigned)nq_e; c_pairs += (unsigned)nq_e * (unsigned)LEAF; }
                    nq_e = 0;
                }
                const long long tp1 = RRL_NOW();
                t_pass += tp1 - tp0;
                t_test += tp1 - ts1;
            }
        }
        tk[5] = RRL_NOW();
        __syncthreads();
        tk[6] = RRL_NOW();
        if (wv == 0 && w.valid) {
            w.best = s_best[lane];
            const bool qnan = (w.qx != w.qx) || (w.qy != w.qy) || (w.qz != w.qz);
            if (qnan || tnan)  // torch.min propagates NaN
                w.best = ((unsigned long long)0x7fc00000u << 32) | (unsigned)(w.best & 0xffffffffu);
            (dir ? best_y : best_x)[(size_t)b * nq + qorig] = w.best;
            mine = (double)__uint_as_float((unsigned)(w.best >> 32));
        }
    }
    if constexpr (COUNT) {
        // one 16-slot row per wavefront, plain stores (same-address atomics from 4096 wavefronts would
        // saturate the memory system and distort the very clocks recorded here); the host adds the rows
        const long long crow = (long long)(blockIdx.y * gridDim.x + blockIdx.x) * NWV + wv;  // rows past the buffer are dropped
        if (lane == 0 && sgq < nsgq && crow < counter_rows) {
            unsigned long long *row = counters + 16 * (size_t)crow;
            row[0] = c_sg; row[1] = c_gt; row[2] = c_ge; row[3] = c_pairs; row[4] = 1;
            row[5] = (unsigned long long)t_stage;
            row[6] = (unsigned long long)(t_test - t_pass);
            row[7] = (unsigned long long)t_pass;
            row[8] = (unsigned long long)((long long)wall_clock64() - wall0);  // 100 MHz ticks of the same span as sum(tk)
            row[9] = (unsigned long long)wall0;  // absolute start (tools/cham_count_vs_plain.py: start-time spread)
            row[15] = (unsigned long long)(RRL_NOW() - tk[0]);
            for (int i = 2; i <= 6; ++i) row[8 + i] = (unsigned long long)(tk[i] - tk[i - 1]);
        }
    }
#undef RRL_NOW
    if (wv != 0) return;  // the minima of the patch are in wavefront 0
    // ---- mean: fixed-order sums, finished INSIDE this launch (round 3; a second, tiny launch added 4.2 us + a launch
    //      boundary to a 22 us walk).  The patch's 64 minima in one wavefront (LDS operations execute in order) ->
    //      partial[workgroup]; the LAST patch of a (sample, direction) to arrive adds that group's partials in index
    //      order -> gpart[group]; the last GROUP adds the 2 B group sums in index order and writes the value.  Two
    //      levels because a thousand arrivals on one word serialise (~12 ns each); the counters live 128 bytes apart.
    //      Stores / loads of what crosses workgroups are agent-scope (sc1), every counter rewinds itself.
    auto wave_total = [&](double v) {  // all lanes must call; lane 0 holds the sum
        red[lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int o = 32; o > 0; o >>= 1) {
            if (lane < o) red[lane] += red[lane + o];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        return red[0];
    };
    auto ld64 = [](const double *p) {
        return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    auto st64 = [](double *p, double v) {
        __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const double psum = wave_total(mine);
    const int wg = (int)(blockIdx.y * gridDim.x + blockIdx.x), grp = (int)blockIdx.x, ngrp = (int)gridDim.x, npatch = (int)gridDim.y;
    if (tk_.top == nullptr) {  // (no counters: leave the partials to the caller)
        if (lane == 0) partial[wg] = psum;
        return;
    }
    uint32_t *gtick = tk_.group + (size_t)b * tk_.stride_b + (size_t)dir * tk_.stride_d;
    int last = 0;
    if (lane == 0) {
        st64(&partial[wg], psum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(gtick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(npatch - 1) ? 1 : 0;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    double acc = 0.0;
    for (int i = lane; i < npatch; i += 64) acc += ld64(&partial[(size_t)i * ngrp + grp]);  // patches of this group, index order per lane
    const double gsum = wave_total(acc);
    last = 0;
    if (lane == 0) {
        __hip_atomic_store(gtick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st64(&gpart[grp], gsum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(tk_.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ngrp - 1) ? 1 : 0;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    acc = 0.0;
    for (int i = lane; i < ngrp; i += 64) acc += ld64(&gpart[i]);
    const double tot = wave_total(acc);
    if (lane == 0) {
        value[0] = (float)(tot / denom);
        __hip_atomic_store(tk_.top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static unsigned long long *g_cham_counters = nullptr;
static long long g_cham_counter_rows = 0;
extern "C" int rrl_chamfer_counters(uint64_t *dev_counters, long long rows) {
    g_cham_counters = (unsigned long long *)dev_counters;
    g_cham_counter_rows = dev_counters ? rows : 0;
    return 0;
}

// best_x [B][N], best_y [B][M]: u64 keys, every entry written exactly once (no initialisation needed).
extern "C" int rrl_chamfer_tree_fwd_ex(const float *x, const float *y, void *ws, size_t ws_bytes, uint64_t *best_x,
                                       uint64_t *best_y, float *value, int B, int N, int M, const int32_t *order_x,
                                       const int32_t *order_y, uint64_t *counters, long long counter_rows, void *stream) {
    unsigned long long *const cnt_buf = (unsigned long long *)counters;  // per-call counter table (NULL: the plain kernel)
    const long long cnt_rows = counters ? counter_rows : 0;
    if (!x || !y || !ws || !best_x || !best_y || !value || B <= 0 || N <= 0 || M <= 0) return RRL_E_ARG;
    if ((N > M ? N : M) > rrl_sort_capacity() || B > 32767) return RRL_E_ARG;
    const ChamLayout L(B, N, M);
    if (ws_bytes < L.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    const int nmax = N > M ? N : M;
    const bool small = nmax <= 4096;  // the sort kernel reads the points itself: no records launch
    int rc = 0;
    if (order_x && order_y) {  // prepared clouds: records at their sorted positions + tree refit, no sort
        const int npadmax = (nmax + SGT - 1) / SGT * SGT;
        hipLaunchKernelGGL(pts_records_sorted_kernel, dim3((unsigned)((npadmax + 255) / 256), (unsigned)B, 2u), dim3(256), 0, s, x, y,
                           order_x, order_y, (float4 *)(w + L.p0s1), (float4 *)(w + L.p0s2), (int32_t *)(w + L.idx1),
                           (int32_t *)(w + L.idx2), (float4 *)(w + L.grp1), (float4 *)(w + L.grp2), (float *)(w + L.apart),
                           (uint32_t *)(w + L.ctrl), (int)(L.ctrl_bytes / 4), B, N, M, L.nblk);
    } else {
    if (!small)
        hipLaunchKernelGGL(pts_records_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B, 2u), dim3(256), 0, s, x,
                           y, (float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart),
                           (uint4 *)(w + L.histg), (L.total - L.histg) / 16, B, N, M, L.nblk);
    rc = rrl_launch_cloud_sort(small ? x : nullptr, small ? y : nullptr, (float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart), L.nblk,
                                   (float4 *)(w + L.p0s1), (float4 *)(w + L.p0s2), (int32_t *)(w + L.idx1),
                                   (int32_t *)(w + L.idx2), (float4 *)(w + L.grp1), (float4 *)(w + L.grp2),
                                   (uint32_t *)(w + L.pmax), (unsigned *)(w + L.histg),
                                   small ? (uint32_t *)(w + L.ctrl) : nullptr, (int)(L.ctrl_bytes / 4),  // (large clouds: cleared by pts_records_kernel)
                                   B, N, M, s);
    }
    if (rc) return rc;
    const int nsgmax = (nmax + SGT - 1) / SGT;
    const ChamTick tick = {(uint32_t *)(w + L.ctrl), (uint32_t *)(w + L.ctrl) + 32, 64, 32};
#define RRL_NN_LAUNCH(COUNT)                                                                                     \
    hipLaunchKernelGGL((chamfer_tree_kernel<COUNT, false>), dim3((unsigned)(2 * B), (unsigned)nsgmax), dim3(64 * NWV), 0, \
                       s, (const float4 *)(w + L.p0s1), (const float4 *)(w + L.p0s2),                            \
                       (const float4 *)(w + L.grp1), (const float4 *)(w + L.grp2), (const float *)(w + L.apart),  \
                       L.nblk, (unsigned long long *)best_x, (unsigned long long *)best_y,                       \
                       (double *)(w + L.partial), B, N, M, cnt_buf, cnt_rows, (const int32_t *)nullptr, \
                       (const int32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, tick,      \
                       (double *)(w + L.gpart), value, (double)B * (double)(N + M))
    if (cnt_buf) RRL_NN_LAUNCH(true);
    else RRL_NN_LAUNCH(false);
#undef RRL_NN_LAUNCH
    RRL_LAUNCH_CHECK();
    return 0;
}
extern "C" int rrl_chamfer_tree_fwd(const float *x, const float *y, void *ws, size_t ws_bytes, uint64_t *best_x,
                                    uint64_t *best_y, float *value, int B, int N, int M, void *stream) {
    return rrl_chamfer_tree_fwd_ex(x, y, ws, ws_bytes, best_x, best_y, value, B, N, M, nullptr, nullptr,
                                   (uint64_t *)g_cham_counters, g_cham_counter_rows, stream);
}

// Chamfer between the two clouds of a loss evaluation WITHOUT sorting them again (include/rrl.h): the loss
// forward left the sorted P0 records, their original indices and the sphere trees of both clouds in its
// workspace -- the first point of every pseudo-triangle IS the point of the cloud (loss.Sample_neighs:
// row = [P, neighbour, neighbour]), and the fused op's source records are the MOVED source.
//   ws_src: workspace of the evaluation (cloud 1);  ws_tar: the workspace that holds cloud 2's records --
//   the same one, or the `target_ws` the evaluation was carried over from (rrl_loss_forward_cached).
// The tree radii include the triangles' thresholds (looser bounds, same minima).  Keys and value are those of
// rrl_chamfer_fwd on (P0 of cloud 1, P0 of cloud 2); a non-finite or overflowing coordinate gives NaN.
extern "C" int rrl_chamfer_from_loss_ex(void *ws_src, const void *ws_tar, size_t loss_ws_bytes, int B, int N, int M,
                                        int L, void *ws, size_t ws_bytes, uint64_t *best_x, uint64_t *best_y,
                                        float *value, uint64_t *counters, long long counter_rows, void *stream) {
    unsigned long long *const cnt_buf = (unsigned long long *)counters;
    const long long cnt_rows = counters ? counter_rows : 0;
    if (!ws_src || !ws_tar || !ws || !best_x || !best_y || !value || B <= 0 || N <= 0 || M <= 0 || L < 0) return RRL_E_ARG;
    if ((N > M ? N : M) > rrl_sort_capacity() || B > 32767) return RRL_E_ARG;  // larger clouds are not sorted by the loss
    const WsLayout lw(B, N, M, L);
    if (loss_ws_bytes < lw.total) return RRL_E_WS;
    const ChamLayout C(B, N, M);
    if (ws_bytes < C.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    const int nmax = N > M ? N : M, nsgmax = (nmax + SGT - 1) / SGT;
    // arrival counters of the mean: words of the evaluation's own workspace that every forward leaves zero and that
    // rewind themselves (MCTL[b][30 + direction] per group, MCTL[0][32] for the groups): any number of calls per forward
    uint32_t *mctl = (uint32_t *)((char *)ws_src + lw.off[RRL_WS_MCTL]);
    const ChamTick tick = {mctl + 32, mctl + 30, 64, 1};
#define RRL_NN_LAUNCH(COUNT)                                                                                     \
    hipLaunchKernelGGL((chamfer_tree_kernel<COUNT, true>), dim3((unsigned)(2 * B), (unsigned)nsgmax), dim3(64 * NWV), 0, \
                       s, (const float4 *)lw.f32(ws_src, RRL_WS_P0S1), (const float4 *)lw.f32(ws_tar, RRL_WS_P0S2), \
                       (const float4 *)lw.f32(ws_src, RRL_WS_GRP1), (const float4 *)lw.f32(ws_tar, RRL_WS_GRP2),  \
                       (const float *)nullptr, 0, (unsigned long long *)best_x, (unsigned long long *)best_y,    \
                       (double *)(w + C.partial), B, N, M, cnt_buf, cnt_rows, lw.i32(ws_src, RRL_WS_IDX1), \
                       lw.i32(ws_tar, RRL_WS_IDX2), (const uint32_t *)lw.i32(ws_src, RRL_WS_PMAX),              \
                       (const uint32_t *)lw.i32(ws_tar, RRL_WS_PMAX) + B, tick, (double *)(w + C.gpart), value,  \
                       (double)B * (double)(N + M))
    if (cnt_buf) RRL_NN_LAUNCH(true);
    else RRL_NN_LAUNCH(false);
#undef RRL_NN_LAUNCH
    RRL_LAUNCH_CHECK();
    return 0;
}
extern "C" int rrl_chamfer_from_loss(void *ws_src, const void *ws_tar, size_t loss_ws_bytes, int B, int N, int M,
                                     int L, void *ws, size_t ws_bytes, uint64_t *best_x, uint64_t *best_y,
                                     float *value, void *stream) {
    return rrl_chamfer_from_loss_ex(ws_src, ws_tar, loss_ws_bytes, B, N, M, L, ws, ws_bytes, best_x, best_y, value,
                                    (uint64_t *)g_cham_counters, g_cham_counter_rows, stream);
}
