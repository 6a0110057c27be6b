// rrl_chamfer.hip -- K7 Chamfer monitor on the spatial structures of the scan
// (code/loss.py:38-52 compute_sqrdis_map_2, :236-252 chamfer_dist).
//
// The brute-force kernel (rrl_geom.hip chamfer_nn_kernel) evaluates all N x M pairs in both
// directions: 2.7e8 pair evaluations at B=8, 4096 x 4096, 79 us + 19 us for the mean, at the issue
// limit of VALU ops with a scalar operand.  Here both clouds are put in grid-cell (Hilbert) order under
// the sphere tree of rrl_tree.h -- the sort kernels of rrl_cull.hip, fed with (x, y, z, index) records
// (built inside the sort kernel for clouds <= 4096 points) -- and a workgroup = one PATCH of 64
// consecutive sorted queries (one supergroup of the query cloud) walks the target's tree with
// wave-uniform control flow, NNW wavefronts sharing the patch
// (every wavefront holds all 64 queries in its lanes; of every target supergroup wavefront k owns leaf
// k = group k of 16 records for NNW = 4):
//   seed      the target supergroup whose centre is nearest to the patch centre is evaluated first
//             (wavefront k its leaf k, minima exchanged through LDS): a good upper bound bd per query;
//   prune 1   lane-parallel over the target supergroups: lane j tests this wavefront's leaf of supergroup
//             j against each of the patch's four query-group spheres with that group's largest bound;
//   prune 2   per surviving leaf every lane tests ITS query against the leaf sphere; the leaf is
//             evaluated when some lane's bound allows it (one ballot);
//   leaves    the records of a leaf arrive through the scalar cache in one request (wave-uniform),
//             11 VALU ops per (query, target) pair, keys reduced as a tree.
// Exactness: a distance is evaluated with the reference's arithmetic ((dx^2 + dy^2) + dz^2, no FMA);
// every pruning test is |q - c|^2 (1 - 1e-4) > (sqrt(bd) (1 + 1e-5) + R)^2 with the tree's conservative
// radii, so a leaf is skipped only if every point in it is STRICTLY farther than the lane's current
// best; the running minimum is the u64 key (distance bits << 32 | ORIGINAL target index), so ties
// resolve to the smallest index in any visiting order == torch.min's first occurrence.  Keys are
// bit-identical to the brute-force kernel's (tests/test_gpu_parity.py).  NaN: a NaN coordinate in the
// target cloud makes every minimum of that sample NaN, a NaN query its own minimum (torch semantics:
// min propagates NaN) -- decided from per-workgroup flags of the records kernel, at no cost in the
// inner loop.  The mean: per-patch double partials, summed in a fixed order by a tiny second launch.
//
// Measured on the way (B=8, 4096 x 4096, profiles/r02_chamfer_notes.txt): one wavefront per patch 88 us;
// a release fence per workgroup for a "last workgroup sums" hand-over +30 us (1024 write-backs of an
// XCD's L2), relaxed tickets on one address +5 us (same-address device atomics serialise at ~12 ns);
// patches of one (sample, direction) spread over all XCDs +10 us; broadcasting the targets through LDS
// (all lanes reading the same 16 bytes costs the full 1 KiB of LDS bandwidth) or with v_readlane: no
// better than the scalar cache; a running u64 minimum instead of the key tree: same.
#include "rrl_tree.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

int rrl_launch_cloud_sort(const float *raw1, const float *raw2, float4 *crec1, float4 *crec2, float *apart, int nblk,
                          float4 *p0s1, float4 *p0s2, int32_t *idx1, int32_t *idx2, float4 *grp1, float4 *grp2,
                          uint32_t *pmax, unsigned *histg, int B, int N, int M, hipStream_t s);
int rrl_sort_capacity(void);

struct ChamLayout {
    size_t crec1, crec2, p0s1, p0s2, idx1, idx2, grp1, grp2, apart, pmax, histg, partial, total;
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
        histg = take(nmax > 4096 ? 4 * 2 * b * 2 * SORT_CELLS : 16);  // cleared by pts_records_kernel
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

#define LB_SCALE 0.9999f  // squared distances to node centres are shrunk: evaluation error of the bound (~1e-6)

struct NNWave {
    const float4 *T;      // sorted target records of the sample
    const float4 *tree;   // target tree
    int nt;               // real target records
    float qx, qy, qz;
    bool valid;
    unsigned long long best;  // (distance bits << 32) | original target index
};

__device__ __forceinline__ unsigned long long point_key(const NNWave &w, float tx, float ty, float tz, float ti) {
    // code/loss.py:51: sum((x - y)**2, -1); (a0 + a1) + a2, no FMA
    const float dx = w.qx - tx, dy = w.qy - ty, dz = w.qz - tz;
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    // NaN bits (> +inf bits) never win: handled by the flags
    return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)__float_as_int(ti);
}
__device__ __forceinline__ unsigned long long kmin(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
// sqrt(current best distance) rounded up; NaN bits (no target seen yet) give NaN: every test then visits
__device__ __forceinline__ float root_of_best(const NNWave &w) {
    return __builtin_amdgcn_sqrtf(__uint_as_float((unsigned)(w.best >> 32))) * 1.00001f;
}

#ifndef NNW
#define NNW 4  // wavefronts per patch: 4 (leaf = group of 16) or 8 (leaf = half of 8)
#endif
#define LEAF (SGT / NNW)
static_assert(NNW == 8 || NNW == 4, "a wavefront owns a half (8 records) or a group (16) of every supergroup");
#define LEAF_NODE(k) (NNW == 8 ? 5 + (k) : 1 + (k))
#ifndef TB
#define TB 1  // per-lane leaf tests per loop iteration (2, 4: no faster -- the kernel is VALU-issue bound, see notes)
#endif

// The `cnt` (<= LEAF) records of one leaf, wave-uniform, through the scalar cache in ONE request (the
// arrays are padded to whole supergroups, so all LEAF rows exist); the keys are independent and
// reduced as a tree.
__device__ __forceinline__ void eval_leaf(NNWave &w, int pos0, int cnt) {
    kptr tp = (kptr)(uintptr_t)(w.T + pos0);
    if (cnt == LEAF) {  // uniform
        float r[4 * LEAF];
#pragma unroll
        for (int i = 0; i < 4 * LEAF; ++i) r[i] = tp[i];
        unsigned long long key[LEAF];
#pragma unroll
        for (int t = 0; t < LEAF; ++t) key[t] = point_key(w, r[4 * t], r[4 * t + 1], r[4 * t + 2], r[4 * t + 3]);
#pragma unroll
        for (int o = LEAF / 2; o > 0; o >>= 1)
#pragma unroll
            for (int t = 0; t < o; ++t) key[t] = kmin(key[t], key[t + o]);
        w.best = kmin(w.best, key[0]);
    } else {  // the ragged last leaf of a cloud
        for (int t = 0; t < cnt; ++t, tp += 4) w.best = kmin(w.best, point_key(w, tp[0], tp[1], tp[2], tp[3]));
    }
}

// COUNT: executed-work counters (rrl_chamfer_counters): [0] patch-level leaf tests (lane-parallel),
// [1] per-lane leaf sphere tests (wave x leaf), [2] leaves evaluated, [3] (query, target) pairs evaluated,
// [4] wavefronts.
template <bool COUNT>
__global__ __launch_bounds__(64 * NNW) void chamfer_tree_kernel(
    const float4 *__restrict__ p0s1, const float4 *__restrict__ p0s2, const float4 *__restrict__ grp1,
    const float4 *__restrict__ grp2, const float *__restrict__ apart, int nblk,
    unsigned long long *__restrict__ best_x, unsigned long long *__restrict__ best_y,
    double *__restrict__ partial, int B, int N, int M, unsigned long long *__restrict__ counters) {
    __shared__ unsigned long long s_best[NNW][64];
    __shared__ double red[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware: workgroups go to the 8 XCDs round-robin by linear id; with (sample, direction) on the fast
    // index every XCD's L2 holds the records and trees of 2 B / 8 of the (cloud pair, direction) combinations only
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int nq = dir ? M : N, nt = dir ? N : M;
    const int nsgq = (nq + SGT - 1) / SGT, nsgt = (nt + SGT - 1) / SGT;
    const int sgq = (int)blockIdx.y;
    double mine = 0.0;
    unsigned c_sg = 0, c_gt = 0, c_ge = 0, c_pairs = 0;
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
        w.qx = qr.x; w.qy = qr.y; w.qz = qr.z;
        w.best = ~0ull;
        kptr qn = (kptr)(uintptr_t)(treeQ + (size_t)sgq * NODE);
        const float cqx = qn[0], cqy = qn[1], cqz = qn[2];
        float qg[SGG][4];  // the patch's four query-group spheres (wave-uniform)
#pragma unroll
        for (int g = 0; g < SGG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) qg[g][c] = qn[4 * (1 + g) + c];
        // does the TARGET cloud hold a NaN coordinate (slot 7 of its AABB partial rows)?
        bool tnan = false;
        if (wave == 0) {
            const int ct = dir ? 0 : 1, nb = (nt + 255) / 256;
            const float *ap = apart + ((size_t)ct * B + b) * nblk * 8;
            for (int j = lane; j < nb; j += 64) tnan |= ap[j * 8 + 7] != 0.0f;
            tnan = __any(tnan);
        }

        // ---- seed: the target supergroup nearest to the patch centre; wavefront k evaluates its leaf k
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
        {
            const int cnt = min(LEAF, nt - seed * SGT - wave * LEAF);
            if (cnt > 0) {
                eval_leaf(w, seed * SGT + wave * LEAF, cnt);
                if constexpr (COUNT) { ++c_ge; c_pairs += 64u * (unsigned)cnt; }
            }
        }
        s_best[wave][lane] = w.best;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NNW; ++k) w.best = kmin(w.best, s_best[k][lane]);
        __syncthreads();

        // ---- the other supergroups
        float sb = root_of_best(w);
        for (int j0 = 0; j0 < nsgt; j0 += 64) {
            const int j = j0 + lane;
            // prune 1, lane-parallel over the target supergroups: this wavefront's leaf of supergroup j against
            // each of the patch's four query GROUPS (16 consecutive lanes = one group of the query tree) with
            // that group's largest bound -- |cg - cj| <= sqrt(bd_g) + Rg + Rj, squared
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
            if (j < nsgt && j != seed) {
                gn = w.tree[(size_t)j * NODE + LEAF_NODE(wave)];
                cand = blind;
#pragma unroll
                for (int g = 0; g < SGG; ++g) {
                    const float dx = qg[g][0] - gn.x, dy = qg[g][1] - gn.y, dz = qg[g][2] - gn.z;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sbg[g] + qg[g][3] + gn.w;
                    cand = cand || !(d2 * LB_SCALE > t * t);  // NaN radius (empty query group / leaf): kept; cnt <= 0 below
                }
            }
            unsigned long long m = __ballot(cand);
            if constexpr (COUNT) c_sg += (unsigned)min(64, nsgt - j0);
            // prune 2, TB candidates per iteration (tests inside a batch use the bound from before the batch)
            while (m) {
                int sl[TB];
                bool need[TB];
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    sl[u] = m ? __ffsll((long long)m) - 1 : -1;
                    if (m) m &= m - 1;
                }
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    need[u] = false;
                    if (sl[u] < 0) continue;  // uniform
                    const int cnt = min(LEAF, nt - (j0 + sl[u]) * SGT - wave * LEAF);
                    if (cnt <= 0) continue;  // uniform: empty leaf
                    // the leaf node was fetched by lane sl above: broadcast it (no dependent load per candidate)
                    const float gx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.x), sl[u]));
                    const float gy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.y), sl[u]));
                    const float gz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.z), sl[u]));
                    const float gr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.w), sl[u]));
                    if constexpr (COUNT) ++c_gt;
                    // per lane: |q - c| <= sqrt(bd) + R, squared; NaN anywhere: visit
                    const float dx = w.qx - gx, dy = w.qy - gy, dz = w.qz - gz;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sb + gr;
                    need[u] = __any(w.valid && !(d2 * LB_SCALE > t * t));
                }
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    if (!need[u]) continue;  // uniform
                    const int s = j0 + sl[u], cnt = min(LEAF, nt - s * SGT - wave * LEAF);
                    eval_leaf(w, s * SGT + wave * LEAF, cnt);
                    if constexpr (COUNT) { ++c_ge; c_pairs += 64u * (unsigned)cnt; }
                }
                sb = root_of_best(w);
            }
        }
        s_best[wave][lane] = w.best;
        __syncthreads();
        if (wave == 0 && w.valid) {
#pragma unroll
            for (int k = 1; k < NNW; ++k) w.best = kmin(w.best, s_best[k][lane]);
            const bool qnan = (w.qx != w.qx) || (w.qy != w.qy) || (w.qz != w.qz);
            if (qnan || tnan)  // torch.min propagates NaN
                w.best = ((unsigned long long)0x7fc00000u << 32) | (unsigned)(w.best & 0xffffffffu);
            (dir ? best_y : best_x)[(size_t)b * nq + __float_as_int(qr.w)] = w.best;
            mine = (double)__uint_as_float((unsigned)(w.best >> 32));
        }
    }
    if constexpr (COUNT) {
        if (lane == 0 && sgq < nsgq) {
            atomicAdd(&counters[0], (unsigned long long)c_sg);
            atomicAdd(&counters[1], (unsigned long long)c_gt);
            atomicAdd(&counters[2], (unsigned long long)c_ge);
            atomicAdd(&counters[3], (unsigned long long)c_pairs);
            atomicAdd(&counters[4], 1ull);
        }
    }
    if (wave != 0) return;  // the minima of the patch are in wavefront 0
    // ---- mean: fixed-order sum of the patch (one wavefront: LDS operations execute in order); the
    //      partials are summed by a second, tiny launch (chamfer_partials_kernel)
    red[lane] = mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int o = 32; o > 0; o >>= 1) {
        if (lane < o) red[lane] += red[lane + o];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// value = (sum of the per-patch partial sums, fixed order) / (B (N + M)): deterministic
__global__ __launch_bounds__(256) void chamfer_partials_kernel(const double *__restrict__ partial, int n,
                                                               float *__restrict__ value, double denom) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) value[0] = (float)(red[0] / denom);
}

static unsigned long long *g_cham_counters = nullptr;
extern "C" int rrl_chamfer_counters(uint64_t *dev_counters) {
    g_cham_counters = (unsigned long long *)dev_counters;
    return 0;
}

// best_x [B][N], best_y [B][M]: u64 keys, every entry written exactly once (no initialisation needed).
extern "C" int rrl_chamfer_tree_fwd(const float *x, const float *y, void *ws, size_t ws_bytes, uint64_t *best_x,
                                    uint64_t *best_y, float *value, int B, int N, int M, void *stream) {
    if (!x || !y || !ws || !best_x || !best_y || !value || B <= 0 || N <= 0 || M <= 0) return RRL_E_ARG;
    if ((N > M ? N : M) > rrl_sort_capacity() || B > 32767) return RRL_E_ARG;
    const ChamLayout L(B, N, M);
    if (ws_bytes < L.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    const int nmax = N > M ? N : M;
    const bool small = nmax <= 4096;  // the sort kernel reads the points itself: no records launch
    if (!small)
        hipLaunchKernelGGL(pts_records_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B, 2u), dim3(256), 0, s, x,
                           y, (float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart),
                           (uint4 *)(w + L.histg), (L.total - L.histg) / 16, B, N, M, L.nblk);
    int rc = rrl_launch_cloud_sort(small ? x : nullptr, small ? y : nullptr, (float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart), L.nblk,
                                   (float4 *)(w + L.p0s1), (float4 *)(w + L.p0s2), (int32_t *)(w + L.idx1),
                                   (int32_t *)(w + L.idx2), (float4 *)(w + L.grp1), (float4 *)(w + L.grp2),
                                   (uint32_t *)(w + L.pmax), (unsigned *)(w + L.histg), B, N, M, s);
    if (rc) return rc;
    const int nsgmax = (nmax + SGT - 1) / SGT;
#define RRL_NN_LAUNCH(COUNT)                                                                                     \
    hipLaunchKernelGGL(chamfer_tree_kernel<COUNT>, dim3((unsigned)(2 * B), (unsigned)nsgmax), dim3(64 * NNW), 0,      \
                       s, (const float4 *)(w + L.p0s1), (const float4 *)(w + L.p0s2),                            \
                       (const float4 *)(w + L.grp1), (const float4 *)(w + L.grp2), (const float *)(w + L.apart),  \
                       L.nblk, (unsigned long long *)best_x, (unsigned long long *)best_y,                       \
                       (double *)(w + L.partial), B, N, M, g_cham_counters)
    if (g_cham_counters) RRL_NN_LAUNCH(true);
    else RRL_NN_LAUNCH(false);
#undef RRL_NN_LAUNCH
    hipLaunchKernelGGL(chamfer_partials_kernel, dim3(1), dim3(256), 0, s, (const double *)(w + L.partial),
                       2 * B * nsgmax, value, (double)B * (double)(N + M));
    RRL_LAUNCH_CHECK();
    return 0;
}
