// rrl_chamfer.hip -- K7 Chamfer monitor on the spatial structures of the scan
// (code/loss.py:38-52 compute_sqrdis_map_2, :236-252 chamfer_dist).
//
// The brute-force kernel (rrl_geom.hip chamfer_nn_kernel) evaluates all N x M pairs in both
// directions: 2.7e8 pair evaluations at B=8, 4096 x 4096, 79 us, at the issue limit of VALU ops
// with a scalar operand.  Here both clouds are put in grid-cell (Hilbert) order under the sphere
// tree of rrl_tree.h -- the sort kernels of rrl_cull.hip, unchanged, fed with (x, y, z, index)
// records -- and a wavefront = 64 CONSECUTIVE SORTED queries (one supergroup: a compact patch)
// walks the target's tree with wave-uniform control flow:
//   seed      the target supergroup whose centre is nearest to the patch centre is evaluated
//             first (64 points, brute force): every lane gets a good upper bound bd;
//   prune 1   lane j tests target supergroup j against the PATCH sphere:
//             (|cq - cj| - Rq - Rj)+^2 > max_lanes(bd) can hold no nearest neighbour of any lane;
//   prune 2   per surviving supergroup, and per group of 16 inside it, every lane tests ITS query:
//             the node is visited when (|q - c| - R)+^2 <= bd for some lane (one ballot);
//   leaves    the 16 points of a surviving group arrive through the scalar cache (wave-uniform),
//             11 VALU ops per (query, target) pair.
// Exactness: a distance is evaluated with the reference's arithmetic ((dx^2 + dy^2) + dz^2, no
// FMA); lower bounds are scaled by (1 - 1e-4) and use the tree's conservative radii, so a node is
// skipped only if every point in it is STRICTLY farther than the lane's current best; the running
// minimum is the u64 key (distance bits << 32 | ORIGINAL target index), so ties resolve to the
// smallest index in any visiting order == torch.min's first occurrence.  Keys are bit-identical
// to the brute-force kernel's (tests/test_gpu_parity.py).  NaN: a NaN coordinate in the target cloud
// makes every minimum of that sample NaN, a NaN query makes its own minimum NaN (torch semantics:
// min propagates NaN) -- decided from per-cloud flags, at no cost in the inner loop.
// The mean is folded into the same launch: per-workgroup double partials, fixed-order sum by the
// last workgroup to arrive (deterministic).
#include "rrl_tree.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

int rrl_launch_cloud_sort(float4 *crec1, float4 *crec2, float *apart, int nblk, float4 *p0s1, float4 *p0s2,
                          int32_t *idx1, int32_t *idx2, float4 *grp1, float4 *grp2, uint32_t *pmax,
                          unsigned *histg, int B, int N, int M, hipStream_t s);
int rrl_sort_capacity(void);

struct ChamLayout {
    size_t crec1, crec2, p0s1, p0s2, idx1, idx2, grp1, grp2, apart, pmax, histg, flags, partial, ticket, total;
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
        // the fields below are cleared by pts_records_kernel (contiguous)
        histg = take(nmax > 4096 ? 4 * 2 * b * 2 * SORT_CELLS : 16);
        flags = take(4 * 2 * b);
        ticket = take(16);
        partial = take(8 * (2 * b * ((nmax + 63) / 64) + 1));  // >= one per workgroup of the NN launch
        total = o;
    }
};

extern "C" size_t rrl_chamfer_workspace_bytes(int B, int N, int M) {
    if (B < 0 || N < 0 || M < 0) return 0;
    return ChamLayout(B, N, M).total;
}

// (x, y, z, original index) records in original order + per-workgroup AABB / max |P|^2 partials (the
// sort kernels' inputs), NaN flags per cloud and sample, and the clearing of the call's counters.
__global__ __launch_bounds__(256) void pts_records_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          float4 *__restrict__ crec1, float4 *__restrict__ crec2,
                                                          float *__restrict__ apart, int32_t *__restrict__ flags,
                                                          uint4 *__restrict__ zero, size_t zero_vec4, int B, int N,
                                                          int M, int nblk) {
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
    if (__any(bad) && lane == 0) atomicOr(&flags[cloud * B + b], 1);
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
        for (int w = 1; w < 4; ++w) r = tid < 3 ? fminf(r, red[w][tid]) : fmaxf(r, red[w][tid]);
        apart[(((size_t)cloud * B + b) * nblk + blockIdx.x) * 8 + tid] = r;
    }
}

#define LB_SCALE 0.9999f  // lower bounds are shrunk: evaluation error of the bound itself (~1e-6 relative)

struct NNWave {
    const float4 *T;      // sorted target records of the sample
    const float4 *tree;   // target tree
    int nt;               // real target records
    float qx, qy, qz;
    bool valid;
    unsigned long long best;  // (distance bits << 32) | original target index
};

// lower bound of the squared distance from this lane's query to anything inside the node
__device__ __forceinline__ float node_lb(const NNWave &w, float cx, float cy, float cz, float Rs) {
    const float dx = w.qx - cx, dy = w.qy - cy, dz = w.qz - cz;
    const float t = sqrtf(dx * dx + dy * dy + dz * dz) - Rs;
    return t > 0.0f ? t * t * LB_SCALE : 0.0f;  // NaN anywhere -> 0: visit
}

__device__ __forceinline__ bool lane_needs(const NNWave &w, float lb) {
    return w.valid && !(lb > __uint_as_float((unsigned)(w.best >> 32)));
}

// the `cnt` records starting at sorted position pos0: wave-uniform targets through the scalar cache
__device__ __forceinline__ void eval_points(NNWave &w, int pos0, int cnt) {
    kptr tp = (kptr)(uintptr_t)(w.T + pos0);
    for (int t = 0; t < cnt; ++t, tp += 4) {
        // code/loss.py:51: sum((x - y)**2, -1); (a0 + a1) + a2, no FMA
        const float dx = w.qx - tp[0], dy = w.qy - tp[1], dz = w.qz - tp[2];
        float s = dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)__float_as_int(tp[3]);
        w.best = key < w.best ? key : w.best;  // NaN bits (> +inf bits) never win: handled by the flags
    }
}

__device__ __forceinline__ void visit_supergroup(NNWave &w, int s, bool check) {
    kptr nd = (kptr)(uintptr_t)(w.tree + (size_t)s * NODE);
    if (check && !__any(lane_needs(w, node_lb(w, nd[0], nd[1], nd[2], nd[3])))) return;
    const int left = w.nt - s * SGT;  // > 0
#pragma unroll
    for (int k = 0; k < SGG; ++k) {
        const int cnt = min(GRP, left - k * GRP);
        if (cnt <= 0) break;  // uniform
        kptr g = nd + 4 * (1 + k);
        if (!__any(lane_needs(w, node_lb(w, g[0], g[1], g[2], g[3])))) continue;
        eval_points(w, s * SGT + k * GRP, cnt);
    }
}

__global__ __launch_bounds__(256) void chamfer_tree_kernel(
    const float4 *__restrict__ p0s1, const float4 *__restrict__ p0s2, const float4 *__restrict__ grp1,
    const float4 *__restrict__ grp2, const int32_t *__restrict__ flags, unsigned long long *__restrict__ best_x,
    unsigned long long *__restrict__ best_y, double *__restrict__ partial, int32_t *__restrict__ ticket,
    float *__restrict__ value, int B, int N, int M) {
    __shared__ double red[256];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = blockDim.x >> 6;
    const int b = blockIdx.z >> 1, dir = blockIdx.z & 1;
    const int nq = dir ? M : N, nt = dir ? N : M;
    const int nsgq = (nq + SGT - 1) / SGT, nsgt = (nt + SGT - 1) / SGT;
    const int sgq = (int)blockIdx.x * nwave + wave;
    double mine = 0.0;
    if (sgq < nsgq) {  // wave-uniform
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
        const float cqx = qn[0], cqy = qn[1], cqz = qn[2], Rq = qn[3];

        // ---- seed: the target supergroup nearest to the patch centre
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
        visit_supergroup(w, seed, false);

        // ---- the other supergroups, pruned against the whole patch first, then per lane
        for (int j0 = 0; j0 < nsgt; j0 += 64) {
            const int j = j0 + lane;
            const float bdmax = wave_max(w.valid ? __uint_as_float((unsigned)(w.best >> 32)) : 0.0f);
            bool cand = false;
            if (j < nsgt && j != seed) {
                const float4 c = w.tree[(size_t)j * NODE];
                const float dx = cqx - c.x, dy = cqy - c.y, dz = cqz - c.z;
                const float t = sqrtf(dx * dx + dy * dy + dz * dz) - Rq - c.w;
                const float lb = t > 0.0f ? t * t * LB_SCALE : 0.0f;
                cand = !(lb > bdmax);
            }
            unsigned long long m = __ballot(cand);
            while (m) {
                const int s = __ffsll((long long)m) - 1;
                m &= m - 1;
                visit_supergroup(w, j0 + s, true);
            }
        }

        // ---- result of this lane's query
        if (w.valid) {
            const bool qnan = (w.qx != w.qx) || (w.qy != w.qy) || (w.qz != w.qz);
            if (qnan || flags[(dir ? 0 : 1) * B + b])  // torch.min propagates NaN
                w.best = ((unsigned long long)0x7fc00000u << 32) | (unsigned)(w.best & 0xffffffffu);
            (dir ? best_y : best_x)[(size_t)b * nq + __float_as_int(qr.w)] = w.best;
            mine = (double)__uint_as_float((unsigned)(w.best >> 32));
        }
    }
    // ---- mean: fixed-order sum inside the workgroup, fixed-order sum of the partials by the last
    //      workgroup to arrive
    red[tid] = mine;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const int nwg = gridDim.x * gridDim.z, me = blockIdx.z * gridDim.x + blockIdx.x;
    if (tid == 0) {
        __hip_atomic_store(&partial[me], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release: the partial is visible device-wide before the ticket; acquire: the last arrival
        // sees every other workgroup's partial
        s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double acc = 0.0;
    for (int i = tid; i < nwg; i += blockDim.x)
        acc += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        value[0] = (float)(red[0] / ((double)B * (double)(N + M)));
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    hipLaunchKernelGGL(pts_records_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B, 2u), dim3(256), 0, s, x, y,
                       (float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart),
                       (int32_t *)(w + L.flags), (uint4 *)(w + L.histg), (L.total - L.histg) / 16, B, N, M, L.nblk);
    int rc = rrl_launch_cloud_sort((float4 *)(w + L.crec1), (float4 *)(w + L.crec2), (float *)(w + L.apart), L.nblk,
                                   (float4 *)(w + L.p0s1), (float4 *)(w + L.p0s2), (int32_t *)(w + L.idx1),
                                   (int32_t *)(w + L.idx2), (float4 *)(w + L.grp1), (float4 *)(w + L.grp2),
                                   (uint32_t *)(w + L.pmax), (unsigned *)(w + L.histg), B, N, M, s);
    if (rc) return rc;
    const int nsgmax = (nmax + SGT - 1) / SGT;
    const int waves = (long)2 * B * nsgmax >= 1024 ? 4 : 1;  // small problems: one wavefront per workgroup
    hipLaunchKernelGGL(chamfer_tree_kernel, dim3((unsigned)((nsgmax + waves - 1) / waves), 1u, (unsigned)(2 * B)),
                       dim3(64 * waves), 0, s, (const float4 *)(w + L.p0s1), (const float4 *)(w + L.p0s2),
                       (const float4 *)(w + L.grp1), (const float4 *)(w + L.grp2), (const int32_t *)(w + L.flags),
                       (unsigned long long *)best_x, (unsigned long long *)best_y, (double *)(w + L.partial),
                       (int32_t *)(w + L.ticket), value, B, N, M);
    RRL_LAUNCH_CHECK();
    return 0;
}
