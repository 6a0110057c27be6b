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
#include "rrl_chamfer_walk.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

int rrl_launch_cloud_sort(const float *raw1, const float *raw2, float4 *crec1, float4 *crec2, float *apart, int nblk,
                          float4 *p0s1, float4 *p0s2, int32_t *idx1, int32_t *idx2, float4 *grp1, float4 *grp2,
                          uint32_t *pmax, unsigned *histg, uint32_t *zwords, int nzwords, int B, int N, int M, hipStream_t s);
int rrl_sort_capacity(void);

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

// G group means of ONE walk (include/rrl.h rrl_chamfer_group_means): the walk leaves the sum of every (sample, direction)
// in its workspace (GPART [2 B] doubles, index 2 b + direction: chamfer_tree_body's second-level partials); group g =
// samples [g B / G, (g + 1) B / G).  One wavefront per group, fixed-order double sums.
__global__ __launch_bounds__(64) void chamfer_group_mean_kernel(const double *__restrict__ gpart, float *__restrict__ values,
                                                                int per, double denom) {
    __shared__ double red[64];
    const int g = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int i = lane; i < 2 * per; i += 64) acc += gpart[(size_t)2 * per * g + i];
    red[lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int o = 32; o > 0; o >>= 1) {
        if (lane < o) red[lane] += red[lane + o];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) values[g] = (float)(red[0] / denom);
}
extern "C" int rrl_chamfer_group_means(const void *cham_ws, size_t cham_ws_bytes, float *values, int G, int B, int N, int M,
                                       void *stream) {
    if (!cham_ws || !values || G <= 0 || B <= 0 || B % G || N < 0 || M < 0 || N + M == 0) return RRL_E_ARG;
    const ChamLayout C(B, N, M);
    if (cham_ws_bytes < C.total) return RRL_E_WS;
    hipLaunchKernelGGL(chamfer_group_mean_kernel, dim3((unsigned)G), dim3(64), 0, (hipStream_t)stream,
                       (const double *)((const char *)cham_ws + C.gpart), values, B / G, (double)(B / G) * (double)(N + M));
    RRL_LAUNCH_CHECK();
    return 0;
}
