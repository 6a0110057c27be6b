// rrl_order.hip -- rrl_cloud_order (include/rrl.h): the spatial ORDER of a cloud, computed ONCE per cloud.
//
// The reference scans every (line, triangle) pair and only remarks that "a tree" would help (code/loss.py:260-262);
// its callers move one source rigidly against a fixed target for thousands of steps (code/test_demo_optimized_Lie_
// Algebra.py:57-62) or for several poses per batch (rpm/Train_RPM.py:207-231).  A rigid motion preserves the spatial
// order of a cloud, so the order is a property of the cloud: built here once, used by every later step
// (rrl_opts.order1 / order2 -> tri_records_sorted_kernel, rrl_cull.hip).
//
// Because it runs once, it can afford a better order than the per-step cell sort (16^3 grid cells along a Hilbert
// curve, arbitrary inside a cell): a full K-D ORDER.  Positions [0, P), P = the power of two >= n, form an implicit
// binary tree of aligned windows of S = P, P/2, ..., 16 positions; at every level each window sorts ITS records by the
// coordinate along the longest axis of their bounding box (pads sort last), so its lower half of positions receives the
// records below the median plane.  The sphere tree of the scan sits on aligned runs of 64 / 16 / 8 positions = k-d
// cells: at the bench shape a line meets 5.1 instead of 8.1 supergroup spheres and 47 instead of 90 point-0 tests per
// cloud (tools/attic/order_sim.py; profiles/r04_kd_order.txt), and clouds beyond 4096 triangles get whole-cloud cells instead
// of the four interleaved 4096-chunks of the per-step sort.
// Any permutation gives the same labels, hit lists and loss -- the order only shapes the tree nodes.
//
// Implementation: (key, index) pairs, key = order-preserving bits of the chosen coordinate, bitonic networks.
//   windows of <= 4096 positions: ONE workgroup runs all remaining levels in LDS (kd_window_kernel);
//   larger windows (clouds beyond 4096): per level a node-AABB pass, a key pass and the network with its strides
//   >= 4096 as global passes and everything below inside 4096-position LDS windows.
// One-off cost: ~0.33 ms per call of 8 clouds of 4096, ~2.8 ms at 16384.
#include "rrl_tree.h"

#define KD_WIN 4096   // positions per LDS window
#define KD_THREADS 1024

__device__ __forceinline__ uint32_t f2ord(float x) {  // order-preserving map float -> uint32 (NaNs sort at the ends)
    const uint32_t u = __float_as_uint(x);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);  // negative: all bits flipped; else: the sign bit set
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float(u ^ (((u >> 31) - 1u) | 0x80000000u));
}

// one compare-exchange of the bitonic network on (key, idx) pairs: element i < l; ascending = the smaller pair ends at i
__device__ __forceinline__ void cmpx(uint32_t &ka, int32_t &ia, uint32_t &kb, int32_t &ib, bool asc) {
    const bool gt = ka > kb || (ka == kb && (uint32_t)ia > (uint32_t)ib);
    if (gt == asc) {
        const uint32_t tk = ka; ka = kb; kb = tk;
        const int32_t ti = ia; ia = ib; ib = ti;
    }
}

// Stages k = k_lo .. k_hi (powers of two) of the network over a window of W positions held in LDS, strides j <= W / 2
// only (larger strides were done by global passes).  gbase: global position of the window's first element; S: the node
// size being sorted (the final merge k == S is ascending everywhere: every node ends up ascending).
__device__ __forceinline__ void lds_bitonic(uint32_t *key, int32_t *idx, int W, int gbase, int k_lo, int k_hi, int S) {
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        for (int j = min(k >> 1, W >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (W >> 1); t += KD_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const bool asc = (((gbase + i) & k) == 0) || k == S;
                uint32_t ka = key[i], kb = key[l];
                int32_t ia = idx[i], ib = idx[l];
                cmpx(ka, ia, kb, ib, asc);
                key[i] = ka; key[l] = kb; idx[i] = ia; idx[l] = ib;
            }
            __syncthreads();
        }
    }
}

// ---- levels inside one window of <= 4096 positions ---------------------------------------------------------------
// Window of W <= 4096 positions of sample blockIdx.y: levels S = S0, S0 / 2, ..., 16 (S0 <= W).  idx_g [B][P] holds the
// current order (-1 = pad).  At the end the window's part of `order` ([B][npad], pads -> 0) is written.
// Every lane keeps its E = W / T elements (T = min(W, 1024) lanes; element i <-> lane i % T, slot i / T) in REGISTERS
// between the stages, so that a compare-exchange at stride j costs
//   j >= T        nothing but the lane's own registers (slots r and r ^ (j / T)),
//   j <  64       two cross-lane reads inside the wavefront (ds_bpermute via __shfl_xor),
//   64 <= j < T   one trip through LDS (write, barrier, read the partner, barrier).
// A full sort of 4096 keys then has 18 barrier stages instead of the 78 of the all-LDS network, all nine levels 52
// instead of 354 (measured: 0.41 -> 0.33 ms per call of 8 clouds of 4096 -- the chain of ~300 dependent in-wavefront
// exchanges and the per-level gathers remain; one workgroup per cloud).
__global__ __launch_bounds__(KD_THREADS) void kd_window_kernel(const float *__restrict__ tri, int32_t *__restrict__ idx_g,
                                                               int32_t *__restrict__ order, int n, int npad, int P, int W, int S0,
                                                               int stride) {
    __shared__ uint32_t xkey[KD_WIN];
    __shared__ int32_t xidx[KD_WIN];
    __shared__ uint32_t bb[KD_WIN / 16 * 6];  // per node: min xyz, max xyz (order-preserving bits)
    const int b = blockIdx.y, w0 = blockIdx.x * W, tid = threadIdx.x;
    const int T = W < KD_THREADS ? W : KD_THREADS, E = W / T;  // E in {1, 2, 4}
    const bool live = tid < T;
    const float *t0 = tri + (size_t)b * n * stride;  // rows of `stride` floats whose first three are the point
    uint32_t key[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    int32_t idx[4] = {-1, -1, -1, -1};
    if (live)
        for (int r = 0; r < E; ++r) idx[r] = idx_g[(size_t)b * P + w0 + tid + T * r];
    // the smaller (keep_min) or larger of the lane's pair (ka, ia) and a partner's pair
    auto take = [](uint32_t &ka, int32_t &ia, uint32_t kb, int32_t ib, bool keep_min) {
        const bool b_less = kb < ka || (kb == ka && (uint32_t)ib < (uint32_t)ia);
        if (b_less == keep_min) { ka = kb; ia = ib; }
    };
    for (int S = S0; S >= 16; S >>= 1) {
        const int nodes = W / S;
        for (int q = tid; q < nodes * 6; q += KD_THREADS) bb[q] = (q % 6) < 3 ? 0xffffffffu : 0u;
        __syncthreads();
        // ---- bounding box of every node's records
        for (int r = 0; r < E; ++r) {
            const int i = tid + T * r, f = live ? idx[r] : -1;
            uint32_t u[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, v[3] = {0u, 0u, 0u};  // (pads: neutral for min / max)
            if (f >= 0) {
                const float *p = t0 + (size_t)f * stride;
#pragma unroll
                for (int c = 0; c < 3; ++c) u[c] = v[c] = f2ord(p[c]);
            }
            if (S >= 64) {  // a wavefront's 64 consecutive positions lie in one node: reduce first, one atomic per value
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    for (int o = 32; o > 0; o >>= 1) {
                        u[c] = min(u[c], (uint32_t)__shfl_xor((int)u[c], o));
                        v[c] = max(v[c], (uint32_t)__shfl_xor((int)v[c], o));
                    }
                }
                if (live && (tid & 63) == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { atomicMin(&bb[(i / S) * 6 + c], u[c]); atomicMax(&bb[(i / S) * 6 + 3 + c], v[c]); }
                }
            } else if (f >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { atomicMin(&bb[(i / S) * 6 + c], u[c]); atomicMax(&bb[(i / S) * 6 + 3 + c], v[c]); }
            }
        }
        __syncthreads();
        // ---- keys: the coordinate along the node's longest axis (pads last)
        for (int r = 0; r < E; ++r) {
            const int i = tid + T * r, f = live ? idx[r] : -1;
            uint32_t k = 0xffffffffu;
            if (f >= 0) {
                const uint32_t *q = bb + (i / S) * 6;
                const float e0 = ord2f(q[3]) - ord2f(q[0]), e1 = ord2f(q[4]) - ord2f(q[1]), e2 = ord2f(q[5]) - ord2f(q[2]);
                const int ax = (e1 > e0 && e1 >= e2) ? 1 : ((e2 > e0 && e2 > e1) ? 2 : 0);
                k = f2ord(t0[(size_t)f * stride + ax]);
                if (k == 0xffffffffu) k = 0xfffffffeu;  // (a negative NaN pattern) keep real records in front of the pads
            }
            key[r] = k;
        }
        // ---- the network: every node of S positions ascending
        for (int k = 2; k <= S; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= T) {  // both elements in this lane's registers (static register indices: no scratch)
                    auto cswap = [&](uint32_t &ka, int32_t &ia, uint32_t &kb, int32_t &ib, int r) {
                        const int i = tid + T * r;
                        const bool asc = ((i & k) == 0) || k == S;
                        const bool gt = ka > kb || (ka == kb && (uint32_t)ia > (uint32_t)ib);
                        if (gt == asc) {
                            const uint32_t tk = ka; ka = kb; kb = tk;
                            const int32_t ti = ia; ia = ib; ib = ti;
                        }
                    };
                    if (j == T) {  // slots (0, 1) and (2, 3)
                        cswap(key[0], idx[0], key[1], idx[1], 0);
                        if (E == 4) cswap(key[2], idx[2], key[3], idx[3], 2);
                    } else {       // j == 2 T: slots (0, 2) and (1, 3)
                        cswap(key[0], idx[0], key[2], idx[2], 0);
                        cswap(key[1], idx[1], key[3], idx[3], 1);
                    }
                } else if (j < 64) {  // partner in the same wavefront
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r < E) {
                            const int i = tid + T * r;
                            const uint32_t kb = (uint32_t)__shfl_xor((int)key[r], j);
                            const int32_t ib = __shfl_xor(idx[r], j);
                            const bool asc = ((i & k) == 0) || k == S;
                            take(key[r], idx[r], kb, ib, ((i & j) == 0) == asc);
                        }
                    }
                } else {  // partner in another wavefront: through LDS
                    if (live)
                        for (int r = 0; r < E; ++r) { xkey[tid + T * r] = key[r]; xidx[tid + T * r] = idx[r]; }
                    __syncthreads();
                    if (live) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (r < E) {
                                const int i = tid + T * r;
                                const bool asc = ((i & k) == 0) || k == S;
                                take(key[r], idx[r], xkey[i ^ j], xidx[i ^ j], ((i & j) == 0) == asc);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
    if (live)
        for (int r = 0; r < E; ++r)
            if (w0 + tid + T * r < npad) order[(size_t)b * npad + w0 + tid + T * r] = max(idx[r], 0);
}

// ---- levels above one LDS window (clouds beyond 4096 positions) --------------------------------------------------
__global__ void kd_init_kernel(int32_t *__restrict__ idx_g, uint32_t *__restrict__ nodebb, int n, int P, int nbb) {
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) idx_g[(size_t)b * P + p] = p < n ? p : -1;
    if (p < nbb) nodebb[(size_t)b * nbb + p] = (p % 6) < 3 ? 0xffffffffu : 0u;
}
__global__ void kd_aabb_kernel(const float *__restrict__ tri, const int32_t *__restrict__ idx_g, uint32_t *__restrict__ nodebb,
                               int n, int P, int S, int nbb, int stride) {
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int f = idx_g[(size_t)b * P + p];
    if (f < 0) return;
    const float *r = tri + ((size_t)b * n + f) * stride;
    uint32_t *q = nodebb + (size_t)b * nbb + (size_t)(p / S) * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t u = f2ord(r[c]);
        atomicMin(&q[c], u);
        atomicMax(&q[3 + c], u);
    }
}
// keys of one level; the node boxes are reset for the next level by the last use (each thread clears nothing: the host
// re-initialises the table with kd_clear_kernel)
__global__ void kd_keys_kernel(const float *__restrict__ tri, const int32_t *__restrict__ idx_g, const uint32_t *__restrict__ nodebb,
                               uint32_t *__restrict__ key_g, int n, int P, int S, int nbb, int stride) {
    const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int f = idx_g[(size_t)b * P + p];
    uint32_t k = 0xffffffffu;
    if (f >= 0) {
        const uint32_t *q = nodebb + (size_t)b * nbb + (size_t)(p / S) * 6;
        const float e0 = ord2f(q[3]) - ord2f(q[0]), e1 = ord2f(q[4]) - ord2f(q[1]), e2 = ord2f(q[5]) - ord2f(q[2]);
        const int ax = (e1 > e0 && e1 >= e2) ? 1 : ((e2 > e0 && e2 > e1) ? 2 : 0);
        k = f2ord(tri[((size_t)b * n + f) * stride + ax]);
        if (k == 0xffffffffu) k = 0xfffffffeu;
    }
    key_g[(size_t)b * P + p] = k;
}
__global__ void kd_clear_kernel(uint32_t *__restrict__ nodebb, int total) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < total) nodebb[p] = (p % 6) < 3 ? 0xffffffffu : 0u;
}
// one stage (k, j >= KD_WIN) of the network in global memory
__global__ void kd_global_stage_kernel(uint32_t *__restrict__ key_g, int32_t *__restrict__ idx_g, int P, int k, int j, int S) {
    const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (P >> 1)) return;
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
    uint32_t *kk = key_g + (size_t)b * P;
    int32_t *ii = idx_g + (size_t)b * P;
    uint32_t ka = kk[i], kb = kk[l];
    int32_t ia = ii[i], ib = ii[l];
    cmpx(ka, ia, kb, ib, ((i & k) == 0) || k == S);
    kk[i] = ka; kk[l] = kb; ii[i] = ia; ii[l] = ib;
}
// stages k_lo .. k_hi with their strides below KD_WIN, inside LDS windows of KD_WIN positions
__global__ __launch_bounds__(KD_THREADS) void kd_local_stages_kernel(uint32_t *__restrict__ key_g, int32_t *__restrict__ idx_g,
                                                                     int P, int k_lo, int k_hi, int S) {
    __shared__ uint32_t key[KD_WIN];
    __shared__ int32_t idx[KD_WIN];
    const int b = blockIdx.y, w0 = blockIdx.x * KD_WIN;
    for (int p = threadIdx.x; p < KD_WIN; p += KD_THREADS) {
        key[p] = key_g[(size_t)b * P + w0 + p];
        idx[p] = idx_g[(size_t)b * P + w0 + p];
    }
    __syncthreads();
    lds_bitonic(key, idx, KD_WIN, w0, k_lo, k_hi, S);
    for (int p = threadIdx.x; p < KD_WIN; p += KD_THREADS) {
        key_g[(size_t)b * P + w0 + p] = key[p];
        idx_g[(size_t)b * P + w0 + p] = idx[p];
    }
}

static int kd_pow2(int n) {
    int P = SGT;
    while (P < n) P <<= 1;
    return P;
}
struct KdScratch {
    size_t idx, key, bb, total;
    int P, nbb;
    KdScratch(int B, int n) {
        P = kd_pow2(n);
        nbb = 6 * (P > KD_WIN ? P / (2 * KD_WIN) : 1);  // nodes of the smallest global level (S = 2 KD_WIN)
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
        idx = take(sizeof(int32_t) * (size_t)B * P);
        key = take(sizeof(uint32_t) * (size_t)B * P);
        bb = take(sizeof(uint32_t) * (size_t)B * nbb);
        total = o;
    }
};

extern "C" size_t rrl_cloud_order_workspace_bytes(int B, int n) {
    return KdScratch(B > 0 ? B : 0, n > 0 ? n : 0).total + 256;
}

static int cloud_order_impl(const float *tri, int stride, int32_t *order, void *ws, size_t ws_bytes, int B, int n, void *stream) {
    if (!tri || !order || !ws || B < 0 || n < 0 || n > SORT_CAP || B > 65535) return RRL_E_ARG;
    if (B == 0 || n == 0) return 0;
    const KdScratch L(B, n);
    if (ws_bytes < L.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    int32_t *idx_g = (int32_t *)((char *)ws + L.idx);
    uint32_t *key_g = (uint32_t *)((char *)ws + L.key), *bb = (uint32_t *)((char *)ws + L.bb);
    const int P = L.P, npad = (n + SGT - 1) / SGT * SGT;
    const dim3 gp((unsigned)((P + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(kd_init_kernel, gp, dim3(256), 0, s, idx_g, bb, n, P, L.nbb);
    for (int S = P; S > KD_WIN; S >>= 1) {  // levels whose windows exceed one workgroup's LDS
        hipLaunchKernelGGL(kd_aabb_kernel, gp, dim3(256), 0, s, tri, idx_g, bb, n, P, S, L.nbb, stride);
        hipLaunchKernelGGL(kd_keys_kernel, gp, dim3(256), 0, s, tri, idx_g, bb, key_g, n, P, S, L.nbb, stride);
        hipLaunchKernelGGL(kd_clear_kernel, dim3((unsigned)((B * L.nbb + 255) / 256)), dim3(256), 0, s, bb, B * L.nbb);
        const dim3 gw((unsigned)(P / KD_WIN), (unsigned)B), gh((unsigned)((P / 2 + 255) / 256), (unsigned)B);
        hipLaunchKernelGGL(kd_local_stages_kernel, gw, dim3(KD_THREADS), 0, s, key_g, idx_g, P, 2, KD_WIN, S);
        for (int k = 2 * KD_WIN; k <= S; k <<= 1) {
            for (int j = k >> 1; j >= KD_WIN; j >>= 1)
                hipLaunchKernelGGL(kd_global_stage_kernel, gh, dim3(256), 0, s, key_g, idx_g, P, k, j, S);
            hipLaunchKernelGGL(kd_local_stages_kernel, gw, dim3(KD_THREADS), 0, s, key_g, idx_g, P, k, k, S);
        }
    }
    const int W = P < KD_WIN ? P : KD_WIN;
    hipLaunchKernelGGL(kd_window_kernel, dim3((unsigned)(P / W), (unsigned)B), dim3(KD_THREADS), 0, s, tri, idx_g, order, n, npad, P, W, W, stride);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
extern "C" int rrl_cloud_order(const float *tri, int32_t *order, void *ws, size_t ws_bytes, int B, int n, void *stream) {
    return cloud_order_impl(tri, 9, order, ws, ws_bytes, B, n, stream);
}
// the same for point clouds pts [B][n][3] (the Chamfer monitor's inputs, rrl_chamfer_tree_fwd_ex)
extern "C" int rrl_cloud_order_points(const float *pts, int32_t *order, void *ws, size_t ws_bytes, int B, int n, void *stream) {
    return cloud_order_impl(pts, 3, order, ws, ws_bytes, B, n, stream);
}
