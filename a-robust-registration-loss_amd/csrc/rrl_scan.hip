// rrl_scan.hip -- K1: dense line <-> pseudo-triangle scan (code/loss.py:68-112, 181-186)
// and K1': prepared triangles (code/loss.py:94-110).
//
// Decomposition (gfx950, 64-wide waves):
//   lane  = R lines held in VGPRs, two per register pair (v_pk_mul_f32 / v_pk_add_f32)
//   wave  = walks a chunk of triangles; a triangle is WAVE-UNIFORM, so its 10 floats arrive
//           through the scalar cache (s_load_dwordx8 + x2) into SGPRs and cost no VGPRs, no
//           LDS and no vector-memory issue slots
//   block = 256 lanes (4 waves) x `chunk` triangles; grid = line tiles x chunks x (2 clouds x B)
//
// strict loop, per (line, triangle): 48 fp32 ops for the three squared distances + 3 integer
// ops (v_max3_u32, v_cmp_lt_u32, v_max3_u32).  A hit is "all three x_k < thr2" which, for
// x_k >= 0, is one unsigned compare of max3(x) against thr2 (non-negative floats order like
// their bit patterns); a negative x_k (the reference's NaN -> exit(0), loss.py:89-91) has its
// sign bit set, can never hit, and is caught by the running unsigned max.
//
// lazy loop: only point 0 (16 fp32 ops) for every pair; points 1 and 2 are evaluated for a
// register pair only in the rare wave-iterations where some lane's point 0 passes.  The
// labels are identical by construction ("all three pass" needs "point 0 passes").
//
// auto mode (default) picks the lazy loop per wavefront when a NaN is PROVABLY impossible for
// all of its lines, so strict's NaN detection is preserved exactly (DESIGN.md §"NaN bound"):
//   with u = 2^-24, |dir|^2 <= 1 + 19.8u and |a|^2 = |P - x0|^2 <= 100:
//   dAC - proj >= -30u |a|^2 >= -1.8e-4 > -2e-4, hence (dAC - proj) + 2e-4 > 0.
//
// Hits are rare (~1e-4 per pair): a lane that finds one bumps count[line] atomically and
// stores the triangle index in the first free slot; the consumer sorts the <= 4 indices,
// which restores nonzero() order (loss.py:125-131) deterministically.
// Nothing of size L*N is ever written: compulsory traffic is a few MB against ~35 GFLOP
// (B=8, N=M=4096, L=10000), so the kernel is fp32-VALU-bound, not HBM-bound.
#include <dlfcn.h>
#include <stdlib.h>

#include "rrl_ws.h"

// ---- roctx ranges (rrl_common.h) ----------------------------------------------------------------
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static int g_roctx_state = -1;  // -1 unknown, 0 off, 1 on
static void roctx_init() {
    g_roctx_state = 0;
    const char *e = getenv("RRL_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) { g_roctx_state = 1; return; }
    }
}
void rrl_range_push(const char *name) {
    if (g_roctx_state < 0) roctx_init();
    if (g_roctx_state == 1) g_roctx_push(name);
}
void rrl_range_pop(void) {
    if (g_roctx_state == 1) g_roctx_pop();
}

// ---------------------------------------------------------------------------------------
// K1' prepared triangles (+ max |P|^2 per cloud and sample for the auto-mode NaN bound)
// ---------------------------------------------------------------------------------------
// 64 triangles per workgroup, staged through LDS so that both the 36-byte input rows and the
// 48-byte output records move with fully coalesced accesses.
// Only the LEGACY instantiation is launched (clouds too large to sort; state cleared by a memset,
// atomicMax on PMAX): the usual, sorted path builds the records inside tri_build_kernel
// (rrl_cull.hip) together with the transform, the sort and the group spheres.
template <bool LEGACY>
__global__ __launch_bounds__(64) void tri_prepare_kernel(const float *__restrict__ tri1,
                                                         const float *__restrict__ tri2,
                                                         float *__restrict__ ptri1,
                                                         float *__restrict__ ptri2,
                                                         uint32_t *__restrict__ pmax,
                                                         uint4 *__restrict__ zero_base,
                                                         size_t zero_vec4, size_t skip_lo,
                                                         size_t skip_hi, int B, int N, int M) {
    __shared__ float sin_[64 * 9];
    __shared__ float sout[64 * PTRI_STRIDE];
    const int cloud = blockIdx.z, b = blockIdx.y, tid = threadIdx.x;
    if constexpr (!LEGACY) {
        const size_t nblk = (size_t)gridDim.x * gridDim.y * gridDim.z;
        const size_t me = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (size_t i = me * 64 + tid; i < zero_vec4; i += nblk * 64)
            if (i < skip_lo || i >= skip_hi) zero_base[i] = make_uint4(0, 0, 0, 0);
    }
    const int n = cloud ? M : N;
    const int f0 = blockIdx.x * 64;
    if (f0 >= n) return;
    const int cnt = min(64, n - f0);
    const float *src = (cloud ? tri2 : tri1) + 9 * ((size_t)b * n + f0);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = k * 64 + tid;
        sin_[i] = i < cnt * 9 ? src[i] : 0.0f;
    }
    __syncthreads();
    float c[9], thr, x, p2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = sin_[tid * 9 + i];
    tri_thresholds(c, &thr, &x);  // code/loss.py:94-110
#pragma unroll
    for (int i = 0; i < 9; ++i) sout[tid * PTRI_STRIDE + i] = c[i];
    sout[tid * PTRI_STRIDE + 9] = x;
    sout[tid * PTRI_STRIDE + 10] = thr;
    sout[tid * PTRI_STRIDE + 11] = __int_as_float(f0 + tid);  // original index
    __syncthreads();
    float *dst = (cloud ? ptri2 : ptri1) + PTRI_STRIDE * ((size_t)b * n + f0);
#pragma unroll
    for (int k = 0; k < PTRI_STRIDE; ++k) {
        const int i = k * 64 + tid;
        if (i < cnt * PTRI_STRIDE) dst[i] = sout[i];
    }
    if constexpr (LEGACY) {
        if (tid < cnt) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                p2 = fmaxf(p2, c[3 * k] * c[3 * k] + c[3 * k + 1] * c[3 * k + 1] + c[3 * k + 2] * c[3 * k + 2]);
            if (!(p2 <= 3.0e38f)) p2 = INFINITY;  // NaN/inf coordinates: never "provably safe"
        }
        for (int o = 32; o > 0; o >>= 1) p2 = fmaxf(p2, __shfl_down(p2, o));
        if (tid == 0) atomicMax(&pmax[cloud * B + b], __float_as_uint(p2));
    }
}

int rrl_launch_pmax_from_partials(void *ws, const WsLayout &w, int B, int N, int M, int clouds, hipStream_t s);
int rrl_launch_tri_build(const float *tri1, const float *tri2, void *ws, const WsLayout &w, int B,
                         int N, int M, int clouds, const RrlXform *xf, const float *line, int L, const RrlCall &o,
                         hipStream_t s);
int rrl_launch_cull_scan(const float *line, void *ws, const WsLayout &w, int B, int N, int M, int L,
                         int clouds, int lmax_ready, const RrlCall &o, hipStream_t s);
int rrl_sort_capacity(void);

// clouds = 2: both clouds; clouds = 1: the source only (the target's scan results are carried
// over from an earlier call with the same target and lines, see rrl_loss_forward_cached).  The
// sorted/legacy decision always looks at both sizes so that a cached call takes the same path.
// xf != NULL: the source is xf->src moved by (xf->R, xf->t); the moved triangles land in TRI1
// (`tri1` is ignored).  Sorted path: ONE launch (tri_build_kernel) does transform + records +
// state clearing + sort + spheres.  Legacy path (a cloud > 65536 triangles): rigid apply,
// memset, tri_prepare_kernel<LEGACY>.
// line != NULL: the records kernel also reduces the samples' lines to the partial maxima the culled scan derives its
// slacks from (the fused forwards); NULL: the scan entry does that itself (rrl_tri_prepare + rrl_line_tri_scan).
int rrl_tri_prepare_clouds(const float *tri1, const float *tri2, void *ws, size_t ws_bytes, int B,
                           int N, int M, int L, int clouds, const RrlXform *xf, const float *line, const RrlCall &o,
                           void *stream) {
    if ((!tri1 && !xf) || !tri2 || !ws || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    const bool sorted = (N > M ? N : M) <= rrl_sort_capacity();
    const int nmax = clouds == 2 && M > N ? M : N;
    uint4 *zb = (uint4 *)((char *)ws + w.off[RRL_WS_STATUS]);
    if (!sorted || B == 0 || nmax == 0) {
        // one fill clears status, nvals, nsel, pmax, count1, count2 (contiguous by construction)
        int rc = rrl_fill(zb, 0u, w.zero_bytes, s);
        if (rc) return rc;
        // ... and one more the tiled reduce's state (MHIST, MCTL, MSUM)
        if ((rc = rrl_fill((char *)ws + w.state_off, 0u, w.state_bytes, s))) return rc;
        if (o.clear_ptr && (rc = rrl_fill(o.clear_ptr, 0u, o.clear_bytes, s))) return rc;
    }
    if (B == 0 || nmax == 0) return 0;
    if (sorted) return rrl_launch_tri_build(tri1, tri2, ws, w, B, N, M, clouds, xf, line, L, o, s);
    if (xf) {
        if (xf->zero_g1) {
            int rc = rrl_fill(w.f32(ws, RRL_WS_GACC), 0u, w.off[RRL_WS_KJC] - w.off[RRL_WS_GACC], s);
            if (rc) return rc;
        }
        tri1 = w.f32(ws, RRL_WS_TRI1);
        int rc = rrl_rigid_apply_fwd(xf->src, xf->R, xf->t, w.f32(ws, RRL_WS_TRI1), B, 3 * N,
                                     xf->transpose_r, 0, stream);
        if (rc) return rc;
    }
    dim3 grid((unsigned)((nmax + 63) / 64), (unsigned)B, (unsigned)clouds);
    hipLaunchKernelGGL(tri_prepare_kernel<true>, grid, dim3(64), 0, s, tri1, tri2,
                       w.f32(ws, RRL_WS_PTRI1), w.f32(ws, RRL_WS_PTRI2),
                       (uint32_t *)w.i32(ws, RRL_WS_PMAX), zb, (size_t)0, (size_t)0, (size_t)0, B, N, M);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_tri_prepare_ex(const float *tri1, const float *tri2, void *ws, size_t ws_bytes,
                                  int B, int N, int M, int L, const rrl_opts *opts, void *stream) {
    RrlCall o = rrl_resolve_opts(opts);
    if (o.prepared() && (!o.order2 || (N > M ? N : M) > rrl_sort_capacity())) o.order1 = o.order2 = nullptr;  // both orders, sorted layout
    o.flags &= ~RRL_F_TARGET_KEPT;  // a stage call builds both clouds
    int rc = rrl_tri_prepare_clouds(tri1, tri2, ws, ws_bytes, B, N, M, L, 2, nullptr, nullptr, o, stream);
    if (rc || !o.prepared() || B <= 0 || (N <= 0 && M <= 0)) return rc;
    // prepared build: PMAX is normally reduced by the culled scan's prologue; a stage call leaves it complete itself
    return rrl_launch_pmax_from_partials(ws, WsLayout(B, N, M, L), B, N, M, 2, (hipStream_t)stream);
}
extern "C" int rrl_tri_prepare(const float *tri1, const float *tri2, void *ws, size_t ws_bytes,
                               int B, int N, int M, int L, void *stream) {
    return rrl_tri_prepare_ex(tri1, tri2, ws, ws_bytes, B, N, M, L, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// K1 scan
// ---------------------------------------------------------------------------------------
typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) {
    return max(max(a, b), c);
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    return min(min(a, b), c);
}

template <typename T>
struct Lanes;
template <>
struct Lanes<float> {
    static constexpr int W = 1;
    static __device__ __forceinline__ float get(float v, int) { return v; }
};
template <>
struct Lanes<v2f> {
    static constexpr int W = 2;
    static __device__ __forceinline__ float get(v2f v, int i) { return i ? v.y : v.x; }
};

template <typename T, int NP>
struct LineRegs {
    T ux[NP], uy[NP], uz[NP], ox[NP], oy[NP], oz[NP];
};

struct HitSink {
    int32_t *cnt, *hit;
    int lbase, L;
    __device__ __forceinline__ void commit(int r, int t) const {
        const int l = lbase + r * 256;
        if (l < L) {
            int pos = atomicAdd(&cnt[l], 1);
            if (pos < RRL_MAX_HITS) hit[(size_t)l * RRL_MAX_HITS + pos] = t;
        }
    }
};

#define DSQ(px, py, pz, i) dist_sq<T>(px, py, pz, r.ux[i], r.uy[i], r.uz[i], r.ox[i], r.oy[i], r.oz[i])

// every (line, triangle) pair fully evaluated; returns the running unsigned max (NaN witness)
template <typename T, int NP>
__device__ __forceinline__ uint32_t scan_strict(const LineRegs<T, NP> &r, kptr tp, int t0, int t1,
                                                const HitSink &sink) {
    constexpr int W = Lanes<T>::W, R = W * NP;
    uint32_t nanacc = 0;
    for (int t = t0; t < t1; ++t, tp += PTRI_STRIDE) {
        const float p0x = tp[0], p0y = tp[1], p0z = tp[2];
        const float p1x = tp[3], p1y = tp[4], p1z = tp[5];
        const float p2x = tp[6], p2y = tp[7], p2z = tp[8];
        const uint32_t thr2 = __float_as_uint(tp[9]);
        bool any = false;
        uint32_t m[R];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            T x0 = DSQ(p0x, p0y, p0z, i), x1 = DSQ(p1x, p1y, p1z, i), x2 = DSQ(p2x, p2y, p2z, i);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint32_t mm = umax3(f2u(Lanes<T>::get(x0, w)), f2u(Lanes<T>::get(x1, w)),
                                    f2u(Lanes<T>::get(x2, w)));
                m[i * W + w] = mm;
                nanacc = max(nanacc, mm);
                any |= mm < thr2;
            }
        }
        if (__builtin_expect(any, 0)) {
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (m[q] < thr2) sink.commit(q, __float_as_int(tp[11]));
        }
    }
    return nanacc;
}

// point 0 for every pair; points 1, 2 per register pair only where some lane's point 0 passes
template <typename T, int NP, bool TRACK_NAN>
__device__ __forceinline__ uint32_t scan_lazy(const LineRegs<T, NP> &r, kptr tp, int t0, int t1,
                                              const HitSink &sink) {
    constexpr int W = Lanes<T>::W, R = W * NP;
    uint32_t nanacc = 0;
    for (int t = t0; t < t1; ++t, tp += PTRI_STRIDE) {
        const float p0x = tp[0], p0y = tp[1], p0z = tp[2];
        const uint32_t thr2 = __float_as_uint(tp[9]);
        uint32_t m[R];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            T x0 = DSQ(p0x, p0y, p0z, i);
#pragma unroll
            for (int w = 0; w < W; ++w) m[i * W + w] = f2u(Lanes<T>::get(x0, w));
        }
        // one compare per triangle: the smallest point-0 value of the lane's R lines
        uint32_t lo = m[0];
        if constexpr (TRACK_NAN) {
            uint32_t hi = m[0];
#pragma unroll
            for (int q = 1; q < R; ++q) { lo = min(lo, m[q]); hi = max(hi, m[q]); }
            nanacc = max(nanacc, hi);
        } else {
#pragma unroll
            for (int q = 1; q + 1 < R; q += 2) lo = umin3(lo, m[q], m[q + 1]);
            if constexpr ((R & 1) == 0) lo = min(lo, m[R - 1]);
        }
        if (__builtin_expect(__any(lo < thr2), 0)) {
            const float p1x = tp[3], p1y = tp[4], p1z = tp[5];
            const float p2x = tp[6], p2y = tp[7], p2z = tp[8];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                bool pass = false;
#pragma unroll
                for (int w = 0; w < W; ++w) pass |= m[i * W + w] < thr2;
                if (!__any(pass)) continue;
                T x1 = DSQ(p1x, p1y, p1z, i), x2 = DSQ(p2x, p2y, p2z, i);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint32_t mm = umax3(m[i * W + w], f2u(Lanes<T>::get(x1, w)),
                                        f2u(Lanes<T>::get(x2, w)));
                    if constexpr (TRACK_NAN) nanacc = max(nanacc, mm);
                    if (mm < thr2) sink.commit(i * W + w, __float_as_int(tp[11]));
                }
            }
        }
    }
    return nanacc;
}

// T = float (1 line per register) or v2f (2 lines per register pair); NP registers per lane.
template <typename T, int NP>
__global__ __launch_bounds__(256) void scan_kernel(
    const float *__restrict__ ptri1, const float *__restrict__ ptri2,
    const float *__restrict__ line, int32_t *__restrict__ count1, int32_t *__restrict__ hit1,
    int32_t *__restrict__ count2, int32_t *__restrict__ hit2, int32_t *__restrict__ status,
    const uint32_t *__restrict__ pmax, int B, int N, int M, int L,
    int chunk, int mode) {
    constexpr int W = Lanes<T>::W;
    constexpr int R = W * NP;  // lines per lane
    const int z = blockIdx.z;
    const int cloud = z >= B ? 1 : 0;
    const int b = z - cloud * B;
    const int n = cloud ? M : N;
    const int t0 = blockIdx.y * chunk;
    if (t0 >= n) return;
    const int t1 = min(n, t0 + chunk);
    const float *tri = (cloud ? ptri2 : ptri1) + (size_t)b * n * PTRI_STRIDE;
    const float *ln = line + (size_t)b * L * 6;
    HitSink sink;
    sink.cnt = (cloud ? count2 : count1) + (size_t)b * L;
    sink.hit = (cloud ? hit2 : hit1) + (size_t)b * L * RRL_MAX_HITS;
    sink.L = L;
    // lines of this lane: l = tile*256*R + r*256 + tid (adjacent lanes = adjacent lines)
    sink.lbase = blockIdx.x * (256 * R) + threadIdx.x;

    LineRegs<T, NP> r;
    bool safe = true;
    const float pm = __uint_as_float(pmax[cloud * B + b]);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float v[W][6];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            int l = sink.lbase + (i * W + w) * 256;
            // out-of-range lanes scan the all-zero line (a legal input) and never write
            const float *q = ln + 6 * (size_t)(l < L ? l : 0);
            float keep = l < L ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 6; ++c) v[w][c] = q[c] * keep;
            safe &= rrl_line_safe(v[w], pm);  // NaN-impossibility bound (see header)
        }
        if constexpr (W == 1) {
            r.ux[i] = v[0][0]; r.uy[i] = v[0][1]; r.uz[i] = v[0][2];
            r.ox[i] = v[0][3]; r.oy[i] = v[0][4]; r.oz[i] = v[0][5];
        } else {
            r.ux[i] = (v2f){v[0][0], v[1][0]}; r.uy[i] = (v2f){v[0][1], v[1][1]};
            r.uz[i] = (v2f){v[0][2], v[1][2]}; r.ox[i] = (v2f){v[0][3], v[1][3]};
            r.oy[i] = (v2f){v[0][4], v[1][4]}; r.oz[i] = (v2f){v[0][5], v[1][5]};
        }
    }

    kptr tp = (kptr)(uintptr_t)(tri + (size_t)t0 * PTRI_STRIDE);
    uint32_t nanacc;
    if (mode == RRL_SCAN_STRICT || (mode == RRL_SCAN_AUTO && !__all(safe)))
        nanacc = scan_strict<T, NP>(r, tp, t0, t1, sink);
    else if (mode == RRL_SCAN_AUTO)
        nanacc = scan_lazy<T, NP, false>(r, tp, t0, t1, sink);  // provably NaN-free
    else
        nanacc = scan_lazy<T, NP, true>(r, tp, t0, t1, sink);
    if (nanacc >= 0x80000000u) atomicOr(&status[0], 1);
}

static int g_scan_variant = 0;  // 0 = default; else lines per lane (1, 2, 4, 8)
int rrl_default_scan_variant(void) { return g_scan_variant; }

extern "C" int rrl_set_scan_variant(int lines_per_lane) {
    if (lines_per_lane != 0 && lines_per_lane != 1 && lines_per_lane != 2 && lines_per_lane != 4 &&
        lines_per_lane != 8)
        return RRL_E_ARG;
    g_scan_variant = lines_per_lane;
    return 0;
}

// ---- optional scan timing ring -----------------------------------------------------------
#define TIMING_RING 1024
static int g_timing_on = 0, g_timing_n = 0, g_timing_seen = 0;
static hipEvent_t g_ev[TIMING_RING][2];
static bool g_ev_made = false;

extern "C" int rrl_scan_timing_enable(int on) {
    if (on && !g_ev_made) {
        for (int i = 0; i < TIMING_RING; ++i)
            for (int k = 0; k < 2; ++k) {
                hipError_t e = hipEventCreate(&g_ev[i][k]);
                if (e != hipSuccess) return (int)e;
            }
        g_ev_made = true;
    }
    g_timing_on = on > 0 ? on : 0;  // on = k: bracket every k-th scan launch
    g_timing_n = g_timing_seen = 0;
    return 0;
}

extern "C" int rrl_scan_timing_collect(float *ms, int max_n) {
    int n = g_timing_n < TIMING_RING ? g_timing_n : TIMING_RING;
    if (n > max_n) n = max_n;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_ev[i][1]) != hipSuccess) return 0;
        if (hipEventElapsedTime(&ms[i], g_ev[i][0], g_ev[i][1]) != hipSuccess) return 0;
    }
    g_timing_n = 0;
    return n;
}

int rrl_line_tri_scan_clouds(const float *line, void *ws, size_t ws_bytes, int B, int N, int M, int L,
                             int mode, int chunk, int clouds, int lmax_ready, const RrlCall &o, void *stream) {
    if (!line || !ws || B < 0 || N < 0 || M < 0 || L < 0 || chunk < 0) return RRL_E_ARG;
    if (mode != RRL_SCAN_STRICT && mode != RRL_SCAN_LAZY && mode != RRL_SCAN_AUTO &&
        mode != RRL_SCAN_CULL)
        return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0 || L == 0 || (N == 0 && (M == 0 || clouds == 1))) return 0;
    const int nmax0 = N > M ? N : M;
    if (mode == RRL_SCAN_CULL && nmax0 > rrl_sort_capacity()) mode = RRL_SCAN_AUTO;
    hipStream_t s = (hipStream_t)stream;
    const bool timed = g_timing_on && (g_timing_seen++ % g_timing_on) == 0 && g_timing_n < TIMING_RING;
    if (mode == RRL_SCAN_CULL) {  // one launch: sphere-culled scan with an inline strict fallback
        if (timed) (void)hipEventRecord(g_ev[g_timing_n][0], s);
        int rc = rrl_launch_cull_scan(line, ws, w, B, N, M, L, clouds, lmax_ready, o, s);
        if (rc) return rc;
        if (timed) (void)hipEventRecord(g_ev[g_timing_n++][1], s);
        return 0;
    }
    int R = o.scan_variant;
    if (R == 0) {
        const char *v = nullptr;
#ifdef RRL_EXPERIMENT  // (sweeps of the dense scans' variants: experimental builds; rrl_set_scan_variant / rrl_opts.scan_variant ship)
        v = getenv("RRL_SCAN_VARIANT");
#endif
        R = v ? atoi(v) : 0;
        if (R != 1 && R != 2 && R != 4 && R != 8) R = 4;  // measured best for every mode (profiles/r02_scan_sweep.jsonl)
    }
    if (chunk == 0) {
        const char *c = nullptr;
#ifdef RRL_EXPERIMENT
        c = getenv("RRL_SCAN_CHUNK");
#endif
        chunk = c ? atoi(c) : 0;
        if (chunk <= 0) chunk = mode == RRL_SCAN_STRICT ? 64 : 128;
    }
    const int nmax = clouds == 2 && M > N ? M : N;
    dim3 grid((unsigned)((L + 256 * R - 1) / (256 * R)), (unsigned)((nmax + chunk - 1) / chunk),
              (unsigned)(clouds * B));
    if (timed) (void)hipEventRecord(g_ev[g_timing_n][0], s);
#define RRL_SCAN_LAUNCH(T, NP)                                                                   \
    hipLaunchKernelGGL((scan_kernel<T, NP>), grid, dim3(256), 0, s, w.f32(ws, RRL_WS_PTRI1),     \
                       w.f32(ws, RRL_WS_PTRI2), line, w.i32(ws, RRL_WS_COUNT1),                  \
                       w.i32(ws, RRL_WS_HIT1), w.i32(ws, RRL_WS_COUNT2), w.i32(ws, RRL_WS_HIT2), \
                       w.i32(ws, RRL_WS_STATUS), (const uint32_t *)w.i32(ws, RRL_WS_PMAX), B, N, \
                       M, L, chunk, mode)
    if (R == 1) RRL_SCAN_LAUNCH(float, 1);
    else if (R == 2) RRL_SCAN_LAUNCH(v2f, 1);
    else if (R == 4) RRL_SCAN_LAUNCH(v2f, 2);
    else RRL_SCAN_LAUNCH(v2f, 4);
#undef RRL_SCAN_LAUNCH
    if (timed) (void)hipEventRecord(g_ev[g_timing_n++][1], s);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_line_tri_scan_ex(const float *line, void *ws, size_t ws_bytes, int B, int N, int M,
                                    int L, int mode, int chunk, const rrl_opts *opts, void *stream) {
    RrlCall o = rrl_resolve_opts(opts);
    if (o.prepared() && (!o.order2 || (N > M ? N : M) > rrl_sort_capacity())) o.order1 = o.order2 = nullptr;
    return rrl_line_tri_scan_clouds(line, ws, ws_bytes, B, N, M, L, mode, chunk, 2, 0, o, stream);
}
extern "C" int rrl_line_tri_scan(const float *line, void *ws, size_t ws_bytes, int B, int N, int M,
                                 int L, int mode, int chunk, void *stream) {
    return rrl_line_tri_scan_ex(line, ws, ws_bytes, B, N, M, L, mode, chunk, nullptr, stream);
}
