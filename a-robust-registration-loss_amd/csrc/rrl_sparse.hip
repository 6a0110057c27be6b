// rrl_sparse.hip -- everything after the dense scan, plus the fused forward/backward entries.
//   K2 line_pair_dist   code/loss.py:115-167   (one lane per line, <= 4x4 block in registers)
//   K3+K4 loss_reduce   code/loss.py:223-230   (radix-select median + Welsch min/mean, one
//                                               workgroup per sample, fixed-point bucket sums)
//   K5 backward         autograd of code/loss.py:170-232
// About 9 % of the lines are selected; these kernels touch O(L) data and are latency bound
// (a few microseconds each) next to the O(L*(N+M)) scan.
#include <stdlib.h>
#include <string.h>

#include "rrl_ws.h"

#define FIX_SHIFT 40  // bucket sums in 2^-40 fixed point: order-independent, bit-deterministic

// sqrt(dist_sq) of the three points of triangle f and the detached weights of
// code/loss.py:92: w_k = d_k / ((d0 + d1) + d2).  Same arithmetic as the scan, so the
// distances are bit-identical to the ones that decided the label.
#ifdef RRL_STAMPS  // experiments only (RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp): 100 MHz time stamps of workgroup 0's lane 0
__device__ unsigned long long g_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_stamps[i] = wall_clock64(); } while (0)
extern "C" int rrl_debug_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_wstamps[12 * 2048];  // [stamp][workgroup]: per-workgroup stamps of the tail kernel
#define STAMPW(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 2048u) g_wstamps[(i) * 2048 + wg_] = wall_clock64(); } } while (0)
extern "C" int rrl_debug_wstamps(unsigned long long *out, int clear) {
    if (clear) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_wstamps)) != hipSuccess) return -1; return hipMemset(p_, 0, sizeof(unsigned long long) * 12 * 2048) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * 12 * 2048) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_pstamps[8 * 2048];  // ... of the per-line stage
#define STAMPP(i) do { if ((threadIdx.x & 63) == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 2048u) g_pstamps[(i) * 2048 + wg_] = wall_clock64(); } } while (0)
extern "C" int rrl_debug_pstamps(unsigned long long *out, int clear) {
    if (clear) { void *p_ = nullptr; if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_pstamps)) != hipSuccess) return -1; return hipMemset(p_, 0, sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -1;
}
#define STAMPC(i) STAMPP(i)  // (the sampler's count pass, rrl_sampler.h: stamps 6, 7 of the same table)
#else
#define STAMP(i)
#define STAMPW(i)
#define STAMPP(i)
#endif
#include "rrl_sampler.h"  // the sampler's count pass as a device function: pair_count_kernel carries it

#include "rrl_stage_pair.inc"    // K2: the per-line stage

// tar_ws != NULL: cloud 2's hit counts / hit lists are read where its scan left them -- the workspace of the evaluation the
// target was carried over from (round 4b: they used to be copied into this workspace first, two launches per evaluation).
static PairArgs pair_args(const float *tri1, const float *tri2, const float *line, void *ws, const WsLayout &w, int B, int N,
                          int M, int L, int s_m, int s_n, int e_m, int e_n, bool tally = true, const void *tar_ws = nullptr,
                          int Bt = 0) {
    PairArgs a;
    a.mhist = tally ? w.u32(ws, RRL_WS_MHIST) : nullptr;
    a.mctl = tally ? w.u32(ws, RRL_WS_MCTL) : nullptr;
    a.tri1 = tri1;
    a.tri2 = tri2;
    a.line = line;
    a.count1 = w.i32(ws, RRL_WS_COUNT1); a.hit1 = w.i32(ws, RRL_WS_HIT1);
    a.count2 = tar_ws ? w.i32(tar_ws, RRL_WS_COUNT2) : w.i32(ws, RRL_WS_COUNT2);
    a.hit2 = tar_ws ? w.i32(tar_ws, RRL_WS_HIT2) : w.i32(ws, RRL_WS_HIT2);
    a.kj = w.u8(ws, RRL_WS_KJ);
    a.sel_out = w.i32(ws, RRL_WS_SEL); a.nsel = w.i32(ws, RRL_WS_NSEL);
    a.hs1 = w.i32(ws, RRL_WS_HS1); a.hs2 = w.i32(ws, RRL_WS_HS2);
    a.w1 = w.f32(ws, RRL_WS_W1); a.w2 = w.f32(ws, RRL_WS_W2);
    a.Q1 = (float4 *)w.f32(ws, RRL_WS_Q1); a.Q2 = (float4 *)w.f32(ws, RRL_WS_Q2);
    a.D = w.f32(ws, RRL_WS_D); a.dc = w.f32(ws, RRL_WS_VALS);
    a.kjc = w.u8(ws, RRL_WS_KJC); a.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    a.lidc = w.u32(ws, RRL_WS_LIDC);
    a.vlist = w.f32(ws, RRL_WS_VLIST); a.vlcnt = w.i32(ws, RRL_WS_VLCNT);
    a.B = B; a.N = N; a.M = M; a.L = L;
    a.s_m = s_m; a.s_n = s_n; a.e_m = e_m; a.e_n = e_n;
    a.st1 = 9; a.st2 = 9;
    a.Bt = Bt;  // multi-pose (RrlCall::problems)
    a.xcd_align = B % 8 == 0 && xcd_align_on();
    a.zc1 = a.zc2 = nullptr;
    return a;
}

static int reduce_kind(int mode, int B, int nblk, int pool, bool with_bwd);

// with_bwd: the reduce that follows will carry the direct backward (rrl_registration_step) -- it decides, with the shape,
// whether the tail kernel runs and wants the dense value lists
static int line_pair_dist_impl(const float *tri1, const float *tri2, const float *line, void *ws, size_t ws_bytes, int B,
                               int N, int M, int L, int s_m, int s_n, int e_m, int e_n, int pool,
                               const RrlCall &o, void *stream, bool with_bwd = false) {
    if (!tri1 || !tri2 || !line || !ws || B < 0 || N < 0 || M < 0 || L < 0 || L >= (1 << 24)) return RRL_E_ARG;  // 24-bit line ids in LDS
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0 || L == 0) return 0;
    PairArgs pa = pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, true, o.tar_ws);
    pa.Bt = o.problems;
    if (o.leave_clean && !o.problems && !o.tar_ws) { pa.zc1 = w.i32(ws, RRL_WS_COUNT1); pa.zc2 = w.i32(ws, RRL_WS_COUNT2); }
    if (reduce_kind(o.reduce_mode, B, (L + 1023) / 1024, pool, with_bwd) != 2) pa.vlist = nullptr;  // only the tail kernel reads VLIST
    if (RrlCountRider *cr = o.count_rider) {  // the next epoch's count pass rides along (pair_count_kernel)
        const int ctiles = (cr->n + 1023) / 1024;
        if (B == 1 && cr->rounds > 0 && cr->n > 0 && (long)ctiles * cr->rounds < 512 && cr->rows && cr->n_rows > 0) {
            const CountKArgs c = {cr->rng_state, cr->r, cr->centers, cr->aabb2, cr->rows, cr->accept, cr->n_rows, cr->n,
                                  cr->rounds, rrl_sample_prefilter(), ctiles, cr->rounds};
            const int pgx = (L + 1023) / 1024;
            hipLaunchKernelGGL(pair_count_kernel, dim3((unsigned)(ctiles * cr->rounds + pgx * B)), dim3(1024), 0,
                               (hipStream_t)stream, pa, c, pgx);
            RRL_LAUNCH_CHECK();
            cr->done = 1;
            return 0;
        }
    }
    hipLaunchKernelGGL(line_pair_dist_kernel, dim3((unsigned)((L + 1023) / 1024), (unsigned)B), dim3(1024), 0,
                       (hipStream_t)stream, pa);
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_line_pair_dist_ex(const float *tri1, const float *tri2, const float *line,
                                     void *ws, size_t ws_bytes, int B, int N, int M, int L, int s_m,
                                     int s_n, int e_m, int e_n, int pool, const rrl_opts *opts, void *stream) {
    return line_pair_dist_impl(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m, e_n, pool, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line,
                                  void *ws, size_t ws_bytes, int B, int N, int M, int L, int s_m,
                                  int s_n, int e_m, int e_n, int pool, void *stream) {
    return rrl_line_pair_dist_ex(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, stream);
}

#include "rrl_stage_reduce.inc"  // K3 + K4: single-workgroup and exchange reduce
#include "rrl_stage_tail.inc"    // K3 + K4 (+ K5): the tail kernel; the single-tile forward kernel

// The DEFAULT reduce mode (include/rrl.h rrl_set_reduce_mode; a call's rrl_opts.reduce_mode overrides it): 0 auto, 1 single,
// 2 tiled (the tail kernel wherever legal), 3 xchg (the exchange kernel wherever legal); reduce_kind() below turns a mode
// and a shape into the kernel.  Env RRL_REDUCE=single|tiled|xchg.
static int g_reduce_mode = -1;  // -1: read RRL_REDUCE once
extern "C" int rrl_set_reduce_mode(int mode) {
    if (mode < 0 || mode > 3) return RRL_E_ARG;
    g_reduce_mode = mode;
    return 0;
}
static int default_reduce_mode() {
    if (g_reduce_mode < 0) {
        const char *e = getenv("RRL_REDUCE");
        g_reduce_mode = !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 't' ? 2 : (e[0] == 'x' ? 3 : 0)));
    }
    return g_reduce_mode;
}
static bool default_deterministic();
// include/rrl.h rrl_opts -> the options of one call (csrc/rrl_ws.h RrlCall).  Fields the caller's struct does not
// reach (struct_bytes), -1 and NULL mean the process-wide default.
RrlCall rrl_resolve_opts(const rrl_opts *p) {
    RrlCall o;
    o.clear_ptr = nullptr;
    o.clear_bytes = 0;
    o.rider = nullptr;  // (set below from rrl_opts.chamfer)
    o.count_rider = nullptr;
    o.write_rider = nullptr;
    o.tar_ws = nullptr;
    rrl_opts v;
    memset(&v, 0, sizeof v);
    v.reduce_mode = v.deterministic = v.sort_parts = v.scan_variant = -1;
    if (p && p->struct_bytes >= 8) memcpy(&v, p, (size_t)p->struct_bytes < sizeof v ? (size_t)p->struct_bytes : sizeof v);
    o.flags = v.flags;
    o.reduce_mode = v.reduce_mode >= 0 && v.reduce_mode <= 3 ? v.reduce_mode : default_reduce_mode();
    o.deterministic = v.deterministic >= 0 ? (v.deterministic ? 1 : 0) : (default_deterministic() ? 1 : 0);
    o.sort_parts = v.sort_parts >= 0 && v.sort_parts <= 16 ? v.sort_parts : rrl_default_sort_parts();
    const int sv = v.scan_variant;
    o.scan_variant = (sv == 0 || sv == 1 || sv == 2 || sv == 4 || sv == 8) ? sv : rrl_default_scan_variant();
    o.order1 = v.order1;
    o.order2 = v.order2;
    if (v.scan_counters) { o.counters = (unsigned long long *)v.scan_counters; o.counter_rows = v.scan_counter_rows; }
    else rrl_default_scan_counters(&o.counters, &o.counter_rows);
    o.rider = v.chamfer;  // (done is the caller's to clear; the scan's launcher sets it when the walk rides along)
    o.payload = v.payload;
    o.payload_in_reduce = 0;
    o.problems = v.problems > 0 ? v.problems : 0;
    o.chain_left = v.chain_left;
    o.leave_clean = o.fused_build = 0;
    o.xf = nullptr;
    o.tri1_in = nullptr;
    return o;
}
// Which reduce kernel: 0 one workgroup per sample, 1 tiled with the candidate exchange (loss_reduce_tiled_kernel), 2 the
// tail kernel (no exchange: every workgroup streams its sample's dense value lists; one 512-lane workgroup or two
// per compute unit, so it serves the small, latency-bound grids: B x tiles <= 256, <= 32 tiles per sample).
// mode 0 (auto): the tail kernel where the direct backward rides along (with_bwd: rrl_registration_step -- measured
// -1.9 .. -3.4 us per step at C2 / L = 4096 / C4, round 5b: -6 us at the demo's 20 tiles; as a reduce alone it is within
// +-1 % of the exchange kernel), else the exchange kernel for >= 2 tiles while the grid
// is co-resident, else the single workgroup; 1: single; 2 ("tiled"): the tail kernel wherever it is legal (also forward
// only, also one tile: tests), exchange beyond; 3 ("xchg"): the exchange kernel wherever it is legal.
// Test hook: polls a waiting workgroup of the exchange reduce makes before it gives up (default 2^18, ~0.3 s); 0 makes
// every hand-off "time out", so the repair path runs on every sample (tests/test_gpu_stress.py).  Env RRL_SPIN_LIMIT.
static long g_spin_limit = -1;
extern "C" int rrl_set_spin_limit(long long polls) {
    if (polls < 0 || polls > 0xffffffffll) return RRL_E_ARG;
    g_spin_limit = (long)polls;
    return 0;
}
static unsigned spin_limit() {
    if (g_spin_limit < 0) {
        const char *e = getenv("RRL_SPIN_LIMIT");
        g_spin_limit = e ? atol(e) : (1l << 18);
        if (g_spin_limit < 0) g_spin_limit = 1l << 18;
    }
    return (unsigned)g_spin_limit;
}
// Workgroups of loss_reduce_tiled_kernel that are co-resident on the CURRENT device when it has the device to itself:
// compute units (as the runtime reports them: a CPX partition or a CU mask reports fewer) x the occupancy the
// runtime computes for this kernel.  (Round 3 hard-coded 1024 = 256 CUs x 4.)  Co-residency is a matter of speed only
// since round 4 -- a workgroup that waits in vain is repaired by its sample's last workgroup -- but a grid beyond the
// capacity would make that slow path the usual one, so the exchange kernel is only chosen within it.
static long xchg_capacity() {
    static long cap[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cap[dev] == 0) {
        hipDeviceProp_t p;
        int per_cu = 0;
        long c = 256;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, loss_reduce_tiled_kernel, 256, 0) == hipSuccess && per_cu > 0)
            c = (long)p.multiProcessorCount * per_cu;
#ifdef RRL_EXPERIMENT
        if (const char *e = getenv("RRL_XCHG_CAPACITY")) c = atol(e);
#endif
        cap[dev] = c < 1 ? 1 : (c > 4096 ? 4096 : c);
    }
    return cap[dev];
}
// (sample, tile) pairs up to which the tail kernel serves a call (experiments: RRL_TAIL_MAX_WG)
static long tail_max_wg() {
    static long v = -1;
    if (v < 0) {
        const char *e = nullptr;
#ifdef RRL_EXPERIMENT
        e = getenv("RRL_TAIL_MAX_WG");
#endif
        v = e ? atol(e) : 256;  // measured at 10 tiles per sample (round 5): B = 16 78.5 -> 77.7, B = 24 98.0 -> 96.7 us per step with the
        if (v < 1) v = 256;     // tail kernel, B = 32 121 -> 128, B = 64 200 -> 213 (the exchange reduce + a backward launch win there)
    }
    return v;
}
static int reduce_kind(int mode, int B, int nblk, int pool, bool with_bwd) {
    if (pool || mode == 1) return 0;
    const bool xchg_ok = (long)B * nblk <= xchg_capacity();
    if (mode == 3) return xchg_ok && nblk >= 1 ? 1 : 0;
    const bool tail_ok = nblk <= TAIL_MAX_TILES && (long)B * nblk <= tail_max_wg();
    if (mode == 2 && tail_ok && nblk >= 1) return 2;
    // (round 5: up to TAIL_MAX_TILES line tiles, not 16 -- the demo's 20: tail 11.1 us against tiled reduce 10.6 + backward 5.8 / 7.2)
    if (mode == 0 && tail_ok && with_bwd && nblk >= 2) return 2;
    return xchg_ok && nblk >= 2 ? 1 : 0;
}
// the direct backward that may ride in the tail kernel's launch (rrl_registration_step)
struct TailBwd {
    const float *grad_loss, *src;
    float *gR, *gt, *payload;
    int transpose_r;
    float *grad_tri1;  // scatter target (rrl_loss_step) instead of (gR, gt)
};

static ReduceArgs reduce_args(void *ws, const WsLayout &w, float *loss, int B, int L, int s_m, int s_n, int e_m, int e_n,
                              int pool) {
    ReduceArgs r;
    r.kjc = w.u8(ws, RRL_WS_KJC); r.dc = w.f32(ws, RRL_WS_VALS); r.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    r.med_out = w.f32(ws, RRL_WS_MED); r.bcnt_out = w.i32(ws, RRL_WS_BCNT); r.bsum_out = w.i64(ws, RRL_WS_BSUM);
    r.info = w.i32(ws, RRL_WS_INFO); r.loss = loss; r.status = w.i32(ws, RRL_WS_STATUS);
    r.B = B; r.nblk = (L + 1023) / 1024;
    r.s_m = s_m; r.s_n = s_n; r.e_m = e_m; r.e_n = e_n; r.pool = pool;
    return r;
}

// tb != NULL: the caller wants the direct backward too; *bwd_done tells whether this launch carried it
static int loss_reduce_impl(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m, int s_n, int e_m,
                            int e_n, int pool, const TailBwd *tb, bool *bwd_done, const RrlCall &o, void *stream) {
    if (bwd_done) *bwd_done = false;
    if (!ws || !loss || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    if (B == 0) return 0;
    const int nblk = (L + 1023) / 1024;
    const int kind = reduce_kind(o.reduce_mode, B, nblk, pool, tb != nullptr);
    if (kind == 2) {
        TailArgs t;
        t.lidc = w.u32(ws, RRL_WS_LIDC); t.dc = w.f32(ws, RRL_WS_VALS); t.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
        t.vlist = w.f32(ws, RRL_WS_VLIST); t.vlcnt = w.i32(ws, RRL_WS_VLCNT);
        t.mhist = w.u32(ws, RRL_WS_MHIST); t.mctl = w.u32(ws, RRL_WS_MCTL);
        t.msum = (unsigned long long *)w.i64(ws, RRL_WS_MSUM);
        t.med_out = w.f32(ws, RRL_WS_MED); t.bcnt_out = w.i32(ws, RRL_WS_BCNT); t.bsum_out = w.i64(ws, RRL_WS_BSUM);
        t.info = w.i32(ws, RRL_WS_INFO); t.loss = loss; t.status = w.i32(ws, RRL_WS_STATUS);
        t.B = B; t.nblk = nblk; t.s_m = s_m; t.s_n = s_n; t.e_m = e_m; t.e_n = e_n;
        t.do_bwd = tb ? 1 : 0; t.N = N; t.L = L; t.transpose_r = tb ? tb->transpose_r : 0;
        t.hs1 = w.i32(ws, RRL_WS_HS1); t.w1 = w.f32(ws, RRL_WS_W1);
        t.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); t.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
        t.grad_loss = tb ? tb->grad_loss : nullptr; t.src = tb ? tb->src : nullptr;
        t.gR = tb ? tb->gR : nullptr; t.gt = tb ? tb->gt : nullptr; t.payload = tb ? tb->payload : nullptr;
        t.grad_tri1 = tb ? tb->grad_tri1 : nullptr;
        t.Bt = o.problems;
        t.xcd_align = B % 8 == 0 && xcd_align_on();
        t.chain = o.leave_clean ? w.u32(ws, RRL_WS_CHAIN) : nullptr;
        t.chain_flags = o.fused_build ? 1 : 0;
        // the next epoch's sampler write pass rides along (tail_write_kernel; rrl_demo_epoch) -- when this launch carries the
        // backward (nothing after it reads the line buffer the pass overwrites) and the ballots of THAT count pass are there
        RrlWriteRider *wr = tb ? o.write_rider : nullptr;
        const int wtiles = wr ? (wr->n + 1023) / 1024 : 0;
        const bool ride = wr && (!o.count_rider || o.count_rider->done) && B == 1 && wr->n > 0 && wr->rounds > 0 &&
                          (long)wtiles * wr->rounds < 512 && sizeof(int32_t) * (size_t)wr->rounds * wtiles <= 32 * 1024;
        if (ride) {
            const WriteKArgs wk = {wr->rng_state, wr->r, wr->centers, wr->accept, wr->lines, wr->filled, wr->n, wr->rounds, wtiles, wr->rounds};
            const dim3 g((unsigned)(wk.gx * wk.gy + nblk * B * TAIL_SUBS));
            const size_t lds = sizeof(int32_t) * (size_t)wk.rounds * wk.gx;
            if (t.grad_tri1) hipLaunchKernelGGL(tail_write_kernel<true>, g, dim3(TAIL_LANES), lds, (hipStream_t)stream, t, wk);
            else hipLaunchKernelGGL(tail_write_kernel<false>, g, dim3(TAIL_LANES), lds, (hipStream_t)stream, t, wk);
            wr->done = 1;
        } else {
            const dim3 g((unsigned)nblk, (unsigned)B, TAIL_SUBS);
            const bool two = TAIL_LANES / nblk >= 48;  // (see tail_body: groups per lane and round)
#define RRL_TAIL(S_, R_) hipLaunchKernelGGL((loss_tail_kernel<S_, R_>), g, dim3(TAIL_LANES), 0, (hipStream_t)stream, t)
            if (t.grad_tri1) { if (two) RRL_TAIL(true, 2); else RRL_TAIL(true, TAIL_RPL); }
            else { if (two) RRL_TAIL(false, 2); else RRL_TAIL(false, TAIL_RPL); }
#undef RRL_TAIL
        }
        RRL_LAUNCH_CHECK();
        if (bwd_done) *bwd_done = tb != nullptr;
        return 0;
    }
    if (kind == 1) {
        TiledArgs t;
        t.kjc = w.u8(ws, RRL_WS_KJC); t.dc = w.f32(ws, RRL_WS_VALS); t.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
        t.mhist = w.u32(ws, RRL_WS_MHIST); t.mctl = w.u32(ws, RRL_WS_MCTL); t.mcand = w.u32(ws, RRL_WS_MCAND);
        t.msum = (unsigned long long *)w.i64(ws, RRL_WS_MSUM);
        t.med_out = w.f32(ws, RRL_WS_MED); t.bcnt_out = w.i32(ws, RRL_WS_BCNT); t.bsum_out = w.i64(ws, RRL_WS_BSUM);
        t.info = w.i32(ws, RRL_WS_INFO); t.loss = loss; t.status = w.i32(ws, RRL_WS_STATUS);
        t.B = B; t.nblk = nblk; t.s_m = s_m; t.s_n = s_n; t.e_m = e_m; t.e_n = e_n;
        t.spin_limit = spin_limit();
        t.xcd_align = B % 8 == 0 && xcd_align_on();
        t.payload = o.payload_in_reduce ? o.payload : nullptr;
        t.chain = o.leave_clean ? w.u32(ws, RRL_WS_CHAIN) : nullptr;
        t.chain_flags = o.fused_build ? 1 : 0;
        hipLaunchKernelGGL(loss_reduce_tiled_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, t);
        RRL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(loss_reduce_kernel, dim3((unsigned)(pool ? 1 : B)), dim3(1024), sizeof(int) * (size_t)(nblk + 1),
                       (hipStream_t)stream, reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, pool));
    RRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int rrl_loss_reduce_ex(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L,
                                  int s_m, int s_n, int e_m, int e_n, int pool, const rrl_opts *opts, void *stream) {
    return loss_reduce_impl(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, nullptr, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_loss_reduce(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L,
                               int s_m, int s_n, int e_m, int e_n, int pool, void *stream) {
    return rrl_loss_reduce_ex(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, nullptr, stream);
}

// K3 + K4 over CALLER-SUPPLIED rows (the merge step of the line-sharded single-sample mode, rrl_hip/dist.py): every rank
// ran the scan and the per-line stage on ITS share of one sample's lines; the selected lines' canonical D tiles and
// (k | j << 4) bytes of all ranks, gathered into one dense list, are reduced here by the single-workgroup kernel exactly
// as if one per-line stage had produced them (full tiles of 1024 rows: blkcnt is filled by a tiny launch).  The median
// is over the same multiset and the bucket sums are order-independent fixed point, so loss, median, bucket counts and
// sums are bit-identical to the unsharded evaluation.
__global__ void rows_blkcnt_kernel(int32_t *__restrict__ blkcnt, int nblk, int nrows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nblk) blkcnt[t] = min(1024, nrows - 1024 * t);
}

extern "C" int rrl_loss_reduce_rows(const float *rows16, const uint8_t *kj, int nrows, int32_t *blkcnt_scratch, float *loss,
                                    float *med, int32_t *bcnt, int64_t *bsum, int32_t *info, const int32_t *status,
                                    int s_m, int s_n, int e_m, int e_n, void *stream) {
    if (!rows16 || !kj || !blkcnt_scratch || !loss || !med || !bcnt || !bsum || !info || !status || nrows < 0) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    const int nblk = nrows > 0 ? (nrows + 1023) / 1024 : 1;
    hipLaunchKernelGGL(rows_blkcnt_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, blkcnt_scratch,
                       nblk, nrows);
    ReduceArgs r;
    r.kjc = kj; r.dc = rows16; r.blkcnt = blkcnt_scratch;
    r.med_out = med; r.bcnt_out = bcnt; r.bsum_out = bsum; r.info = info; r.loss = loss; r.status = status;
    r.B = 1; r.nblk = nblk; r.s_m = s_m; r.s_n = s_n; r.e_m = e_m; r.e_n = e_n; r.pool = 0;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), sizeof(int) * (size_t)(nblk + 1), (hipStream_t)stream, r);
    RRL_LAUNCH_CHECK();
    return 0;
}

#include "rrl_stage_bwd.inc"     // K5: backward variants

static int g_deterministic = -1;  // -1: read RRL_DETERMINISTIC once
extern "C" int rrl_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}
static bool default_deterministic() {
    if (g_deterministic < 0) {
        const char *e = getenv("RRL_DETERMINISTIC");
        g_deterministic = (e && e[0] == '1') ? 1 : 0;
    }
    return g_deterministic == 1;
}

static ScatArgs scat_args(const void *ws, const WsLayout &w, const float *grad_loss, float *g1, float *g2, int N, int M, int L) {
    ScatArgs a;
    a.lidc = w.u32(ws, RRL_WS_LIDC); a.blkcnt = w.i32(ws, RRL_WS_BLKCNT);
    a.hs1 = w.i32(ws, RRL_WS_HS1); a.hs2 = w.i32(ws, RRL_WS_HS2); a.bcnt = w.i32(ws, RRL_WS_BCNT); a.info = w.i32(ws, RRL_WS_INFO);
    a.w1 = w.f32(ws, RRL_WS_W1); a.w2 = w.f32(ws, RRL_WS_W2); a.D = w.f32(ws, RRL_WS_D); a.med = w.f32(ws, RRL_WS_MED);
    a.grad_loss = grad_loss;
    a.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); a.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
    a.g1 = g1; a.g2 = g2; a.N = N; a.M = M; a.L = L;
    a.fx = nullptr; a.fxbits = 0; a.fxB = 0;
    return a;
}

static int scat_fx_bits_host(int L) {
    int lg = 1;
    while ((1 << lg) < L && lg < 30) ++lg;
    return 62 - lg;
}

// deterministic: include/rrl.h rrl_set_deterministic -- the scatter accumulates in the workspace's fixed-point field (which this
// call clears and therefore WRITES: the one entry that touches the workspace of a finished forward) and one more launch
// converts; grad_tri1 / grad_tri2 are overwritten, not accumulated.
static int loss_backward_impl(const float *tri1, const float *tri2, const void *ws,
                              size_t ws_bytes, const float *grad_loss, float *grad_tri1,
                              float *grad_tri2, int B, int N, int M, int L, int pool, bool zero1,
                              void *stream, bool deterministic = false) {
    if (!tri1 || !tri2 || !ws || !grad_loss || !grad_tri1) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (zero1 && (rc = rrl_fill(grad_tri1, 0u, sizeof(float) * 9 * (size_t)B * N, s))) return rc;
    if (grad_tri2 && (rc = rrl_fill(grad_tri2, 0u, sizeof(float) * 9 * (size_t)B * M, s))) return rc;
    if (B == 0 || L == 0) return 0;
    ScatArgs sa = scat_args(ws, w, grad_loss, grad_tri1, grad_tri2, N, M, L);
    if (deterministic && (N + M) > 0) {
        sa.fx = (unsigned long long *)((char *)const_cast<void *>(ws) + w.off[RRL_WS_GFIX]);
        sa.fxbits = scat_fx_bits_host(L);
        sa.fxB = B;
        if ((rc = rrl_fill(sa.fx, 0u, 8 * (size_t)B * (N + M) * 9 + 8 * (size_t)B, s))) return rc;
    }
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((L + 1023) / 1024), (unsigned)B, BWDS_SUBS), dim3(256), 0, s, sa, B, pool,
                       B % 8 == 0 && xcd_align_on() ? 1 : 0);
    RRL_LAUNCH_CHECK();
    if (sa.fx) {
        const int nmax = grad_tri2 && M > N ? M : N;
        hipLaunchKernelGGL(scatter_fix_to_float_kernel, dim3((unsigned)(((size_t)nmax * 9 + 255) / 256), (unsigned)B, grad_tri2 ? 2u : 1u),
                           dim3(256), 0, s, sa, B, pool);
        RRL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int rrl_loss_backward(const float *tri1, const float *tri2, const void *ws,
                                 size_t ws_bytes, const float *grad_loss, float *grad_tri1,
                                 float *grad_tri2, int B, int N, int M, int L, int pool,
                                 void *stream) {
    return loss_backward_impl(tri1, tri2, ws, ws_bytes, grad_loss, grad_tri1, grad_tri2, B, N, M, L,
                              pool, true, stream, default_deterministic());
}

int rrl_fused_backward(int B, int N, int M);
int rrl_launch_reg_bwd(const float *src, const float *R, float *g1, float *grad_src, float *partial,
                       float *gR, float *gt, float *payload, const float *loss, const int32_t *info,
                       int32_t *done, int B, int N, int transpose_r, hipStream_t s);

// ---------------------------------------------------------------------------------------
// workspace + fused forward
// ---------------------------------------------------------------------------------------
extern "C" size_t rrl_workspace_bytes(int B, int N, int M, int L) { return WsLayout(B, N, M, L).total; }

extern "C" int rrl_workspace_layout(int B, int N, int M, int L, size_t *offsets) {
    if (!offsets || B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    for (int i = 0; i < RRL_WS_FIELDS; ++i) offsets[i] = w.off[i];
    return 0;
}

int rrl_tri_prepare_clouds(const float *tri1, const float *tri2, void *ws, size_t ws_bytes, int B,
                           int N, int M, int L, int clouds, const RrlXform *xf, const float *line, const RrlCall &o,
                           void *stream);
int rrl_line_tri_scan_clouds(const float *line, void *ws, size_t ws_bytes, int B, int N, int M, int L,
                             int mode, int chunk, int clouds, int lmax_ready, const RrlCall &o, void *stream);
int rrl_sort_capacity(void);
int rrl_cull_scan_can_fuse(int B, int N, int M, int L, const RrlCall &o);  // rrl_cull.hip

// target_ws != NULL: a workspace of the same (B, N, M, L) that already went through a forward with
// the SAME tri2 and line (RPM / FMR evaluate several source poses against one target and one
// line set, rpm/Train_RPM.py:204-231): the target's hit counts and hit lists are copied from it
// and only the source cloud is prepared, sorted and scanned.
// xf != NULL: tri1 is the workspace field TRI1, filled by the prepare step from xf->src.
static int loss_forward_impl(const float *tri1, const float *tri2, const float *line, void *ws,
                             size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                             int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                             const void *target_ws, const RrlXform *xf, RrlCall o, void *stream,
                             const TailBwd *tb = nullptr, bool *bwd_done = nullptr) {
    if (bwd_done) *bwd_done = false;
    if (!tri1 || !tri2 || !line || !ws || !loss) return RRL_E_ARG;
    if (s_m < 1 || s_n < 1 || e_m > RRL_MAX_HITS + 1 || e_n > RRL_MAX_HITS + 1) return RRL_E_RANGE;
    if (target_ws == ws) return RRL_E_ARG;
    const int clouds = target_ws ? 1 : 2;
    o.tar_ws = target_ws;
    // multi-pose evaluation (rrl_opts.problems = Bt): the B instances are B / Bt poses of Bt problems; the inputs have Bt
    // entries.  Served by the sorted layout of scan mode cull through the fused entries that move the source (xf); anything
    // else is an argument error (the caller evaluates pose after pose then)
    if (o.problems >= B) o.problems = 0;
    if (o.problems > 0 && (B % o.problems != 0 || !xf || pool || target_ws || mode != RRL_SCAN_CULL ||
                           (N > M ? N : M) > rrl_sort_capacity() || N <= 0 || M <= 0))
        return RRL_E_ARG;
    // prepared clouds (include/rrl.h rrl_opts): honoured by the sorted layout of scan mode cull, with the orders of every
    // cloud this call builds; anything else takes the plain path (same results)
    if (o.prepared() && (mode != RRL_SCAN_CULL || (N > M ? N : M) > rrl_sort_capacity() ||
                         (clouds == 2 && !o.order2 && !(o.flags & RRL_F_TARGET_KEPT))))
        o.order1 = o.order2 = nullptr;
    // a kept target: cloud 2's records / tree / partials stay as the previous call on this workspace left them
    const int build_clouds = o.target_kept() ? 1 : clouds;
    // Chained steps (include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED).  The chain lives where the per-line stage + the tail kernel or
    // the exchange reduce serve the call: they are the ones that leave COUNT1 / COUNT2 and the CHAIN words cleared.
    const bool chain_path = B > 0 && L > 1024 && !pool && clouds == 2 && !o.problems && mode == RRL_SCAN_CULL &&
                            (N > M ? N : M) <= rrl_sort_capacity() && N > 0 && M > 0 &&
                            reduce_kind(o.reduce_mode, B, (L + 1023) / 1024, pool, tb != nullptr) >= 1;
    o.leave_clean = (o.flags & RRL_F_CHAIN) && chain_path ? 1 : 0;
    // ... and a step that FINDS them cleared runs source records + target scan + source scan as ONE launch
    o.fused_build = (o.flags & RRL_F_CHAINED) && chain_path && o.target_kept() && !o.count_rider && !o.write_rider &&
                    rrl_cull_scan_can_fuse(B, N, M, L, o) ? 1 : 0;
    if (o.chain_left) *o.chain_left = o.leave_clean | (o.fused_build << 1);  // bit 0: leaves the workspace chain-clean; bit 1: THIS call's build is fused
    o.xf = xf;
    o.tri1_in = tri1;
    int rc;
    RrlRange step("rrl forward");
    if (!o.fused_build) {
        RrlRange r("K1' records + sort + tree");
        if ((rc = rrl_tri_prepare_clouds(tri1, tri2, ws, ws_bytes, B, N, M, L, build_clouds, xf, line, o, stream))) return rc;
    }
    // (a carried-over target: the per-line stage reads cloud 2's hit counts and lists in `target_ws` itself -- pair_args)
    {
        RrlRange r("K1 line<->triangle scan");
        // (the records kernel reduced the lines' maxima whenever it ran: the sorted path)
        const int lmax_ready = (N > M ? N : M) <= rrl_sort_capacity() && B > 0 && (clouds == 2 && M > N ? M : N) > 0 && L > 0;
        if ((rc = rrl_line_tri_scan_clouds(line, ws, ws_bytes, B, N, M, L, mode, chunk, clouds, lmax_ready, o, stream))) return rc;
    }
    if (L >= 1 && L <= 1024 && !pool && B > 0 && o.reduce_mode < 2) {  // one tile of lines per sample: K2 + K3 + K4 in one launch
        RrlRange r("K2 + K3 + K4 (single tile)");
        WsLayout w(B, N, M, L);
        if (ws_bytes < w.total) return RRL_E_WS;
        if (tb && tb->grad_tri1) {  // ... and the scatter backward to points1.grad too (rrl_loss_step_ex)
            hipLaunchKernelGGL(pair_reduce_scatter_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                               pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                               reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0),
                               scat_args(ws, w, tb->grad_loss, tb->grad_tri1, nullptr, N, M, L), tb->payload, w.u32(ws, RRL_WS_MCTL));
            RRL_LAUNCH_CHECK();
            if (bwd_done) *bwd_done = true;
            return 0;
        }
        if (tb && !tb->grad_tri1) {  // ... and the direct backward too (rrl_registration_step)
            SoloBwd sb;
            sb.kj = w.u8(ws, RRL_WS_KJ); sb.sel = w.i32(ws, RRL_WS_SEL); sb.nsel = w.i32(ws, RRL_WS_NSEL);
            sb.hs1 = w.i32(ws, RRL_WS_HS1); sb.bcnt = w.i32(ws, RRL_WS_BCNT); sb.info = w.i32(ws, RRL_WS_INFO);
            sb.w1 = w.f32(ws, RRL_WS_W1); sb.D = w.f32(ws, RRL_WS_D); sb.med = w.f32(ws, RRL_WS_MED);
            sb.grad_loss = tb->grad_loss; sb.src = tb->src; sb.loss = loss;
            sb.Q1 = (const float4 *)w.f32(ws, RRL_WS_Q1); sb.Q2 = (const float4 *)w.f32(ws, RRL_WS_Q2);
            sb.gR = tb->gR; sb.gt = tb->gt; sb.payload = tb->payload; sb.mctl = w.u32(ws, RRL_WS_MCTL);
            sb.B = B; sb.N = N; sb.L = L; sb.transpose_r = tb->transpose_r; sb.Bt = o.problems;
            hipLaunchKernelGGL(pair_reduce_bwd_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                               pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                               reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0), sb);
            RRL_LAUNCH_CHECK();
            if (bwd_done) *bwd_done = true;
            return 0;
        }
        hipLaunchKernelGGL(pair_reduce_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * 2, (hipStream_t)stream,
                           pair_args(tri1, tri2, line, ws, w, B, N, M, L, s_m, s_n, e_m, e_n, false, target_ws, o.problems),
                           reduce_args(ws, w, loss, B, L, s_m, s_n, e_m, e_n, 0));
        RRL_LAUNCH_CHECK();
        return 0;
    }
    {
        RrlRange r("K2 per-line distances");
        if ((rc = line_pair_dist_impl(tri1, tri2, line, ws, ws_bytes, B, N, M, L, s_m, s_n, e_m,
                                      e_n, pool, o, stream, tb != nullptr)))
            return rc;
    }
    RrlRange r("K3+K4 median + Welsch reduce");
    return loss_reduce_impl(ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, pool, tb, bwd_done, o, stream);
}

extern "C" int rrl_loss_forward_ex(const float *tri1, const float *tri2, const float *line,
                                   void *ws, size_t ws_bytes, float *loss, int B, int N, int M,
                                   int L, int s_m, int s_n, int e_m, int e_n, int pool, int mode,
                                   int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    return loss_forward_impl(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                             pool, mode, chunk, target_ws, nullptr, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_loss_forward_cached(const float *tri1, const float *tri2, const float *line,
                                       void *ws, size_t ws_bytes, float *loss, int B, int N, int M,
                                       int L, int s_m, int s_n, int e_m, int e_n, int pool, int mode,
                                       int chunk, const void *target_ws, void *stream) {
    return rrl_loss_forward_ex(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                               pool, mode, chunk, target_ws, nullptr, stream);
}

// The drop-in call (code/loss.py:170-232 as the reference's callers use it: one sample, a Python-level
// decision on the result) in ONE entry: the forward, then INFO[0 .. 4 G) (nbuckets, nselected, nvalues, NaN
// flag per group) copied to host_info, then a wait for the stream -- the only entry of the library that
// synchronises, because the reference's return value (a tensor, or None when no bucket is populated, or an
// exit on NaN) is a host-side decision by contract.  host_info: 4 G int32 in host memory (pinned memory makes
// the copy asynchronous up to the wait).
extern "C" int rrl_loss_forward_info(const float *tri1, const float *tri2, const float *line, void *ws,
                                     size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                                     int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                                     const void *target_ws, int32_t *host_info, void *stream) {
    if (!host_info) return RRL_E_ARG;
    int rc = loss_forward_impl(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n,
                               pool, mode, chunk, target_ws, nullptr, rrl_resolve_opts(nullptr), stream);
    if (rc) return rc;
    const int G = pool ? 1 : B;
    if (G <= 0) return 0;
    WsLayout w(B, N, M, L);
    hipError_t e = hipMemcpyAsync(host_info, w.i32(ws, RRL_WS_INFO), sizeof(int32_t) * 4 * (size_t)G,
                                  hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rrl_loss_forward(const float *tri1, const float *tri2, const float *line, void *ws,
                                size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                                int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                                void *stream) {
    return rrl_loss_forward_cached(tri1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m,
                                   e_n, pool, mode, chunk, nullptr, stream);
}

// ---------------------------------------------------------------------------------------
// fused training op: rigid transform of the source + loss, and its backward to (dR, dt)
// ---------------------------------------------------------------------------------------
extern "C" int rrl_registration_forward_ex(const float *src, const float *R, const float *t,
                                           const float *tri2, const float *line, void *ws,
                                           size_t ws_bytes, float *loss, int B, int N, int M,
                                           int L, int transpose_r, int s_m, int s_n, int e_m,
                                           int e_n, int mode, int chunk, const void *target_ws,
                                           const rrl_opts *opts, void *stream) {
    if (!src || !R || !t || !tri2 || !line || !ws || !loss) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    // the transform runs inside the prepare step
    const RrlXform xf = {src, R, t, transpose_r, 1};  // 1: clear GACC for the backward's atomics
    return loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m,
                             s_n, e_m, e_n, 0, mode, chunk, target_ws, &xf, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_registration_forward_cached(const float *src, const float *R, const float *t,
                                               const float *tri2, const float *line, void *ws,
                                               size_t ws_bytes, float *loss, int B, int N, int M,
                                               int L, int transpose_r, int s_m, int s_n, int e_m,
                                               int e_n, int mode, int chunk, const void *target_ws,
                                               void *stream) {
    return rrl_registration_forward_ex(src, R, t, tri2, line, ws, ws_bytes, loss, B, N, M, L, transpose_r, s_m, s_n,
                                       e_m, e_n, mode, chunk, target_ws, nullptr, stream);
}

extern "C" int rrl_registration_forward(const float *src, const float *R, const float *t,
                                        const float *tri2, const float *line, void *ws,
                                        size_t ws_bytes, float *loss, int B, int N, int M, int L,
                                        int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                        int chunk, void *stream) {
    return rrl_registration_forward_cached(src, R, t, tri2, line, ws, ws_bytes, loss, B, N, M, L,
                                           transpose_r, s_m, s_n, e_m, e_n, mode, chunk, nullptr,
                                           stream);
}

static int registration_backward_impl(const float *src, const float *R, const float *tri2,
                                      void *ws, size_t ws_bytes, const float *loss,
                                      const float *grad_loss, float *grad_src, float *gR, float *gt,
                                      float *payload, int B, int N, int M, int L, int transpose_r,
                                      const RrlCall &o, void *stream);

// Forward + direct backward of the fused training op in ONE call (dL/dloss is an input, so nothing has to come back
// to the host in between): when the tail kernel serves the shape, the backward rides in its launch (5 launches per
// step instead of 6, and the reduce's and the backward's chains of dependent loads overlap); otherwise exactly
// rrl_registration_forward_cached followed by rrl_registration_backward.  gR, gt (and payload) should be the
// workspace's GACC field, which the forward's first launch clears; other buffers are cleared here first.
extern "C" int rrl_registration_step_ex(const float *src, const float *R, const float *t, const float *tri2,
                                        const float *line, void *ws, size_t ws_bytes, float *loss,
                                        const float *grad_loss, float *gR, float *gt, float *payload, int B, int N,
                                        int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                        int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    return rrl_registration_step_call(src, R, t, tri2, line, ws, ws_bytes, loss, grad_loss, gR, gt, payload, B, N, M, L,
                                      transpose_r, s_m, s_n, e_m, e_n, mode, chunk, target_ws, rrl_resolve_opts(opts), stream);
}
int rrl_registration_step_call(const float *src, const float *R, const float *t, const float *tri2, const float *line,
                               void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *gR, float *gt,
                               float *payload, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n,
                               int mode, int chunk, const void *target_ws, const RrlCall &o, void *stream) {
    if (!src || !R || !t || !tri2 || !line || !ws || !loss || !grad_loss || !gR || !gt) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    const int nblk = (L + 1023) / 1024;
    bool done = false;
    int rc;
    const bool solo = L <= 1024 && o.reduce_mode < 2;  // one tile of lines: per-line stage + reduce + backward by one workgroup per sample
    if (B > 0 && L > 0 && !o.deterministic && (solo || reduce_kind(o.reduce_mode, B, nblk, 0, true) == 2)) {
        float *gacc = w.f32(ws, RRL_WS_GACC);
        hipStream_t s = (hipStream_t)stream;
        if (gR != gacc || gt != gacc + 9 * (size_t)B || (payload && payload != gacc + 12 * (size_t)B)) {
            if ((rc = rrl_fill(gR, 0u, sizeof(float) * 9 * (size_t)B, s))) return rc;
            if ((rc = rrl_fill(gt, 0u, sizeof(float) * 3 * (size_t)B, s))) return rc;
            if (payload && (rc = rrl_fill(payload, 0u, sizeof(float) * 14, s))) return rc;
        }
        const RrlXform xf = {src, R, t, transpose_r, 1};
        const TailBwd tb = {grad_loss, src, gR, gt, payload, transpose_r, nullptr};
        rc = loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0,
                               mode, chunk, target_ws, &xf, o, stream, &tb, &done);
        if (rc || done) return rc;
    } else {
        const RrlXform xf = {src, R, t, transpose_r, 1};
        rc = loss_forward_impl(w.f32(ws, RRL_WS_TRI1), tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0,
                               mode, chunk, target_ws, &xf, o, stream);
        if (rc) return rc;
    }
    return registration_backward_impl(src, R, tri2, ws, ws_bytes, loss, grad_loss, nullptr, gR, gt, payload, B, N, M, L,
                                      transpose_r, o, stream);
}
extern "C" int rrl_registration_step(const float *src, const float *R, const float *t, const float *tri2,
                                     const float *line, void *ws, size_t ws_bytes, float *loss,
                                     const float *grad_loss, float *gR, float *gt, float *payload, int B, int N,
                                     int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode,
                                     int chunk, const void *target_ws, void *stream) {
    return rrl_registration_step_ex(src, R, t, tri2, line, ws, ws_bytes, loss, grad_loss, gR, gt, payload, B, N, M, L,
                                    transpose_r, s_m, s_n, e_m, e_n, mode, chunk, target_ws, nullptr, stream);
}

// SURVEY 8(d)'s own definition -- T-apply + S + P + median + Welsch reduce + backward to points1.grad -- in ONE call
// (include/rrl.h rrl_loss_step_ex): the forward of rrl_loss_forward_ex / rrl_registration_forward_ex with the scatter
// backward of rrl_loss_backward riding in the reduce's launch where the tail kernel serves the shape (the records launch
// clears grad_tri1 with the per-call state: 4 launches with prepared orders); elsewhere, and when grad_tri2 is wanted,
// forward + loss_bwd_kernel.  Bit-identical loss; gradients to the rounding of the float atomics.
extern "C" int rrl_loss_step_ex(const float *tri1, const float *R, const float *t, const float *tri2, const float *line,
                                void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *grad_tri1,
                                float *grad_tri2, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m,
                                int e_n, int mode, int chunk, const void *target_ws, const rrl_opts *opts, void *stream) {
    if (!tri1 || !tri2 || !line || !ws || !loss || !grad_loss || !grad_tri1 || ((R == nullptr) != (t == nullptr))) return RRL_E_ARG;
    if (B < 0 || N < 0 || M < 0 || L < 0) return RRL_E_ARG;
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    RrlCall o = rrl_resolve_opts(opts);
    o.clear_ptr = grad_tri1;
    o.clear_bytes = sizeof(float) * 9 * (size_t)B * N;
    const int nblk = (L + 1023) / 1024;
    // the shard payload of the step (rrl_opts.payload): in the workspace's GACC field the records launch clears it with the
    // rest of the accumulator (the fused op's convention); any other buffer is cleared here first
    float *payload = o.payload;
    const bool pay_in_ws = payload && R && payload == w.f32(ws, RRL_WS_GACC) + 12 * (size_t)B;
    if (payload && !pay_in_ws) {
        int rcf = rrl_fill(payload, 0u, sizeof(float) * 14, (hipStream_t)stream);
        if (rcf) return rcf;
    }
    const RrlXform xf = {tri1, R, t, transpose_r, pay_in_ws ? 1 : 0};
    const float *p1 = R ? w.f32(ws, RRL_WS_TRI1) : tri1;  // points1: the moved source, or the caller's triangles as given
    // (the tail kernel, or -- one tile of lines per sample -- the single-tile kernel: loss_forward_impl's own conditions)
    // (deterministic: the fixed-point scatter of loss_bwd_kernel + its conversion launch, never the riding float atomics)
    const bool ride = B > 0 && L > 0 && !grad_tri2 && !o.deterministic &&
                      (L > 1024 ? reduce_kind(o.reduce_mode, B, nblk, 0, true) == 2 : o.reduce_mode < 2);
    // no riding backward, but the exchange reduce serves the call: its last arrivers add the payload (no payload launch)
    const bool pay_in_reduce = payload && !ride && B > 0 && L > 1024 && reduce_kind(o.reduce_mode, B, nblk, 0, false) == 1;
    o.payload_in_reduce = pay_in_reduce ? 1 : 0;
    const TailBwd tb = {grad_loss, nullptr, nullptr, nullptr, payload, 0, grad_tri1};
    bool done = false;
    int rc = loss_forward_impl(p1, tri2, line, ws, ws_bytes, loss, B, N, M, L, s_m, s_n, e_m, e_n, 0, mode, chunk, target_ws,
                               R ? &xf : nullptr, o, stream, ride ? &tb : nullptr, &done);
    if (rc || done) return rc;
    // (grad_tri1 was cleared by the build step's first launch -- or by its fill on the unsorted path; an empty batch /
    //  cloud launches nothing: clear here)
    if (B == 0 || (N == 0 && M == 0)) return rrl_fill(grad_tri1, 0u, o.clear_bytes, (hipStream_t)stream);
    rc = loss_backward_impl(p1, tri2, ws, ws_bytes, grad_loss, grad_tri1, grad_tri2, B, N, M, L, 0, false, stream, o.deterministic != 0);
    if (rc || !payload || L <= 0 || pay_in_reduce) return rc;
    return rrl_shard_payload(loss, ws, ws_bytes, nullptr, nullptr, payload, B, N, M, L, stream);
}

extern "C" int rrl_registration_backward_ex(const float *src, const float *R, const float *tri2,
                                            void *ws, size_t ws_bytes, const float *loss,
                                            const float *grad_loss, float *grad_src, float *gR, float *gt,
                                            float *payload, int B, int N, int M, int L, int transpose_r,
                                            const rrl_opts *opts, void *stream) {
    return registration_backward_impl(src, R, tri2, ws, ws_bytes, loss, grad_loss, grad_src, gR, gt, payload, B, N, M, L,
                                      transpose_r, rrl_resolve_opts(opts), stream);
}
extern "C" int rrl_registration_backward(const float *src, const float *R, const float *tri2,
                                         void *ws, size_t ws_bytes, const float *loss,
                                         const float *grad_loss, float *grad_src, float *gR, float *gt,
                                         float *payload, int B, int N, int M, int L, int transpose_r,
                                         void *stream) {
    return rrl_registration_backward_ex(src, R, tri2, ws, ws_bytes, loss, grad_loss, grad_src, gR, gt, payload, B, N, M, L,
                                        transpose_r, nullptr, stream);
}

static int registration_backward_impl(const float *src, const float *R, const float *tri2,
                                      void *ws, size_t ws_bytes, const float *loss,
                                      const float *grad_loss, float *grad_src, float *gR, float *gt,
                                      float *payload, int B, int N, int M, int L, int transpose_r,
                                      const RrlCall &o, void *stream) {
    if (!src || !R || !tri2 || !ws || !grad_loss || !gR || !gt) return RRL_E_ARG;
    if (payload && !loss) return RRL_E_ARG;
    if (o.problems > 0 && o.problems < B && grad_src) return RRL_E_ARG;  // multi-pose: the direct backward only (dL/dsrc would sum over the poses)
    WsLayout w(B, N, M, L);
    if (ws_bytes < w.total) return RRL_E_WS;
    float *g1 = w.f32(ws, RRL_WS_G1);
    hipStream_t s = (hipStream_t)stream;
    RrlRange step("K5 rrl backward");
    if (!grad_src && B > 0 && L > 0) {
        // only dL/dR, dL/dt (+ payload): ONE launch, straight from the selected lines; the outputs
        // are accumulated with atomics -- clear them unless they are the workspace's GACC field,
        // which the forward left zeroed
        float *gacc = w.f32(ws, RRL_WS_GACC);
        const int nblk = o.deterministic ? 16 * ((L + 1023) / 1024) : (L + BWD_LINES - 1) / BWD_LINES;
#define RRL_BWD_RT(DET, PART)                                                                                  \
        hipLaunchKernelGGL(loss_bwd_rt_kernel<DET>, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, s,            \
                           w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL), w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), \
                           w.f32(ws, RRL_WS_W1), (const float4 *)w.f32(ws, RRL_WS_Q1),                             \
                           (const float4 *)w.f32(ws, RRL_WS_Q2), w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_MED),       \
                           w.i32(ws, RRL_WS_BCNT), w.i32(ws, RRL_WS_INFO), grad_loss, src, gR, gt, payload, loss,  \
                           B, N, L, transpose_r, PART, o.problems, B % 8 == 0 && xcd_align_on() ? 1 : 0)
        // the next epoch's sampler write pass rides along (bwd_write_kernel; rrl_demo_epoch)
        RrlWriteRider *wr = o.write_rider;
        const int wtiles = wr ? (wr->n + 1023) / 1024 : 0;
        const bool ride = wr && (!o.count_rider || o.count_rider->done) &&  // (the ballots of THAT count pass: it must have ridden)
                          B == 1 && wr->n > 0 && wr->rounds > 0 && (long)wtiles * wr->rounds < 512 &&
                          sizeof(int32_t) * (size_t)wr->rounds * wtiles <= 48 * 1024;
        const WriteKArgs wk = ride ? WriteKArgs{wr->rng_state, wr->r, wr->centers, wr->accept, wr->lines, wr->filled, wr->n,
                                                wr->rounds, wtiles, wr->rounds}
                                   : WriteKArgs{};
#define RRL_BWD_WRITE(DET, PART)                                                                               \
        do {                                                                                                   \
            const BwdKArgs ba = {w.u8(ws, RRL_WS_KJ), w.i32(ws, RRL_WS_SEL), w.i32(ws, RRL_WS_NSEL), w.i32(ws, RRL_WS_HS1), \
                                 w.f32(ws, RRL_WS_W1), (const float4 *)w.f32(ws, RRL_WS_Q1),                    \
                                 (const float4 *)w.f32(ws, RRL_WS_Q2), w.f32(ws, RRL_WS_D), w.f32(ws, RRL_WS_MED), \
                                 w.i32(ws, RRL_WS_BCNT), w.i32(ws, RRL_WS_INFO), grad_loss, src, gR, gt, payload, loss, \
                                 B, N, L, transpose_r, PART, nblk};                                             \
            hipLaunchKernelGGL(bwd_write_kernel<DET>, dim3((unsigned)(wk.gx * wk.gy + nblk * B)), dim3(256),     \
                               sizeof(int32_t) * (size_t)wk.rounds * wk.gx, s, ba, wk);                         \
            wr->done = 1;                                                                                      \
        } while (0)
        if (o.deterministic) {
            // partials in VALS (the reduce kernel's input tiles: dead after the forward; B * Lp * 16 floats
            // >= B * 16 ceil(L / 1024) * 12), fixed-order sums by a second launch: nothing to clear
            float *part = w.f32(ws, RRL_WS_VALS);
            if (ride) RRL_BWD_WRITE(true, part);
            else RRL_BWD_RT(true, part);
            hipLaunchKernelGGL(loss_bwd_rt_finalize_kernel, dim3(1), dim3(256), 0, s, part, w.i32(ws, RRL_WS_INFO), loss,
                               gR, gt, payload, B, nblk);
            RRL_LAUNCH_CHECK();
            return 0;
        }
        if (gR != gacc || gt != gacc + 9 * (size_t)B || (payload && payload != gacc + 12 * (size_t)B)) {
            int rc;
            if ((rc = rrl_fill(gR, 0u, sizeof(float) * 9 * (size_t)B, s))) return rc;
            if ((rc = rrl_fill(gt, 0u, sizeof(float) * 3 * (size_t)B, s))) return rc;
            if (payload && (rc = rrl_fill(payload, 0u, sizeof(float) * 14, s))) return rc;
        }
        if (ride) RRL_BWD_WRITE(false, nullptr);
        else RRL_BWD_RT(false, nullptr);
#undef RRL_BWD_RT
#undef RRL_BWD_WRITE
        RRL_LAUNCH_CHECK();
        return 0;
    }
    if (rrl_fused_backward(B, N, M)) {
        // dL/dsrc wanted too: scatter of the line gradients into G1 (cleared first), then rigid
        // backward + payload (reg_bwd_kernel)
        int rc = loss_backward_impl(w.f32(ws, RRL_WS_TRI1), tri2, ws, ws_bytes, grad_loss, g1, nullptr,
                                    B, N, M, L, 0, true, stream);
        if (rc) return rc;
        return rrl_launch_reg_bwd(src, R, g1, grad_src, w.f32(ws, RRL_WS_RPART), gR, gt, payload, loss,
                                  w.i32(ws, RRL_WS_INFO), w.i32(ws, RRL_WS_STATUS) + 3, B, N, transpose_r,
                                  s);
    }
    int rc = rrl_loss_backward(w.f32(ws, RRL_WS_TRI1), tri2, ws, ws_bytes, grad_loss, g1, nullptr, B,
                               N, M, L, 0, stream);
    if (rc) return rc;
    rc = rrl_rigid_apply_bwd(src, R, g1, grad_src, gR, gt, w.f32(ws, RRL_WS_RPART), B, 3 * N,
                             transpose_r, 0, stream);
    if (rc) return rc;
    if (payload) rc = rrl_shard_payload(loss, ws, ws_bytes, gR, gt, payload, B, N, M, L, stream);
    return rc;
}
