/*
 * rrl.h -- C ABI of librrl_hip.so: the MI355X (gfx950) implementation of the
 * intersected-line robust registration loss.
 *
 * The reference (Dengzhi-USTC/A-robust-registration-loss) is pure Python and has
 * no FFI; its boundary is the import surface of code/loss.py.  These entry
 * points are what a binding for that path attaches to: each one names the
 * reference lines it replaces.  The Python drop-in
 * (a-robust-registration-loss_amd/loss.py) calls them through ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's
 *     allocator); the library allocates no device memory and never
 *     synchronises -- except rrl_loss_forward_info, whose purpose is the drop-in call's one read-back (the optional
 *     scan-timing hook owns a few hipEvents);
 *   - `stream` is a hipStream_t (pass torch's current stream);
 *   - return value: 0 ok, <0 argument error (RRL_E_*), >0 a hipError_t;
 *   - all floating-point data is fp32, dense and contiguous;
 *   - re-entrant and thread-safe per stream AND per workspace: no entry keeps state between calls except what the
 *     caller's buffers hold, the options of a call are resolved once at its top (rrl_opts below), and two host threads
 *     may drive different streams with different options concurrently (tests/test_gpu_threads.py).  Calls that share
 *     a workspace or an output buffer must be ordered by the caller.  Process-wide state: only the DEFAULTS that the
 *     rrl_set_* setters / RRL_* environment variables choose (read once per call), the rrl_scan_counters /
 *     rrl_chamfer_counters hooks (profiling; use the per-call counter fields / *_ex entries from threads) and the
 *     rrl_scan_timing ring (a profiling hook, not thread-safe);
 *
 * Layouts
 *   tri   [B][N][9]   pseudo-triangles, row = P0 P1 P2 (xyz interleaved)
 *   line  [B][L][6]   dir(3) (unit length, or all zero), x0(3)
 *   ws    one caller-allocated workspace of rrl_workspace_bytes() bytes that
 *         carries every intermediate from forward to backward; field offsets
 *         come from rrl_workspace_layout().
 */
#ifndef RRL_H
#define RRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRL_MAX_HITS 4 /* hits kept per line and cloud: callers use buckets 1..4 */
#define RRL_E_ARG (-1)
#define RRL_E_RANGE (-2) /* bucket range outside 1..RRL_MAX_HITS */
#define RRL_E_WS (-3)    /* workspace too small */

/* scan modes */
#define RRL_SCAN_STRICT 0 /* all 3 points of every (line, triangle) are evaluated */
#define RRL_SCAN_LAZY 1   /* points 1,2 only where point 0 passes: same labels; a NaN (negative
                             sqrt argument) is reported only if it occurs in an evaluated pair */
#define RRL_SCAN_AUTO 2   /* per wavefront: lazy where a NaN is provably impossible for its
                             lines (|dir|^2 <= 1+1e-6 and (|x0| + max|P|)^2 <= 100), else strict.
                             Same results AND same NaN detection as strict. */
#define RRL_SCAN_CULL 3   /* default.  Triangles are sorted by grid cell (Hilbert order) under a three-
                             level sphere tree; a line looks only at the halves of 8 whose sphere it can
                             reach, found by a conservative test that is valid at any finite data scale
                             (DESIGN.md "culling bound"); there a conservative FMA prefilter on point 0
                             picks the candidates (~3 %) and the reference's own arithmetic decides on
                             all three points of those: labels, hit lists and the loss equal strict's
                             bit for bit.  NaN detection equals strict's at
                             every scale (round 3): where (|x0| + max|P|)^2 < 111 and |dir|^2 <= 1 + 1e-6 a NaN
                             is provably impossible; lines with |dir|^2 > 1 + 1e-6 or non-finite data send
                             their wavefront of 128 lines through the strict loop (STATUS[1] counts such
                             wavefronts); for unit directions at larger scale (the demo's full-diagonal
                             radius) the walk is widened by every triangle's NaN reach (DEL1 / DEL2), so
                             that each triangle one of whose three points could see a negative sqrt
                             argument is evaluated exactly (csrc/rrl_cull.hip, "NaN").
                             Needs N, M <= 65536, else behaves like AUTO. */

/* workspace fields (indices into rrl_workspace_layout's offset array) */
enum {
    RRL_WS_STATUS = 0, /* int32[4]   [0] = NaN seen (reference exit(0), loss.py:89-91); [1] = wavefronts of the
                          culled scan that fell back to the strict loop; [2] = samples whose exchange reduce was repaired
                          by their last workgroup after a hand-off time-out (rrl_set_spin_limit); [3] = internal ticket */
    RRL_WS_NVALS,      /* int32[B]   (unused since the compact-slot layout; kept for ABI stability) */
    RRL_WS_NSEL,       /* int32[B]   selected lines per sample (length of SEL[b])           */
    RRL_WS_PMAX,       /* uint32[2][B] bits of max |P|^2 per cloud and sample                */
    RRL_WS_COUNT1,     /* int32[B][L] hit count, cloud 1 (loss.py:185)                      */
    RRL_WS_COUNT2,     /* int32[B][L]                                                      */
    RRL_WS_HIT1,       /* int32[B][L][4] unordered hit indices                              */
    RRL_WS_HIT2,
    RRL_WS_PTRI1,      /* float[B][N][12] 9 coords, thr2, thr, original triangle index (int): the scans' records.  Rows in
                          original order after a cold build, at the triangles' SORTED positions after a prepared build
                          (rrl_opts.order1 / order2) -- slot 7 of the cloud's first APART row says which (0 / 1)  */
    RRL_WS_PTRI2,      /* float[B][M][12]                                                   */
    RRL_WS_P0S1,       /* float[B][64*NSG1][4] P0 + thr2 in grid-cell (Hilbert curve) order -- clouds of more than 4096
                          triangles: each chunk of 4096 (by original index) in its own order; NSG = ceil(N/64)   */
    RRL_WS_P0S2,       /*   supergroups of 64 sorted triangles; pad records have thr2 = 0 (never hit)           */
    RRL_WS_IDX1,       /* int32[B][64*NSG1]  original triangle index of each sorted position     */
    RRL_WS_IDX2,
    RRL_WS_GRP1,       /* float[B][NSG1][13][4] sphere tree (centre, conservative radius; NaN = empty): per   */
    RRL_WS_GRP2,       /*   supergroup [0] its own sphere, [1..4] its groups of 16, [5..12] their halves of 8   */
    RRL_WS_CREC1,      /* float[B][16*NG1][4] P0 + thr2 in original order (input of the sort)   */
    RRL_WS_CREC2,
    RRL_WS_APART,      /* float[2][B][ceil(max(N,M)/256)][8] per-workgroup AABB / max |P|^2 partials (slot 7: PTRI layout) */
    RRL_WS_KJ,         /* uint8[B][L]  k | j<<4, 0 = line not selected                      */
    RRL_WS_SEL,        /* int32[B][L]  indices of the selected lines, compacted (any order)  */
    RRL_WS_HS1,        /* int32[B][L][4] ascending hit indices (nonzero() order)            */
    RRL_WS_HS2,
    RRL_WS_W1,         /* float[B][L][4][3] weights d / sum d (loss.py:92)                  */
    RRL_WS_W2,
    RRL_WS_Q1,         /* float[B][L][4][4] intersection points q (xyz, 0) (loss.py:155-163)  */
    RRL_WS_Q2,
    RRL_WS_D,          /* float[B][L][16] k x j block of |q1-q2|^2 (loss.py:165-166)        */
    RRL_WS_VALS,       /* float[B][Lp][16] canonical 4x4 D tiles (+inf padded) of the selected lines
                          at compact slots: slot = 1024 * x + rank for the x-th 1024-line tile,
                          Lp = 1024 * ceil(L/1024) (input of the reduce kernel)                */
    RRL_WS_MED,        /* float[G]  lower median (loss.py:223-224)                          */
    RRL_WS_BCNT,       /* int32[G][16] lines per (k,j) bucket                               */
    RRL_WS_BSUM,       /* int64[G][16][2] bucket sums of row / column minima, 2^-40 fixed pt */
    RRL_WS_INFO,       /* int32[G][4] nbuckets, nselected, nvalues, STATUS[0] (the scan's NaN flag; after a CHAINED step's fused
                          launch: the sample's OWN flag, CHAIN[b][1])  */
    RRL_WS_TRI1,       /* float[B][N][9] transformed source triangles (rrl_registration_*)   */
    RRL_WS_G1,         /* float[B][N][9] gradient w.r.t. TRI1 (rrl_registration_backward)    */
    RRL_WS_RPART,      /* float[B][nblk][12] rigid-apply backward partial sums              */
    RRL_WS_GACC,       /* float[12 B + 16]  dL/dR [B][9], dL/dt [B][3], shard payload [14]: zeroed by
                          rrl_registration_forward, accumulated by rrl_registration_backward     */
    RRL_WS_KJC,        /* uint8[B][Lp]  k | j<<4 at the compact slots                            */
    RRL_WS_BLKCNT,     /* int32[B][ceil(L/1024)] selected lines per 1024-line tile               */
    RRL_WS_HISTG,      /* uint32[2 B][2][4096] cell counts and cursors of the wide sort (clouds > 4096) */
    RRL_WS_DEL1,       /* float[B][N]  NaN reach of a triangle: max(|P1-P0|, |P2-P0|) - thr, clamped at 0, rounded up: how   */
    RRL_WS_DEL2,       /* float[B][M]  much farther than thr points 1, 2 can sit from point 0 (culled scan, NaN detection);
                          rows laid out like PTRI's (original order / sorted positions)                                  */
    RRL_WS_MHIST,      /* uint32[B][2048] histogram of the D values' bits 30..20, accumulated by the per-line stage (the
                          median's first radix pass); MHIST, MCTL, MSUM are contiguous and cleared per call                 */
    RRL_WS_MCTL,       /* uint32[B][64]  [0..15] lines per (k,j) bucket (per-line stage); [16] candidate cursor, [17] / [18]
                          arrival counters of the tiled reduce, [19] its error flag (spin time-out)                          */
    RRL_WS_MSUM,       /* uint64[B][32]  bucket sums of the tiled reduce (2^-40 fixed point, device atomics)                */
    RRL_WS_MCAND,      /* uint32[B][2048] D values (bit patterns) of the median's bin, gathered by the tiled reduce          */
    RRL_WS_LMAX,       /* float[B][64][2] (max |dir|^2, max |x0|^2) over 1/64 of a sample's cullable lines: the culled scan's
                          slacks come from their maxima (written by the records kernel, or by the scan entry itself)       */
    RRL_WS_LIDC,       /* uint32[B][Lp]  line index | (k | j<<4) << 24 at the compact slots (the tail kernel's way from a
                          compact slot back to the per-line arrays)                                                        */
    RRL_WS_VLIST,      /* float[B][ceil(L/1024)][16384]  the valid D values of each 1024-line tile as a dense list (arbitrary order,
                          padded with -1 to a multiple of 4): what the tail kernel streams to find the median                */
    RRL_WS_VLCNT,      /* int32[B][ceil(L/1024)]  their number per tile                                                     */
    RRL_WS_CHAIN,      /* uint32[B][4]  per-sample words of a CHAINED step (RRL_F_CHAIN / RRL_F_CHAINED, round 6): [0] records
                          workgroups of the sample that have finished in the step's build + scan launch, [1] the scan's NaN
                          flag of the sample, [2] its wavefronts that fell back to the strict loop, [3] wait time-outs; all
                          zero between calls (cleared on exit by the sample's last tail workgroup)                          */
    RRL_WS_GFIX,       /* int64[B][N + M][9] + int32[2 B]  deterministic scatter backward (rrl_opts.deterministic with rrl_loss_step_ex /
                          rrl_loss_backward): fixed-point accumulators of dL/dpoints1 (and dL/dpoints2), order-independent like MSUM;
                          behind them one non-finite flag per sample and cloud                                              */
    RRL_WS_FIELDS
};

const char *rrl_version(void);

/* ---- per-call options ------------------------------------------------------------------------
 * Every entry point that has an `_ex` twin takes `const rrl_opts *opts` in front of the stream; NULL (and the
 * plain entry) means "the library defaults", i.e. what rrl_set_* / the RRL_* environment variables selected.
 * A field left at -1 / NULL also means "default".  The options are resolved ONCE at the top of a call and travel
 * through all of its stages by value, so a call never observes a setter running in another thread half way
 * through, and two host threads may drive different streams with different options at the same time (the only
 * process-wide state left are the defaults themselves and the rrl_scan_timing ring, a profiling hook).
 * Thread-safety contract: re-entrant; thread-safe per stream and per workspace -- two calls that
 * share a workspace or output buffers must be ordered by the caller (same stream, or events). */
#define RRL_F_TARGET_KEPT 1  /* with order1: cloud 2's prepared records, sphere tree and partials in `ws` are those
                                the PREVIOUS call on this workspace built from the same tri2 -- the target has not moved
                                (code/test_demo_optimized_Lie_Algebra.py:57-62, rpm/Train_RPM.py:207-231 move only the
                                source) -- so nothing of cloud 2 is rebuilt; order2 is not read.  Same results bit for bit. */
/* CHAINED steps (round 6): a loop that calls rrl_loss_step_ex / rrl_registration_step_ex again and again on ONE workspace
 * with a kept target (the demo, code/test_demo_optimized_Lie_Algebra.py:48-62; a trainer's inner iterations).  The target
 * half of the scan (code/loss.py:181-184: the two scans are independent) needs nothing this step's records launch
 * produces -- only cleared hit counts.  RRL_F_CHAIN asks a step to LEAVE them cleared: the per-line stage zeroes
 * COUNT1 / COUNT2 behind its own read, the sample's last tail workgroup zeroes the CHAIN words.  RRL_F_CHAINED tells a
 * step that it FINDS them cleared: records, target scan and source scan then run as ONE launch -- the source records' body
 * in the leading workgroups of the scan's grid, the target-cloud workgroups next, the source-cloud workgroups last, behind
 * the sample's ready word (CHAIN[b][0]); the line slacks come from the scan tile's own lines instead of LMAX.  Three
 * launches instead of four, same labels / hit lists / loss bits.
 *   - *chain_left (rrl_opts, host memory, written when the call is ISSUED): bit 0 = this call leaves the workspace
 *     chain-clean (the shape is served by the per-line stage + tail kernel / exchange reduce and RRL_F_CHAIN was set); bit 1 =
 *     THIS call's build ran fused (RRL_F_CHAINED was honoured);
 *   - RRL_F_CHAINED is valid only when the PREVIOUS call on this workspace reported chain_left bit 0 and nothing else has
 *     written the workspace since; it needs RRL_F_TARGET_KEPT and is ignored (plain 4-launch step) whenever the fused
 *     launch cannot serve the call (rider, counters, multi-pose, other reduce kernels, thin grids);
 *   - after a step with RRL_F_CHAIN COUNT1 / COUNT2 read zero (KJ / HS1 / HS2 hold what the per-line stage read), so its
 *     workspace cannot serve as another call's target_ws; after a step whose build was fused STATUS is not updated:
 *     INFO[b][3] carries each sample's OWN NaN flag (the plain step reports the batch-wide STATUS[0] in every row), and a
 *     source workgroup that gave up waiting for its records (RRL_CHAIN_SPIN polls; CHAIN[b][3]) makes that sample's loss NaN. */
#define RRL_F_CHAIN 2
#define RRL_F_CHAINED 4
/* A Chamfer walk carried by the evaluation's own scan launch (round 4b).  rrl_chamfer_from_loss -- the monitor every caller
 * of the reference computes next to the loss (rpm/Train_RPM.py:223-224, dcp/Train_DCP.py:246, fmr/model.py:293,
 * test_demo_optimized_Lie_Algebra.py:68) -- needs the evaluation's records launch only, not its scan, but launches of one
 * stream never overlap on this stack; handed to an `_ex` forward / step through rrl_opts.chamfer, the walk's workgroups
 * are issued in the culled scan's grid (one launch for both, the walk's time hidden beside the scan's).  Same arithmetic,
 * same keys and value as rrl_chamfer_from_loss after the call.  done: clear it before the call; the call sets it to 1
 * (host side, before it returns) when the walk rode along -- also with a carried-over target (`target_ws`: only the source is
 * scanned, the walk reads the target in the workspace that holds it).  Still 0: it did not (scan mode other than cull,
 * counters, fewer than 1024 lines, clouds beyond the sort capacity, an entry that runs no scan) -- call
 * rrl_chamfer_from_loss as before. */
typedef struct rrl_chamfer_rider {
    void *ws;                  /* Chamfer workspace, rrl_chamfer_workspace_bytes(B, N, M) */
    size_t ws_bytes;
    uint64_t *best_x, *best_y; /* [B][N], [B][M] keys (distance bits << 32 | argmin), as rrl_chamfer_from_loss */
    float *value;              /* [1] the mean */
    int32_t done;              /* out */
} rrl_chamfer_rider;

typedef struct rrl_opts {
    int32_t struct_bytes;    /* sizeof(rrl_opts) of the caller's header (fields beyond it are defaults) */
    int32_t flags;           /* RRL_F_* */
    int32_t reduce_mode;     /* -1 default; 0 auto, 1 single, 2 tiled, 3 xchg (rrl_set_reduce_mode) */
    int32_t deterministic;   /* -1 default; 0 / 1 (rrl_set_deterministic) */
    int32_t sort_parts;      /* -1 default; 0 automatic, 1..16 (rrl_set_sort_parts) */
    int32_t scan_variant;    /* -1 default; 0, 1, 2, 4, 8 (rrl_set_scan_variant) */
    /* Prepared clouds: order [B][64 ceil(n / 64)] from rrl_cloud_order -- sorted position -> triangle, computed ONCE
     * per cloud in any rigid frame of it.  Both given (or order1 + RRL_F_TARGET_KEPT), scan mode cull: the cell sort
     * leaves the step; one wide launch moves the source, writes the records at their sorted positions and refits
     * the sphere tree.  The first n entries of every row MUST be a permutation of [0, n) (the kernels only clamp the
     * index range: a duplicated or missing triangle silently changes labels and loss).  ANY permutation gives the same
     * labels, hit lists and loss (the tree is a conservative filter, the reference's arithmetic decides): an order taken
     * in another pose of the cloud, or a stale one, only costs time. */
    const int32_t *order1, *order2;
    uint64_t *scan_counters;       /* per-call counter table of the culled scan (see rrl_scan_counters) */
    long long scan_counter_rows;
    rrl_chamfer_rider *chamfer;    /* NULL, or the evaluation's Chamfer walk to be carried by its scan launch (above) */
    /* rrl_loss_step_ex only (round 5): NULL, or float[14] that receives the batch-shard payload of the section-8(d) step
     * { sum of the valid losses, #valid, 0 x 12 } -- the buffer a rank contributes to the all-reduce of the scalar loss
     * (points1.grad stays local, so the (dR, dt) slots of rrl_registration_step's payload stay zero).
     * Ideally the workspace's GACC + 12 B floats (cleared by the step's first launch when R, t are given); any other
     * buffer is cleared by a fill launch first.  Written in the reduce's launch where the scatter rides in it, else by
     * one small launch behind the backward. */
    float *payload;
    /* MULTI-POSE evaluation (round 5): 0, or Bt with B % Bt == 0 and Bt < B.  The iterative trainers evaluate several
     * poses of the SAME source against the SAME target along the SAME lines (rpm/Train_RPM.py:207-231 num_iter,
     * fmr/model.py:292-308 the last three estimates), and all poses are known before the first loss call.  With
     * problems = Bt the fused entries that move the source (rrl_registration_forward_ex / _backward_ex / _step_ex,
     * rrl_loss_step_ex with R, t) take src [Bt][N][9], tri2 [Bt][M][9], line [Bt][L][6] and the orders [Bt][..] with
     * R [B][3][3], t [B][3]: instance s = pose (s / Bt) of problem (s % Bt).  Every output (loss [B], gR [B][9], gt [B][3],
     * grad_tri1 [B][N][9], the workspace of B instances) is per instance and bit-identical to evaluating the B / Bt poses one
     * after the other; the target's scan runs ONCE per problem (instances < Bt), the sources' scans side by side in the same
     * launch.  Scan mode cull, clouds within the sort capacity, no target_ws, pool = 0; RRL_E_ARG otherwise. */
    int32_t problems;
    int32_t *chain_left;     /* NULL, or a HOST int32 that receives the two bits above at issue time (RRL_F_CHAIN) */
} rrl_opts;

size_t rrl_workspace_bytes(int B, int N, int M, int L);
/* offsets[RRL_WS_FIELDS] in bytes from the workspace base */
int rrl_workspace_layout(int B, int N, int M, int L, size_t *offsets);

/* ---- fused entry points (what loss.py calls) ------------------------------------------ */

/* Forward of cal_loss_intersection_batch_whole_median_pts_lines (code/loss.py:170-232) for
 * B samples in 4 launches: prepare -> scan -> per-line distances -> median + Welsch reduce.
 *   loss [G], G = pool ? 1 : B.  pool != 0 reproduces the reference's own B>1 behaviour
 *   (all lines pooled, LAST sample's median: SURVEY.md Q2); pool == 0 gives B independent
 *   losses (what the reference's callers compute by looping B = 1 calls).
 *   loss[g] is 0 and INFO[g][0] == 0 where no (k,j) bucket is populated (the reference
 *   returns (None, None, None), loss.py:231-232).
 * chunk = triangles per workgroup of the scan (0 = default). */
int rrl_loss_forward(const float *tri1, const float *tri2, const float *line, void *ws,
                     size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m, int s_n,
                     int e_m, int e_n, int pool, int mode, int chunk, void *stream);

/* Closed-form backward (autograd of code/loss.py:170-232).
 * grad_loss [G]; grad_tri1 [B][N][9] is zeroed then accumulated; grad_tri2 may be NULL. */
int rrl_loss_backward(const float *tri1, const float *tri2, const void *ws, size_t ws_bytes,
                      const float *grad_loss, float *grad_tri1, float *grad_tri2, int B, int N,
                      int M, int L, int pool, void *stream);

/* Fused training op (the rigid transform of the call sites + the loss: rpm/Train_RPM.py:205-231,
 * dcp/Train_DCP.py:233-270, fmr/model.py:265-313; code/loss.py:458-463 for the demo): the source
 * pseudo-triangles src [B][N][9] are moved by per-sample (R [B][3][3], t [B][3]) -- x R + t, or
 * x R^T + t when transpose_r -- into the workspace (TRI1) and the loss is evaluated against
 * tri2, all in one call.  The backward returns dL/dR, dL/dt (deterministic reduction), optionally
 * dL/dsrc (may be NULL), and, when payload != NULL, the 14-float batch-shard payload
 * { sum of valid losses, #valid, sum_b dR, sum_b dt } for the all-reduce.  pool must be 0.
 * With grad_src == NULL the gradients are accumulated with float atomics in one launch: pass
 * gR = GACC, gt = GACC + 9 B, payload = GACC + 12 B (the workspace field the forward zeroed) to
 * avoid three extra clearing launches; any other buffers are cleared by the call first. */
int rrl_registration_forward(const float *src, const float *R, const float *t, const float *tri2,
                             const float *line, void *ws, size_t ws_bytes, float *loss, int B,
                             int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m,
                             int e_n, int mode, int chunk, void *stream);
/* The same two forwards with the TARGET's scan carried over: target_ws is a workspace of equal
 * (B, N, M, L) that already ran a forward with the same tri2 and line (RPM and FMR evaluate
 * num_iter source poses against one target and one line set, rpm/Train_RPM.py:204-231,
 * fmr/model.py:295-310).  Only the source cloud is prepared/sorted/scanned; the target's hit
 * counts and lists are READ in target_ws by the per-line stage (round 4b; they used to be copied: two launches) -- so
 * target_ws must be the workspace of the FULL evaluation that scanned the target (not of another carried-over one), and
 * it must stay untouched until this call's kernels have run (same stream, or an event).  COUNT2 / HIT2 of `ws` are not
 * written.  target_ws == NULL: identical to the plain call.  The result is bit-identical to the plain call either way. */
int rrl_loss_forward_cached(const float *tri1, const float *tri2, const float *line, void *ws,
                            size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                            int s_n, int e_m, int e_n, int pool, int mode, int chunk,
                            const void *target_ws, void *stream);
int rrl_registration_forward_cached(const float *src, const float *R, const float *t,
                                    const float *tri2, const float *line, void *ws, size_t ws_bytes,
                                    float *loss, int B, int N, int M, int L, int transpose_r,
                                    int s_m, int s_n, int e_m, int e_n, int mode, int chunk,
                                    const void *target_ws, void *stream);
/* rrl_loss_forward_cached followed by the read-back the drop-in call needs: INFO (4 int32 per group: nbuckets,
 * nselected, nvalues, NaN flag) is copied to host_info [4 G] (host memory, ideally pinned) and the call waits
 * for the stream.  The reference's callable returns a tensor, None (no populated bucket, code/loss.py:231-232)
 * or exits (NaN, :88-91): a host-side decision per call by contract, so this is the one entry that synchronises. */
int rrl_loss_forward_info(const float *tri1, const float *tri2, const float *line, void *ws,
                          size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m, int s_n,
                          int e_m, int e_n, int pool, int mode, int chunk, const void *target_ws,
                          int32_t *host_info, void *stream);
/* Opt-in bit-reproducible backward (the reference's CPU autograd is deterministic).  on != 0 (or rrl_opts.deterministic = 1)
 *   - replaces the float atomics of rrl_registration_backward's direct route (grad_src == NULL) by per-workgroup partial
 *     sums added in a fixed order by a second, tiny launch;
 *   - (round 6) makes the SCATTER backward to points1.grad / points2.grad -- rrl_loss_backward, rrl_loss_step_ex -- accumulate
 *     in 64-bit fixed point (workspace field GFIX; the unit is a power of two derived per sample from |dL/dloss|, the
 *     bucket count and the median, so that 2^24 contributions cannot overflow): integer sums do not depend on the order of the
 *     atomics.  One more launch converts them to fp32; rrl_loss_step_ex then runs forward + backward + conversion instead of
 *     carrying the scatter in its reduce launch.  Every contribution is rounded to the unit (2^-38 .. 2^-61 of the largest
 *     possible one): the result agrees with the float-atomic one to ~1e-6 of the largest entry and reproduces bit for bit.
 * Env RRL_DETERMINISTIC=1 sets the initial state.  The forward is deterministic either way. */
int rrl_set_deterministic(int on);
int rrl_registration_backward(const float *src, const float *R, const float *tri2, void *ws,
                              size_t ws_bytes, const float *loss, const float *grad_loss,
                              float *grad_src, float *gR, float *gt, float *payload, int B, int N,
                              int M, int L, int transpose_r, void *stream);

/* Forward + direct backward of the fused training op in ONE call -- what a training step does when dL/dloss is
 * known up front (grad_loss [B], usually ones): rpm/Train_RPM.py:226-259, dcp/Train_DCP.py:246-270 compute the loss
 * and call backward() right away.  Same arguments and results as rrl_registration_forward_cached followed by
 * rrl_registration_backward(grad_src = NULL); where the tail kernel serves the shape (auto mode: 2 .. 32 line tiles per
 * sample and B x tiles <= 256; not in deterministic mode) the backward rides in the reduce's launch -- 4 launches per
 * step with prepared orders (rrl_opts), 5 without -- and the two kernels' chains of dependent loads overlap; a single
 * tile of lines (L <= 1024) is finished by one workgroup per sample (per-line stage + reduce + backward); every other
 * shape runs the forward and then the backward launch.  gR [B][9], gt [B][3], payload [14] or NULL: ideally the
 * workspace's GACC field. */
int rrl_registration_step(const float *src, const float *R, const float *t, const float *tri2,
                          const float *line, void *ws, size_t ws_bytes, float *loss, const float *grad_loss,
                          float *gR, float *gt, float *payload, int B, int N, int M, int L, int transpose_r,
                          int s_m, int s_n, int e_m, int e_n, int mode, int chunk, const void *target_ws,
                          void *stream);

/* ---- prepared clouds (round 4) ----------------------------------------------------------------
 * The reference itself points at a spatial structure (code/loss.py:260-262), and every caller moves the SAME source
 * rigidly, step after step, against a target that never moves (test_demo_optimized_Lie_Algebra.py:57-62,
 * rpm/Train_RPM.py:207-231).  A rigid motion preserves the spatial order of a cloud, so the order is computed once
 * per cloud (dataset item / demo start) and handed to every later call through rrl_opts.order1 / order2.
 * rrl_cloud_order: order [B][64 ceil(n/64)] int32 = sorted position -> triangle index (positions >= n hold 0) for
 * the clouds tri [B][n][9] in the frame they are given in.  Because it runs once it builds a better order than the
 * per-step cell sort: a full K-D ORDER (csrc/rrl_order.hip) -- positions form an implicit binary tree of aligned
 * power-of-two windows, every window sorted along the longest axis of its records' bounding box, down to halves of
 * 8 -- so the scan's tree nodes (aligned runs of 64 / 16 / 8 positions) are compact k-d cells of the WHOLE cloud
 * (the per-step sort orders clouds beyond 4096 triangles in four or more interleaved chunks).  Only the first point
 * of a row places it.  ws: scratch of rrl_cloud_order_workspace_bytes(B, n) bytes.  n <= 65536. */
size_t rrl_cloud_order_workspace_bytes(int B, int n);
int rrl_cloud_order(const float *tri, int32_t *order, void *ws, size_t ws_bytes, int B, int n, void *stream);
/* the same for point clouds pts [B][n][3] (the Chamfer monitor's inputs: rrl_chamfer_tree_fwd_ex); a cloud of
 * pseudo-triangles and the cloud of their first points have the same order */
int rrl_cloud_order_points(const float *pts, int32_t *order, void *ws, size_t ws_bytes, int B, int n, void *stream);

/* The fused entries with per-call options (rrl_opts above; NULL = defaults = the plain entries). */
int rrl_loss_forward_ex(const float *tri1, const float *tri2, const float *line, void *ws, size_t ws_bytes,
                        float *loss, int B, int N, int M, int L, int s_m, int s_n, int e_m, int e_n, int pool,
                        int mode, int chunk, const void *target_ws, const rrl_opts *opts, void *stream);
int rrl_registration_forward_ex(const float *src, const float *R, const float *t, const float *tri2,
                                const float *line, void *ws, size_t ws_bytes, float *loss, int B, int N, int M,
                                int L, int transpose_r, int s_m, int s_n, int e_m, int e_n, int mode, int chunk,
                                const void *target_ws, const rrl_opts *opts, void *stream);
int rrl_registration_backward_ex(const float *src, const float *R, const float *tri2, void *ws, size_t ws_bytes,
                                 const float *loss, const float *grad_loss, float *grad_src, float *gR, float *gt,
                                 float *payload, int B, int N, int M, int L, int transpose_r, const rrl_opts *opts,
                                 void *stream);
int rrl_registration_step_ex(const float *src, const float *R, const float *t, const float *tri2,
                             const float *line, void *ws, size_t ws_bytes, float *loss, const float *grad_loss,
                             float *gR, float *gt, float *payload, int B, int N, int M, int L, int transpose_r,
                             int s_m, int s_n, int e_m, int e_n, int mode, int chunk, const void *target_ws,
                             const rrl_opts *opts, void *stream);

/* SURVEY 8(d) by direct issue: forward + backward to points1.grad in ONE call -- what
 * `loss = cal_loss_intersection_batch_whole_median_pts_lines(...); loss.backward()` delivers through autograd
 * (code/loss.py:170-232: points1.grad (B, N, 9)), with the rigid apply of the call sites in front when R, t are given
 * (code/loss.py:458-463, rpm/Train_RPM.py:205-212): points1 = tri1 R + t (x R^T + t when transpose_r), kept in the
 * workspace field TRI1; R == t == NULL: points1 = tri1 as given.  grad_loss [B] = dL/dloss (usually ones),
 * grad_tri1 [B][N][9] = dL/dpoints1 -- cleared by the call's first launch, accumulated by float atomics like
 * rrl_loss_backward --, grad_tri2 [B][M][9] or NULL.  Where the tail kernel serves the shape (2 .. 32 line tiles,
 * B x tiles <= 256) and grad_tri2 == NULL the scatter rides in the reduce's launch: 4 launches per step with prepared
 * orders (opts), 5 without; a single tile of lines (L <= 1024) is finished by one workgroup per sample (per-line stage +
 * reduce + scatter, one launch: 3 / 4 per step); otherwise forward + the scatter kernel of rrl_loss_backward.  Loss, median, bucket sums
 * bit-identical to rrl_loss_forward / rrl_registration_forward; gradients equal rrl_loss_backward's to the rounding of
 * the atomics.  pool semantics: independent samples (pool = 0). */
int rrl_loss_step_ex(const float *tri1, const float *R, const float *t, const float *tri2, const float *line,
                     void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *grad_tri1,
                     float *grad_tri2, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m,
                     int e_n, int mode, int chunk, const void *target_ws, const rrl_opts *opts, void *stream);

/* ---- the four forward stages, individually (tests, profiling) ------------------------- */

/* K1': prepared triangles for both clouds + zeroing of the per-call state.
 * thr = mean edge * 1.731 / 2 (code/loss.py:94-110); thr2 = the smallest fp32 x with
 * sqrt(x) >= thr, so that "sqrt(x) < thr" (loss.py:107-110) is decided exactly by x < thr2. */
int rrl_tri_prepare(const float *tri1, const float *tri2, void *ws, size_t ws_bytes, int B, int N,
                    int M, int L, void *stream);

/* With options: sort parts; orders (both): the prepared build -- records at their sorted positions + tree refit in one
 * launch, and PMAX completed by a tiny second launch (inside a fused forward the culled scan's prologue does that).
 * rrl_line_tri_scan_ex must then be given the same opts. */
int rrl_tri_prepare_ex(const float *tri1, const float *tri2, void *ws, size_t ws_bytes, int B, int N,
                       int M, int L, const rrl_opts *opts, void *stream);

/* K1: dense line <-> pseudo-triangle scan of both clouds in one launch
 * (code/loss.py:68-112 for points1 and points2, :181-186).  Emits per line the hit count and
 * the (unordered) indices of the first RRL_MAX_HITS hits; nothing of size L*N is written. */
int rrl_line_tri_scan(const float *line, void *ws, size_t ws_bytes, int B, int N, int M, int L,
                      int mode, int chunk, void *stream);

int rrl_line_tri_scan_ex(const float *line, void *ws, size_t ws_bytes, int B, int N, int M, int L,
                         int mode, int chunk, const rrl_opts *opts, void *stream);
/* (K2 below with options: the reduce mode decides whether the dense value lists of the tail kernel are written, so the
 * stage calls of ONE evaluation must be given the same opts) */
int rrl_line_pair_dist_ex(const float *tri1, const float *tri2, const float *line, void *ws,
                          size_t ws_bytes, int B, int N, int M, int L, int s_m, int s_n, int e_m,
                          int e_n, int pool, const rrl_opts *opts, void *stream);

/* K2: per-line sparse stage (code/loss.py:115-167): for lines whose two hit counts fall in
 * [s_m,e_m) x [s_n,e_n): hits sorted ascending (nonzero() order), weights w = d / sum d
 * (loss.py:92), intersection points q = mean_k w_k P_k (loss.py:155-163),
 * D[a][b] = |q1_a - q2_b|^2 (loss.py:165-166), and the D values appended to VALS. */
int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line, void *ws,
                       size_t ws_bytes, int B, int N, int M, int L, int s_m, int s_n, int e_m,
                       int e_n, int pool, void *stream);

/* K3+K4: lower median (torch.median: sorted[(n-1)/2], loss.py:223-224), Welsch weighting
 * 1 - exp(-(D/med)/2) with symmetric min/mean per bucket (loss.py:20-21, 226-229), bucket
 * weights exp(-|k-j|/2) and the final division by the number of non-empty buckets
 * (loss.py:215-217, 230).  One workgroup per sample; bucket sums in 2^-40 fixed point, so
 * the result does not depend on summation order. */
int rrl_loss_reduce(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                    int s_n, int e_m, int e_n, int pool, void *stream);

int rrl_loss_reduce_ex(void *ws, size_t ws_bytes, float *loss, int B, int N, int M, int L, int s_m,
                       int s_n, int e_m, int e_n, int pool, const rrl_opts *opts, void *stream);

/* K3 + K4 over caller-supplied rows: rows16 [nrows][16] = canonical 4 x 4 D tiles (+inf outside the k x j block) and
 * kj [nrows] = k | j << 4, i.e. what rrl_line_pair_dist leaves at its compact slots (fields VALS, KJC, BLKCNT) -- the merge
 * step of the line-sharded single-sample mode ("within one sample, lines could also be sharded ... one all-gather
 * of <= 16 S D values"; rrl_hip/dist.py line_sharded_loss): every rank evaluates its share of the lines up to the per-line
 * stage, the rows of all ranks are gathered and reduced here.  Outputs as the fields MED [1], BCNT [16], BSUM [32], INFO [4]
 * (INFO[3] = status[0]) and loss [1]: point them at a rank's own workspace fields and its backward entries use the merged
 * statistics.  blkcnt_scratch: int32 [ceil(nrows / 1024) + 1] device scratch.  Bit-identical to the unsharded loss. */
int rrl_loss_reduce_rows(const float *rows16, const uint8_t *kj, int nrows, int32_t *blkcnt_scratch, float *loss,
                         float *med, int32_t *bcnt, int64_t *bsum, int32_t *info, const int32_t *status, int s_m,
                         int s_n, int e_m, int e_n, void *stream);

/* Which reduce kernel rrl_loss_reduce (and the fused forwards) launch -- the DEFAULT; a call's rrl_opts.reduce_mode
 * overrides it.  Three kernels, bit-identical median, loss and bucket sums (csrc/rrl_sparse.hip reduce_kind):
 *   single   one 1024-lane workgroup per sample (always legal; the only one for pool != 0);
 *   xchg     loss_reduce_tiled_kernel: one 256-lane workgroup per 1024-line tile; the median's first radix pass comes as a
 *            histogram from the per-line stage, the tiles exchange the values of the median's bin through the workspace
 *            (two tickets, bounded spins; a hand-off that times out is repaired by the sample's last workgroup, see
 *            rrl_set_spin_limit).  Legal while B x tiles <= the number of its workgroups that are co-resident on the
 *            device (compute units x occupancy, queried once per device; 1280 on a whole MI355X);
 *   tail     loss_tail_kernel: no exchange (every workgroup streams its sample's dense D-value lists and selects the
 *            median itself, one ticket, no spin), and the direct / scatter backward can ride in the same launch.
 *            Legal for <= 32 tiles per sample and B x tiles <= 256.
 * mode 0 (auto): tail where a backward rides along (rrl_registration_step, rrl_loss_step), the sample has 2 .. 32 tiles
 *   and B x tiles <= 256; else xchg for >= 2 tiles within its capacity; else single.  A forward alone never takes the
 *   tail kernel in auto mode (as a reduce alone it is within +-1 % of xchg: the per-line stage then writes the value lists).  A single tile of lines (L <= 1024) is
 *   finished by one workgroup per sample together with the per-line stage (and the direct backward).
 * mode 1: single everywhere.  mode 2 ("tiled"): tail wherever it is legal (also forward only, also one tile), xchg
 * beyond, else single.  mode 3 ("xchg"): xchg wherever it is legal (also one tile), else single.
 * Env RRL_REDUCE=single|tiled|xchg sets the initial default. */
int rrl_set_reduce_mode(int mode);

/* Workgroups per cloud of the cell sort + sphere-tree kernel (they share nothing but their input: each owns a range
 * of supergroups): 0 = default (one: more measured no faster, csrc/rrl_cull.hip sort_parts), k = 1..16 forced.  Any value gives the same labels, loss and Chamfer
 * keys; the order of records INSIDE a grid cell may differ.  Env RRL_SORT_PARTS sets the initial state. */
int rrl_set_sort_parts(int parts);

/* Test hooks of the in-launch hand-offs.  rrl_set_spin_limit: polls a waiting workgroup of the exchange reduce makes
 * before it gives up (default 2^18, a fraction of a second; env RRL_SPIN_LIMIT).  A hand-off that times out -- its
 * partners were not co-resident in time: a partitioned device, a CU mask, another process holding the slots -- is
 * REPAIRED inside the same launch by the sample's last workgroup (bit-identical result; STATUS[2] counts the repaired
 * samples); 0 makes every hand-off time out, which is how the tests drive that path.  rrl_debug_occupy: a filler launch
 * of `workgroups` x `lanes` threads that hold their compute-unit slots until the 100 MHz wall clock has advanced by
 * `ticks` (contention tests: a second stream / process that leaves the library only part of the device). */
int rrl_set_spin_limit(long long polls);
int rrl_debug_occupy(int workgroups, int lanes, long long ticks, void *stream);

/* Tuning/testing knobs.  rrl_set_scan_variant: lines per lane of the scan (1 = scalar fp32,
 * 2 / 4 / 8 = one / two / four packed v_pk_*_f32 pairs); 0 = default.  All variants give
 * identical results.  Env RRL_SCAN_VARIANT / RRL_SCAN_CHUNK override the defaults. */
int rrl_set_scan_variant(int lines_per_lane);

/* Profiling hook: enable(k > 0) brackets every k-th scan launch with a hipEvent pair (ring of
 * 1024) recorded on the launch stream; enable(0) turns it off.  collect() synchronises them,
 * writes up to max_n durations in milliseconds and resets the ring.  Returns the number written. */
int rrl_scan_timing_enable(int on);
int rrl_scan_timing_collect(float *ms, int max_n);

/* Profiling hook: executed work of the culled scan.  While dev_counters != NULL every culled scan
 * launches an instrumented instantiation of the same kernel in which every wavefront WRITES one row of 16
 * uint64 (plain stores; row = linear workgroup id x wavefronts per workgroup + wavefront; rows >= `rows`
 * are dropped; the caller clears the buffer and adds the rows up):
 *   [0] level-A sphere tests (line x supergroup)   [1] level-B (line x group)   [2] level-C (line x half)
 *   [3] point-0 prefilter tests (line x record)    [4] candidates resolved exactly (all 3 points)
 *   [5] 1 (the wavefront ran)                      [6] 1 if it took the strict fallback
 *   [7] (line, triangle) pairs evaluated by the fallback   [8], [9] its start / end on the 100 MHz wall clock;
 *   [10..14] phase stamps on the same clock: slice staged, level A done, final drains of levels B / C / D done.
 * NULL switches back to the plain kernel.  bench.py derives the executed flops of a launch from
 * these (12 per sphere test, 11 per prefilter test, 48 per resolved candidate, 48 per fallback pair). */
int rrl_scan_counters(uint64_t *dev_counters, long long rows);

/* Batch-shard payload: out[14] = { sum of valid losses, number of valid
 * samples, sum_b gR[b] (9), sum_b gt[b] (3) } in one launch, fixed summation order; this is
 * the buffer a rank hands to the RCCL all-reduce.  gR / gt may be NULL (zeros). */
int rrl_shard_payload(const float *loss, const void *ws, size_t ws_bytes, const float *gR,
                      const float *gt, float *out, int B, int N, int M, int L, void *stream);

/* ---- rigid apply ----------------------------------------------------------------------- */
/* code/loss.py:460-461; rpm/common/math_torch/se3.py:67-72; code/utils.py:32-37;
 * fmr/se_math/se3.py:110-124.
 *   transpose_r = 0: y = x R + t   (row-vector convention of Reconstruction_point)
 *   transpose_r = 1: y = x R^T + t (= R x + t per point: RPM/DCP/FMR)
 *   channel_first = 0: x,y are [B][n][3];  1: [B][3][n] (DCP)
 * R [B][3][3], t [B][3]. */
int rrl_rigid_apply_fwd(const float *x, const float *R, const float *t, float *y, int B, int n,
                        int transpose_r, int channel_first, void *stream);
/* gx may be NULL.  partial [B][nblk][12] scratch, nblk = rrl_rigid_bwd_blocks(n);
 * gR [B][3][3], gt [B][3] are overwritten (deterministic two-stage reduction). */
int rrl_rigid_bwd_blocks(int n);
int rrl_rigid_apply_bwd(const float *x, const float *R, const float *gy, float *gx, float *gR,
                        float *gt, float *partial, int B, int n, int transpose_r,
                        int channel_first, void *stream);

/* ---- pose kernels of the single-pair demo ------------------------------------------------- */
/* code/loss.py:437-463 (Reconstruction_point.Transform), code/LieAlgebra/se3.py:83-106 (exp3),
 * sinc.py:5-17, 91-103, 120-132.  xi [B][6] = (w, v) -> R [B][3][3] (row-major), T [B][3].
 * rrl_se3_exp_bwd: gxi [B][6] = d<gR, R> + d<gT, T> / dxi (gR or gT may be NULL = zero). */
int rrl_se3_exp(const float *xi, float *R, float *T, int B, void *stream);
int rrl_se3_exp_bwd(const float *xi, const float *gR, const float *gT, float *gxi, int B, void *stream);
/* torch.optim.Adam's step (test_demo_optimized_Lie_Algebra.py:42, 64-66) on p [n] with device-side
 * scalars: state[0] = step count (float), lr[0]; b1, b2, eps are DOUBLES like the Python floats torch.optim.Adam
 * holds (its bias corrections are double arithmetic; in fp32 1 - 0.999^t is 3e-5 off at small t); the update is
 * skipped when gate != NULL and gate[0] <= 0 (the demo's `if loss_di is not None`; gate = INFO[0] of the loss). */
int rrl_adam_gated(float *p, const float *g, float *m, float *v, float *state, const float *lr,
                   const int32_t *gate, int n, double b1, double b2, double eps, void *stream);

/* One row of the demo's per-epoch log (code/test_demo_optimized_Lie_Algebra.py:72-82 prints / logs
 * loss and Chamfer) written on the device, so a captured step needs no host read-back:
 * table[cursor[0]][0..2] = (loss[0], value[0], info[0] > 0 ? 1 : 0), cursor[0] += 1; rows outside
 * [0, nrows) are dropped; row (3 floats, may be NULL) receives a copy. */
int rrl_log_row(const float *loss, const float *value, const int32_t *info, float *table,
                long long *cursor, long long nrows, float *row, void *stream);

/* The pose side of one epoch of the demo loop (code/test_demo_optimized_Lie_Algebra.py:55-82: backward through
 * model.Transform(), optimizer.step(), the next epoch's Transform(), the printed/logged scalars) for ONE pose in one
 * launch: gxi = d/dxi <gR, R(xi)> + <gT, T(xi)> (rrl_se3_exp_bwd), the gated Adam step on xi (rrl_adam_gated),
 * (R, T) = exp(updated xi) (rrl_se3_exp) and, when table and cursor are given, the log row (rrl_log_row with
 * info = gate).  Bit-identical to the four calls.  gxi, loss, value, table, cursor, row may be NULL.
 * box (6 floats, optional): min xyz, max xyz over aabb_rows [n_aabb_rows][8] -- the per-workgroup partial rows (APART
 * field, cloud 1 of sample 0: rows of (min xyz, max xyz, ..)) that the loss step's records launch just wrote for the
 * MOVED source's first points: the next epoch's sampler box (test_demo…:46-51 samples against the previous epoch's moved
 * source) without a rigid-apply + AABB launch of its own. */
int rrl_se3_adam_step(float *xi, const float *gR, const float *gT, float *m, float *v, float *state,
                      const float *lr, const int32_t *gate, double b1, double b2, double eps, float *R, float *T,
                      float *gxi, const float *loss, const float *value, float *table, long long *cursor,
                      long long nrows, float *row, const float *aabb_rows, int n_aabb_rows, float *box, void *stream);

/* One epoch of the single-pair demo (code/test_demo_optimized_Lie_Algebra.py:46-82) as ONE call: line sampler with the
 * library's generator (rrl_sample_lines_rng, against box1 = the previous epoch's moved source) -> fused registration step
 * (rrl_registration_step_ex; pass prepared orders in opts) with the Chamfer walk between the step's own sorted clouds
 * riding in its scan launch (rrl_chamfer_rider; = rrl_chamfer_from_loss: requires the point sets to be the triangles' first
 * points; RRL_DEMO_RIDE=0 in the environment: the separate launch) -> pose step (rrl_se3_adam_step: exp-map backward,
 * gated Adam on xi, (R, T) = exp(updated xi), log row, box1 = the moved source's AABB for the next epoch).  Nothing but
 * the entries' own kernels, 8 launches issued back to back (6 with `pipeline`: the next epoch's sampler passes ride in this
 * epoch's per-line and backward launches): a loop pays one host call per epoch.  One pair
 * (B = 1); every pointer a device pointer as in the four entries; struct_bytes = sizeof(rrl_demo_epoch_args). */
typedef struct rrl_demo_epoch_args {
    int32_t struct_bytes, N, M, L, rounds, transpose_r;
    /* sampler */
    uint64_t *rng_state; const float *radius, *centers; float *box1; const float *box2; float *lines; int32_t *filled, *tile_counts;
    /* loss step: src_tri [N][9], tar_tri [M][9], pose (R [9], T [3]: in / out), workspace of rrl_workspace_bytes(1, N, M, L) */
    const float *src_tri, *tar_tri; float *R, *T; void *ws; size_t ws_bytes; float *loss; const float *grad_loss; float *gR, *gt;
    const rrl_opts *opts;
    /* Chamfer monitor from the loss state: scratch of rrl_chamfer_workspace_bytes(1, N, M) */
    void *cham_ws; size_t cham_ws_bytes; uint64_t *best_x, *best_y; float *cham_value;
    /* pose: xi [6], Adam moments / state / lr on the device, log table [table_rows][3] + cursor, row [3] */
    float *xi, *m, *v, *adam_state; const float *lr; double b1, b2, eps; float *table; long long *cursor; long long table_rows; float *row;
    /* NULL, or one HOST int32 the caller zeroes before the first epoch and then leaves alone: with it the sampler is software-
     * pipelined across epochs -- the COUNT pass of epoch k + 1 (it needs the moved source's box = the partial rows of epoch
     * k's records launch, not box1 from the pose launch) rides in the per-line launch of epoch k, its WRITE pass in the
     * direct-backward launch of epoch k (nothing after the per-line stage reads `lines`); the word remembers what the next
     * call may skip (bit 0: the count pass, bit 1: the write pass too).  radius / centers / box2 / L / rounds must not
     * change while it is non-zero, and `lines` belongs to the library between two calls. */
    int32_t *pipeline;
} rrl_demo_epoch_args;
int rrl_demo_epoch(const rrl_demo_epoch_args *args, void *stream);

/* ---- Chamfer monitor (code/loss.py:38-52, 236-252) -------------------------------------- */
/* best_x [B][N], best_y [B][M] are u64 keys (dist bits << 32 | argmin), set to all-ones by
 * the call itself.  value[0] = mean of all B*(N+M) minima. */
int rrl_chamfer_fwd(const float *x, const float *y, uint64_t *best_x, uint64_t *best_y,
                    float *value, int B, int N, int M, void *stream);
/* The same result (keys bit-identical, value from a fixed-order sum) through the scan's spatial
 * structures: both clouds in grid-cell (Hilbert) order under the sphere tree, nearest neighbours by a
 * pruned tree walk, the mean folded into the same launch (2 launches for N, M <= 4096).  ws: scratch of
 * rrl_chamfer_workspace_bytes(B, N, M) bytes.  N, M in [1, 65536].  best_x / best_y need no
 * initialisation.  A NaN coordinate in a target cloud makes every minimum of that sample NaN and a
 * NaN query its own minimum, as torch.min does. */
size_t rrl_chamfer_workspace_bytes(int B, int N, int M);
int rrl_chamfer_tree_fwd(const float *x, const float *y, void *ws, size_t ws_bytes, uint64_t *best_x,
                         uint64_t *best_y, float *value, int B, int N, int M, void *stream);
/* Chamfer between the two clouds of a LOSS EVALUATION without sorting them again: the loss forward
 * (rrl_loss_forward*, rrl_registration_forward*, or rrl_tri_prepare) left the sorted P0 records, their
 * original indices and the sphere trees of both clouds in its workspace, and the first point of every
 * pseudo-triangle is the point of the cloud it was built from (code/loss.py:473-485: row = [P, neighbour,
 * neighbour]); the fused op's source records are the MOVED source.  This replaces the trainers'
 * chamfer_dist(moved source points, target points) next to the loss (rpm/Train_RPM.py:223-224) when the
 * point sets ARE the triangles' first points: keys and value equal rrl_chamfer_fwd on (P0 of cloud 1,
 * P0 of cloud 2).  ws_src: the evaluation's workspace; ws_tar: the workspace holding cloud 2's records
 * (the same one, or the target_ws the evaluation was carried over from); both of layout (B, N, M, L).
 * ws: scratch of rrl_chamfer_workspace_bytes(B, N, M).  N, M <= 65536 (larger clouds are not sorted).
 * ONE launch (round 3: the mean is finished by the last workgroups to arrive; their counters are words of ws_src's MCTL
 * field, which is why ws_src is not const).  A non-finite (or overflowing) coordinate anywhere in a cloud gives NaN minima. */
int rrl_chamfer_from_loss(void *ws_src, const void *ws_tar, size_t loss_ws_bytes, int B, int N, int M,
                          int L, void *ws, size_t ws_bytes, uint64_t *best_x, uint64_t *best_y, float *value,
                          void *stream);
/* Profiling hook like rrl_scan_counters: while dev_counters != NULL rrl_chamfer_tree_fwd runs an
 * instrumented instantiation that WRITES one row of 16 uint64 per wavefront of the walk (row index =
 * workgroup * wavefronts per workgroup + wavefront; a full table has 8 * 2 B * ceil(max(N,M)/64) rows; rows >= `rows`
 * are dropped; cleared by the caller): [0] patch-level leaf tests, [1] per-lane leaf tests, [2] (query, leaf)
 * entries evaluated, [3] (query, target) pairs evaluated, [4] 1, [5..7] / [9..14] shader clocks of the phases. */
int rrl_chamfer_counters(uint64_t *dev_counters, long long rows);
/* values[g] = the Chamfer mean (code/loss.py:249-252) over samples [g B / G, (g + 1) B / G) of the evaluation whose walk
 * (rrl_chamfer_from_loss / _ex, or the rider of an `_ex` forward) ran on the Chamfer workspace cham_ws -- the monitor per
 * ITERATION of a multi-pose evaluation (rrl_opts.problems: group g = pose g of every problem; rpm/Train_RPM.py:223-224 logs
 * the distance of every iteration's moved source).  From the per-(sample, direction) sums the walk left in cham_ws. */
int rrl_chamfer_group_means(const void *cham_ws, size_t cham_ws_bytes, float *values, int G, int B, int N, int M,
                            void *stream);
/* The two tree-walk entries with the counter table given PER CALL (NULL: the plain kernel) instead of through the
 * process-wide hook above: two threads / streams can profile independently.  rrl_chamfer_tree_fwd_ex also takes
 * PREPARED clouds: order_x [B][64 ceil(N/64)], order_y [B][64 ceil(M/64)] from rrl_cloud_order on the same clouds in any
 * rigid pose (point clouds: call it on (B, n, 9) rows whose first three floats are the points, or on the pseudo-
 * triangles they are the first points of) -- both given, the per-call sort is replaced by one wide launch that writes
 * the records at their sorted positions and refits the sphere tree; keys and value are the same bits (any permutation
 * gives the same minima and first-occurrence argmins). */
int rrl_chamfer_tree_fwd_ex(const float *x, const float *y, void *ws, size_t ws_bytes, uint64_t *best_x,
                            uint64_t *best_y, float *value, int B, int N, int M, const int32_t *order_x,
                            const int32_t *order_y, uint64_t *counters, long long counter_rows, void *stream);
int rrl_chamfer_from_loss_ex(void *ws_src, const void *ws_tar, size_t loss_ws_bytes, int B, int N, int M,
                             int L, void *ws, size_t ws_bytes, uint64_t *best_x, uint64_t *best_y, float *value,
                             uint64_t *counters, long long counter_rows, void *stream);
int rrl_chamfer_bwd(const float *x, const float *y, const uint64_t *best_x,
                    const uint64_t *best_y, const float *grad_value, float *gx, float *gy, int B,
                    int N, int M, void *stream);

/* ---- dense (line x triangle) tables of code/loss.py:68-112 ------------------------------- */
/* What cal_intersection_batch2_points_with_line returns besides the expanded view of its input:
 * norm_d [B*L][N][3] = d_k / ((d_0 + d_1) + d_2), label [B][L][N] (0/1 bytes), status[0] |= 1 on
 * a NaN distance.  The loss path never materialises these; provided for callers of that public
 * function.  L, B <= 65535. */
int rrl_dense_scan(const float *tri, const float *line, float *norm_d, uint8_t *label,
                   int32_t *status, int B, int N, int L, void *stream);

/* ---- line sampler (code/loss.py:265-432) ------------------------------------------------- */
/* rrl_aabb: per-sample min/max -> aabb [B][6] = min xyz, max xyz (loss.py:325-351).
 * rrl_sample_lines: all `rounds` rejection rounds in two wide launches, slot order == the
 *   reference's candidate order.
 *   rands [rounds][4][B][n] uniform [0,1) draws in the reference's stream order
 *   r [B], centers [B][3], aabb1/aabb2 [B][6] (both NULL: keep every candidate, loss.py:384-412)
 *   lines [B][n][6] (unfilled rows zeroed by the call), filled [B] = accepted (may exceed n)
 *   tile_counts: scratch, 8-byte aligned, int32 [B * rounds * ceil(n/1024) * 32] (one 64-bit accept
 *     ballot per wavefront of every tile of 1024 candidates) */
int rrl_aabb(const float *v, float *aabb, int B, int n, void *stream);
/* y = x R + t (rrl_rigid_apply_fwd, row layout) and aabb [B][6] of y (rrl_aabb) in one launch, one workgroup per
 * sample: for loops whose clouds are small enough that either is launch latency (the demo's epoch). */
int rrl_rigid_apply_aabb(const float *x, const float *R, const float *t, float *y, float *aabb, int B, int n,
                         int transpose_r, void *stream);
/* The resampler's accept test on caller-supplied lines (code/loss.py:265-322, 415-432: label1 * label2):
 * lines [B][n][6], aabb1/aabb2 [B][6] -> mask [B][n] (bit 0: the line crosses >= 1 of the 12 triangles
 * of box 1 by the reference's sub-area test; bit 1: same for box 2 -- accepted == both; bit 2: the
 * sampler's conservative slab pre-test passes) and, when hits != NULL, hits [B][n][2] = the number of
 * box triangles crossed (the reference's label1, label2).  Same device code as rrl_sample_lines. */
int rrl_box_accept(const float *lines, const float *aabb1, const float *aabb2, uint8_t *mask,
                   int32_t *hits, int B, int n, void *stream);
int rrl_sample_lines(const float *rands, const float *r, const float *centers, const float *aabb1,
                     const float *aabb2, float *lines, int32_t *filled, int32_t *tile_counts, int B,
                     int n, int rounds, void *stream);
/* The same with the uniforms drawn INSIDE the kernels by the library's counter-based generator (Philox4x32-10) instead
 * of read from `rands` -- the GPU-side stream for training loops and captured steps (same distribution as torch.rand:
 * 24-bit uniforms in [0, 1); a different stream from both torch generators).  rng_state: 4 uint64 on the device,
 * [0] seed, [1] call counter (every call draws a fresh block and advances it), [2] internal ticket (zero), [3] unused.
 * A captured graph replays correctly: the counter lives on the device. */
int rrl_sample_lines_rng(uint64_t *rng_state, const float *r, const float *centers, const float *aabb1,
                         const float *aabb2, float *lines, int32_t *filled, int32_t *tile_counts, int B,
                         int n, int rounds, void *stream);

/* ---- pseudo-triangle builder (code/loss.py:473-485 + code/utils.py:275-296) --------------- */
/* Farthest-point sampling of S <= n points per cloud, starting at start[b] (the reference draws it
 * with torch.randint): out_idx [B][S] in selection order.  pts [B][n][3]; dist_scratch [B][n]. */
int rrl_fps(const float *pts, const int32_t *start, int32_t *out_idx, float *dist_scratch, int B,
            int n, int S, void *stream);
/* 3 nearest neighbours (itself first) of the points query_idx [B][S] among the n points of their
 * cloud: nn [B][S][3], ascending distance, ties to the lower index (sklearn KDTree.query, k=3). */
int rrl_knn3(const float *pts, const int32_t *query_idx, int32_t *nn, int B, int n, int S,
             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RRL_H */
