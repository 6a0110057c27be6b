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
 *     allocator); the library allocates nothing and never synchronises;
 *   - `stream` is a hipStream_t (pass torch's current stream);
 *   - return value: 0 ok, <0 argument error (RRL_E_*), >0 a hipError_t;
 *   - all floating-point data is fp32, dense and contiguous;
 *   - re-entrant and thread-safe per stream.
 *
 * Layouts
 *   tri   [B][N][9]   pseudo-triangles, row = P0 P1 P2 (xyz interleaved)
 *   line  [B][L][6]   dir(3) (unit length, or all zero), x0(3)
 *   ptri  [B][N][12]  prepared triangles: 9 coords, thr2, thr, 0
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

/* status word indices (int32 status[RRL_STATUS_WORDS], zeroed by rrl_loss_begin) */
#define RRL_STATUS_NAN 0 /* negative sqrt argument seen: reference exit(0), loss.py:89-91 */
#define RRL_STATUS_WORDS 4

/* scan modes */
#define RRL_SCAN_STRICT 0 /* evaluate all 3 points of every (line, triangle): exact NaN flag */
#define RRL_SCAN_LAZY 1   /* points 1,2 only where point 0 passes; same labels, NaN flag
                             covers evaluated points only */

const char *rrl_version(void);

/* Prepared triangles: per-triangle threshold thr = mean edge * 1.731 / 2
 * (code/loss.py:94-110) and thr2 = the smallest fp32 x with sqrt(x) >= thr, so
 * that "sqrt(x) < thr" (loss.py:107-110) is decided exactly by "x < thr2". */
int rrl_tri_prepare(const float *tri, float *ptri, int B, int N, void *stream);

/* Zero the per-call state (count1/count2, status, bucket sums). */
int rrl_loss_begin(int32_t *count1, int32_t *count2, int32_t *status, int64_t *bsum,
                   int32_t *bcnt, int B, int L, void *stream);

/* Dense line <-> pseudo-triangle scan of both clouds in one launch
 * (code/loss.py:68-112 for points1 and points2, :181-186).  Emits per line the
 * hit count and the (unordered) indices of the first RRL_MAX_HITS hits; nothing
 * of size L*N is materialised.  count/hit must have been zeroed by rrl_loss_begin.
 * chunk = triangles per workgroup (0 = default). */
int rrl_line_tri_scan(const float *ptri1, const float *ptri2, const float *line,
                      int32_t *count1, int32_t *hit1, int32_t *count2, int32_t *hit2,
                      int32_t *status, int B, int N, int M, int L, int mode, int chunk,
                      void *stream);

/* Tuning/testing knob: lines per lane of the scan kernel (1 = scalar fp32, 2 / 4 = one / two
 * packed v_pk_*_f32 pairs).  All variants produce identical results.  Default 2, or env
 * RRL_SCAN_VARIANT. */
int rrl_set_scan_variant(int lines_per_lane);

/* Per-line sparse stage (code/loss.py:115-167): for lines whose two hit counts
 * fall in [s_m,e_m) x [s_n,e_n): sort hits ascending (nonzero() order), weights
 * w = d / sum d (loss.py:92), intersection points q = mean_k w_k P_k
 * (loss.py:155-163) and D[a][b] = |q1_a - q2_b|^2 (loss.py:165-166).
 *   kj   [B][L]       k | j<<4, 0 = line not selected
 *   hs1  [B][L][4]    sorted hit indices (cloud 1), hs2 likewise
 *   w1   [B][L][4][3] weights, w2 likewise
 *   D    [B][L][16]   row-major k x j block
 *   bcnt [G][16]      lines per bucket (G = pool ? 1 : B)                    */
int rrl_line_pair_dist(const float *tri1, const float *tri2, const float *line,
                       const int32_t *count1, const int32_t *hit1, const int32_t *count2,
                       const int32_t *hit2, uint8_t *kj, int32_t *hs1, int32_t *hs2, float *w1,
                       float *w2, float *D, int32_t *bcnt, int B, int N, int M, int L, int s_m,
                       int s_n, int e_m, int e_n, int pool, void *stream);

/* Lower median (torch.median: sorted[(n-1)/2]) of all selected D values per sample
 * (code/loss.py:223-224).  pool != 0 reproduces the reference's B>1 behaviour:
 * one median, taken over the LAST sample's values.
 *   med [G], nval [G] */
int rrl_lower_median(const uint8_t *kj, const float *D, float *med, int32_t *nval, int B, int L,
                     int pool, void *stream);

/* Welsch weighting + symmetric min/mean reduction (code/loss.py:20-21, 226-230).
 * Bucket sums are accumulated in 2^-40 fixed point (bit-deterministic);
 * rrl_loss_finalize turns them into loss[G], nbuckets[G]. */
int rrl_welsch_reduce_fwd(const uint8_t *kj, const float *D, const float *med, int64_t *bsum,
                          int B, int L, int pool, void *stream);
int rrl_loss_finalize(const int64_t *bsum, const int32_t *bcnt, float *loss, int32_t *nbuckets,
                      int G, int s_m, int s_n, int e_m, int e_n, void *stream);

/* Closed-form backward of the whole loss (autograd of code/loss.py:170-232;
 * SURVEY.md section 8a row G).  grad_tri1 [B][N][9] must be zeroed by the caller;
 * grad_tri2 may be NULL.  grad_loss [G]. */
int rrl_welsch_reduce_bwd(const float *tri1, const float *tri2, const uint8_t *kj,
                          const int32_t *hs1, const int32_t *hs2, const float *w1, const float *w2,
                          const float *D, const float *med, const int32_t *bcnt,
                          const int32_t *nbuckets, const float *grad_loss, float *grad_tri1,
                          float *grad_tri2, int B, int N, int M, int L, int pool, void *stream);

/* Rigid apply (code/loss.py:460-461; rpm/common/math_torch/se3.py:67-72;
 * code/utils.py:32-37; fmr/se_math/se3.py:110-124).
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

/* Chamfer monitor (code/loss.py:38-52, 236-252).  best_x [B][N], best_y [B][M] are
 * u64 keys (dist bits << 32 | argmin), set to all-ones by the call itself.
 * value[0] = mean of all B*(N+M) minima. */
int rrl_chamfer_fwd(const float *x, const float *y, uint64_t *best_x, uint64_t *best_y,
                    float *value, int B, int N, int M, void *stream);
int rrl_chamfer_bwd(const float *x, const float *y, const uint64_t *best_x,
                    const uint64_t *best_y, const float *grad_value, float *gx, float *gy, int B,
                    int N, int M, void *stream);

/* Line sampler (code/loss.py:265-432).
 * rrl_aabb: per-sample min/max -> aabb [B][6] = min xyz, max xyz (loss.py:325-351).
 * rrl_sample_lines: all `rounds` rejection rounds in one launch.
 *   rands [rounds][4][B][n] uniform [0,1) draws in the reference's stream order
 *   r [B], centers [B][3], aabb1/aabb2 [B][6] (both NULL: keep every candidate, loss.py:384-412)
 *   lines [B][n][6] (zeroed by the call), filled [B] = accepted so far (may exceed n) */
int rrl_aabb(const float *v, float *aabb, int B, int n, void *stream);
int rrl_sample_lines(const float *rands, const float *r, const float *centers, const float *aabb1,
                     const float *aabb2, float *lines, int32_t *filled, int B, int n, int rounds,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RRL_H */
