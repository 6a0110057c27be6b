// rrl_ws.h -- layout of the caller-allocated workspace (fields: include/rrl.h RRL_WS_*).
// The first six fields (status, nvals, nsel, pmax, count1, count2) are contiguous so that one
// hipMemsetAsync clears all per-call state.  Every field starts on a 256-byte boundary.
#pragma once
#include "rrl_common.h"

struct WsLayout {
    size_t off[RRL_WS_FIELDS];
    size_t total, zero_bytes;
    size_t state_off, state_bytes;  // MHIST .. MSUM: per-call state of the tiled reduce (cleared by the records kernel)

    __host__ WsLayout(int B, int N, int M, int L) {
        const size_t b = (size_t)(B > 0 ? B : 0), n = (size_t)(N > 0 ? N : 0),
                     m = (size_t)(M > 0 ? M : 0), l = (size_t)(L > 0 ? L : 0);
        const size_t bytes[RRL_WS_FIELDS] = {
            4 * 4,               // STATUS
            4 * b,               // NVALS
            4 * b,               // NSEL
            4 * 2 * b,           // PMAX
            4 * b * l,           // COUNT1
            4 * b * l,           // COUNT2
            4 * b * l * 4,       // HIT1
            4 * b * l * 4,       // HIT2
            4 * b * n * 12,      // PTRI1
            4 * b * m * 12,      // PTRI2
            16 * b * ((n + 63) / 64) * 64,  // P0S1 (padded to whole supergroups of 64)
            16 * b * ((m + 63) / 64) * 64,  // P0S2
            4 * b * ((n + 63) / 64) * 64,   // IDX1
            4 * b * ((m + 63) / 64) * 64,   // IDX2
            16 * b * ((n + 63) / 64) * 13,  // GRP1 (sphere tree: 13 float4 per supergroup)
            16 * b * ((m + 63) / 64) * 13,  // GRP2
            16 * b * ((n + 15) / 16) * 16,  // CREC1
            16 * b * ((m + 15) / 16) * 16,  // CREC2
            4 * 2 * b * 8 * (((n > m ? n : m) + 255) / 256),  // APART
            b * l,               // KJ
            4 * b * l,           // SEL
            4 * b * l * 4,       // HS1
            4 * b * l * 4,       // HS2
            4 * b * l * 12,      // W1
            4 * b * l * 12,      // W2
            4 * b * l * 16,      // Q1
            4 * b * l * 16,      // Q2
            4 * b * l * 16,      // D
            4 * b * ((l + 1023) / 1024) * 1024 * 16,  // VALS (canonical D tiles by compact slot)
            4 * b,               // MED   (G <= B)
            4 * b * 16,          // BCNT
            8 * b * 32,          // BSUM
            4 * b * 4,           // INFO
            4 * b * n * 9,       // TRI1
            4 * b * n * 9,       // G1
            4 * b * 12 * ((3 * n + 1023) / 1024 + 1),    // RPART
            4 * (b * 12 + 16),                           // GACC (followed by KJC: see rrl_launch_tri_build)
            b * ((l + 1023) / 1024) * 1024,              // KJC
            4 * b * ((l + 1023) / 1024 + 1),             // BLKCNT
            (n > 4096 || m > 4096) ? 4 * 2 * b * 2 * 4096 : 16,  // HISTG (wide sort of large clouds)
            4 * b * n,           // DEL1
            4 * b * m,           // DEL2
            4 * b * 2048,        // MHIST  (MHIST .. MSUM: contiguous, cleared per call: state_bytes)
            4 * b * 64,          // MCTL
            8 * b * 32,          // MSUM
            4 * b * 2048,        // MCAND
            4 * b * 64 * 2,      // LMAX
            4 * b * ((l + 1023) / 1024) * 1024,  // LIDC
            4 * b * ((l + 1023) / 1024) * 16384, // VLIST
            4 * b * ((l + 1023) / 1024 + 1),     // VLCNT
            4 * b * 4,                           // CHAIN (chained steps: include/rrl.h RRL_F_CHAIN)
            8 * b * (n + m) * 9 + 4 * 2 * b,     // GFIX (deterministic scatter backward)
        };
        size_t o = 0;
        for (int i = 0; i < RRL_WS_FIELDS; ++i) {
            off[i] = o;
            o += (bytes[i] + 255) & ~(size_t)255;
            if (i == RRL_WS_COUNT2) zero_bytes = o;
            if (i == RRL_WS_MSUM) { state_off = off[RRL_WS_MHIST]; state_bytes = o - state_off; }
        }
        total = o;
    }
    __host__ float *f32(void *ws, int f) const { return (float *)((char *)ws + off[f]); }
    __host__ int32_t *i32(void *ws, int f) const { return (int32_t *)((char *)ws + off[f]); }
    __host__ uint8_t *u8(void *ws, int f) const { return (uint8_t *)((char *)ws + off[f]); }
    __host__ int64_t *i64(void *ws, int f) const { return (int64_t *)((char *)ws + off[f]); }
    __host__ uint32_t *u32(void *ws, int f) const { return (uint32_t *)((char *)ws + off[f]); }
    __host__ const float *f32(const void *ws, int f) const { return (const float *)((const char *)ws + off[f]); }
    __host__ const int32_t *i32(const void *ws, int f) const { return (const int32_t *)((const char *)ws + off[f]); }
    __host__ const uint8_t *u8(const void *ws, int f) const { return (const uint8_t *)((const char *)ws + off[f]); }
    __host__ const uint32_t *u32(const void *ws, int f) const { return (const uint32_t *)((const char *)ws + off[f]); }
};

// The options of ONE call, resolved once at its top (include/rrl.h rrl_opts; defaults = what the rrl_set_* setters /
// RRL_* environment variables selected) and handed through its stages by value: no stage reads a process-wide knob.
typedef rrl_chamfer_rider RrlChamRider;  // include/rrl.h: the evaluation's Chamfer walk, carried by the culled scan's launch

// (internal, rrl_demo_epoch) The COUNT pass of the NEXT epoch's line sampler, carried by this evaluation's per-line launch
// (pair_count_kernel): it needs the moved source's box -- the records launch's partial rows -- and the sampler's static
// geometry, not the scan; the per-line stage and the count pass both run 1024-lane workgroups.
struct RrlCountRider {
    const unsigned long long *rng_state;
    const float *r, *centers, *aabb2;
    const float *rows;  // APART rows of cloud 1 (the moved source's partial boxes)
    int n_rows;
    unsigned long long *accept;  // the sampler's ballots (tile_counts)
    int n, rounds;
    int done;
};

// (internal, rrl_demo_epoch) ... and the WRITE pass of the next epoch's sampler, carried by this evaluation's direct-backward
// launch (bwd_write_kernel; 256-lane workgroups): it needs the count pass's ballots (which rode in the per-line launch) and
// overwrites the line buffer, which nothing after the per-line stage reads.
struct RrlWriteRider {
    unsigned long long *rng_state;
    const float *r, *centers;
    const unsigned long long *accept;
    float *lines;
    int32_t *filled;
    int n, rounds;
    int done;
};

struct RrlXform;
struct RrlCall {
    int flags;
    int reduce_mode;    // 0 auto, 1 single, 2 tiled, 3 xchg
    int deterministic;  // 0 / 1
    int sort_parts;     // 0 automatic, k forced
    int scan_variant;   // 0 default, else lines per lane
    const int32_t *order1, *order2;
    unsigned long long *counters;
    long long counter_rows;
    // (internal, not part of rrl_opts) a caller-owned buffer the build step's first launch clears along with the per-call
    // state: the scatter target of rrl_loss_step (grad_tri1), so that no fill launch precedes the step
    void *clear_ptr;
    size_t clear_bytes;  // multiple of 4
    RrlChamRider *rider;  // rrl_opts.chamfer
    RrlCountRider *count_rider;  // (internal) see RrlCountRider
    RrlWriteRider *write_rider;  // (internal) see RrlWriteRider
    int problems;         // rrl_opts.problems (multi-pose evaluation): 0, or Bt < B with B % Bt == 0
    float *payload;       // rrl_opts.payload (rrl_loss_step_ex): [sum of valid losses, #valid, 0 x 12], or NULL
    int payload_in_reduce;  // (internal) rrl_loss_step_ex: the tiled reduce's last arrivers add payload[0 .. 1] (no payload launch)
    const void *tar_ws;   // (internal) the workspace that holds cloud 2's records when the target's scan is carried over
                          // (rrl_*_forward_cached: `target_ws`): the riding walk takes the target from there
    int32_t *chain_left;  // rrl_opts.chain_left (host int, or NULL)
    // (internal, chained steps: include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED) decided once per call by loss_forward_impl:
    int leave_clean;      //   the per-line stage zeroes COUNT1 / COUNT2 behind its read, the tail kernel the CHAIN words
    int fused_build;      //   records + target scan + source scan as ONE launch (rrl_launch_cull_scan issues it; the
    const RrlXform *xf;   //   source's transform for its records body)
    const float *tri1_in; //   ... and the caller's source rows when there is no transform
    __host__ bool prepared() const { return order1 != nullptr; }
    __host__ bool target_kept() const { return order1 != nullptr && (flags & RRL_F_TARGET_KEPT); }
};
RrlCall rrl_resolve_opts(const rrl_opts *o);  // rrl_sparse.hip
// the sampler's two passes on their own (rrl_geom.hip; rrl_sample_lines_rng = both): rrl_demo_epoch pipelines them
int rrl_sample_count_pass(const uint64_t *rng_state, const float *r, const float *centers, const float *aabb1, const float *aabb2,
                          int32_t *tile_counts, int B, int n, int rounds, void *stream);
int rrl_sample_write_pass(uint64_t *rng_state, const float *r, const float *centers, float *lines, int32_t *filled,
                          int32_t *tile_counts, int B, int n, int rounds, void *stream);
int rrl_sample_prefilter(void);
// rrl_registration_step_ex with the call's options already resolved (rrl_sparse.hip; rrl_epoch.hip adds a rider)
int rrl_registration_step_call(const float *src, const float *R, const float *t, const float *tri2, const float *line,
                               void *ws, size_t ws_bytes, float *loss, const float *grad_loss, float *gR, float *gt,
                               float *payload, int B, int N, int M, int L, int transpose_r, int s_m, int s_n, int e_m, int e_n,
                               int mode, int chunk, const void *target_ws, const RrlCall &o, void *stream);
// the process-wide defaults, one accessor per translation unit that owns one
int rrl_default_sort_parts(void);                                           // rrl_cull.hip
void rrl_default_scan_counters(unsigned long long **buf, long long *rows);  // rrl_cull.hip
int rrl_default_scan_variant(void);                                         // rrl_scan.hip

// Rigid transform of the source cloud folded into the prepare step (the fused training op):
// tri1 = src moved by (R, t) per sample, stored into the workspace field TRI1.
struct RrlXform {
    const float *src, *R, *t;
    int transpose_r;
    int zero_g1;  // also clear GACC, the (dR, dt, payload) accumulator of the direct backward
};
