// rrl_epoch.hip -- one epoch of the single-pair demo as ONE C call (include/rrl.h rrl_demo_epoch).
//
// code/test_demo_optimized_Lie_Algebra.py:46-82 does, per epoch: sample lines through both clouds' boxes (against the
// PREVIOUS epoch's moved source), move the source by the current pose, evaluate the loss, backward, Adam step, Chamfer
// monitor, log.  Every piece has its entry in this library already; this file only issues them back to back on the
// caller's stream -- sampler (2 launches) -> fused registration step on prepared clouds (4; 5 at the demo's 20 line tiles),
// the Chamfer walk of the step's clouds riding in its scan launch (round 4b; a launch of its own before) -> pose step (1:
// exp-map backward, gated Adam, next exp map, log row, next sampler box) -- so that a loop pays one
// host call (~5 us) per epoch instead of a hipGraph replay (~8 us fixed + ~1.5 us per node on this stack,
// tools/attic/graph_node_cost.py) or eight Python-level calls.  No new arithmetic: the results are those of the four entries.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include "rrl_ws.h"

extern "C" int rrl_demo_epoch(const rrl_demo_epoch_args *a, void *stream) {
    if (!a || a->struct_bytes < (int32_t)offsetof(rrl_demo_epoch_args, pipeline)) return RRL_E_ARG;  // (pipeline: appended in round 4b)
    const int N = a->N, M = a->M, L = a->L;
    if (N <= 0 || M <= 0 || L <= 0 || a->rounds <= 0) return RRL_E_ARG;
    const char *env = getenv("RRL_DEMO_RIDE");  // (read per call: tests switch it between epochs) 0: no launch carries another's work
    const bool rides = !(env && env[0] == '0');
    int32_t *pipe = a->struct_bytes >= (int32_t)sizeof(rrl_demo_epoch_args) && rides ? a->pipeline : nullptr;
    if (pipe && ((long)((L + 1023) / 1024) * a->rounds >= 512 || L <= 1024)) pipe = nullptr;  // (the rider's limits: line_pair_dist_impl)
    int rc;
    if (pipe && *pipe == 3) {  // ... and its backward launch the write pass too: this epoch's lines are in place
        rc = 0;
    } else if (pipe && (*pipe & 1)) {  // the previous epoch's per-line launch carried this epoch's count pass: the write pass remains
        rc = rrl_sample_write_pass(a->rng_state, a->radius, a->centers, a->lines, a->filled, a->tile_counts, 1, L, a->rounds, stream);
    } else {
        rc = rrl_sample_lines_rng(a->rng_state, a->radius, a->centers, a->box1, a->box2, a->lines, a->filled, a->tile_counts,
                                  1, L, a->rounds, stream);
    }
    if (a->pipeline && a->struct_bytes >= (int32_t)sizeof(rrl_demo_epoch_args)) *a->pipeline = 0;
    if (rc) return rc;
    // The Chamfer monitor needs the step's records launch only (the sorted moved source + the kept target), and launches of
    // one stream never overlap on this stack: its walk RIDES in the culled scan's launch (rrl_ws.h RrlChamRider; same
    // arithmetic, same value) -- one launch and ~12 us per epoch less.  RRL_DEMO_RIDE=0: the separate launch, as before.
    RrlCall o = rrl_resolve_opts(a->opts);
    RrlChamRider rider = {a->cham_ws, a->cham_ws_bytes, a->best_x, a->best_y, a->cham_value, 0};
    o.rider = rides ? &rider : nullptr;
    // ... and so does the NEXT epoch's count pass, in the per-line launch (RrlCountRider): it samples against the moved source
    // of THIS epoch (code/test_demo_optimized_Lie_Algebra.py:46-51), whose box is in the records launch's partial rows
    const WsLayout wl(1, N, M, L);
    if (a->ws_bytes < wl.total) return RRL_E_WS;
    RrlCountRider counter = {(const unsigned long long *)a->rng_state, a->radius, a->centers, a->box2, wl.f32(a->ws, RRL_WS_APART),
                             (N + 255) / 256, (unsigned long long *)a->tile_counts, L, a->rounds, 0};
    // ... and its WRITE pass in the direct backward's launch (RrlWriteRider): the ballots are there by then (the per-line
    // launch precedes it), and nothing after the per-line stage reads the line buffer it overwrites
    RrlWriteRider writer = {(unsigned long long *)a->rng_state, a->radius, a->centers, (const unsigned long long *)a->tile_counts,
                            a->lines, a->filled, L, a->rounds, 0};
    if (pipe && (((uintptr_t)a->tile_counts) & 7) == 0) {
        o.count_rider = &counter;
        o.write_rider = &writer;
    }
    rc = rrl_registration_step_call(a->src_tri, a->R, a->T, a->tar_tri, a->lines, a->ws, a->ws_bytes, a->loss, a->grad_loss,
                                    a->gR, a->gt, nullptr, 1, N, M, L, a->transpose_r, 1, 1, 5, 5, RRL_SCAN_CULL, 0, nullptr, o,
                                    stream);
    if (rc) return rc;
    if (pipe) *pipe = counter.done ? (writer.done ? 3 : 1) : 0;  // (a write pass without its count pass cannot have ridden)
    if (!rider.done) {
        rc = rrl_chamfer_from_loss(a->ws, a->ws, a->ws_bytes, 1, N, M, L, a->cham_ws, a->cham_ws_bytes, a->best_x, a->best_y,
                                   a->cham_value, stream);
        if (rc) return rc;
    }
    const WsLayout w(1, N, M, L);
    if (a->ws_bytes < w.total) return RRL_E_WS;
    const int32_t *info = w.i32(a->ws, RRL_WS_INFO);        // gate: the loss's bucket count (`if loss_di is not None`)
    const float *rows = w.f32(a->ws, RRL_WS_APART);         // cloud 1, sample 0: the moved first points' partial boxes
    return rrl_se3_adam_step(a->xi, a->gR, a->gt, a->m, a->v, a->adam_state, a->lr, info, a->b1, a->b2, a->eps, a->R, a->T,
                             nullptr, a->loss, a->cham_value, a->table, a->cursor, a->table_rows, a->row, rows,
                             (N + 255) / 256, a->box1, stream);
}
