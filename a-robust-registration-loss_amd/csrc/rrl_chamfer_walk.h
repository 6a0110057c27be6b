// rrl_chamfer_walk.h -- the tree walk of the Chamfer monitor (K7; header of rrl_chamfer.hip explains it) as a device
// function, shared by chamfer_tree_kernel (rrl_chamfer.hip) and by the launch that carries the walk beside the culled
// scan's workgroups (cull_scan_chamfer_kernel, rrl_cull.hip: launches of one stream never overlap on this stack, so two
// INDEPENDENT 512-lane kernels only run side by side inside one launch).
#pragma once
#include "rrl_tree.h"

typedef const float __attribute__((address_space(4))) * kptr;  // constant AS -> s_load

// the Chamfer workspace (include/rrl.h rrl_chamfer_workspace_bytes)
struct ChamLayout {
    size_t crec1, crec2, p0s1, p0s2, idx1, idx2, grp1, grp2, apart, pmax, histg, partial, gpart, ctrl, ctrl_bytes, total;
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
        gpart = take(8 * 2 * b);  // per (sample, direction) sum of its patches' partials
        histg = take(nmax > 4096 ? 4 * 2 * b * 2 * SORT_CELLS : 16);  // cleared by pts_records_kernel
        ctrl_bytes = 4 * (32 + 64 * b);  // arrival counters of the mean (ChamTick): zero before the walk -- inside the
        ctrl = take(ctrl_bytes);         //   range pts_records_kernel clears; the small-cloud sort kernel clears them itself
        total = o;
    }
};


#define LB_SCALE 0.9999f  // squared distances to node centres are shrunk: evaluation error of the bound (~1e-6)

struct NNWave {
    const float4 *T;      // sorted target records of the sample
    const float4 *tree;   // target tree
    int nt;               // real target records
    float qx, qy, qz;
    bool valid;
    unsigned long long best;  // (distance bits << 32) | original target index
};

__device__ __forceinline__ unsigned long long kmin(unsigned long long a, unsigned long long b) { return b < a ? b : a; }

#ifndef NNW
#define NNW 4  // wavefronts per patch: 4 (leaf = group of 16) or 8 (leaf = half of 8)
#endif
#ifndef SPLIT
#define SPLIT 2  // the candidate supergroups of a patch are dealt to SPLIT sets of NNW wavefronts (j % SPLIT): the kernel's
                 // duration is the time of its SLOWEST patch (misaligned clouds: queries far from the target see
                 // most of the tree), and both the per-lane tests and the entry passes of a patch split this way
#endif
#define NWV (NNW * SPLIT)  // wavefronts per workgroup
#define LEAF (SGT / NNW)
static_assert(NNW == 8 || NNW == 4, "a wavefront owns a half (8 records) or a group (16) of every supergroup");
#define LEAF_NODE(k) (NNW == 8 ? 5 + (k) : 1 + (k))
#ifndef TB
#define TB 1  // per-lane leaf tests per loop iteration (2, 4: no faster -- the kernel is VALU-issue bound, see notes)
#endif

// ---- per-lane leaf evaluation (round 2, v9) -------------------------------------------------------
// Evaluating a leaf for the whole wavefront whenever ANY of its 64 queries needs it wastes ~2/3 of the pair
// evaluations (a leaf is typically needed by ~14 of the 64 lanes).  Instead, like the culled scan's levels:
// a wavefront stages its candidate leaves in LDS (CHK at a time, coalesced loads, one memory latency), every
// lane tests ITS query against each staged leaf and pushes (query, leaf slot) ENTRIES into the wavefront's LDS
// queue (ballot + rank); whenever 64 entries wait, every lane pops one and evaluates "its" 16 records
// for "its" query -- all lanes busy on pairs that are actually needed -- and folds the key into the
// query's minimum with an LDS atomicMin on the u64 key (several lanes may hold the same query).
#ifndef CHK
#define CHK 12                 // candidate leaves staged per round and wavefront (12: 32 KiB of LDS per 8-wavefront workgroup -> 4 per CU)
#endif
#define LROW (LEAF + 1)        // float4 per staged leaf row: +1 of padding spreads the rows over the banks
#define QCAP (64 + 63)         // entries: < 64 left-overs + one push round

struct NNShared {
    float4 *q;                   // [64] the patch's queries (xyz, original index)
    unsigned long long *best;    // [64] u64 keys, LDS atomicMin
    float4 *rec;                 // this wavefront's staged leaves [CHK][LROW]
    int *pos;                    // this wavefront's staged leaf positions [CHK] (sorted position of record 0)
    unsigned *queue;             // this wavefront's entries: query << 8 | slot
};

// pops up to 64 entries [base, base + take) and evaluates them, one per lane
__device__ __forceinline__ void eval_entries(const NNShared &sh, int base, int take, int nt, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < take) {
        const unsigned e = sh.queue[base + lane];
        const int q = (int)(e >> 8), slot = (int)(e & 255u);
        const float4 qv = sh.q[q];
        const int cnt = min(LEAF, nt - sh.pos[slot]);
        const float4 *row = sh.rec + slot * LROW;
        unsigned long long key[LEAF];
#pragma unroll
        for (int t = 0; t < LEAF; ++t) {
            const float4 r = row[t];
            // code/loss.py:51: sum((x - y)**2, -1); (a0 + a1) + a2, no FMA
            const float dx = qv.x - r.x, dy = qv.y - r.y, dz = qv.z - r.z;
            float s = dx * dx;
            s = s + dy * dy;
            s = s + dz * dz;
            key[t] = t < cnt ? (((unsigned long long)__float_as_uint(s) << 32) | (unsigned)__float_as_int(r.w)) : ~0ull;
        }
#pragma unroll
        for (int o = LEAF / 2; o > 0; o >>= 1)
#pragma unroll
            for (int t = 0; t < o; ++t) key[t] = kmin(key[t], key[t + o]);
        atomicMin(&sh.best[q], key[0]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Arrival counters of the in-launch mean (chamfer_tree_kernel's ending): one per (sample, direction) group at
// group + b * stride_b + dir * stride_d, one for the groups.  All zero before the launch; they rewind themselves.
struct ChamTick {
    uint32_t *top, *group;
    int stride_b, stride_d;
};

// COUNT: executed-work counters (rrl_chamfer_counters): [0] patch-level leaf tests (lane-parallel),
// [1] per-lane leaf sphere tests (wave x leaf), [2] (query, leaf) entries evaluated, [3] (query, target)
// pairs evaluated, [4] wavefronts.
// IDX = true (rrl_chamfer_from_loss): the records are the LOSS workspace's sorted (P0, thr2) records; the
// original indices come from its IDX arrays and "a NaN coordinate in the cloud" from its PMAX entries
// (pm1 = &PMAX[0] of the source's workspace, pm2 = &PMAX[B] of the target's).
// The walk's LDS (32 KiB per 8-wavefront workgroup): one object, so that a launch which carries the walk beside another
// kernel's workgroups (cull_scan_chamfer_kernel, rrl_cull.hip) can overlay it with that kernel's.
struct ChamLds {
    unsigned long long s_best[64];
    __attribute__((aligned(16))) float4 s_q[64];
    __attribute__((aligned(16))) float4 s_rec[NWV][CHK * LROW];
    int s_pos[NWV][CHK];
    unsigned s_queue[NWV][QCAP + 1];
    double red[64];
};

// The walk of one workgroup (bx = 2 sample + direction, by = patch; gx, gy = the walk's grid): the kernel below, and --
// round 4b -- the rider of the culled scan's launch in the demo epoch (IDX = true with pm1 == NULL: max |P|^2 of the target
// then comes from the loss workspace's partial rows `apart`, because that launch's scan is still rebuilding PMAX).
template <bool COUNT, bool IDX>
__device__ __forceinline__ void chamfer_tree_body(
    ChamLds &lds_, const float4 *__restrict__ p0s1, const float4 *__restrict__ p0s2, const float4 *__restrict__ grp1,
    const float4 *__restrict__ grp2, const float *__restrict__ apart, int nblk,
    unsigned long long *__restrict__ best_x, unsigned long long *__restrict__ best_y,
    double *__restrict__ partial, int B, int N, int M, unsigned long long *__restrict__ counters, long long counter_rows,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2, const uint32_t *__restrict__ pm1,
    const uint32_t *__restrict__ pm2, const ChamTick tk_, double *__restrict__ gpart, float *__restrict__ value, double denom,
    const int bx, const int by, const int gx, const int gy, const float *__restrict__ apart_tar = nullptr) {
    unsigned long long (&s_best)[64] = lds_.s_best;
    float4 (&s_q)[64] = lds_.s_q;
    float4 (&s_rec)[NWV][CHK * LROW] = lds_.s_rec;
    int (&s_pos)[NWV][CHK] = lds_.s_pos;
    unsigned (&s_queue)[NWV][QCAP + 1] = lds_.s_queue;
    double (&red)[64] = lds_.red;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wavefront of the workgroup
    const int wave = wv % NNW, cls = wv / NNW;                  // its leaf within a supergroup, its supergroup class
    // XCD-aware: workgroups go to the 8 XCDs round-robin by linear id; with (sample, direction) on the fast
    // index every XCD's L2 holds the records and trees of 2 B / 8 of the (cloud pair, direction) combinations only
    const int b = bx >> 1, dir = bx & 1;
    const int nq = dir ? M : N, nt = dir ? N : M;
    const int nsgq = (nq + SGT - 1) / SGT, nsgt = (nt + SGT - 1) / SGT;
    const int sgq = by;
    double mine = 0.0;
    unsigned c_sg = 0, c_gt = 0, c_ge = 0, c_pairs = 0;
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_stage = 0, t_test = 0, t_pass = 0;  // COUNT: phase clocks
#define RRL_NOW() (COUNT ? (long long)__builtin_readcyclecounter() : 0ll)
    tk[0] = RRL_NOW();
    const long long wall0 = COUNT ? (long long)wall_clock64() : 0ll;
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
        const int32_t *idxT = IDX ? (dir ? idx1 : idx2) + (size_t)b * nsgt * SGT : nullptr;
        const int qorig = IDX ? ((dir ? idx2 : idx1) + (size_t)b * nsgq * SGT)[qi] : __float_as_int(qr.w);
        // a staged target record carries its original index in .w
        auto target = [&](int pos) {
            float4 r = w.T[pos];
            if constexpr (IDX) r.w = __int_as_float(idxT[pos]);
            return r;
        };
        w.qx = qr.x; w.qy = qr.y; w.qz = qr.z;
        w.best = ~0ull;
        if (wv == 0) { s_q[lane] = qr; s_best[lane] = ~0ull; }
        const NNShared sh = {s_q, s_best, s_rec[wv], s_pos[wv], s_queue[wv]};
        kptr qn = (kptr)(uintptr_t)(treeQ + (size_t)sgq * NODE);
        const float cqx = qn[0], cqy = qn[1], cqz = qn[2];
        float qg[SGG][4];  // the patch's four query-group spheres (wave-uniform)
#pragma unroll
        for (int g = 0; g < SGG; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) qg[g][c] = qn[4 * (1 + g) + c];
        // does the TARGET cloud hold a NaN coordinate (slot 7 of its AABB partial rows)?
        bool tnan = false;
        if (wv == 0) {
            if constexpr (IDX) {  // max |P|^2 of the target cloud: +inf when a coordinate is NaN (or overflows)
                if (pm1 != nullptr) {  // uniform
                    tnan = !(__uint_as_float((dir ? pm1 : pm2)[b]) <= 3.0e38f);
                } else {  // (riding in the scan's launch) ... from the records launch's per-workgroup partial rows, slot 6
                    const int ct = dir ? 0 : 1, nb = (nt + 255) / 256;  // (cloud 2's rows may live in another workspace)
                    const float *ap = (ct && apart_tar ? apart_tar : apart) + ((size_t)ct * B + b) * nblk * 8 + 6;
                    float pmv = 0.0f;
                    for (int j = lane; j < nb; j += 64) pmv = fmaxf(pmv, ap[j * 8]);
                    tnan = __any(!(pmv <= 3.0e38f));
                }
            } else {
                const int ct = dir ? 0 : 1, nb = (nt + 255) / 256;
                const float *ap = apart + ((size_t)ct * B + b) * nblk * 8;
                for (int j = lane; j < nb; j += 64) tnan |= ap[j * 8 + 7] != 0.0f;
                tnan = __any(tnan);
            }
        }

        tk[1] = RRL_NOW();
        // ---- seed: the target supergroup nearest to the patch centre; wavefront k evaluates its leaf k for
        //      all 64 queries (every query needs a bound; the records arrive through the scalar cache)
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
        tk[2] = RRL_NOW();
        __syncthreads();  // s_q / s_best initialised by wavefront 0
        {   // every query needs a bound: 64 entries (lane, slot 0) on this wavefront's leaf of the seed supergroup
            const int cnt = min(LEAF, nt - seed * SGT - wave * LEAF);
            if (cnt > 0 && cls == 0) {
                if (lane == 0) sh.pos[0] = seed * SGT + wave * LEAF;
                if (lane < LEAF) sh.rec[lane] = target(seed * SGT + wave * LEAF + lane);
                sh.queue[lane] = (unsigned)lane << 8;
                eval_entries(sh, 0, nq - sgq * SGT < 64 ? nq - sgq * SGT : 64, nt, lane);
                if constexpr (COUNT) { c_ge += 64u; c_pairs += 64u * (unsigned)cnt; }
            }
        }
        tk[3] = RRL_NOW();
        __syncthreads();
        tk[4] = RRL_NOW();

        // ---- the other supergroups
        int nq_e = 0;  // entries waiting in this wavefront's queue (uniform)
        for (int j0 = 0; j0 < nsgt; j0 += 64) {
            const int j = j0 + lane;
            // prune 1, lane-parallel over the target supergroups: this wavefront's leaf of supergroup j against
            // each of the patch's four query GROUPS (16 consecutive lanes = one group of the query tree) with
            // that group's largest bound -- |cg - cj| <= sqrt(bd_g) + Rg + Rj, squared
            float sb = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned)(s_best[lane] >> 32))) * 1.00001f;
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
            if (j < nsgt && j != seed && (j % SPLIT) == cls && nt - j * SGT - wave * LEAF > 0) {  // (an empty leaf is no candidate)
                gn = w.tree[(size_t)j * NODE + LEAF_NODE(wave)];
                cand = blind;
#pragma unroll
                for (int g = 0; g < SGG; ++g) {
                    const float dx = qg[g][0] - gn.x, dy = qg[g][1] - gn.y, dz = qg[g][2] - gn.z;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sbg[g] + qg[g][3] + gn.w;
                    cand = cand || !(d2 * LB_SCALE > t * t);  // NaN radius (empty query group): kept
                }
            }
            unsigned long long m = __ballot(cand);
            if constexpr (COUNT) c_sg += (unsigned)min(64, nsgt - j0);
            while (m) {
                const long long ts0 = RRL_NOW();
                // ---- stage the next <= CHK candidate leaves of this wavefront (one round of coalesced loads)
                const int nc = min(CHK, __popcll(m));
                {
                    const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (cand && ((m >> lane) & 1ull) && rk < nc) sh.pos[rk] = (j0 + lane) * SGT + wave * LEAF;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (int i = lane; i < nc * LEAF; i += 64) {
                    const int slot = i / LEAF, t = i % LEAF;
                    sh.rec[slot * LROW + t] = target(sh.pos[slot] + t);  // rows exist up to the supergroup boundary
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if constexpr (COUNT) { (void)sh.rec[lane].x; }
                const long long ts1 = RRL_NOW();
                t_stage += ts1 - ts0;
                // ---- per staged leaf: prune 2 per lane, entries for the lanes that need it
                for (int c = 0; c < nc; ++c) {
                    const int sl = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    // the leaf node was fetched by lane sl above: broadcast it (no dependent load per candidate)
                    const float gx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.x), sl));
                    const float gy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.y), sl));
                    const float gz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.z), sl));
                    const float gr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gn.w), sl));
                    if constexpr (COUNT) ++c_gt;
                    // |q - c| <= sqrt(bd) + R, squared, with the query's CURRENT bound (all wavefronts fold into it);
                    // NaN anywhere: visit
                    sb = __builtin_amdgcn_sqrtf(__uint_as_float((unsigned)(s_best[lane] >> 32))) * 1.00001f;
                    const float dx = w.qx - gx, dy = w.qy - gy, dz = w.qz - gz;
                    const float d2 = dx * dx + dy * dy + dz * dz, t = sb + gr;
                    const bool need = w.valid && !(d2 * LB_SCALE > t * t);
                    const unsigned long long nm = __ballot(need);
                    if (need) {
                        const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
                        sh.queue[nq_e + rk] = ((unsigned)lane << 8) | (unsigned)c;
                    }
                    nq_e += __popcll(nm);
                    if (nq_e >= 64) {  // uniform: a full pass
                        nq_e -= 64;
                        const long long tp0 = RRL_NOW();
                        eval_entries(sh, nq_e, 64, nt, lane);
                        t_pass += RRL_NOW() - tp0;
                        if constexpr (COUNT) { c_ge += 64u; c_pairs += 64u * (unsigned)LEAF; }
                    }
                }
                // the staged rows are overwritten by the next round: drain the queue
                const long long tp0 = RRL_NOW();
                if (nq_e > 0) {
                    eval_entries(sh, 0, nq_e, nt, lane);
                    if constexpr (COUNT) { c_ge += (unsigned)nq_e; c_pairs += (unsigned)nq_e * (unsigned)LEAF; }
                    nq_e = 0;
                }
                const long long tp1 = RRL_NOW();
                t_pass += tp1 - tp0;
                t_test += tp1 - ts1;
            }
        }
        tk[5] = RRL_NOW();
        __syncthreads();
        tk[6] = RRL_NOW();
        if (wv == 0 && w.valid) {
            w.best = s_best[lane];
            const bool qnan = (w.qx != w.qx) || (w.qy != w.qy) || (w.qz != w.qz);
            if (qnan || tnan)  // torch.min propagates NaN
                w.best = ((unsigned long long)0x7fc00000u << 32) | (unsigned)(w.best & 0xffffffffu);
            (dir ? best_y : best_x)[(size_t)b * nq + qorig] = w.best;
            mine = (double)__uint_as_float((unsigned)(w.best >> 32));
        }
    }
    if constexpr (COUNT) {
        // one 16-slot row per wavefront, plain stores (same-address atomics from 4096 wavefronts would
        // saturate the memory system and distort the very clocks recorded here); the host adds the rows
        const long long crow = (long long)(by * gx + bx) * NWV + wv;  // rows past the buffer are dropped
        if (lane == 0 && sgq < nsgq && crow < counter_rows) {
            unsigned long long *row = counters + 16 * (size_t)crow;
            row[0] = c_sg; row[1] = c_gt; row[2] = c_ge; row[3] = c_pairs; row[4] = 1;
            row[5] = (unsigned long long)t_stage;
            row[6] = (unsigned long long)(t_test - t_pass);
            row[7] = (unsigned long long)t_pass;
            row[8] = (unsigned long long)((long long)wall_clock64() - wall0);  // 100 MHz ticks of the same span as sum(tk)
            row[9] = (unsigned long long)wall0;  // absolute start (tools/cham_count_vs_plain.py: start-time spread)
            row[15] = (unsigned long long)(RRL_NOW() - tk[0]);
            for (int i = 2; i <= 6; ++i) row[8 + i] = (unsigned long long)(tk[i] - tk[i - 1]);
        }
    }
#undef RRL_NOW
    if (wv != 0) return;  // the minima of the patch are in wavefront 0
    // ---- mean: fixed-order sums, finished INSIDE this launch (round 3; a second, tiny launch added 4.2 us + a launch
    //      boundary to a 22 us walk).  The patch's 64 minima in one wavefront (LDS operations execute in order) ->
    //      partial[workgroup]; the LAST patch of a (sample, direction) to arrive adds that group's partials in index
    //      order -> gpart[group]; the last GROUP adds the 2 B group sums in index order and writes the value.  Two
    //      levels because a thousand arrivals on one word serialise (~12 ns each); the counters live 128 bytes apart.
    //      Stores / loads of what crosses workgroups are agent-scope (sc1), every counter rewinds itself.
    auto wave_total = [&](double v) {  // all lanes must call; lane 0 holds the sum
        red[lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int o = 32; o > 0; o >>= 1) {
            if (lane < o) red[lane] += red[lane + o];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        return red[0];
    };
    auto ld64 = [](const double *p) {
        return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    auto st64 = [](double *p, double v) {
        __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const double psum = wave_total(mine);
    const int wg = by * gx + bx, grp = bx, ngrp = gx, npatch = gy;
    if (tk_.top == nullptr) {  // (no counters: leave the partials to the caller)
        if (lane == 0) partial[wg] = psum;
        return;
    }
    uint32_t *gtick = tk_.group + (size_t)b * tk_.stride_b + (size_t)dir * tk_.stride_d;
    int last = 0;
    if (lane == 0) {
        st64(&partial[wg], psum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(gtick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(npatch - 1) ? 1 : 0;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    double acc = 0.0;
    for (int i = lane; i < npatch; i += 64) acc += ld64(&partial[(size_t)i * ngrp + grp]);  // patches of this group, index order per lane
    const double gsum = wave_total(acc);
    last = 0;
    if (lane == 0) {
        __hip_atomic_store(gtick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st64(&gpart[grp], gsum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(tk_.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ngrp - 1) ? 1 : 0;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    acc = 0.0;
    for (int i = lane; i < ngrp; i += 64) acc += ld64(&gpart[i]);
    const double tot = wave_total(acc);
    if (lane == 0) {
        value[0] = (float)(tot / denom);
        __hip_atomic_store(tk_.top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool COUNT, bool IDX>
__global__ __launch_bounds__(64 * NWV) void chamfer_tree_kernel(
    const float4 *__restrict__ p0s1, const float4 *__restrict__ p0s2, const float4 *__restrict__ grp1,
    const float4 *__restrict__ grp2, const float *__restrict__ apart, int nblk,
    unsigned long long *__restrict__ best_x, unsigned long long *__restrict__ best_y,
    double *__restrict__ partial, int B, int N, int M, unsigned long long *__restrict__ counters, long long counter_rows,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2, const uint32_t *__restrict__ pm1,
    const uint32_t *__restrict__ pm2, const ChamTick tk_, double *__restrict__ gpart, float *__restrict__ value, double denom) {
    __shared__ ChamLds lds_;
    chamfer_tree_body<COUNT, IDX>(lds_, p0s1, p0s2, grp1, grp2, apart, nblk, best_x, best_y, partial, B, N, M, counters,
                                  counter_rows, idx1, idx2, pm1, pm2, tk_, gpart, value, denom, (int)blockIdx.x,
                                  (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}
