"""Randomised soak of the CHAINED step (include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED): random (B, N, M, L, scale) shapes on both sides
of rrl_cull_scan_can_fuse's rule, ops.LossStep chained against ops.LossStep(chain=False) on the same inputs -- lines and poses
alternate between two sets from call to call, so every chained call consumes the clean state the previous one left -- loss,
info, KJ, hit lists, median, bucket sums bit for bit, gradient to the rounding of its atomics.  Product-only (no oracle).
usage (GPU box): python tools/chain_soak.py [seed] [cases]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(seed=0, n_cases=200, log=print):
    import loss as L
    from rrl_hip import ops
    from test_gpu_prepared import _pairs
    from test_gpu_chain import _new_lines, _poses, _snapshot, _assert_same
    rng = np.random.default_rng(seed)
    bad = fused = 0
    t0 = time.time()
    for case in range(n_cases):
        B = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 12, 16]))
        n = int(rng.choice([64, 300, 1000, 1024, 2048, 3000, 4096, 4096, 5000, 9000]))
        m = int(rng.choice([64, 260, 1000, 1024, 2048, 2600, 4096, 4096, 7000]))
        nl = int(rng.choice([1025, 1100, 2048, 3000, 4096, 6000, 8192, 10000, 10000, 14000, 20000]))
        if B * (n + m) * nl > 16 * 8192 * 10000:
            B = max(1, B // 4)
        scale = float(rng.choice([0.3, 1.0, 1.0, 1.0, 3.0, 30.0]))
        prs, src, tar = _pairs(int(rng.integers(0, 10**6)), B, n, m)
        src, tar = src * scale, tar * scale
        for p in prs:
            p["src"] = np.asarray(p["src"], np.float32) * np.float32(scale); p["tar"] = np.asarray(p["tar"], np.float32) * np.float32(scale)
            p["radius"] = float(p["radius"]) * scale; p["center"] = np.asarray(p["center"], np.float32) * np.float32(scale)
        ln = [_new_lines(L, prs, nl, it) for it in range(2)]
        poses = [_poses(B, it) for it in range(2)]
        if rng.random() < 0.1:  # a non-unit direction: one wavefront takes the strict fallback inside the chained launch
            ln[1] = ln[1].clone(); ln[1][0, 5, :3] *= 1.5
        ref = ops.LossStep(src, tar, nl, chain=False)
        want = [_snapshot(ref, ref(*poses[it], ln[it])) for it in range(2)]
        st = ops.LossStep(src, tar, nl)
        ok = True
        try:
            for call in range(6):
                it = call & 1
                got = _snapshot(st, st(*poses[it], ln[it]))
                torch.cuda.synchronize()
                def nn(d):  # NaN == NaN; info[:, 3] is the batch-wide NaN flag after a plain step, each sample's own after a
                    d = {k: torch.nan_to_num(v, nan=-7.0) if v.dtype.is_floating_point else v for k, v in d.items()}  # chained one
                    d["info"] = torch.cat([d["info"][:, :3], d["info"][:, 3:].max().expand(d["info"].shape[0], 1)], 1)
                    return d
                _assert_same(nn(want[it]), nn(got), (case, call))
        except AssertionError as exc:
            ok = False
            log("MISMATCH " + str(dict(case=case, B=B, N=n, M=m, L=nl, scale=scale, fused=bool(st.fused))) + " " + str(exc)[:200])
        fused += int(bool(st.fused))
        bad += int(not ok)
    log(f"{n_cases} cases ({fused} ran the one-launch build), {bad} mismatches, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 200) else 0)
