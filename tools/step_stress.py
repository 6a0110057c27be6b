"""Race hunt for the one-call steps (tail kernel: tickets, fixed-point sums, payload by atomicMax; prepared build with a
kept target; the scatter of ops.LossStep): thousands of repeated steps at a few shapes; every call must reproduce the
loss, median, info, bucket sums and payload[0 .. 1] of the first call bit for bit and the gradients within the rounding
of their float atomics; the first call is checked against the two-call path and the cold (sorting) build.
usage (GPU box): python tools/step_stress.py [iterations]      (tests/test_gpu_stress.py runs a bounded version)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]

SHAPES = ((8, 4096, 10000), (8, 4096, 4096), (3, 1500, 16000), (16, 2048, 8000), (1, 1024, 3000))


def _close(a, b):
    return bool(((a - b).abs() <= 2e-5 * b.abs() + 2e-6 * float(b.abs().max())).all())


def run(iters=3000, shapes=SHAPES, log=print):
    """Returns the number of mismatches (0 = every step reproduced the first one)."""
    import bench
    from rrl_hip import ops
    dev = torch.device("cuda", 0)
    bad = 0
    for (B, N, L) in shapes:
        w = bench.make_workload(B, N, N, L, 0, dev)
        R = torch.eye(3, device=dev).repeat(B, 1, 1)
        t = torch.zeros(B, 3, device=dev)
        try:
            ops.RegistrationStep.ONE_CALL = False
            two = ops.RegistrationStep(w["tri1"], w["tri2"], L, want_payload=True, prepared=False)  # two calls, cold build
            ref2 = [x.clone() for x in two(R, t, w["lines"])[:4]] + [two.st.med.clone(), two.st.bsum.clone()]
        finally:
            ops.RegistrationStep.ONE_CALL = True
        # one call, prepared build, kept target; CHAINED like the LossStep below (round 6: from the second iteration on the source's
        # records, the target's scan and the source's scan are one launch with an in-launch hand-off)
        one = ops.RegistrationStep(w["tri1"], w["tri2"], L, want_payload=True, chain=True)
        ls = ops.LossStep(w["tri1"], w["tri2"], L)
        first = lfirst = None
        t0 = time.time()
        for it in range(iters):
            out = one(R, t, w["lines"])
            cur = [out[0].clone(), one.st.med.clone(), out[4].clone(), one.st.bsum.clone(), out[3][:2].clone(), out[1].clone(), out[2].clone()]
            lo = ls(R, t, w["lines"])
            lcur = [lo[0].clone(), ls.st.med.clone(), lo[2].clone(), ls.st.bsum.clone(), lo[1].clone()]
            if first is None:
                torch.cuda.synchronize()
                first, lfirst = cur, lcur
                ok = torch.equal(cur[0], ref2[0]) and torch.equal(cur[1], ref2[4]) and torch.equal(cur[3], ref2[5]) and _close(cur[5], ref2[1])
                ok = ok and torch.equal(lcur[0], ref2[0]) and torch.equal(lcur[3], ref2[5])
                if not ok:
                    bad += 1
                    log(f"MISMATCH vs the two-call cold path {(B, N, L)}")
                continue
            same = all(torch.equal(a, b) for a, b in zip(cur[:5], first[:5])) and all(torch.equal(a, b) for a, b in zip(lcur[:4], lfirst[:4]))
            close = all(_close(a, b) for a, b in zip(cur[5:], first[5:])) and _close(lcur[4], lfirst[4])
            if not (same and close):
                bad += 1
                log(f"MISMATCH {(B, N, L)} iteration {it} bits {same} grad {close}")
                if bad > 10:
                    return bad
        torch.cuda.synchronize()
        log(f"B={B} N=M={N} L={L}: {iters} steps of RegistrationStep + LossStep, {(time.time() - t0) / iters * 1e6:.1f} us per "
            f"iteration incl. the checks, status {one.st.status.tolist()}, mismatches so far {bad}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 3000) else 0)
