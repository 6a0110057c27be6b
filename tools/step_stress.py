"""Race hunt for the one-call step (tail kernel: tickets, fixed-point sums, payload by atomicMax): thousands of repeated
steps at a few shapes; every call must reproduce the loss, median, info, bucket sums and payload[0 .. 1] of the first call
bit for bit and (dR, dt) within the rounding of their float atomics; every 100th call is checked against the two-call path.
usage (GPU box): python tools/step_stress.py [iterations]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import bench
from rrl_hip import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
bad = 0
for (B, N, L) in ((8, 4096, 10000), (8, 4096, 4096), (3, 1500, 16000), (16, 2048, 8000), (1, 1024, 3000)):
    w = bench.make_workload(B, N, N, L, 0, dev)
    R = torch.eye(3, device=dev).repeat(B, 1, 1)
    t = torch.zeros(B, 3, device=dev)
    ops.RegistrationStep.ONE_CALL = False
    two = ops.RegistrationStep(w["tri1"], w["tri2"], L, want_payload=True)
    ref2 = [x.clone() for x in two(R, t, w["lines"])[:4]] + [two.st.med.clone(), two.st.bsum.clone()]
    ops.RegistrationStep.ONE_CALL = True
    one = ops.RegistrationStep(w["tri1"], w["tri2"], L, want_payload=True)
    first = None
    t0 = time.time()
    for it in range(iters):
        out = one(R, t, w["lines"])
        cur = [out[0].clone(), one.st.med.clone(), out[4].clone(), one.st.bsum.clone(), out[3][:2].clone(), out[1].clone(), out[2].clone()]
        if first is None:
            torch.cuda.synchronize()
            first = cur
            ok = torch.equal(cur[0], ref2[0]) and torch.equal(cur[1], ref2[4]) and torch.equal(cur[3], ref2[5])
            ok = ok and bool(((cur[5] - ref2[1]).abs() <= 2e-5 * ref2[1].abs() + 2e-6 * float(ref2[1].abs().max())).all())
            if not ok:
                bad += 1
                print("MISMATCH vs two calls", (B, N, L))
            continue
        same = all(torch.equal(a, b) for a, b in zip(cur[:5], first[:5]))
        close = all(bool(((a - b).abs() <= 2e-5 * b.abs() + 2e-6 * float(b.abs().max())).all()) for a, b in zip(cur[5:], first[5:]))
        if not (same and close):
            bad += 1
            print("MISMATCH", (B, N, L), "iteration", it, "bits", same, "grad", close)
            if bad > 10:
                sys.exit(1)
    torch.cuda.synchronize()
    print(f"B={B} N=M={N} L={L}: {iters} steps, {(time.time() - t0) / iters * 1e6:.1f} us per step incl. the checks, mismatches so far {bad}")
sys.exit(1 if bad else 0)
