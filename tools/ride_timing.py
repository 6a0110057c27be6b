"""The fused step WITH the trainers' Chamfer monitor, per step: the walk riding in the scan's launch (ops.ChamferRide)
against a launch of its own after the step (ops.chamfer_from_state).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch
import bench
from rrl_hip import ops
dev = torch.device("cuda", 0)
for (B, N, L) in ((8, 4096, 10000), (1, 1024, 20000), (8, 2048, 10000)):
    w = bench.make_workload(B, N, N, L, 0, dev)
    R, T = w["R"].detach(), w["T"].detach()
    o1, o2 = ops.cloud_order(w["tri1"]), ops.cloud_order(w["tri2"])
    res = {}
    for name, kw in (("step alone", {}), ("step, then chamfer_from_state", {}), ("step with the walk riding", dict(chamfer=True))):
        st = ops.RegistrationStep(w["tri1"], w["tri2"], L, transpose_r=True, src_order=o1, tar_order=o2, **kw)
        def one():
            st(R, T, w["lines"])
            if name.startswith("step, then"):
                return ops.chamfer_from_state(st.st)
            return st.chamfer_value
        for _ in range(20):
            v = one()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            v = one()
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / n * 1e6, None if v is None else float(v))
    print(f"B={B} N=M={N} L={L}: " + "; ".join(f"{k} {a:.1f} us" + (f" (chamfer {c:.6f})" if c is not None else "") for k, (a, c) in res.items()))
