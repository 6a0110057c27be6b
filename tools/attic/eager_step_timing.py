"""Host cost of the EAGER paths a drop-in user calls (no graph): the public loss (forward + backward
to points1.grad), the fused op (forward + backward to dR, dt), per call, at the bench shape.
usage (GPU box): python tools/eager_step_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch
import bench
import loss as Lmod
from rrl_hip import ops
dev = torch.device("cuda", 0)
for B in (1, 8):
    w = bench.make_workload(B, 4096, 4096, 10000, 0, dev)
    ones = torch.ones(B, device=dev)
    tri1 = w["tri1"].clone().requires_grad_(True)

    def fused():
        w["R"].grad = w["T"].grad = None
        loss, info, _ = ops.registration_loss(w["tri1"], w["R"], w["T"], w["tri2"], w["lines"], (1, 1, 5, 5), transpose_r=True)
        torch.autograd.backward([loss], [ones])

    def dropin():
        tri1.grad = None
        l = Lmod.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri1[:1], w["tri2"][:1], w["lines"][:1], dev)
        l.backward()

    for name, fn in (("fused op fwd+bwd", fused), ("drop-in loss fwd+bwd (B=1 slice)", dropin)):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for _ in range(n): fn()
        t_issue = (time.perf_counter() - t0) / n
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
        print(f"B={B} {name}: host issue {t_issue * 1e6:.0f} us/call, wall {t_all * 1e6:.0f} us/call")
