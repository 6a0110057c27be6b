#!/usr/bin/env python3
"""cProfile of the host side of the EAGER drop-in call (public loss, B=1, forward + backward to points1.grad)
and of the eager fused op -- what an unmodified trainer pays per sample (run on the GPU box)."""
import cProfile, pstats, sys, os, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import torch
import bench
import loss as Lmod
from rrl_hip import ops
dev = torch.device("cuda", 0)
w = bench.make_workload(1, 4096, 4096, 10000, 0, dev)
tri1 = w["tri1"].clone().requires_grad_(True)
def dropin():
    tri1.grad = None
    l = Lmod.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri1, w["tri2"], w["lines"], dev)
    l.backward()
for _ in range(30): dropin()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): dropin()
torch.cuda.synchronize()
print(f"drop-in loss fwd+bwd, B=1: {(time.perf_counter() - t0) / 300 * 1e6:.0f} us per call (wall, eager)")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): dropin()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
