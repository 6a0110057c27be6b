#!/usr/bin/env python3
"""Where the reference trainers' literal loop (bench.py variants.dropin_loop: rigid apply, eight B=1 reference-signature
calls on [j:j+1] slices, Python sum, one backward) spends its host time -- wall-clock stamps per phase + cProfile.
Run on the GPU box."""
import cProfile, pstats, sys, os, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import torch
import bench
import loss as Lmod
from rrl_hip import ops
dev = torch.device("cuda", 0)
B, N, L = 8, 4096, 10000
w = bench.make_workload(B, N, N, L, 0, dev)
w["R"].requires_grad_(True); w["T"].requires_grad_(True)
stamps = [0.0] * 6


def loop_step(timed=False):
    t0 = time.perf_counter()
    w["R"].grad = w["T"].grad = None
    tri1 = ops.rigid_apply(w["tri1"].reshape(B, 3 * N, 3), w["R"], w["T"], transpose_r=True).reshape(B, N, 9)
    t1 = time.perf_counter()
    total = 0
    tfirst = None
    for j in range(B):
        one = Lmod.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri1[j:j + 1], w["tri2"][j:j + 1], w["lines"][j:j + 1], dev)
        if tfirst is None:
            tfirst = time.perf_counter()
        if one is not None:
            total = total + one
    t2 = time.perf_counter()
    total.backward()
    t3 = time.perf_counter()
    if timed:
        for i, v in enumerate((t1 - t0, tfirst - t1, t2 - tfirst, t3 - t2)):
            stamps[i] += v
    return total


for _ in range(20):
    loop_step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    loop_step(True)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / n * 1e6
print(f"loop: {tot:.0f} us per step; host phases (us): rigid apply {stamps[0] / n * 1e6:.0f} | first call (whole batch + sync) "
      f"{stamps[1] / n * 1e6:.0f} | seven served calls + sums {stamps[2] / n * 1e6:.0f} | backward {stamps[3] / n * 1e6:.0f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    loop_step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
