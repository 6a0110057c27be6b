#!/usr/bin/env python3
"""cProfile of the reference trainers' literal per-sample loop (bench.py variants.dropin_loop): where the host
time of `for j in range(B): loss += cal_loss_...(p1[j:j+1], ...)` + one backward goes (run on the GPU box)."""
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

def loop_step(backward=True):
    w["R"].grad = w["T"].grad = None
    tri1 = ops.rigid_apply(w["tri1"].reshape(B, 3 * N, 3), w["R"], w["T"], transpose_r=True).reshape(B, N, 9)
    total = 0
    for j in range(B):
        one = Lmod.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri1[j:j + 1], w["tri2"][j:j + 1], w["lines"][j:j + 1], dev)
        if one is not None:
            total = total + one
    if backward:
        total.backward()
    return total

def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

print(f"loop fwd+bwd: {timeit(loop_step):.0f} us per B=8 step;  forward only: {timeit(lambda: loop_step(False)):.0f} us")
with torch.no_grad():
    print(f"forward only, no_grad: {timeit(lambda: loop_step(False)):.0f} us")
one = lambda: Lmod.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, w["tri1"][:1], w["tri2"][:1], w["lines"][:1], dev)
print(f"one call, contiguous B=1 inputs, no grad: {timeit(one, 300):.0f} us")
st = ops.loss_forward_raw(w["tri1"][:1], w["tri2"][:1], w["lines"][:1])
def raw():
    ops.loss_forward_raw(w["tri1"][:1], w["tri2"][:1], w["lines"][:1])
print(f"raw forward without sync (host issue): {timeit(raw, 300):.0f} us")
def raw_sync():
    ops.loss_forward_raw(w["tri1"][:1], w["tri2"][:1], w["lines"][:1]); torch.cuda.synchronize()
print(f"raw forward + synchronize: {timeit(raw_sync, 300):.0f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): loop_step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
