#!/usr/bin/env python3
"""cProfile of the bench step's host side (run on the GPU box)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import torch
import bench
from rrl_hip import ops, dist as rdist
dev = torch.device("cuda", 0)
B, N, L = 8, 4096, 10000
w = bench.make_workload(B, N, N, L, 0, dev)
ones = torch.ones(B, device=dev)
src_pts = w["tri1"].reshape(B, -1, 3)
def step():
    w["R"].grad = w["T"].grad = None
    loss, info, _ = ops.registration_loss(w["tri1"], w["R"], w["T"], w["tri2"], w["lines"], (1, 1, 5, 5),
                                          transpose_r=True, want_payload=True)
    torch.autograd.backward([loss], [ones])
    return rdist.reduce_payload(ops.last_state().payload)
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:5000])
