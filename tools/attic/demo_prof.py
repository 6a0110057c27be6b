import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a-robust-registration-loss_amd"))
import loss as L
from rrl_hip import synth, ops
pr = synth.make_pair(1, 1024, 1024)
dev = torch.device("cuda:0")
v1 = torch.from_numpy(pr["src"]).to(dev); v2 = torch.from_numpy(pr["tar"]).to(dev)
R = torch.tensor([[2.0]], device=dev); c = v2.mean(0)
def T(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
ms, rands = T(lambda: L._uniform_rounds(1, 20000, 10)); print("cpu rand rounds ms", ms)
ms, rg = T(lambda: rands.to(dev)); print("H2D ms", ms)
bb1, bb2 = ops.aabb(v1[None]), ops.aabb(v2[None])
ms, _ = T(lambda: ops.sample_lines(rands, R.reshape(1), c.reshape(1, 3), bb1, bb2)); print("sample_lines (incl. upload) ms", ms)
ms, lines = T(lambda: L.Random_uniform_distribution_lines_batch_efficient_resample(R, c.reshape(1, 3), 20000, v1[None], v2[None], dev)); print("full sampler ms", ms)
tri1 = torch.from_numpy(pr["src_tri"]).to(dev)[None].requires_grad_(True); tri2 = torch.from_numpy(pr["tar_tri"]).to(dev)[None]
def fb():
    tri1.grad = None
    l = L.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri1, tri2, lines, dev)
    l.backward(); return l
ms, _ = T(fb); print("drop-in loss fwd+bwd ms", ms)
ms, _ = T(lambda: L.chamfer_dist(v1[None], v2[None]).item()); print("chamfer+item ms", ms)
