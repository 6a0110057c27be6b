import os, sys, time, cProfile, pstats
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import callsites as C, ops, synth
from LieAlgebra import se3
import loss as L
B, n, nl = 8, 4096, 10000
prs = [synth.make_pair(b, n, n) for b in range(B)]
cu = lambda k, f=lambda x: x: torch.from_numpy(np.stack([f(p[k]) for p in prs])).cuda()
d = {"points_src_sample": cu("src"), "points_tar_sample": cu("tar"),
     "points_based_neighs_src": cu("src_tri", lambda x: x.reshape(-1, 3)),
     "points_based_neighs_tar": cu("tar_tri", lambda x: x.reshape(-1, 3))}
d["tar_box"] = L.generate_bbox(d["points_tar_sample"]).cuda()
d["centers"] = d["points_tar_sample"].mean(1)
d["order_src"] = ops.cloud_order(d["points_based_neighs_src"].reshape(B, -1, 9))
d["order_tar"] = ops.cloud_order(d["points_based_neighs_tar"].reshape(B, -1, 9))
d["p0_rows"] = torch.ones(B, dtype=torch.bool)
gen = torch.Generator().manual_seed(0)
Rs, ts = se3.exp3(0.05 * torch.randn(3 * B, 6, generator=gen))
Rs, ts = Rs.reshape(3, B, 3, 3).cuda().requires_grad_(True), ts.reshape(3, B, 3).cuda().requires_grad_(True)
torch.manual_seed(0)
lines = C.draw_lines(C.bounding_radius(d["tar_box"]), d["centers"], nl, d["points_src_sample"], d["points_tar_sample"])
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def frag():
    Rs.grad = ts.grad = None
    pred = [torch.cat([Rs[i], ts[i][..., None]], -1) for i in range(k)]
    out = C.rpm_intersection_loss(pred, d, lines=lines)
    out["loss_intersection"].backward()
for _ in range(20): frag()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200): frag()
torch.cuda.synchronize(); print("ms per call", (time.perf_counter()-t0)/200*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): frag()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
