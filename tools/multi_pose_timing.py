"""Round 5: the iterative trainers' poses as ONE multi-pose evaluation (callsites.multi_pose_loss, rrl_opts.problems)
against round 4's loop pose after pose (target scan carried over) -- RPM's fragment (rpm/Train_RPM.py:188-259) at the C2
shape (B = 8, N = M = 4096, 10000 lines given, the dataset's orders in the dict), num_iter = 1, 2, 3: forward + backward to
the predicted transforms, Chamfer monitor included.  Eager (what a trainer that swaps its fragment gets) and as a hipGraph
replay (device time).  usage: multi_pose_timing.py [B n L]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import callsites as C, ops, synth
from rrl_hip.graph import GraphedStep
from LieAlgebra import se3
import loss as L

B, n, nl = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 4096, 10000)
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


def timeit(f, n_=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n_):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_ * 1e3


keep = {}
for k in (1, 2, 3):
    def frag():
        Rs.grad = ts.grad = None
        pred = [torch.cat([Rs[i], ts[i][..., None]], -1) for i in range(k)]
        out = C.rpm_intersection_loss(pred, d, lines=lines)
        out["loss_intersection"].backward()
        keep["out"] = (out["loss_intersection"].detach(), out["loss_chamfer"], Rs.grad)
        return keep["out"]
    row = {}
    for multi in (True, False):
        C.MULTI_POSE = multi
        e = timeit(frag)
        val = [x.clone() for x in keep["out"]]
        try:
            g = GraphedStep(frag)
            gms = timeit(g)
        except Exception as exc:
            gms = float("nan")
            print("capture failed:", type(exc).__name__, exc)
        row[multi] = (e, gms, val)
    same = torch.equal(row[True][2][0], row[False][2][0])
    print(f"B={B} N=M={n} L={nl} RPM fragment, {k} pose(s): multi-pose {row[True][0]:.3f} ms eager / {row[True][1]:.3f} ms graph | "
          f"pose after pose {row[False][0]:.3f} / {row[False][1]:.3f} | loss bits equal {same} | "
          f"loss {float(row[True][2][0]):.6f} chamfer {float(row[True][2][1]):.6f}")
C.MULTI_POSE = True
