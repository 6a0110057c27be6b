"""End-to-end time of the trainers' loss fragments (rrl_hip.callsites) at B=8: transform + line
drawing (every call, as the trainers do) + per-sample loss for every pose + Chamfer monitor +
backward to the predicted transforms.  Eager (host-bound) numbers: what a trainer that simply
swaps its fragment for one call gets."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import callsites as C, synth
from LieAlgebra import se3
import loss as L

def data_for(B, n):
    prs = [synth.make_pair(b, n, n) for b in range(B)]
    cu = lambda k, f=lambda x: x: torch.from_numpy(np.stack([f(p[k]) for p in prs])).cuda()
    d = {"points_src_sample": cu("src"), "points_tar_sample": cu("tar"),
         "points_based_neighs_src": cu("src_tri", lambda x: x.reshape(-1, 3)),
         "points_based_neighs_tar": cu("tar_tri", lambda x: x.reshape(-1, 3))}
    d["tar_box"] = L.generate_bbox(d["points_tar_sample"]).cuda()
    d["centers"] = d["points_tar_sample"].mean(1)
    return d

def timeit(f, n=100):
    for _ in range(10): f()  # (the caching allocator needs a few rounds to settle: every evaluation leases a 48 MB state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

if os.environ.get("RRL_FRAGMENT_NO_RIDE"):  # A/B: the monitor as a launch of its own after every evaluation (round 3 / 4a)
    C._ride_monitor = lambda *a, **k: False
    print("(the Chamfer monitor does NOT ride in the scan launches)")
for B, n in ((8, 1024), (8, 4096)):
    d = data_for(B, n)
    gen = torch.Generator().manual_seed(0)
    Rs, ts = se3.exp3(0.05 * torch.randn(3 * B, 6, generator=gen))
    Rs, ts = Rs.reshape(3, B, 3, 3).cuda().requires_grad_(True), ts.reshape(3, B, 3).cuda().requires_grad_(True)
    def rpm():
        Rs.grad = ts.grad = None
        pred = [torch.cat([Rs[i], ts[i][..., None]], -1) for i in range(3)]
        out = C.rpm_intersection_loss(pred, d, n_lines=10000)
        out["loss_intersection"].backward()
    def dcp():
        Rs.grad = ts.grad = None
        dd = {k: (v.transpose(2, 1).contiguous() if k.startswith("points_") else v) for k, v in d.items()}
        loss, cd, _, _ = C.dcp_intersection_loss(dd, Rs[0], ts[0], n_lines=15000)
        loss.backward()
    for rng_name, flag in (("GPU RNG", True), ("CPU RNG stream", False)):
        C.DEVICE_RNG = flag
        print(f"B={B} N=M={n}: RPM fragment (3 poses, 10000 lines, target scan reused) {timeit(rpm):.2f} ms | "
              f"DCP fragment (15000 lines) {timeit(dcp):.2f} ms   [{rng_name}]")
