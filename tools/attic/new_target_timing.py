"""The fused step when EVERY step brings new clouds that come with their orders (a trainer's batches from the dataset shim:
pre_dataloader adds order_src / order_tar per item) -- prepared build of BOTH clouds, nothing kept -- against the kept
target (a registration loop on one pair) and the cold step (no orders).  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch
import bench
from rrl_hip import ops
dev = torch.device("cuda", 0)
B, N, L = 8, 4096, 10000
w = bench.make_workload(B, N, N, L, 0, dev)
R, T = w["R"].detach(), w["T"].detach()
o1, o2 = ops.cloud_order(w["tri1"]), ops.cloud_order(w["tri2"])
srcs = [w["tri1"], w["tri1"].clone()]
tars = [w["tri2"], w["tri2"].clone()]
def run(name, st, fn, n=400):
    for _ in range(20): fn(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per step")
st = ops.RegistrationStep(w["tri1"], w["tri2"], L, transpose_r=True, src_order=o1, tar_order=o2)
run("prepared, target kept (one pair, many poses)", st, lambda i: st(R, T, w["lines"]))
run("prepared, NEW source and target every step (orders from the dataset)", st,
    lambda i: st(R, T, w["lines"], src_tri=srcs[i & 1], tar_tri=tars[i & 1], src_order=o1, tar_order=o2))
cold = ops.RegistrationStep(w["tri1"], w["tri2"], L, transpose_r=True, prepared=False)
run("cold (no orders: records + cell sort every step)", cold, lambda i: cold(R, T, w["lines"], src_tri=srcs[i & 1], tar_tri=tars[i & 1]))
