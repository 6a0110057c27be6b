"""A few monitored steps at C2 (ops.RegistrationStep(chamfer=True): the Chamfer walk rides in the scan's launch) -- the
workload of the PMC passes on cull_scan_chamfer_kernel (tools/pmc_any.sh <tag> cull_scan_chamfer -- tools/ride_step.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch
import bench
from rrl_hip import ops
dev = torch.device("cuda", 0)
B, N, L = 8, 4096, 10000
w = bench.make_workload(B, N, N, L, 0, dev)
st = ops.RegistrationStep(w["tri1"], w["tri2"], L, transpose_r=True, chamfer=True)
for _ in range(12):
    st(w["R"].detach(), w["T"].detach(), w["lines"])
torch.cuda.synchronize()
print("chamfer", float(st.chamfer_value), "rode", st.ride.done)
