"""Executed work of the culled scan (in-kernel counters, rrl_scan_counters) and its launch time under the per-step cell
order and under the prepared k-d order (rrl_cloud_order) at one shape.  usage: order_counters.py [B,N,M,L]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import bench
from rrl_hip import ops

B, N, M, L = (int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (8, 4096, 4096, 10000)
dev = torch.device("cuda", 0)
w = bench.make_workload(B, N, M, L, 0, dev)
t1, t2, ln = w["tri1"], w["tri2"], w["lines"]
torch.cuda.synchronize(); t0 = time.perf_counter()
o1, o2 = ops.cloud_order(t1), ops.cloud_order(t2)
torch.cuda.synchronize(); first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(20):
    ops.cloud_order(t1)
torch.cuda.synchronize()
print(f"rrl_cloud_order: first call (both clouds) {first * 1e3:.2f} ms, then {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call of B={B} clouds of {N}")
names = ["sphere_A", "half_B", "halves_passed", "point0_tests", "candidates"]
for tag, opts in (("cell order (per-step sort)", None), ("k-d order (prepared)", ops.make_opts(order1=o1, order2=o2))):
    st = ops.LossState(B, N, M, L, B, dev)
    ops.scan_counters(True)
    ops.loss_forward_raw(t1, t2, ln, opts=opts, state=st)
    torch.cuda.synchronize()
    c = ops.scan_counters(False).cpu().numpy().astype(np.float64)
    ops.scan_timing(1)
    for _ in range(30):
        ops.loss_forward_raw(t1, t2, ln, opts=opts, state=st)
    torch.cuda.synchronize()
    t = ops.scan_timing_collect()
    ops.scan_timing(0)
    per = c[:5] / (2 * B * L)
    print(f"{tag:28s} scan {np.mean(t[5:]) * 1e3:6.1f} us (HIP events)  per (line, cloud): " + "  ".join(f"{n}={v:.1f}" for n, v in zip(names, per))
          + f"  | per launch: point0 {c[3] / 1e6:.2f} M, loss_sum {float(st.loss.sum()):.10f}")
