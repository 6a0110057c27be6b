#!/bin/bash
# usage (GPU box): tools/ab_prof.sh libA.so libB.so ...   A/B of prebuilt libraries on the SAME box
# (kernel averages differ by ~1.5 us between boxes / runs): each library is copied over
# lib/librrl_hip.so in turn, alternating twice, and the eager bench is profiled.
L=a-robust-registration-loss_amd/lib
cp $L/librrl_hip.so /tmp/rrl_keep.so
for rep in 1 2; do
for so in "$@"; do
  cp $so $L/librrl_hip.so
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab -o w -- python3 /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1 )
  python3 - "$so" <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("/root/repo/gpurun_out/ab/w_kernel_stats.csv"))}
keys = ["tri_records_kernel", "tri_sort_kernel<4>", "cull_scan_kernel", "line_pair_dist_kernel", "loss_reduce_kernel", "loss_bwd_rt_kernel"]
print(f"{sys.argv[1][-28:]:28s} " + " ".join(f"{k.split('_kernel')[0][-8:]}={rows.get(k, 0):5.1f}" for k in keys) + f"  sum={sum(rows.get(k, 0) for k in keys):.1f}")
PY
done
done
cp /tmp/rrl_keep.so $L/librrl_hip.so
