#!/bin/bash
# usage (on the GPU box, from the repo root): tools/quick_prof.sh <tag>  -> kernel averages of an eager bench run
tag=${1:-q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -o w -- python3 /root/repo/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > /root/repo/gpurun_out/bench_$tag.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_$tag/w_kernel_stats.csv")))
for r in rows[:9]: print(f"{r['Name'][:42]:42s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:8.1f}")
PY
