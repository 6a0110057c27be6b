#!/bin/bash
# usage (GPU box, repo root): [ENV=...] tools/kt.sh <tag> <B,N,M,L> [steps] [diag]  -> per-kernel averages of the fused step
TAG=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python3 $R/tools/step_loop.py "$@" 2>/dev/null | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$TAG -o s -- python3 $R/tools/step_loop.py "$@" > /dev/null 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/kt_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0])))
keep = [r for r in rows if int(r["Calls"]) >= 100 and "rocclr" not in r["Name"] and "at::" not in r["Name"]]
print("  " + "  ".join(f"{r['Name'].replace('void ', '').split('(')[0][:22]}={float(r['AverageNs'])/1e3:.2f}" for r in keep),
      " SUM=%.1f" % sum(float(r["AverageNs"]) / 1e3 for r in keep))
PY
