#!/bin/bash
# usage (GPU box, repo root): tools/demo_kt.sh [diag] -> per-kernel averages of the one-call demo epoch
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_demo -o s -- python3 $R/tools/demo_kernels.py "$@" > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/kt_demo/**/*kernel_stats.csv", recursive=True)[0])))
keep = [r for r in rows if int(r["Calls"]) >= 2500]
for r in keep:
    print(f"  {r['Name'].replace('void ', '').split('(')[0][:40]:40s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:7.2f} us")
print("  SUM per epoch = %.1f us" % sum(float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 3000 for r in keep))
PY
