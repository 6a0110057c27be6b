#!/bin/bash
# usage (GPU box, repo root): [RRL_FRAGMENT_NO_RIDE=1] tools/fragment_kt.sh -> kernel totals of tools/fragment_timing.py under rocprofv3
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_frag -o s -- python3 $R/tools/fragment_timing.py > $O/kt_frag.log 2>&1
cd $R
tail -5 $O/kt_frag.log | cut -c1-200
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/kt_frag/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:14]:
    print(f"  {r['Name'].replace('void ', '').split('(')[0][:44]:44s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:8.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
