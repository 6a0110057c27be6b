#!/bin/bash
# usage (GPU box): tools/knob_prof.sh <kernel-substring> "<flags>" ...  rebuilds per flag set, prints that kernel's average
K=$1; shift
for flags in "$@"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/kp -o w -- python3 /root/repo/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-graph > /dev/null 2>&1 )
  python3 - "$K" "$flags" <<'PY'
import csv, sys
for r in csv.DictReader(open("/root/repo/gpurun_out/kp/w_kernel_stats.csv")):
    if sys.argv[1] in r["Name"]: print(f"[{sys.argv[2]}] {r['Name'][:40]} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f}")
PY
done
