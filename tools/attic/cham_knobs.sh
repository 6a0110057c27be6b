#!/bin/bash
# GPU box: rebuild with -D knobs, time the tree Chamfer (graph replay) and take rocprof kernel stats (experiments)
R=$PWD
for k in "-DNNW=4" "-DNNW=8" "$@"; do
  export RRL_HIPCC_FLAGS="$k"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1
  echo "== $k"
  python3 tools/chamfer_prof.py 8 4096 4096 50 1 2>&1 | grep -v amdgpu.ids
  (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ck; CHAM_NO_ALIGNED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ck -o s -- python3 $R/tools/chamfer_prof.py 8 4096 4096 50 1 > /dev/null 2>&1; python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/ck/**/*kernel_stats.csv", recursive=True)[0])):
    if float(r["Percentage"]) > 1: print("   bench clouds only: %-40s calls %5s avg %8.1f us min %8.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
)
done
unset RRL_HIPCC_FLAGS  # (experimental builds live in lib_exp/: the default library was never touched)
