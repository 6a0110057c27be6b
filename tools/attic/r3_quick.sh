#!/bin/bash
# usage (GPU box, repo root): tools/r3_quick.sh <tag> [pytest-args...]
# GPU tests (bounded), then the bench line and the per-kernel averages of the timed step alone.
TAG=${1:-q}; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -q -x "$@" > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -5 $O/${TAG}_pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_step -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras > /dev/null 2> $O/${TAG}_bench_step.err
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys, json
tag = sys.argv[1]
try:
    rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/{tag}_stats_step/**/*kernel_stats.csv", recursive=True)[0])))
    for r in rows[:14]:
        print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us")
except Exception as e:
    print("no stats", e)
try:
    j = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", j["ms_per_step"], "value", j["value"], j["config"]["issue"])
    print({k: v for k, v in j["variants"].items()})
    print({k: v for k, v in j["extras"].items() if "ms" in k})
except Exception as e:
    print("no bench", e)
PY
