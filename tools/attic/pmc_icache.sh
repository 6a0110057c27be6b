#!/bin/bash
# usage (GPU box, repo root): tools/pmc_icache.sh <tag> <kernel-substring>   -- instruction-cache counters of one
# kernel of the bench step (own pass, --kernel-trace only); prints per-launch means
TAG=$1; KER=$2
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_IFETCH_LEVEL" | sort -u > $O/${TAG}_icache_names.txt
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_ic$i -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-extras --no-dist > $O/${TAG}_ic$i.log 2>&1
done
cd $R
python3 - "$TAG" "$KER" <<'PY'
import csv, collections, glob, json, sys
tag, ker = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(f"gpurun_out/{tag}_ic*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if ker in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
out = {c: sum(d.values()) / len(d) for c, d in agg.items()}
print(json.dumps(out, indent=1)); print(open(f"gpurun_out/{tag}_icache_names.txt").read())
PY
tail -3 gpurun_out/${TAG}_ic1.log
