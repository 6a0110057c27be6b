#!/bin/bash
# usage (GPU box, repo root): tools/pmc_any.sh <tag> <kernel-substring> -- <python script + args>
# SQ counter passes (own runs, --kernel-trace only) for one kernel of an arbitrary script; prints per-launch means.
TAG=$1; KER=$2; shift 3
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_pmc$i -o p -- python3 $R/"$@" > $O/${TAG}_pmc$i.log 2>&1
done
cd $R
python3 - "$TAG" "$KER" <<'PY'
import csv, collections, glob, json, sys
tag, ker = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(f"gpurun_out/{tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if ker in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
out = {c: sum(d.values()) / len(d) for c, d in agg.items()}
out["_launches"] = {c: len(d) for c, d in agg.items()}
json.dump(out, open(f"gpurun_out/{tag}_pmc_{ker}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
