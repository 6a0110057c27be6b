#!/bin/bash
# GPU box: LDS / issue counters of the culled scan (own passes, --kernel-trace only)
R=$PWD; O=$R/gpurun_out; TAG=${1:-lds}
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_p$i -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-extras --no-dist > $O/${TAG}_p$i.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(f"gpurun_out/{tag}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "cull_scan_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
out = {c: sum(d.values()) / len(d) for c, d in agg.items()}
print(json.dumps(out, indent=1))
PY
