#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> <outdir-tag>   (run on the GPU box via gpurun)
# Collects two SQ counter passes for one kernel of bench.py and prints per-dispatch sums.
K=$1; TAG=$2
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -o b -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -o b -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_b.log 2>&1
cd $R
python3 - "$K" "$TAG" <<'PY'
import csv, collections, sys
K, TAG = sys.argv[1], sys.argv[2]
for d in ("a", "b"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{TAG}_{d}/b_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if K in r["Kernel_Name"]:
            agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    o = agg[sorted(agg)[-1]]
    print({k: "%.3g" % v for k, v in o.items()})
PY
