#!/bin/bash
# usage (GPU box, repo root): tools/scan_prologue_pmc.sh <tag>   (is the scan's prologue rate- or latency-bound?)
# (1) cache-path counters of the C2 step's scan launch, plain (cull_scan_kernel) and chained (cull_scan_build_kernel): own
#     rocprofv3 --pmc passes with --kernel-trace only; (2) the delay experiment: the plain scan with every second workgroup
#     entering ~2 us late (experimental build -DCULL_DELAY_HALF), in-kernel stamps by tools/scan_tail.py.
TAG=${1:-r06}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-other --issue direct --no-dist --no-fresh --no-parity"
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum"; do
  i=$((i+1))
  for c in 1 0; do
    RRL_CHAIN=$c timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_pp_c${c}_$i -o p -- python3 $R/bench.py $ARGS > $O/${TAG}_pp_c${c}_$i.log 2>&1
  done
done
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
out = {}
for c in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in glob.glob(f"gpurun_out/{tag}_pp_c{c}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "cull_scan" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in agg.items():
        m = {cn: sum(d.values()) / len(d) for cn, d in v.items()}
        if m.get("TCP_TCC_READ_REQ_sum"):
            m["avg_L1_to_L2_read_latency_cycles"] = m.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / m["TCP_TCC_READ_REQ_sum"]
        if m.get("TCC_REQ_sum"):
            m["L2_hit_rate"] = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
        out[("chained " if c else "plain ") + k] = m
json.dump(out, open(f"gpurun_out/{tag}_scan_prologue_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
export RRL_HIPCC_FLAGS="-DCULL_DELAY_HALF"
python3 -c "
import sys; sys.path.insert(0, 'a-robust-registration-loss_amd')
from rrl_hip import build; build.build_lib(force=True)" > /dev/null 2>&1
(echo "# -DCULL_DELAY_HALF: every second workgroup of the plain C2 scan enters ~2 us late"; RRL_DELAY_REPORT=1 python3 tools/scan_tail.py 2>&1 | grep -v amdgpu) > $O/${TAG}_scan_delay.txt
unset RRL_HIPCC_FLAGS
(echo "# default build"; python3 tools/scan_tail.py 2>&1 | grep -v amdgpu) >> $O/${TAG}_scan_delay.txt
cat $O/${TAG}_scan_delay.txt
