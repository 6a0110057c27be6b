#!/bin/bash
# usage (GPU box, repo root): tools/refine_ab.sh  -- A/B of the local k-d block refinement (RRL_REFINE=0/1):
# config sweep step times and the per-kernel averages of the timed bench step
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in 0 1; do
  RRL_REFINE=$v python3 tools/config_sweep.py > $O/refine${v}_sweep.jsonl 2> $O/refine${v}_sweep.err
  RRL_REFINE=$v python3 tools/config_sweep.py 64,4096,4096,10000 >> $O/refine${v}_sweep.jsonl 2>> $O/refine${v}_sweep.err
done
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  export RRL_REFINE=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/refine${v}_stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/refine${v}_bench.json 2> $O/refine${v}_bench.err
done
cd $R
for v in 0 1; do echo "== RRL_REFINE=$v"; cat $O/refine${v}_sweep.jsonl | cut -c1-160; cat $O/refine${v}_bench.json | cut -c1-200; f=$(find $O/refine${v}_stats -name '*kernel_stats.csv' | head -1); head -14 $f | cut -d, -f1-5 ; done
