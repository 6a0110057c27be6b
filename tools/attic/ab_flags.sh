#!/bin/bash
# usage (GPU box, repo root): tools/ab_flags.sh "<shape>" "<flags A>" "<flags B>" [rounds] -- the fused step's kernel averages
# (tools/kt.sh) under two builds of the library, alternating on the SAME box (box-to-box differences are ~0.5 us)
SHAPE=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do
  for flags in "$A" "$B"; do
    export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "[$flags] BUILD FAILED"; continue; }
    echo "[$flags] $(tools/kt.sh ab $SHAPE 400 | tail -1)"
  done
done
unset RRL_HIPCC_FLAGS
python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1
