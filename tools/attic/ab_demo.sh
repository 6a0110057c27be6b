#!/bin/bash
# usage (GPU box, repo root): tools/ab_demo.sh "<flags A>" "<flags B>" -- kernel averages of the one-call demo epoch under two builds
for flags in "$1" "$2"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "[$flags] BUILD FAILED"; continue; }
  echo "[$flags]"; tools/demo_kt.sh | grep "se3_adam\|sample_\|SUM"
done
unset RRL_HIPCC_FLAGS
python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1
