#!/bin/bash
# usage (GPU box, repo root): tools/rec_exp.sh ["flags" ...] -- records-kernel anatomy by compile-time knobs (experimental builds)
for flags in "$@"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo "== [$flags]"
  tools/kt.sh rec 8,4096,4096,10000 300 | tail -1
done
unset RRL_HIPCC_FLAGS
python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1
