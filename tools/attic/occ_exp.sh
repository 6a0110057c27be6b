#!/bin/bash
# usage (GPU box, repo root): tools/occ_exp.sh ["flags" ...] -- culled-scan queue / occupancy experiments (compile-time
# knobs, experimental builds in lib_exp/)
[ $# -eq 0 ] && set -- "" "-DQA_CAP=384" "-DQA_CAP=384 -DQC_CAP=320" "-DQA_CAP=320 -DWCCAP=192" "-DQA_CAP=512 -DQC_CAP=384 -DWCCAP=64"
for flags in "$@"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo "== [$flags]"
  tools/kt.sh occ 8,4096,4096,10000 200 | grep -o "shape.*per step\|cull_scan_kernel[^ ]*=[0-9.]*"
  tools/kt.sh occ 64,4096,4096,10000 40 | grep -o "shape.*per step\|cull_scan_kernel[^ ]*=[0-9.]*"
done
unset RRL_HIPCC_FLAGS
