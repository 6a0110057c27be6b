#!/bin/bash
# usage (GPU box, repo root): tools/attic/scan16_knobs.sh -- queue capacities of the fat scan variant (scan16), one at a time
# around the shipped (QA 384, QC 256, WCCAP 128); experimental builds in lib_exp/.  Prints the step and scan time at B = 16 / 64.
[ $# -eq 0 ] && set -- "" "-DSCAN16_QA_CAP=320" "-DSCAN16_QA_CAP=448" "-DSCAN16_QA_CAP=512" "-DSCAN16_QC_CAP=192" "-DSCAN16_QC_CAP=320" "-DSCAN16_QC_CAP=384" "-DSCAN16_WCCAP=96" "-DSCAN16_WCCAP=160" "-DSCAN16_WCCAP=192"
for flags in "$@"; do
  if [ -n "$flags" ]; then export RRL_HIPCC_FLAGS="$flags"; else unset RRL_HIPCC_FLAGS; fi
  python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo -n "[$flags] "
  for sh in 16,4096,4096,10000 64,4096,4096,10000; do
    RRL_STEP=loss tools/kt.sh s16 $sh 100 | grep -o ": [0-9.]* us per step\|cull_scan_kerne[^ ]*=[0-9.]*" | tr '\n' ' '
  done; echo
done
unset RRL_HIPCC_FLAGS
