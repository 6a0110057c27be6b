#!/bin/bash
# usage (GPU box, repo root): tools/attic/spw_exp2.sh -- where does the fat-slice variant (tools/attic/spw_exp.sh) start to pay?
for case in "|" "-DSPW=16 -DCULL_REGLINES=1 -DQA_CAP=384 -DQC_CAP=256|8,16"; do
  flags="${case%%|*}"; geom="${case##*|}"
  if [ -n "$flags" ]; then export RRL_HIPCC_FLAGS="$flags"; else unset RRL_HIPCC_FLAGS; fi
  if [ -n "$geom" ]; then export RRL_CULL_GEOM="$geom"; else unset RRL_CULL_GEOM; fi
  python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo "== [$flags] geom [$geom]"
  for sh in 12,4096,4096,10000 16,4096,4096,10000 24,4096,4096,10000 32,4096,4096,10000 8,4096,4096,20000 8,16384,16384,512 8,2048,1024,10000 16,2048,2048,10000 8,8192,8192,10000; do
    RRL_STEP=loss tools/kt.sh spw $sh 100 | grep -o "shape [0-9,]*\|: [0-9.]* us per step\|cull_scan_kernel[^ ]*=[0-9.]*" | tr '\n' ' '; echo
  done
done
unset RRL_HIPCC_FLAGS RRL_CULL_GEOM
