#!/bin/bash
# usage (GPU box, repo root): tools/scan_levels.sh [B,N,M,L ...]  -- where the culled scan's time goes, level by level: the same
# step with the walk cut off after the prologue (staging barrier), after level A, after level B, after level D without the
# candidates' exact resolution, and whole (experimental builds: results of the cut-off runs are invalid; the library is rebuilt
# on the box for every variant).  Plain (unchained) step: the scan is a launch of its own.
SHAPES=${@:-"8,4096,4096,10000 64,4096,4096,10000"}
export RRL_CHAIN=0 RRL_STEP=loss
for FL in "-DCULL_STOP_STAGE" "-DCULL_STOP_A" "-DCULL_STOP_C" "-DCULL_NO_RESOLVE" "-DRRL_EXPERIMENT"; do
  export RRL_HIPCC_FLAGS="$FL"
  python3 -c "
import sys; sys.path.insert(0, 'a-robust-registration-loss_amd')
from rrl_hip import build; build.build_lib(force=True)" > /dev/null 2>&1
  for S in $SHAPES; do
    echo "== $FL  $S"
    tools/kt.sh r06_lv $S 100 2>&1 | grep -v amdgpu | tail -1
  done
done
