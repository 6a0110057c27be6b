#!/bin/bash
# NOTE: written against the tree BEFORE the scan was split into two variants (commit "Culled scan: second geometry variant"): there -DSPW / -DCULL_REGLINES
# configured the one scan kernel; today the shipped library holds both variants (rrl_cull_scan.inc) and RRL_CULL_FAT=0/1 selects one.
# usage (GPU box, repo root): tools/attic/spw_exp.sh -- round 5: ONE generation of fatter workgroups for the culled scan?  The C2 grid
# is 1280 workgroups of 8 wavefronts, 768 fit (3 per CU, LDS-bound): 1.67 generations, the launch lasts two wavefront
# lifetimes.  Slices of 16 supergroups halve the grid (640 workgroups: one generation) at twice the work per wavefront and
# one prologue instead of two.  Compile-time knobs, experimental builds in lib_exp/; "flags|geom" per case.
[ $# -eq 0 ] && set -- "|" "-DSPW=16 -DCULL_REGLINES=1 -DQA_CAP=512 -DQC_CAP=384|8,16" "-DSPW=16 -DQA_CAP=512 -DQC_CAP=384|8,16" \
   "-DSPW=16 -DCULL_REGLINES=1 -DQA_CAP=384 -DQC_CAP=256|8,16" "-DCULL_REGLINES=1|" "-DSPW=16 -DCULL_REGLINES=1 -DQA_CAP=512 -DQC_CAP=384 -DWCCAP=192|8,16"
for case in "$@"; do
  flags="${case%%|*}"; geom="${case##*|}"
  if [ -n "$flags" ]; then export RRL_HIPCC_FLAGS="$flags"; else unset RRL_HIPCC_FLAGS; fi
  if [ -n "$geom" ]; then export RRL_CULL_GEOM="$geom"; else unset RRL_CULL_GEOM; fi
  python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo "== [$flags] geom [$geom]"
  for sh in 8,4096,4096,10000 64,4096,4096,10000 1,1024,1024,20000; do
    RRL_STEP=loss tools/kt.sh spw $sh 200 | grep -o "shape.*per step\|loss_sum [0-9.]*\|cull_scan_kernel[^ ]*=[0-9.]*" | tr '\n' ' '; echo
  done
done
unset RRL_HIPCC_FLAGS RRL_CULL_GEOM
