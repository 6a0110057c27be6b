#!/bin/bash
# usage (GPU box, repo root): tools/spw_sweep.sh  -- supergroups per slice of the culled scan (compile-time SPW):
# rebuilds per setting, step time at the bench shape / B=64 / L=20000 / C4 and one parity test
for flags in "-DSPW=8" "-DSPW=10" "-DSPW=11" "-DSPW=12 -DWCCAP=64" "-DSPW=16 -DWCCAP=64"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo -n "$flags  "
  python3 tools/config_sweep.py 8,4096,4096,10000 64,4096,4096,10000 8,4096,4096,20000 8,2048,1024,10000 1,1024,1024,20000 2>/dev/null | python3 -c "
import sys,json
print('  '.join('%s: %.1f' % (json.loads(l)['config'].replace(' N=','/').replace(' M=','/').replace(' L=','/'), json.loads(l)['us_per_step']) for l in sys.stdin))"
  python3 -m pytest tests/test_gpu_parity.py -x -q -k "baseline_configs or soak or cull_is_exact" 2>&1 | tail -1
done
unset RRL_HIPCC_FLAGS  # (experimental builds live in lib_exp/: the default library was never touched)
