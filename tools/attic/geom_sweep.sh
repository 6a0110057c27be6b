#!/bin/bash
# usage (GPU box, repo root): tools/geom_sweep.sh   -- culled-scan launch geometry (RRL_CULL_GEOM="wavefronts per
# workgroup,supergroups per slice") at the bench shape and at B=64: step time (hipGraph) per setting
for g in "8,8" "8,4" "8,2" "4,8" "4,4" "6,8" "2,8" "8,1"; do
  echo -n "RRL_CULL_GEOM=$g  "
  RRL_CULL_GEOM=$g python3 tools/config_sweep.py 8,4096,4096,10000 64,4096,4096,10000 2>/dev/null | python3 -c "
import sys,json
print('  '.join('%s: %.1f us' % (json.loads(l)['config'], json.loads(l)['us_per_step']) for l in sys.stdin))"
done
