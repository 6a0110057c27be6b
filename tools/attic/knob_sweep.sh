#!/bin/bash
# usage (GPU box): tools/knob_sweep.sh "-DBGRP=16" "-DBGRP=8" ...   rebuilds the library per flag set, times the step
for flags in "$@"; do
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  echo "== $flags"
  python3 tools/reuse_timing.py 2>/dev/null | head -1
  python3 - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, "a-robust-registration-loss_amd"); sys.path.insert(0, ".")
from rrl_hip import ops, synth
import loss as Lmod
B, N, L = 8, 4096, 10000
prs = [synth.make_pair(b, N, N) for b in range(B)]
lns = []
for b, p in enumerate(prs):
    torch.manual_seed(b)
    lns.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), L, torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda(); tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda(); ln = torch.stack(lns)
ops.scan_timing(1)
for _ in range(30): st = ops.loss_forward_raw(src, tar, ln)
torch.cuda.synchronize()
t = ops.scan_timing_collect(); ops.scan_timing(0)
print("   scan launch us: mean %.1f min %.1f  loss0 %.6f" % (np.mean(t[5:]) * 1e3, np.min(t) * 1e3, float(st.loss[0])))
PY
done
