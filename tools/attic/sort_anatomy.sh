#!/bin/bash
# usage (GPU box, repo root): tools/sort_anatomy.sh -- tri_sort_kernel cut off after each phase (-DSORT_STOP=k:
# 1 loads + AABB, 2 + cell histogram, 3 + scan, 4 + scatter into LDS, 5 + copy-out, none = + tree), rocprofv3 averages
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/sort_drv.py <<'PY'
import os, sys, numpy as np, torch
R = os.environ["RRL_ROOT"]
sys.path.insert(0, os.path.join(R, "a-robust-registration-loss_amd")); sys.path.insert(0, R)
from rrl_hip import ops, synth
B, N, L = 8, 4096, 2000
prs = [synth.make_pair(b, N, N) for b in range(B)]
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda(); tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
ln = torch.randn(B, L, 6, device="cuda"); ln[..., :3] = torch.nn.functional.normalize(ln[..., :3], dim=-1)
lib = __import__("rrl_hip._lib", fromlist=["x"]).load()
st = ops.LossState(B, N, N, L, B, src.device)
for _ in range(60):
    lib.rrl_tri_prepare(ops._p(src), ops._p(tar), ops._p(st.ws), st.nbytes, B, N, N, L, ops._stream())
torch.cuda.synchronize()
PY
export RRL_ROOT=$R
for k in 1 2 3 4 5 0; do
  flags=""; [ $k != 0 ] && flags="-DSORT_STOP=$k"
  export RRL_HIPCC_FLAGS="$flags"; python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || { echo "$flags: BUILD FAILED"; continue; }
  (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/sortan$k -o s -- python3 /tmp/sort_drv.py > /dev/null 2>&1)
  f=$(find $O/sortan$k -name '*kernel_stats.csv' | head -1)
  echo -n "SORT_STOP=$k  "; grep -E "tri_sort|tri_records" $f | awk -F, '{printf "%s avg %.2f us   ", substr($1,1,28), $4/1000}'; echo
done
unset RRL_HIPCC_FLAGS  # (experimental builds live in lib_exp/: the default library was never touched)
