#!/bin/bash
# usage (GPU box, repo root): tools/ab_tree.sh <shape> [rounds] -- kernel averages of the fused step built from tools/ab_src/ (an older
# copy of csrc/, made before the call and not committed: for f in $(git ls-tree --name-only <rev> a-robust-registration-loss_amd/csrc/);
# do git show <rev>:$f > tools/ab_src/$(basename $f); done) and from csrc/, alternating on the same box
SHAPE=$1; R=${2:-2}
for i in $(seq $R); do
  export RRL_HIPCC_FLAGS="-DAB_OLD=1" RRL_CSRC=$PWD/tools/ab_src
  python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || echo "old: BUILD FAILED"
  echo "[old] $(tools/kt.sh ab $SHAPE 400 | tail -1)"
  unset RRL_CSRC; export RRL_HIPCC_FLAGS="-DAB_NEW=1"
  python3 a-robust-registration-loss_amd/rrl_hip/build.py > /dev/null 2>&1 || echo "new: BUILD FAILED"
  echo "[new] $(tools/kt.sh ab $SHAPE 400 | tail -1)"
done
unset RRL_HIPCC_FLAGS RRL_CSRC
