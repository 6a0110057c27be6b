#!/bin/bash
# usage (GPU box, repo root): [EXPS="0 1 3"] [FLAGS="-DRRL_CHAIN_EXP ..."] tools/chain_exp.sh [B,N,M,L]  -- where the chained step's fused
# launch spends its time: the same launch with parts of the hand-over switched off (experimental build -DRRL_CHAIN_EXP; results of
# those runs are invalid)
S=${1:-8,4096,4096,10000}
export RRL_HIPCC_FLAGS="${FLAGS:--DRRL_CHAIN_EXP}" RRL_STEP=loss
for e in ${EXPS:-0 1 3}; do echo "RRL_CHAIN_EXP=$e ($RRL_HIPCC_FLAGS)"; RRL_CHAIN_EXP=$e tools/kt.sh r06_chain_exp$e $S 400; done
echo "RRL_CHAIN=0"; RRL_CHAIN=0 tools/kt.sh r06_chain_off $S 400
