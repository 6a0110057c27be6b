#!/bin/bash
# usage (a node with N MI355X, repo root): tools/scale_run.sh N [--global-batch 64] [bench.py flags...]
# Starts the batch-shard bench exactly as the driver does for N > 1: the launcher is chosen BEFORE anything touches a
# GPU (N = 1: plain python; N > 1: python -m torch.distributed.run, one rank per GPU over RCCL), and nothing re-execs
# afterwards.  Weak scaling by default (B = 8 per GPU); --global-batch 64 is BASELINE configs[2] (strong scaling).
# Every rank prints its RCCL evidence (nranks / rank / version) to stderr; rank 0 prints the JSON line.
set -e
N=${1:?number of GPUs}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
if [ "$N" -le 1 ]; then
  exec python3 "$R/bench.py" --gpus 1 "$@"
fi
PORT=${MASTER_PORT:-$((29500 + RANDOM % 1000))}
exec python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
  "$R/bench.py" --gpus "$N" "$@"
