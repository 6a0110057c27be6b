#!/bin/bash
# usage (build container, repo root; CPU only): oracle/sanitize.sh -- the CPU oracle (oracle/rrl_oracle.c) built with
# AddressSanitizer + UndefinedBehaviorSanitizer, run against the reference's golden vectors (tests/test_oracle_golden.py);
# the regular build is restored afterwards.  (GPU sanitizers are not available on this pool.)
set -e
cd "$(dirname "$0")/.."
cp oracle/librrl_oracle.so /tmp/librrl_oracle.so.bak
trap 'cp /tmp/librrl_oracle.so.bak oracle/librrl_oracle.so' EXIT
gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -Wall -Wextra -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o oracle/librrl_oracle.so oracle/rrl_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider
