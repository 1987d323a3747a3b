#!/bin/bash
# A/B of library builds on one box: configs[4] kernels (kernel_bench at 16368 / N_CYC 8), alternating
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
for v in "$@"; do
  export GPSMI_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib$v.so
  echo "== $v $round: $(timeout -k 10 200 python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 10 2>&1 | tail -1)"
done
done
