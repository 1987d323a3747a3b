#!/bin/bash
# A/B of library builds on one box with bench.py itself (isolated pipeline), alternating
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; mkdir -p $out
for round in 1 2 3; do
for v in "$@"; do
  export GPSMI_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib$v.so
  timeout -k 10 200 python3 bench.py --no-cpu --no-extra > $out/bench_$v$round.json 2> $out/bench_$v$round.err || { echo "$v failed"; tail -3 $out/bench_$v$round.err; }
  python3 - $out/bench_$v$round.json $v $round <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print('==',sys.argv[2],sys.argv[3],'ms_per_step',d['ms_per_step'],d['kernels_ms'],'ok',d['checks']['replay_equals_closed_loop'])
P
done
done
