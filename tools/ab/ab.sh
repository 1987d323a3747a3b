#!/bin/bash
# A/B of library builds on one box: kernel_bench under rocprofv3 --kernel-trace per library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab; rm -rf $out; mkdir -p $out
for round in 1 2; do
for v in "$@"; do
  export GPSMI_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t_$v$round -- python3 tools/kernel_bench.py --iters 30 > $out/kb_$v$round.txt 2>&1 || { echo "$v failed"; tail -3 $out/kb_$v$round.txt; }
  echo "== $v round $round: $(tail -1 $out/kb_$v$round.txt)"
  python3 tools/prof_summary.py $(find $out/t_$v$round -name "*kernel_trace.csv" | head -1) | grep -E "trk_corr|trk_span_kernel|epilogue" 
  rm -rf $out/t_$v$round
done
done
