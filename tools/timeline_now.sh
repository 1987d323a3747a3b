#!/bin/bash
# step timeline of the bench step (kernel trace of a short bench run) -> gpurun_out/<tag>_timeline.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-now}
out=gpurun_out/tl_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-extra --no-cpu $BENCH_ARGS > $out/line.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 tools/step_timeline.py $out/stats 2 28 > gpurun_out/${tag}_timeline.txt
python3 tools/prof_summary.py $(find $out/stats -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_kernel_table.md
rm -rf $out/stats
head -24 gpurun_out/${tag}_timeline.txt
