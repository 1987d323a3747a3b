#!/bin/bash
# Round-end evidence on a gpurun box: bench line, rocprofv3 kernel stats of the same
# command, PMC passes of the tracking replay.  Everything lands in gpurun_out/round/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round
rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench_line.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py > $out/bench_prof_line.json 2> $out/bench_prof.err || { echo prof failed; tail -5 $out/bench_prof.err; exit 1; }
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/pass$i -- python3 tools/kernel_bench.py --iters 3 --settle 0 > $out/pmc_pass$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $out/pmc_pass$i.log; }
done
python3 tools/pmc_summary.py $out/pmc > $out/replay_pmc_counters.txt
tools/probe/span_prof 1024 > $out/span_prof.txt 2>&1 || echo "span_prof failed"
python3 tools/kernel_bench.py > $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --delays aligned >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 5 >> $out/kernel_bench.txt 2>&1
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
# how the durations of the two big kernels drift from a cold start (no settling steps)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/long -- python3 bench.py --no-cpu --no-extra --steps 1500 --warmup 5 --settle-steps 0 > $out/bench_cold_long.json 2> $out/bench_cold_long.err \
  && python3 tools/span_series.py $out/long 50 > $out/clock_settling.txt; rm -rf $out/long
cat $out/bench_line.json | cut -c1-300
