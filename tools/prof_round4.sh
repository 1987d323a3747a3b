#!/bin/bash
# Round-4 evidence on a gpurun box.  Everything lands in gpurun_out/round4/; tools/install_profiles4.py
# copies what is judged into profiles/round4/.
#   bench line (settled and cold), rocprofv3 kernel stats of the same command, PMC passes over the
#   tracking replay at 2.048 and 16.368 Msps and over the acquisition searches, the probes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round4
part=${1:-all}            # bench | pmc | rest | all: three calls fit gpurun's 20-minute limit
mkdir -p $out
if [ $part = bench ] || [ $part = all ]; then
timeout -k 10 900 python3 bench.py > $out/bench_line.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
echo "bench done"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py > $out/bench_prof_line.json 2> $out/bench_prof.err || { echo prof failed; tail -5 $out/bench_prof.err; exit 1; }
echo "bench under rocprofv3 done"
python3 tools/prof_summary.py $(find $out/stats -name "*kernel_trace.csv" | head -1) > $out/bench_kernel_table.md
python3 tools/step_timeline.py $out/stats 2 28 > $out/step_timeline.txt
python3 tools/step_timeline.py $out/stats 2 0 > $out/step_timeline_overlapped.txt
python3 tools/timed_region.py $out/stats $out/bench_prof_line.json > $out/bench_kernel_summary.md
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
rm -rf $out/stats
cut -c1-300 $out/bench_line.json
fi
if [ $part = pmc ] || [ $part = all ]; then
rm -rf $out/pmc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/replay_pass$i -- python3 tools/kernel_bench.py --iters 3 --settle 0 > $out/pmc_replay$i.log 2>&1 || { echo "pmc replay pass $i failed"; tail -5 $out/pmc_replay$i.log; }
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/hirate_pass$i -- python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 3 --settle 0 > $out/pmc_hirate$i.log 2>&1 || { echo "pmc hirate pass $i failed"; tail -5 $out/pmc_hirate$i.log; }
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/acq_pass$i -- python3 tools/acq_bench.py --iters 3 --grid cfg4 > $out/pmc_acq$i.log 2>&1 || { echo "pmc acq pass $i failed"; tail -5 $out/pmc_acq$i.log; }
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc/acq2_pass$i -- python3 tools/acq_bench.py --iters 3 --grid cfg2 > $out/pmc_acq2$i.log 2>&1 || { echo "pmc acq2 pass $i failed"; tail -5 $out/pmc_acq2$i.log; }
  echo "pmc pass $i done"
done
python3 tools/pmc_summary.py $out/pmc --stamp > $out/pmc_counters.txt
rm -rf $out/pmc
fi
if [ $part = rest ] || [ $part = all ]; then
tools/probe/span_prof 1024 > $out/span_prof.txt 2>&1 || echo "span_prof failed"
tools/probe/span_prof_nc8 4096 1024 > $out/span_prof_nc8.txt 2>&1 || echo "span_prof_nc8 failed"    # (-DPROBE_NC=8)
tools/probe/pfa_prof 6144 > $out/pfa_prof.txt 2>&1 || echo "pfa_prof failed"
tools/probe/chain_floor > $out/chain_floor.txt 2>&1 || echo "chain_floor failed"
python3 tools/kernel_bench.py > $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --delays aligned >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 5 >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --n-cyc 16 --blocks 2048 >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --n-cyc 8 --blocks 4096 >> $out/kernel_bench.txt 2>&1
GPSMI_STREAM_MFMA=0 python3 tools/kernel_bench.py --n-cyc 16 --blocks 2048 >> $out/kernel_bench.txt 2>&1
GPSMI_STREAM_MFMA=0 python3 tools/kernel_bench.py --n-cyc 8 --blocks 4096 >> $out/kernel_bench.txt 2>&1
for ch in 0 64 128; do echo "fold_chunk $ch" >> $out/kernel_bench.txt; GPSMI_FOLD_CHUNK=$ch python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 8 2>&1 | tail -1 >> $out/kernel_bench.txt; done
python3 tools/prof_feed.py > $out/dropin_profile.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/feed -- python3 tools/feed_trace.py > $out/feed_trace.log 2>&1 \
  && python3 tools/trace_summary.py $out/feed > $out/feed_trace_summary.txt; rm -rf $out/feed
python3 tools/acq_bench.py --hirate > $out/acq_bench.txt 2>&1
python3 tools/batched_bench.py > $out/batched_bench.txt 2>&1
python3 tools/stream_bench.py > $out/stream_bench.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/cfg5 -- python3 tools/cfg5_bench.py 8 > $out/cfg5_line.json 2>$out/cfg5.err \
  && python3 tools/prof_summary.py $(find $out/cfg5 -name "*kernel_trace.csv" | head -1) > $out/cfg5_kernel_table.md; rm -rf $out/cfg5
# how the durations of the two big kernels drift from a cold start (no settling steps)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/long -- python3 bench.py --no-cpu --no-extra --steps 1500 --warmup 5 --settle-steps 0 > $out/bench_cold_long.json 2> $out/bench_cold_long.err \
  && python3 tools/span_series.py $out/long 50 > $out/clock_settling.txt; rm -rf $out/long
fi
ls $out
