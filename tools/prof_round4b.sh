#!/bin/bash
# Round 4, second session: the pieces of prof_round4.sh's `rest` part that the session's changes touch
# (batch epilogue, code-phase correlation, stream layout), the bench line once more behind the PMC pass
# (so that it carries `traffic`), and a two-rank rehearsal of the N > 1 path on one GPU.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round4
mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench_line.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
echo "bench done"
python3 tools/kernel_bench.py > $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --delays aligned >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --code-samples 16368 --n-cyc 8 --blocks 512 --iters 5 >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --n-cyc 16 --blocks 2048 >> $out/kernel_bench.txt 2>&1
python3 tools/kernel_bench.py --n-cyc 8 --blocks 4096 >> $out/kernel_bench.txt 2>&1
for f in 1 0; do echo "epilogue_form $f" >> $out/kernel_bench.txt; GPSMI_EPILOGUE_FORM=$f python3 tools/kernel_bench.py --iters 30 2>&1 | tail -1 >> $out/kernel_bench.txt; done
echo "kernel_bench done"
python3 tools/batched_bench.py > $out/batched_bench.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/cfg5 -- python3 tools/cfg5_bench.py 8 > $out/cfg5_line.json 2>$out/cfg5.err \
  && python3 tools/prof_summary.py $(find $out/cfg5 -name "*kernel_trace.csv" | head -1) > $out/cfg5_kernel_table.md; rm -rf $out/cfg5
python3 tools/feed_split.py > $out/feed_split.txt 2>&1
echo "side benches done"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 20 --warmup 3 --rehearse --no-extra > $out/rehearse2.json 2> $out/rehearse2.err || { echo "rehearsal failed"; tail -8 $out/rehearse2.err; }
tail -c 1500 $out/rehearse2.json
ls $out
