#!/bin/bash
# counter passes over tools/kernel_bench.py for two delay patterns (mixed-wave study)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_mixed
mkdir -p $out
for d in aligned onewave; do
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${d}_pass$i -- python3 tools/kernel_bench.py --iters 3 --delays $d > $out/${d}_pass$i.log 2>&1 || { echo "$d pass $i failed"; tail -3 $out/${d}_pass$i.log; }
done
done
python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1 || true
tail -60 $out/summary.txt
