#!/bin/bash
# rocprofv3 counter passes over tools/kernel_bench.py (one --pmc set per run).
# usage: tools/prof_pmc.sh <tag>   -> gpurun_out/pmc_<tag>/passN/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-x}
out=gpurun_out/pmc_$tag
mkdir -p $out
python3 tools/kernel_bench.py --iters 10 | tee $out/plain.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -- python3 tools/kernel_bench.py --iters 3 > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; }
done
find $out -name "*counter_collection.csv" | head
