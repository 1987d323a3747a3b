#!/bin/bash
# trk_corr_kernel under its diagnostic flags (GPSMI_DEBUG_FLAGS 32 = fold without loads,
# 64 = no transforms): kernel-trace durations of the replay batch, one run per setting.
# The switches exist in the diagnostics build only: `make -C gps-sdr-receiver_amd diag` first
# (lib/libgpsmi_diag.so travels to the GPU box like the product library).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPSMI_LIB_PATH=$GRAFT_REPO_ROOT/gps-sdr-receiver_amd/lib/libgpsmi_diag.so
out=gpurun_out/corr_diag
rm -rf $out; mkdir -p $out
for f in ${1:-0 32 64 96}; do
  export GPSMI_DEBUG_FLAGS=$f
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/f$f -- python3 tools/kernel_bench.py --iters 5 > $out/f$f.log 2>&1 || { echo "flags $f failed"; tail -5 $out/f$f.log; exit 1; }
  echo "== flags $f" >> $out/summary.txt
  python3 tools/prof_summary.py $(find $out/f$f -name "*kernel_trace.csv") | grep -E "trk_corr|kernel \|" >> $out/summary.txt
done
cat $out/summary.txt
