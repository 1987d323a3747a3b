#!/usr/bin/env python3
"""Copy the evidence of the last tools/prof_round.sh run (gpurun_out/round/) into
profiles/round2/ and rebuild the per-(kernel, grid) table at the end of
profiles/round2/bench_kernel_summary.md.  The prose of that file and the numbers quoted in
DESIGN.md are edited by hand from what this prints."""
import glob
import json
import os
import shutil
import subprocess
import sys

R, P = 'gpurun_out/round/', 'profiles/round2/'
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
trace = newest(R + 'stats/*/*_kernel_trace.csv')
ksum = subprocess.check_output([sys.executable, 'tools/prof_summary.py', trace], text=True)
os.makedirs(P, exist_ok=True)
shutil.copy(R + 'bench_line.json', P + 'bench_line.json')
shutil.copy(R + 'bench_prof_line.json', P + 'bench_line_under_rocprof.json')
shutil.copy(newest(R + 'stats/*/*_kernel_stats.csv'), P + 'bench_kernel_stats.csv')
for f in ('replay_pmc_counters.txt', 'kernel_bench.txt', 'span_prof.txt', 'clock_settling.txt'):
    if os.path.exists(R + f):
        shutil.copy(R + f, P + f)
md = P + 'bench_kernel_summary.md'
if os.path.exists(md):
    s = open(md).read()
    s = s[:s.index('| kernel | workgroups')] + ksum
    open(md, 'w').write(s)
d = json.load(open(P + 'bench_line.json'))
p = json.load(open(P + 'bench_line_under_rocprof.json'))
for line in ksum.splitlines():
    if 'trk_span_kernel<8' in line or 'trk_corr_kernel<4' in line:
        print(line)
print('events (unprofiled / profiled run):', d['roofline']['kernel_ms'], p['roofline']['kernel_ms'],
      '| value', d['value'], 'ms_per_step', d['ms_per_step'], '| closed loop', d['closed_loop']['us_per_block'])
