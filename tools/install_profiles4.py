#!/usr/bin/env python3
"""Copy the evidence of tools/prof_round4.sh (gpurun_out/round4/) into profiles/round4/ -- what the
round's numbers in DESIGN.md and bench.py's `traffic` / issue rooflines are read from."""
import json
import os
import shutil

R, P = 'gpurun_out/round4/', 'profiles/round4/'
os.makedirs(P, exist_ok=True)
names = {'bench_line.json': 'bench_line.json', 'bench_prof_line.json': 'bench_line_under_rocprof.json',
         'bench_kernel_stats.csv': 'bench_kernel_stats.csv', 'bench_kernel_table.md': 'bench_kernel_table.md',
         'step_timeline.txt': 'step_timeline.txt', 'step_timeline_overlapped.txt': 'step_timeline_overlapped.txt', 'bench_kernel_summary.md': 'bench_kernel_summary.md', 'pmc_counters.txt': 'pmc_counters.txt',
         'span_prof.txt': 'span_prof.txt', 'span_prof_nc8.txt': 'span_prof_nc8.txt', 'pfa_prof.txt': 'pfa_prof.txt', 'kernel_bench.txt': 'kernel_bench.txt',
         'acq_bench.txt': 'acq_bench.txt', 'batched_bench.txt': 'batched_bench.txt',
         'stream_bench.txt': 'stream_bench.txt', 'cfg5_kernel_table.md': 'cfg5_kernel_table.md',
         'cfg5_line.json': 'cfg5_line.json', 'clock_settling.txt': 'clock_settling.txt',
         'chain_floor.txt': 'chain_floor.txt', 'feed_trace_summary.txt': 'dropin_feed_trace.txt', 'dropin_profile.txt': 'dropin_profile.txt', 'feed_split.txt': 'dropin_feed_split.txt',
         'feed_trace_report_block.txt': 'dropin_feed_trace_report_block.txt'}
for src, dst in names.items():
    if os.path.exists(R + src):
        shutil.copy(R + src, P + dst)
        print('installed', dst)
for extra in ('gpurun_out/lds_conflict_probe.md',):
    if os.path.exists(extra):
        shutil.copy(extra, P + os.path.basename(extra))
        print('installed', os.path.basename(extra))
if os.path.exists(P + 'bench_line.json'):
    d = json.load(open(P + 'bench_line.json'))
    print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'],
          'cold', d['roofline'].get('frac_cold'), '| closed loop', d['closed_loop']['us_per_block'])
