#!/usr/bin/env python3
"""Copy the evidence of the last tools/prof_round.sh run (gpurun_out/round/) into
profiles/round1/ and refresh the numbers quoted from it in DESIGN.md and the summary."""
import glob, json, os, re, shutil, subprocess, sys
R, P = 'gpurun_out/round/', 'profiles/round1/'
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
trace = newest(R + 'stats/runc/*_kernel_trace.csv')
ksum = subprocess.check_output([sys.executable, 'tools/prof_summary.py', trace], text=True)
shutil.copy(R + 'bench_line.json', P + 'bench_line.json')
shutil.copy(R + 'bench_prof_line.json', P + 'bench_line_under_rocprof.json')
shutil.copy(newest(R + 'stats/runc/*_kernel_stats.csv'), P + 'bench_kernel_stats.csv')
shutil.copy(R + 'replay_pmc_counters.txt', P + 'replay_pmc_counters.txt')
shutil.copy(R + 'kernel_bench.txt', P + 'kernel_bench.txt')
d = json.load(open(P + 'bench_line.json'))
p = json.load(open(P + 'bench_line_under_rocprof.json'))
avg = None
for line in ksum.splitlines():
    c = [x.strip() for x in line.split('|')]
    if len(c) > 8 and c[1].startswith('trk_stream_mfma_kernel') and c[2] == '1024':
        avg = float(c[7])
ev_p, ev_u = p['kernels_ms']['correlator'] * 1e3, d['kernels_ms']['correlator'] * 1e3
s = open(P + 'bench_kernel_summary.md').read()
s = re.sub(r"\*\*[\d.]+ us\*\* average over the 23 batch launches in the trace,\n[\d.]+ us from the HIP "
           r"events inside that same \(profiled\) run,\n[\d.]+ us from the HIP events",
           f"**{avg:.1f} us** average over the 23 batch launches in the trace,\n{ev_p:.1f} us from the HIP "
           f"events inside that same (profiled) run,\n{ev_u:.1f} us from the HIP events", s)
s = s[:s.index('| kernel | workgroups')] + ksum
open(P + 'bench_kernel_summary.md', 'w').write(s)
t = open('DESIGN.md').read()
t = re.sub(r"the same command agrees \([\d.]+ µs in the trace vs [\d.]+ µs from the events of that "
           r"profiled\nrun, [\d.]+ µs unprofiled",
           f"the same command agrees ({avg:.1f} µs in the trace vs {ev_p:.1f} µs from the events of that "
           f"profiled\nrun, {ev_u:.1f} µs unprofiled", t)
t = re.sub(r"\*\*[\d.]+ Gsamples/s\*\*\n\([\d,]+ × real time\) whole job, [\d.]+ ms per 1024-block step",
           f"**{d['value'] / 1e3:.1f} Gsamples/s**\n({d['x_realtime']:,.0f} × real time) whole job, "
           f"{d['ms_per_step']} ms per 1024-block step", t)
tbs = 5.36870912e8 / (avg * 1e-6) / 1e12
t = re.sub(r"correlator [\d.]+ TB/s by the HIP events \(\*\*[\d.]+ % of HBM\npeak\*\*; [\d.]+ TB/s = \d+ % "
           r"by the trace\)",
           f"correlator {d['roofline']['achieved'] / 1e3:.2f} TB/s by the HIP events "
           f"(**{d['roofline']['frac'] * 100:.1f} % of HBM\npeak**; {tbs:.2f} TB/s = {tbs / 8 * 100:.0f} % "
           f"by the trace)", t)
t = re.sub(r"closed loop\n\d+ µs/block \(\d+ × real time",
           f"closed loop\n{d['closed_loop']['us_per_block']:.0f} µs/block "
           f"({d['closed_loop']['x_realtime']:.0f} × real time", t)
open('DESIGN.md', 'w').write(t)
print('trace avg', avg, 'events', ev_p, ev_u, 'value', d['value'], d['ms_per_step'], d['closed_loop'])
