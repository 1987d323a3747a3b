#!/usr/bin/env python3
"""Duration of every batch launch of the span correlator in a rocprofv3 kernel trace, in
launch order, averaged in groups: shows how the kernel's duration drifts over a run.
usage: span_series.py <dir with *kernel_trace.csv> [group]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
for r in csv.DictReader(open(f)):
    if 'trk_span_kernel<8, 4, 0' in r['Kernel_Name'] or 'trk_corr_kernel<4, 0>' in r['Kernel_Name']:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'span' if 'span' in r['Kernel_Name'] else 'corr'))
rows.sort()
t0 = rows[0][0]
print("""# Mean duration in us of the batch launches of the correlator (trk_span_kernel<8,4>) and of the
# code-phase correlation (trk_corr_kernel<4>) in groups of %d launches, in launch order; in
# brackets the time in ms since the first launch.  Made by tools/prof_round.sh from
#   rocprofv3 --kernel-trace -- python3 bench.py --no-cpu --no-extra --steps 1500 --warmup 5 --settle-steps 0
# i.e. a cold start: the device needs ~40 ms under load to reach its steady clocks, from there
# the durations are flat.  bench.py drives the device with --settle-steps (300) untimed steps
# ahead of its warm-up steps for that reason.""" % grp)
for kind in ('span', 'corr'):
    d = [((e - s) / 1e3, (s - t0) / 1e6) for s, e, k in rows if k == kind]
    print(kind, len(d), 'launches; mean us per group of', grp, '(start ms of the group)')
    print('  ' + ' '.join('%.1f(%.0f)' % (sum(x[0] for x in d[i:i + grp]) / len(d[i:i + grp]), d[i][1]) for i in range(0, len(d), grp)))
