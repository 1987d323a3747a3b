#!/usr/bin/env python3
"""Average duration of the headline step's kernels over the TIMED region of a bench.py run (its
`steps` batch launches in front of the `pipeline.other_pass_steps` launches of the pipeline that is
not the timed one) from a rocprofv3 --kernel-trace CSV, next to the average over all launches
(which includes the settle and warm-up steps from a cold device) and over that other pass.
usage: timed_region.py <trace dir> <bench_line.json>"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
line = json.load(open(sys.argv[2]))
steps = int(line['steps'])
other = int(line.get('pipeline', {}).get('other_pass_steps', 0))
rows = {}
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('gpsmi::', '')
    wgs = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    if name in ('trk_span_kernel<8, 4, 0, 0, 32>', 'trk_span_kernel<8, 4, 0, 0>', 'trk_corr_kernel<4, 0>',
                'trk_epilogue_kernel') and wgs in (512, 3072):
        rows.setdefault(name, []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
print('| kernel | launches | avg us, all launches | avg us, the %d timed launches (isolated pipeline) | '
      'avg us, the %d launches of the other (overlapped) pass | bench.py (events) |' % (steps, other))
print('|---|---|---|---|---|---|')
for name, v in rows.items():
    v.sort()
    d = [x[1] for x in v]
    # the complex64 batch launches end with the timed region (the raw-u8 and other legs use other kernels)
    timed = d[-(steps + other):-other] if other else d[-steps:]
    tail = d[-other:] if other else []
    ev = ''
    if name.startswith('trk_span'):
        ev = '%.1f us (profiled run: kernel_ms of its own line)' % (line['roofline']['kernel_ms'] * 1e3)
    tl = f'{sum(tail) / len(tail) / 1e3:.2f}' if tail else ''
    print(f'| {name} | {len(d)} | {sum(d) / len(d) / 1e3:.2f} | {sum(timed) / len(timed) / 1e3:.2f} | {tl} | {ev} |')
