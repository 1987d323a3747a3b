#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/feed_trace.py: the kernels before each long idle gap
(the report block's flush), times relative to the start of the gap.  usage: trace_around_gaps.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:]) for r in csv.DictReader(open(f))
              if 'trk_' in r['Kernel_Name'] or 'stage_copy' in r['Kernel_Name']))
rows = rows[len(rows) // 2:]
shown = 0
for i in range(1, len(rows)):
    gap = rows[i][0] - rows[i - 1][1]
    if gap > 150_000 and i > 40 and shown < 2:
        shown += 1
        t0 = rows[i - 1][1]
        print(f'--- idle gap of {gap / 1e3:.0f} us; the 16 kernels before it and 6 after (us relative to its start)')
        for s, e, n in rows[i - 20:i + 9]:
            print(f'{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:6.1f} us  {n}')
