#!/usr/bin/env python3
"""Per-(kernel, grid) duration summary of a rocprofv3 --kernel-trace CSV.
bench.py launches the tracking kernels both per block (closed loop, tiny
grids) and once per batch (replay); rocprofv3's own --stats averages the two
together, this keeps them apart.  usage: prof_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('gpsmi::', '')
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    wg = int(r['Workgroup_Size_X'])
    rows[(name, grid // wg, wg, r['VGPR_Count'], r['LDS_Block_Size'])].append(
        int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('| kernel | workgroups | threads | VGPR | LDS B | calls | avg us | min us | max us |')
print('|---|---|---|---|---|---|---|---|---|')
for (name, wgs, wg, vgpr, lds), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f'| {name} | {wgs} | {wg} | {vgpr} | {lds} | {len(d)} | {sum(d) / len(d) / 1e3:.2f} | '
          f'{min(d) / 1e3:.2f} | {max(d) / 1e3:.2f} |')
