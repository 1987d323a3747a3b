#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files: mean counter value per
kernel (short name) over its dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
if '--stamp' in sys.argv:          # first line: the kernel sources the counters were taken from
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    print('# csrc_sha256', bench.csrc_sha256())
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('gpsmi::', '').replace('void ', '')
        tag = f[len(root):].strip('/').split('/')[0]
        tag = tag.split('_pass')[0] if '_pass' in tag else ''
        if not any(w in k for w in ('stream', 'span', 'corr', 'epilogue', 'acq_', 'pfa', 'fold', 'big_')):
            continue
        acc[(tag + ' ' + k).strip()][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})')
