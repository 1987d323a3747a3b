#!/usr/bin/env python3
"""SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS per probe kernel of
tools/probe/lds_conflict_probe.hip from a rocprofv3 counter_collection.csv tree."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print('| kernel | LDS instructions | IDX_ACTIVE / instr | BANK_CONFLICT / instr | conflict / active |')
print('|---|---|---|---|---|')
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = m.get('SQ_INSTS_LDS', 0) or 1
    a, c = m.get('SQ_LDS_IDX_ACTIVE', 0), m.get('SQ_LDS_BANK_CONFLICT', 0)
    print(f'| {k} | {n:.0f} | {a / n:.2f} | {c / n:.2f} | {c / a if a else 0:.2f} |')
