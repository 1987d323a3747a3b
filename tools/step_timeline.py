"""Kernel timeline of the last few steps of a bench.py run from a rocprofv3 --kernel-trace CSV:
start / end of every kernel relative to the start of a step's code-phase correlation."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # headline-shaped launches behind the timed region
rows = []
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('gpsmi::', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
rows.sort()
# the batch launches of the complex64 headline step (the raw-u8 leg runs <4, 1>); the timed steps are
# the last of them
corrs = [i for i, r in enumerate(rows) if r[2].startswith('trk_corr_kernel<4, 0>') and r[3] == 3072]
i0 = corrs[-(nsteps + 3 + skip)]
t0 = rows[i0][0]
for s, e, n, g in rows[i0:]:
    if s > rows[corrs[-(3 + skip)]][0]:
        break
    print(f'{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  {n} [{g}]')
