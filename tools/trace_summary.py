import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]           # steady part
dur = collections.defaultdict(list)
for r in rows:
    dur[r['Kernel_Name'][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    print(f'{k:62s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f} us')
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
print('span', (t1 - t0) / 1e3, 'us  busy', busy / 1e3, 'us  kernels', len(rows))
# gaps between consecutive kernels
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
names = [(a['Kernel_Name'][:25], b['Kernel_Name'][:25]) for a, b in zip(rows, rows[1:])]
g = collections.defaultdict(list)
for n, x in zip(names, gaps):
    g[n].append(x)
for n, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f'gap {n[0]:25s} -> {n[1]:25s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f} us')
