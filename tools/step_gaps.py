import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    name=r['Kernel_Name'].split('(')[0].replace('void ','').replace('gpsmi::','')
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),name))
rows.sort()
spans=[r for r in rows if r[2].startswith('trk_span_kernel<8, 4, 0')]
corrs=[r for r in rows if r[2].startswith('trk_corr_kernel<4, 0>')]
import bisect
cs=[c[0] for c in corrs]
gaps=[];cd=[];sd=[];per=[]
for sp in spans[-101:-1]:                 # the timed region: the last launches of the run
    i=bisect.bisect_left(cs,sp[1]-1000)
    if i<len(corrs):
        gaps.append((corrs[i][0]-sp[1])/1e3)
    sd.append((sp[1]-sp[0])/1e3)
for a,b in zip(spans[-101:-2],spans[-100:-1]): per.append((b[0]-a[0])/1e3)
for c in corrs[-101:-1]: cd.append((c[1]-c[0])/1e3)
import statistics as st
print('gap span->next corr: median %.1f'%st.median(gaps),'span %.1f'%st.median(sd),'corr %.1f'%st.median(cd),'period %.1f'%st.median(per))
