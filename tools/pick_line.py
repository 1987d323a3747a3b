import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    p=d['pipeline']
    print(f, d['steps'], d['ms_per_step'], 'ovl',p['overlapped_ms_per_step'],'host',p.get('host_us_per_step'),'wait',p.get('host_wait_us_per_step'), 'corr',d['kernels_ms']['correlator'],'cp',d['kernels_ms']['codephase_correlation'])
