import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd'), os.path.join(ROOT, 'tests')]
from conftest import load_golden, scene_for
from gpsmi.engine import TrkEngine, OUT_DTYPE
g = load_golden('ref_default.npz')
sc = scene_for('default')
nch = len(g['trk_init'])
e = TrkEngine(max_ch=nch)
for c, (sv, f0, d0) in enumerate(g['trk_init']):
    e.open(c, int(sv), float(f0), int(d0))
print('itemsize', OUT_DTYPE.itemsize)
for i in range(12):
    o = e.process(sc.block(5 + i))
    print(i, 'ms_count', o['ms_count'].tolist(), 'mask', [hex(int(x)) for x in o['edge_mask']], 's0', o['edge_sign0'].tolist(), 'locked', o['phase_locked'].tolist())
for c in range(3):
    st = e.get_state(c)
    print('state', c, st['edge_state'], st['prev_signal'], st['std_dev'], st['reserved'])
