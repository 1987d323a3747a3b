import sys, numpy as np
sys.path[:0] = ['gps-sdr-receiver_amd', 'tests']
from conftest import scene_for, load_golden
from gpsmi.engine import TrkEngine, DeviceBuffer, STATE_DTYPE
g = load_golden('ref_default.npz')
sc = scene_for('default')
nb = 24
res = {}
for fmt in (False, True):
    eng = TrkEngine(max_ch=12)
    eng.set_input_format(fmt)
    for c, (sv, f0, d0) in enumerate(g['trk_init']):
        eng.open(c, int(sv), float(f0), int(d0))
    st = np.zeros((nb, 12), dtype=STATE_DTYPE)
    for c in range(12):
        st[:, c] = eng.get_state(c)
    blks = [sc.block_raw(5 + i) if fmt else sc.block(5 + i) for i in range(nb)]
    buf = DeviceBuffer(nb * blks[0].nbytes)
    for i, b in enumerate(blks):
        buf.upload(b, i * b.nbytes)
    res[fmt, 'batch'] = eng.replay(buf.ptr, nb, st, None)
    res[fmt, 'single'] = eng.replay(buf.ptr, 1, st[:1], None)
    eng.close()
for form in ('batch', 'single'):
    a, b = res[False, form], res[True, form]
    bad = [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
    print(form, 'fields that differ:', bad)
    if 'dumps' in bad:
        d = np.abs(a['dumps'] - b['dumps'])
        print('  dumps: per-block max diff', d.reshape(a.shape[0], -1).max(axis=1)[:6], 'per dump index (block 0, ch 0):', d[0, 0][:12])
print('c64 batch vs single equal:', res[False, 'batch'][:1].tobytes() == res[False, 'single'].tobytes())
