#!/usr/bin/env python3
"""Host-side cost of enqueueing one bench step (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))
from gpsmi import engine as E
from gpsmi.acquisition import Acquisition

nb, nch, NGPS = 1024, 12, 65536
rng = np.random.default_rng(1)
trk = E.TrkEngine(max_ch=nch)
acq = Acquisition()
buf = E.DeviceBuffer(nb * NGPS * 8)
chunk = (rng.standard_normal((16, NGPS, 2)) * 0.25).astype(np.float32)
for i in range(0, nb, 16):
    buf.upload(chunk, i * NGPS * 8)
for c in range(nch):
    trk.open(c, 2 + c, -4000.0 + 700.0 * c, (1137 * c + 11) % 2048)
st = np.zeros((nb, nch), dtype=E.STATE_DTYPE)
for c in range(nch):
    st[:, c] = trk.get_state(c)
dly = np.broadcast_to(st['delay'][0], (nb, nch)).copy()
trk.replay_load(nb, st, dly)
pins = [E.PinnedArray((nb, nch), E.OUT_DTYPE) for _ in range(2)]
f41 = [-5000.0 + 250.0 * i for i in range(41)]
shard = list(range(1, 33))
acq_pin = E.PinnedArray((41, 32), E.PEAK_DTYPE)
T = {'search': 0.0, 'run': 0.0, 'fetch': 0.0, 'acqwait': 0.0, 'waitprev': 0.0}
N = 50
for k in range(N + 5):
    if k == 5:
        for key in T: T[key] = 0.0
        t_all = time.perf_counter()
    t0 = time.perf_counter()
    if k > 0:
        acq.engine.wait()
    t1 = time.perf_counter()
    acq.engine.search_async(buf.ptr, NGPS, shard, f41, 1, acq_pin.array, None)
    t2 = time.perf_counter()
    trk.replay_run_async(buf.ptr, nb)
    t3 = time.perf_counter()
    trk.replay_fetch_async(pins[k & 1].array)
    t4 = time.perf_counter()
    trk.wait_prev()
    t5 = time.perf_counter()
    T['acqwait'] += t1 - t0; T['search'] += t2 - t1; T['run'] += t3 - t2
    T['fetch'] += t4 - t3; T['waitprev'] += t5 - t4
trk.wait()
tot = time.perf_counter() - t_all
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, 'us per step; total', round(tot / N * 1e6, 1), 'us; kernels', trk.last_ms())
