#!/usr/bin/env python3
"""Experiment: alternate tracking batches between two handles (two compute streams) so
that the code-phase correlation of batch k+1 can run beside the correlator of batch k."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))
from gpsmi import engine as E

nb, nch, NGPS = 1024, 12, 65536
rng = np.random.default_rng(1)
buf = E.DeviceBuffer(nb * NGPS * 8)
chunk = (rng.standard_normal((16, NGPS, 2)) * 0.25).astype(np.float32)
for i in range(0, nb, 16):
    buf.upload(chunk, i * NGPS * 8)
def make():
    trk = E.TrkEngine(max_ch=nch)
    for c in range(nch):
        trk.open(c, 2 + c, -4000.0 + 700.0 * c, (1137 * c + 11) % 2048)
    st = np.zeros((nb, nch), dtype=E.STATE_DTYPE)
    for c in range(nch):
        st[:, c] = trk.get_state(c)
    dly = np.broadcast_to(st['delay'][0], (nb, nch)).copy()
    trk.replay_load(nb, st, dly)
    return trk
for nh in (1, 2):
    hs = [make() for _ in range(nh)]
    pins = [[E.PinnedArray((nb, nch), E.OUT_DTYPE) for _ in range(2)] for _ in range(nh)]
    N = 60
    for k in range(N + 6):
        if k == 6:
            for h in hs: h.wait()
            E.sync(0)
            t0 = time.perf_counter()
        h = hs[k % nh]
        h.replay_run_async(buf.ptr, nb)
        h.replay_fetch_async(pins[k % nh][(k // nh) & 1].array)
        h.wait_prev()
    for h in hs: h.wait()
    dt = time.perf_counter() - t0
    print(f'{nh} handle(s): {dt / N * 1e3:.4f} ms per batch; last kernels {hs[0].last_ms()}')
    for h in hs: h.close()
