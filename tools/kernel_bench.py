#!/usr/bin/env python3
"""Micro-benchmark of the tracking kernels in replay: random IQ, synthetic state
table, no closed loop -- for profiling and tuning only (bench.py is the
contract benchmark).  Prints per-kernel HIP-event times."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))
from gpsmi import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--blocks', type=int, default=1024)
ap.add_argument('--channels', type=int, default=12)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--settle', type=int, default=150,
                help='untimed runs first: the device needs ~40 ms under load to reach steady clocks')
ap.add_argument('--code-samples', type=int, default=2048)
ap.add_argument('--n-cyc', type=int, default=32)
ap.add_argument('--delays', default='spread',
                help='spread | aligned (multiples of 512: no mixed wave) | balanced | onewave')
a = ap.parse_args()

CS = a.code_samples
NGPS = CS * a.n_cyc
nb, nch = a.blocks, a.channels
rng = np.random.default_rng(1)
trk = E.TrkEngine(E.Config(code_samples=CS, n_cyc=a.n_cyc), max_ch=nch)
buf = E.DeviceBuffer(nb * NGPS * 8)
chunk = (rng.standard_normal((16, NGPS, 2)) * 0.25).astype(np.float32)
for i in range(0, nb, 16):
    n = min(16, nb - i)
    buf.upload(chunk[:n], i * NGPS * 8)
def delay_of(c):
    if a.delays == 'aligned':
        return (512 * c) % CS
    if a.delays == 'balanced':          # 2,2,1,1 mixed channels per wave in each group of six
        return [100, 300, 700, 900, 1200, 1800][c % 6] % CS
    if a.delays == 'onewave':
        return (40 * (c % 6) + 20) % CS
    return (1137 * c + 11) % CS


for c in range(nch):
    trk.open(c, 2 + c, -4000.0 + 700.0 * c, delay_of(c))
st = np.zeros((nb, nch), dtype=E.STATE_DTYPE)
for c in range(nch):
    st[:, c] = trk.get_state(c)
st['phase'] = rng.uniform(0, 6.28, (nb, nch)).astype(np.float32)
dly = np.broadcast_to(st['delay'][0], (nb, nch)).copy()
trk.replay_load(nb, st, dly)
tot, cor = [], []
trk.set_timing(False)
for i in range(a.settle):
    trk.replay_run_async(buf.ptr, nb)
trk.wait()
trk.set_timing(True)
for i in range(a.iters + 3):
    trk.replay_run(buf.ptr, nb)
    if i >= 3:
        t, c = trk.last_ms()
        tot.append(t)
        cor.append(c)
gb = nb * NGPS * 8 / 1e9
print(f'delays {a.delays} cs {CS} n_cyc {a.n_cyc} blocks {nb} channels {nch}: correlator {np.median(cor):.4f} ms '
      f'({gb / np.median(cor) * 1e3:.0f} GB/s, min {min(cor):.4f}), '
      f'all tracking kernels {np.median(tot):.4f} ms')
