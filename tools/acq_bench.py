#!/usr/bin/env python3
"""The acquisition searches of BASELINE configs[1] (32 SV x 41 bins x 1 ms), configs[3]
(32 SV x 201 bins x 10 ms, all SVs on this GPU) and configs[1]'s grid at 16.368 Msps, on random
IQ resident in HBM: device time per search (HIP events), for kernel traces and PMC passes."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))
from gpsmi import engine as E  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--hirate', action='store_true', help='also CODE_SAMPLES = 16368 (configs[4]\'s rate)')
ap.add_argument('--grid', default='all', help='all | cfg2 (41 bins x 1 ms) | cfg4 (201 bins x 10 ms)')
a = ap.parse_args()
rng = np.random.default_rng(4)
prns = list(range(1, 33))
for cs, n_cyc in ((2048, 32),) + (((16368, 8),) if a.hirate else ()):
    acq = E.AcqEngine(E.Config(code_samples=cs, n_cyc=n_cyc))
    n = cs * n_cyc
    iq = (rng.standard_normal((n, 2)) * 0.3).astype(np.float32)
    buf = E.DeviceBuffer(iq.nbytes)
    buf.upload(iq)
    grids = []
    if a.grid in ('all', 'cfg2'):
        grids.append(('32 SV x 41 bins x 1 ms', [-5000.0 + 250.0 * i for i in range(41)], 1))
    if cs == 2048 and a.grid in ('all', 'cfg4'):
        grids.append(('32 SV x 201 bins x 10 ms', [-5000.0 + 50.0 * i for i in range(201)], 10))
    for name, freqs, n_avg in grids:
        ms = []
        for i in range(a.iters + 3):
            acq.search((buf.ptr, n), prns, freqs, n_avg)
            if i >= 3:
                ms.append(acq.last_ms())
        print(f'cs {cs}: {name}: {np.median(ms) * 1e3:.1f} us per search (min {min(ms) * 1e3:.1f})')
    acq.close()
    buf.free()
