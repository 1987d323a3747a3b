"""Closed loop of R batched receivers (gpsmi_trk_set_streams) on random IQ: us per step and
Msamples/s per R.  `GPSMI_SPAN_SINGLE_MAX=n python3 tools/batched_bench.py` moves the point at
which the span correlator switches from its single-block to its batch form."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))

import bench  # noqa: E402
from gpsmi import engine as E  # noqa: E402

if __name__ == '__main__':
    Rs = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 8, 32, 64, 128, 256]
    nb = 1024
    rng = np.random.default_rng(3)
    buf = E.DeviceBuffer(nb * bench.NGPS * 8, 0)
    chunk = (rng.standard_normal((32, bench.NGPS, 2)) * 0.3).astype(np.float32)
    for i in range(0, nb, 32):
        buf.upload(chunk, i * bench.NGPS * 8)
    chans = [(2 + c, -4000.0 + 700.0 * c, (137 * c + 11) % 2048) for c in range(12)]
    out = bench.measure_batched(E, 0, lambda b: buf.at(b * bench.NGPS * 8), nb, chans, Rs=Rs, max_steps=32)
    print(json.dumps({'span_single_max': os.environ.get('GPSMI_SPAN_SINGLE_MAX', 'default'), 'batched': out}))
