"""The "streamed from host" leg of bench.py alone, on random raw blocks."""
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
    Rs = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 8, 32]
    rng = np.random.default_rng(2)
    raw = rng.integers(0, 65536, (96, bench.NGPS), dtype=np.uint16)
    chans = [(2 + c, -4000.0 + 700.0 * c, (137 * c + 11) % 2048) for c in range(12)]
    print(json.dumps(bench.measure_streamed(E, 0, raw, chans, Rs=Rs)))
