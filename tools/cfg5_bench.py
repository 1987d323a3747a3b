"""BASELINE configs[4] alone (12-channel tracking at 16.368 Msps, N_CYC = 8, 512 MiB of IQ):
the measure_cfg5 leg of bench.py without the rest of the bench, for kernel traces
(`rocprofv3 --kernel-trace --stats -- python3 tools/cfg5_bench.py`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))

import bench  # noqa: E402
from gpsmi import engine as E  # noqa: E402

if __name__ == '__main__':
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    print(json.dumps(bench.measure_cfg5(E, 0, iters=iters)))
