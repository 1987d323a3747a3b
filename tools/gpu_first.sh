#!/bin/bash
# First-contact script for a gpurun box: load the library, run the GPU tests,
# keep the logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/first_contact.log
import sys, time
sys.path.insert(0, 'gps-sdr-receiver_amd')
from gpsmi import engine
print('devices:', engine.device_count(), engine.device_name(0))
PY
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log
