"""How long the results copy of one bench step takes on this box (4.6 MB of gpsmi_trk_out records
from HBM into page-locked host memory on the copy stream), alone: the isolated step has ~0.23 ms
to hide it in.  Prints the GPU's NUMA node and this process's CPUs beside it."""
import glob, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gps-sdr-receiver_amd'))
import numpy as np
from gpsmi import engine as E

nb, nch = 1024, 12
trk = E.TrkEngine(max_ch=nch)
for c in range(nch):
    trk.open(c, c + 1, 0.0, 0)
buf = E.DeviceBuffer(nb * 65536 * 8)
st = np.zeros((nb, nch), dtype=E.STATE_DTYPE)
st['prn'] = np.arange(1, nch + 1)[None, :]
st['df_len'] = 1
trk.replay_load(nb, st, np.zeros((nb, nch), np.int32))
trk.replay_run(buf.ptr, nb)
pin = E.PinnedArray((nb, nch), E.OUT_DTYPE)
ts = []
for i in range(30):
    t0 = time.perf_counter()
    trk.replay_fetch(pin.array)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[5:]) * 1e3
mb = pin.array.nbytes / 1e6
print(f'results copy {mb:.2f} MB: min {ts.min():.3f} ms  median {np.median(ts):.3f} ms  max {ts.max():.3f} ms'
      f'  -> {mb / np.median(ts):.1f} GB/s')
for f in glob.glob('/sys/class/drm/card*/device/numa_node'):
    print(f, open(f).read().strip())
print('cpus', sorted(os.sched_getaffinity(0)))
try:
    for f in sorted(glob.glob('/sys/devices/system/node/node*/cpulist')):
        print(f, open(f).read().strip())
except OSError:
    pass
