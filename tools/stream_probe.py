import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, 'gps-sdr-receiver_amd'); sys.path.insert(0, '.')
import bench
from gpsmi import engine as E
NGPS = 65536
chans = [(2 + c, -4000.0 + 700.0 * c, (137 * c + 11) % 2048) for c in range(12)]
rng = np.random.default_rng(2)
for R in (1, 2, 4, 8):
    for mode in ('stream', 'zerocopy'):
        trk = E.TrkEngine(E.Config(), max_ch=12, streams=R)
        trk.set_input_format(True)
        for r in range(R):
            for c, (s, f, d) in enumerate(chans):
                trk.open(c, s, f, d, stream=r)
        trk.set_timing(False)
        ring = [E.PinnedArray((R, NGPS), np.uint16) for _ in range(3)]
        for p in ring:
            p.array[:] = rng.integers(0, 65536, (R, NGPS), dtype=np.uint16)
        best = None
        for rep in range(3):
            E.sync(0)
            t0 = time.perf_counter()
            for i in range(48):
                if mode == 'stream':
                    trk.process_stream(ring[i % 3].array)
                else:
                    trk.process(C.c_void_p(ring[i % 3].array.ctypes.data), want_out=False)
            trk.wait()
            t = (time.perf_counter() - t0) / 48
            best = t if best is None else min(best, t)
        print(R, mode, round(best * 1e6, 1), 'us/step')
        trk.close()
        for p in ring: p.free()
