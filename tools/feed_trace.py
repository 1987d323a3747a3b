import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd')]
from gpsmi import synth, engine as E
from gpsmi.pipeline import Receiver
sc = synth.default_scene(12, seed=7)
raw = [sc.block_raw(b) for b in range(69)]
rx = Receiver(E.Config(max_sat=12), raw_u8=True)
for b in raw[:5]:
    rx.feed(b)
t0 = time.perf_counter()
for rep in range(4):
    for b in raw[5:]:
        rx.feed(b)
rx.drain()
print('us/block', (time.perf_counter() - t0) / 256 * 1e6)
rx.close()
