import sys, os, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd')]
from gpsmi import synth, engine as E
from gpsmi.pipeline import Receiver
LAG = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # Receiver(report_lag=LAG)
sc = synth.default_scene(12, seed=7)
raw = [sc.block_raw(b) for b in range(133)]
for raw_u8 in (True, False):
    blocks = [r if raw_u8 else synth.raw_to_c64(r) for r in raw]
    rx = Receiver(E.Config(max_sat=12), raw_u8=raw_u8, report_lag=LAG)
    for b in blocks[:5]:
        rx.feed(b)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for rep in range(5):
        for b in blocks[5:]:
            rx.feed(b)
    rx.drain()
    pr.disable()
    t1 = time.perf_counter()
    print('raw_u8', raw_u8, 'us/block', (t1 - t0) / (5 * 128) * 1e6)
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
    # without the profiler
    t0 = time.perf_counter()
    for rep in range(5):
        for b in blocks[5:]:
            rx.feed(b)
    rx.drain()
    t1 = time.perf_counter()
    print('raw_u8', raw_u8, 'report_lag', LAG, 'unprofiled us/block', (t1 - t0) / (5 * 128) * 1e6)
    n = rx.pool.trk.get_option('stat_stream_steps')
    print('   steps', n, 'wait us/step', rx.pool.trk.get_option('stat_stream_wait_ns') / n / 1e3, 'launch us/step', rx.pool.trk.get_option('stat_stream_launch_ns') / n / 1e3)
    rx.close()
