#!/usr/bin/env python3
"""Where the drop-in path's time goes: wall time of Receiver.feed per block, split into the
report blocks (one a second: wait + absorb + hand-off) and the others (copy + enqueue)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gps-sdr-receiver_amd')]
from gpsmi import synth, engine as E, receiver as R
from gpsmi.pipeline import Receiver

sc = synth.default_scene(12, seed=7)
raw = [sc.block_raw(b) for b in range(133)]
for raw_u8 in (True, False):
    blocks = [r if raw_u8 else synth.raw_to_c64(r) for r in raw]
    rx = Receiver(E.Config(max_sat=12), raw_u8=raw_u8)
    for b in blocks[:5]:
        rx.feed(b)
    t_rep, t_oth, n_rep, n_oth = 0.0, 0.0, 0, 0
    parts = {'wait': 0.0, 'extract': 0.0, 'absorb': 0.0}
    orig_wait, orig_extract = rx.pool.trk.wait, R.extract_records

    def timed_wait():
        t = time.perf_counter(); orig_wait(); parts['wait'] += time.perf_counter() - t

    def timed_extract(*a):
        t = time.perf_counter(); x = orig_extract(*a); parts['extract'] += time.perf_counter() - t; return x
    rx.pool.trk.wait = timed_wait
    R.extract_records = timed_extract
    g = rx.pool.trk.get_option
    keys = ('stat_waits', 'stat_quiesce_ns', 'stat_evwait_ns', 'stat_backlog', 'stat_stream_steps', 'stat_stream_wait_ns', 'stat_stream_launch_ns')
    for b in blocks[5:70]:
        rx.feed(b)                      # (past the first launches and their one-off costs)
    rx.drain()
    base = {k: g(k) for k in keys}
    t_rep, t_oth, n_rep, n_oth = 0.0, 0.0, 0, 0
    parts = {'wait': 0.0, 'extract': 0.0, 'absorb': 0.0}
    t0 = time.perf_counter()
    for rep in range(5):
        for b in blocks[5:]:
            t = time.perf_counter()
            dg = rx.feed(b)
            dt = time.perf_counter() - t
            if dg is not None:
                t_rep += dt; n_rep += 1
            else:
                t_oth += dt; n_oth += 1
    rx.drain()
    tot = time.perf_counter() - t0
    R.extract_records = orig_extract
    n = n_rep + n_oth
    print(f'raw_u8 {raw_u8}: {tot / n * 1e6:.1f} us/block; {n_oth} plain blocks {t_oth / n_oth * 1e6:.1f} us each; '
          f'{n_rep} report blocks {t_rep / n_rep * 1e6:.0f} us each (wait {parts["wait"] / n_rep * 1e6:.0f}, '
          f'extract {parts["extract"] / n_rep * 1e6:.0f}); per block: plain {t_oth / n * 1e6:.1f} + report {t_rep / n * 1e6:.1f}')
    d = {k: g(k) - base[k] for k in keys}
    nw, ns = max(1, d['stat_waits']), max(1, d['stat_stream_steps'])
    print(f"   worker per step: wait {d['stat_stream_wait_ns'] / ns / 1e3:.1f} us, runtime calls {d['stat_stream_launch_ns'] / ns / 1e3:.1f} us; "
          f"gpsmi_trk_wait at a report block: {d['stat_backlog'] / nw:.1f} jobs queued, quiesce {d['stat_quiesce_ns'] / nw / 1e3:.0f} us, "
          f"event {d['stat_evwait_ns'] / nw / 1e3:.0f} us ({nw} waits)")
    rx.close()
