#!/usr/bin/env python3
"""A recording in, position fixes out: the reference's file mode (README.md:177-187,
gpsglob.LIVE_MEAS = False / BIN_DATA; gpsrecv.streamData -> processData -> UDP / pickle ->
gpseval) as one command on the GPU path.

    python tools/run_file.py <recording.bin> [--seconds S] [--start-stream K] [--save-pickle P]
                             [--ephemeris gpsEphem.json] [--cpu-acq] [--json]

    <recording.bin>   what gpsbin.py records and streamData reads (gpsrecv.py:162-173):
                      little-endian uint16 per sample, low byte I, high byte Q, 2.048 Msps
                      (1 min = 245.76 MB).  `data/test.bin` of the reference when it is there.

Flow: ingest.read_raw_blocks (the file loop of streamData, START_STREAM honoured) ->
pipeline.Receiver(raw_u8=True).feed (cold sweep, channel selection, tracking, hand-off
datagrams; the u8 decode happens inside the GPU kernels) -> position.PositionSolver.feed
(prepCodePhase ... leastSquaresPos, the evaluation side's data path) -> fixes; the mean of
the second half of the fixes is printed as latitude / longitude / height.  --save-pickle
writes the datagram list exactly as SAVE_PICKLE does (gpsrecv.py:205-212): an unmodified
gpseval.py replays it with LOAD_PICKLE.  --cpu-acq times BASELINE configs[0] beside it: the
cold acquisition of the reference's numpy path (oracle restatement, test infrastructure) on
the first five blocks of the same file, on one host core.

There is no recording in this repository (data/test.bin is absent from the reference
checkout, SURVEY F2, and too short for a fix even upstream): tests/test_run_file.py writes a
synthetic one with gpsmi.synth_nav and checks the fix this command prints."""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gps-sdr-receiver_amd'))


def cpu_cold_acquisition(path, n_blocks=5):
    """BASELINE configs[0]: 31 SV x 50 bins x 4 ms first-hit sweep of the reference's numpy path
    on the first blocks of the file, one core (the oracle: checker / CPU baseline only)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import gps_oracle as orc
    from gpsmi import ingest, synth
    p = orc.Params()
    t = orc.sec_time(p)
    spectra = {s: orc.fft_cacode(s) for s in range(2, 33)}
    blocks = []
    for raw in ingest.read_raw_blocks(path):
        blocks.append(synth.raw_to_c64(raw))
        if len(blocks) == n_blocks:
            break
    t0 = time.perf_counter()
    sat_lst, found, freq = list(range(2, 33)), [], p.min_freq
    for blk in blocks:
        ready, freq, found = orc.sweep_all_sats(blk, freq, sat_lst, found, p.it_sweep_all, p, spectra, t)
    return {'wall_ms': round((time.perf_counter() - t0) * 1e3, 1), 'cores': 1, 'blocks': len(blocks),
            'found': [(int(s), float(f), int(d)) for _, s, f, d in found]}


def run(path, seconds=None, start_stream=0, save_pickle=None, ephemerides=None, cpu_acq=False, report_lag=16):
    from gpsmi import ingest, position as P
    from gpsmi.engine import Config
    from gpsmi.pipeline import Receiver, save_results
    cfg = Config()
    # (a recording: the datagrams may come out `report_lag` blocks behind the block they belong to, the
    # reader then runs ahead of the GPU instead of stalling it once a second -- pipeline.Receiver)
    rx = Receiver(cfg, raw_u8=True, report_lag=report_lag)
    solver = P.PositionSolver(cfg.code_samples, cfg.n_cyc, ephemerides=ephemerides)
    max_blocks = None if seconds is None else int(seconds * 1000 // cfg.n_cyc)
    fixes, n_dg, n_blocks, found = [], 0, 0, None
    t0 = time.perf_counter()
    for raw in ingest.read_raw_blocks(path, cfg.ngps, start_stream):
        rx.feed(raw)
        n_blocks += 1
        if found is None and not rx.sweep_all_freq:
            found = list(rx.found_sats)
        for dg in rx.result_list[n_dg:]:           # (every datagram, in order: RESULT_LIST)
            n_dg += 1
            fixes += solver.feed(pickle.loads(dg))
        if max_blocks is not None and n_blocks >= max_blocks:
            break
    rx.drain()
    for dg in rx.result_list[n_dg:]:
        n_dg += 1
        fixes += solver.feed(pickle.loads(dg))
    wall = time.perf_counter() - t0
    if save_pickle:
        save_results(save_pickle, rx.result_list)
    sats = sorted(rx.act_sat_set)
    rx.close()
    out = {'file': os.path.basename(path), 'blocks': n_blocks, 'signal_s': round(n_blocks * cfg.n_cyc / 1000.0, 3),
           'wall_s': round(wall, 3), 'x_realtime': round(n_blocks * cfg.n_cyc / 1000.0 / wall, 1) if wall else None,
           'acquired': [(int(s), float(f), int(d)) for _, s, f, d in (found or [])],
           'tracked': sats, 'datagrams': n_dg, 'fixes': len(fixes),
           'ephemerides_decoded': sorted(s for s, o in solver.orbits.items() if o.data.ephem_ok)}
    if fixes:
        xyz = np.array([f[1:] for f in fixes])
        late = xyz[len(xyz) // 2:]
        mean = late.mean(axis=0)
        lat, lon, alt = P.ecef_to_geo(mean)
        out['position'] = {'lat_deg': round(float(lat), 6), 'lon_deg': round(float(lon), 6),
                           'height_m': round(float(alt), 1), 'ecef_m': [round(float(v), 2) for v in mean],
                           'from': f'mean of the last {len(late)} fixes',
                           'sd_of_mean_m': round(float(np.linalg.norm(late.std(axis=0)) / np.sqrt(len(late))), 2)}
    if cpu_acq:
        out['cpu_cold_acquisition'] = cpu_cold_acquisition(path)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('recording')
    ap.add_argument('--seconds', type=float, default=None, help='stop after this much signal')
    ap.add_argument('--start-stream', type=int, default=0, help='START_STREAM (gpsglob.py:16)')
    ap.add_argument('--save-pickle', default=None, help='write the datagram list as SAVE_PICKLE does')
    ap.add_argument('--ephemeris', default=None, help='a saved EPHEM_FILE (gpsEphem.json) to start from')
    ap.add_argument('--cpu-acq', action='store_true', help='time the CPU cold acquisition (configs[0]) too')
    ap.add_argument('--report-lag', type=int, default=16,
                    help='blocks a datagram may trail the block it belongs to (0: none, as a live receiver)')
    ap.add_argument('--json', action='store_true', help='one JSON line instead of text')
    a = ap.parse_args()
    eph = None
    if a.ephemeris:
        with open(a.ephemeris) as f:
            eph = {int(k): v for k, v in json.load(f).items()}
    out = run(a.recording, a.seconds, a.start_stream, a.save_pickle, eph, a.cpu_acq, a.report_lag)
    if a.json:
        print(json.dumps(out))
        return
    print(f"{out['file']}: {out['blocks']} blocks = {out['signal_s']} s of signal in {out['wall_s']} s "
          f"({out['x_realtime']} x real time)")
    print(f"acquired {len(out['acquired'])} satellites: " + ', '.join(f'PRN {s} {f:+.0f} Hz' for s, f, _ in out['acquired']))
    print(f"tracked {out['tracked']}; {out['datagrams']} datagrams; ephemerides decoded for {out['ephemerides_decoded']}")
    if 'position' in out:
        p = out['position']
        print(f"{out['fixes']} fixes; {p['from']}: {p['lat_deg']:.6f} N {p['lon_deg']:.6f} E, height {p['height_m']:.1f} m "
              f"(SD of mean {p['sd_of_mean_m']} m)")
    else:
        print('no position fix (too little signal for subframes 1-3 of four satellites, or no ephemerides)')
    if 'cpu_cold_acquisition' in out:
        c = out['cpu_cold_acquisition']
        print(f"CPU cold acquisition (configs[0], numpy path, {c['cores']} core): {c['wall_ms']} ms, {len(c['found'])} satellites")


if __name__ == '__main__':
    main()
