"""Position fix: gpsmi.position against what the reference's own
SatPos / SatOrbit / leastSquaresPos4 / ecefToGeo returned on a constructed scene
(tests/golden/ref_position.npz, made by oracle/make_golden.py position), and the
end-to-end property: noise-free observations of a known receiver position give it back.
The gpseval-level glue (code-phase clean-up, grouping) has no importable reference
here; its parity is unpinned and it is covered through the end-to-end fix and the
behavioural cases at the bottom."""
import json

import numpy as np
import pytest

from conftest import load_golden
from gpsmi import position as P


@pytest.fixture(scope='module')
def g():
    z = load_golden('ref_position.npz')
    d = {k: z[k] for k in ('truth', 'geo', 'ecf', 'orb')}
    d['ephs'] = {int(k): v for k, v in json.loads(str(z['ephs'])).items()}
    d['datagrams'] = [(sk, fl, {int(k): [tuple(e) for e in v] for k, v in co.items()})
                      for sk, fl, co in json.loads(str(z['datagrams']))]
    d['cpl'] = [{int(k): [tuple(e) for e in v] for k, v in c.items()}
                for c in json.loads(str(z['cpl']))]
    d['tuples'] = json.loads(str(z['tuples']))
    d['fixes'] = json.loads(str(z['fixes']))
    return d


def test_broadcast_orbit_matches_reference(g):
    for row in g['orb']:
        s, tow, DT = int(row[0]), int(row[1]), float(row[2])
        for rel, ref in ((True, row[3:7]), (False, row[7:11])):
            got = P.sat_ecef(tow, g['ephs'][s], DT=DT, rel_corr=rel)
            np.testing.assert_allclose(got[:3], ref[:3], rtol=0, atol=1e-6)   # metres
            assert abs(got[3] - ref[3]) < 1e-15                               # seconds


def test_geodetic_conversions_match_reference(g):
    for x, y, z, lat, lon, alt in g['geo']:
        got = P.ecef_to_geo((x, y, z))
        np.testing.assert_allclose(got, (lat, lon, alt), rtol=0, atol=1e-9)
    for lat, lon, alt, x, y, z in g['ecf']:
        np.testing.assert_allclose(P.geo_to_ecef(lat, lon, alt), (x, y, z), rtol=0, atol=1e-8)
    lat, lon, alt = P.ecef_to_geo(g['truth'])
    assert abs(lat - 49.082961) < 1e-9 and abs(lon - 8.307581) < 1e-9 and abs(alt - 160) < 1e-6


def test_code_phase_tuples_match_reference(g):
    """OrbitTracker on the subframes and the cleaned lists the reference's SatOrbit got:
    the same (sat, TOW, x, y, z, sample time, week, cycle, std) tuples."""
    orbits = {s: P.OrbitTracker(s) for s in g['ephs']}
    n = 0
    for (_, frames, _), cpl, ref in zip(g['datagrams'], g['cpl'], g['tuples']):
        for sf in frames:
            orbits[sf['SAT']].read_frame(sf)
        got = []
        for s in cpl:
            got += orbits[s].eval_code_phase(list(cpl[s]))
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert (a[0], a[1], a[6], a[7]) == (b[0], b[1], b[6], b[7])
            np.testing.assert_allclose(a[2:5], b[2:5], rtol=0, atol=1e-6)
            assert abs(a[5] - b[5]) < 1e-12 and abs(a[8] - b[8]) < 1e-9
            n += 1
    assert n > 5000


def test_fixes_match_reference_and_truth(g):
    solver = P.PositionSolver()
    t_first, all_fix = None, []
    for dg, ref in zip(g['datagrams'], g['fixes']):
        fixes = solver.feed(dg)
        assert len(fixes) == len(ref)
        for f, r in zip(fixes, ref):
            np.testing.assert_allclose(f[1:], r[3:6], rtol=0, atol=1e-5)      # 10 um
        all_fix += fixes
    assert len(all_fix) > 500 and not solver.fail_lst
    err = np.linalg.norm(np.array([f[1:] for f in all_fix]) - g['truth'], axis=1)
    # the code-phase slope is averaged over five lists before it is used: the first
    # seconds are off by up to (16 code periods) x (drift per ms)
    assert err.max() < 25.0
    assert err[-200:].max() < 0.10                    # centimetres once the slope is known
    lat, lon, alt = P.ecef_to_geo(np.mean([f[1:] for f in all_fix[-200:]], axis=0))
    assert abs(lat - 49.082961) < 1e-6 and abs(lon - 8.307581) < 1e-6 and abs(alt - 160) < 0.10


def test_least_squares_recovers_receiver_from_exact_ranges():
    rng = np.random.default_rng(5)
    rec = np.array(P.geo_to_ecef(-12.3, 130.9, 55.0))
    up = rec / np.linalg.norm(rec)
    sats = []
    while len(sats) < 6:
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        if d.dot(up) > 0.3:
            sats.append(rec + d * rng.uniform(2.0e7, 2.5e7))
    sats = np.array(sats).T
    v = np.array([-rec[1], rec[0], 0.0]) * P.OMEGA_EARTH
    tau = np.full(6, 0.07)
    for _ in range(5):
        tau = np.linalg.norm(sats - rec[:, None] - np.outer(v, tau), axis=0) / P.GPS_C
    got, resid, ranges, _ = P.least_squares_pos4(sats, 1000.0 + tau, max_it=15)
    assert np.linalg.norm(got[1:] - rec) < 1e-4 and resid[-1] < 1e-6
    np.testing.assert_allclose(ranges, tau * P.GPS_C, atol=1e-3)


# ---- gpseval-level behaviour (parity unpinned: restated from src/gpseval.py:377-457)

def test_code_phase_rollover_inside_a_list_is_unwrapped():
    s = P.PositionSolver()
    lst = [(n, (2040.0 + 0.9 * n) % 2048) for n in range(1, 20)]
    out = s.prep_code_phase({7: lst})[7]
    cps = np.array([c for _, c in out])
    assert np.all(np.diff(cps) > 0.8) and np.all(np.diff(cps) < 1.0)          # one straight line
    assert cps[-1] > 2048


def test_short_lists_are_ignored_and_correlated_jumps_reset_the_reference():
    s = P.PositionSolver()
    for sat in (3, 9, 17, 21):
        s.orbit(sat)
    assert s.prep_code_phase({3: [(1, 100.0), (2, 100.1)]}) == {}              # < N_CYC/4 entries
    good = [(n, 500.0 + 0.1 * n) for n in range(1, 12)]
    jump = [(n, 500.0 + 0.1 * n + (40.0 if n >= 6 else 0.0)) for n in range(1, 12)]
    out = s.prep_code_phase({3: jump, 9: jump, 17: jump, 21: good})
    assert s.n_phase_err == 1
    assert all(out[sat] == [(6, None)] for sat in (3, 9, 17, 21))
    # the tracker then drops its time reference and ignores older frames
    o = s.orbit(3)
    o.ref_time = (5, 12345)
    assert o.eval_code_phase(out[3]) == [] and o.ref_time is None and o.phase_err == [6]
    assert o.read_frame({'SAT': 3, 'ID': 4, 'tow': 7, 'ST': 2 * 65536}) == P.FLAWED_FRAME
