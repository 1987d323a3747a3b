"""Whole chain on a signal that carries real subframes: synthetic satellites
whose 50 bit/s data are IS-GPS-200 encoded frames -> tracking -> edge list ->
20-ms bits -> preamble search -> subframe fields.  CPU: the oracle recovers the
frames that were sent.  GPU: the worker-pool mirror returns the same frames as
the oracle, in the same blocks, with the same sample-time stamps."""
import numpy as np
import pytest

import gps_oracle as orc
from gpsmi import navbits as nb, synth

N_BLOCKS = 420          # 13.4 s: lock-in + two whole 6-s subframes
_CACHE = {}


def _message(seed):
    """Five chained subframes (IDs 1..5) with increasing tow, as 0/1 bits."""
    rng = np.random.default_rng(seed)
    out, ds29, ds30 = [], 0, 0
    for k in range(5):
        w = rng.integers(0, 2, (10, 24)).astype(np.int8)
        w[0, :8] = nb.PREAMBLE_BITS
        tow = 1000 + k
        w[1, :17] = [(tow >> (16 - i)) & 1 for i in range(17)]
        sid = k + 1
        w[1, 19:22] = [(sid >> 2) & 1, (sid >> 1) & 1, sid & 1]
        w[1, 22:24] = 0
        f = nb.encode_subframe(w, ds29, ds30)
        ds29, ds30 = int(f[298]), int(f[299])
        out.append(f)
    return np.concatenate(out)


def _scene():
    sats = [synth.Sat(prn=9, doppler=1234.5, delay=700.3, amp=0.09, phase0=0.4,
                      nav_bits=_message(1)),
            synth.Sat(prn=23, doppler=-2711.0, delay=1501.8, amp=0.09, phase0=2.0,
                      nav_bits=_message(2))]
    return synth.Scene(sats=sats, seed=21)


def _blocks():
    if 'b' not in _CACHE:
        sc = _scene()
        _CACHE['b'] = [sc.block(i) for i in range(N_BLOCKS)]
    return _CACHE['b']


def _oracle_frames():
    if 'o' not in _CACHE:
        out = {}
        for prn, f0, d0 in ((9, 1200.0, 700), (23, -2800.0, 1502)):
            ss = orc.SatStream(prn, f0, orc.Params(), delay=d0)
            got = []
            for i, blk in enumerate(_blocks()):
                _, frames, _, _ = ss.process(blk, np.int64((i + 1) * 65536))
                got += [(i, f) for f in frames if 'ID' in f]
            out[prn] = got
        _CACHE['o'] = out
    return _CACHE['o']


def test_oracle_recovers_the_transmitted_subframes():
    for prn, got in _oracle_frames().items():
        ids = [f['ID'] for _, f in got]
        tows = [f['tow'] for _, f in got]
        assert len(got) >= 1, prn
        assert all(1000 <= t <= 1004 for t in tows)
        assert all(i == t - 999 for i, t in zip(ids, tows))       # ID k+1 carries tow 1000+k
        assert tows == sorted(tows)
        for _, f in got:
            assert f['SAT'] == prn and f['SWP'] is False and f['ST'] > 0


@pytest.mark.gpu
def test_gpu_pool_returns_the_same_subframes():
    from gpsmi import receiver as R
    ref = _oracle_frames()
    found = [(20.0, 9, 1200.0, 700), (19.0, 23, -2800.0, 1502)]
    pool, n, worker = R.initMultiProcPool(2)
    worker, act = R.initPoolStreams(pool, n, worker, set(), {9, 23}, found)
    got = {9: [], 23: []}
    for i, blk in enumerate(_blocks()):
        for sw, sat, frames, cp, cq in R.satCalc(act, pool, worker, blk,
                                                 np.int64((i + 1) * 65536)):
            got[sat] += [(i, f) for f in frames if 'ID' in f]
    R.closeMultiProcPool(pool)
    for prn in (9, 23):
        assert len(got[prn]) == len(ref[prn]) >= 1
        for (ia, a), (ib, b) in zip(got[prn], ref[prn]):
            assert ia == ib                                        # same block
            assert list(a.keys()) == list(b.keys())
            for k in a:
                if k in ('AMP', 'CRM', 'FRQ'):
                    assert abs(float(a[k]) - float(b[k])) < 5e-2 * max(1.0, abs(float(b[k])))
                else:
                    assert a[k] == b[k], k                         # ID, tow, fields, ST
