"""Host control logic of gpsmi.receiver.HostChannel fed with the reference's own
per-block numbers (from the golden fixture) in place of engine output: edge
list, millisecond counter, correlation-quality averages and report frames must
come out exactly as gpslib.SatStream produced them (gpslib.py:1124-1138,
:1331-1339, :1394-1398, :1421-1436, :1451-1492).  No GPU involved."""
import numpy as np

import gps_oracle as orc
from gpsmi._lib import OUT_DTYPE
from gpsmi.engine import Config
from gpsmi.receiver import HostChannel, fit_code_phase


_EDGE_CACHE = {}


def edge_fields(g):
    """What the device epilogue's edge scan reports for the fixture's dumps -- decodeData's rule
    (gpslib.py:1394-1398, :1421-1436) on the float32 dumps, block after block per channel:
    (edge_mask [nch, nb] as Python ints, edge_sign0, ms_count).  The kernel's scan is checked
    against the reference's MS_TIME / edge counts on the GPU (tests/test_gpu_receiver.py)."""
    key = id(g)
    if key in _EDGE_CACHE:
        return _EDGE_CACHE[key]
    nch, nb = g['trk_delay'].shape
    mask = [[0] * nb for _ in range(nch)]
    sign0 = np.zeros((nch, nb), np.int32)
    msc = np.zeros((nch, nb), np.int32)
    for c in range(nch):
        e0, prev_sign, prev_signal, std_dev, locked = 0, 0, np.float32(0), np.float32(0.005), False
        for i in range(nb):
            nd = int(g['trk_n_dumps'][c, i])
            if locked:
                thr = np.float32(3) * std_dev
                for k in range(nd):
                    re = np.float32(g['trk_dumps'][c, i, k].real)
                    sgn = np.sign(re)
                    if e0 == 0:
                        e0 = prev_sign = sgn
                        if sgn != 0 and sign0[c, i] == 0:
                            sign0[c, i] = int(sgn)
                    elif sgn != prev_sign and prev_sign * prev_signal > 0 and abs(re - prev_signal) > thr:
                        mask[c][i] |= 1 << k
                        prev_sign = sgn
                    prev_signal = re
                msc[c, i] = nd
            std_dev = np.float32(g['trk_std_dev'][c, i])
            locked = bool(g['trk_locked'][c, i])
    _EDGE_CACHE[key] = (mask, sign0, msc)
    return _EDGE_CACHE[key]


def _record(g, c, i):
    r = np.zeros(1, dtype=OUT_DTYPE)[0]
    mask, sign0, msc = edge_fields(g)
    r['edge_mask'] = mask[c][i] & 0xFFFFFFFF
    r['edge_mask_hi'] = mask[c][i] >> 32
    r['edge_sign0'] = sign0[c, i]
    r['ms_count'] = msc[c, i]
    nps_prev = int(g['trk_nps'][c, i - 1]) if i else 0
    n1 = nps_prev + int(g['trk_delay'][c, i])
    r['first_len'] = n1 if n1 else 2048
    nd = int(g['trk_n_dumps'][c, i])
    d = g['trk_dumps'][c, i, :nd]
    r['n_dumps'] = nd
    r['dumps'][0:2 * nd:2] = d.real
    r['dumps'][1:2 * nd:2] = d.imag
    r['code_phase'] = g['trk_code_phase'][c, i]
    r['delay_used'] = int(g['trk_delay'][c, i])
    r['std_dev'] = g['trk_std_dev'][c, i]
    r['amplitude'] = g['trk_amplitude'][c, i]
    r['norm_max_corr'] = g['trk_norm'][c, i]
    r['nps'] = int(g['trk_nps'][c, i])
    r['phase_locked'] = int(g['trk_locked'][c, i])
    r['freq'] = g['trk_freq'][c, i]
    return r


def test_host_channel_reproduces_reference_control_state(golden_default):
    g = golden_default
    cfg = Config()
    nch, nb = g['trk_delay'].shape
    frames_ref = eval(str(g['trk_frames_repr']), {'np': np})
    for c in range(nch):
        sv, f0, d0 = g['trk_init'][c]
        hc = HostChannel(int(sv), float(f0), int(d0), cfg)
        for i in range(nb):
            smp = np.int64((5 + i + 1) * cfg.ngps)
            stream_no = smp // cfg.ngps
            if stream_no - 1 != hc.PREV_STREAM_NO:
                hc.erasePrevData()
            hc.PREV_STREAM_NO = stream_no
            sweep, frames, cp = hc.absorb(_record(g, c, i), smp)
            where = (c, i)
            assert not sweep
            assert hc.MS_TIME == g['trk_ms_time'][c, i], where
            assert len(hc.EDGES) == g['trk_n_edges'][c, i], where
            assert float(hc.CORR_Q) == g['trk_corr_q'][c, i], where
            assert float(hc.CORR_L) == g['trk_corr_l'][c, i], where
            ref = [dict(f) for cc, ii, f in frames_ref if (cc, ii) == (c, i)]
            assert len(frames) == len(ref), where
            for a, b in zip(frames, ref):
                assert a['SAT'] == b['SAT'] and a['SWP'] == b['SWP']
                assert float(a['AMP']) == float(b['AMP'])
                # the engine record carries normMaxCorr as float32
                assert float(a['CRM']) == float(np.float32(b['CRM']))
                assert float(a['FRQ']) == float(b['FRQ'])


def test_gap_resets_carry_and_edges():
    hc = HostChannel(5, 100.0, 7, Config())
    hc.EDGES = [1.0, (3, 4), (23, 9)]
    hc.nps = 100
    hc.erasePrevData()
    assert hc.EDGES == [0] and hc.nps == 0 and len(hc.GPSBITS) == 0


def test_logical_bits_matches_oracle():
    """gpslib.py:1465-1492: 20-ms slicing with the r > 17 rounding rule."""
    edges = [1.0, (5, 100), (25, 200), (64, 300), (124, 400), (143, 500)]
    hc = HostChannel(5, 0.0, 0, Config())
    hc.EDGES = list(edges)
    ss = orc.SatStream(5, 0.0)
    ss.edges = list(edges)
    b1, s1 = hc.logicalBits()
    b2, s2 = ss.logical_bits()
    assert np.array_equal(b1, b2) and np.array_equal(s1, s2)
    assert list(b1) == [1, -1, -1, 1, 1, 1, -1]
    assert hc.EDGES == ss.edges == [1.0, (143, 500)]


def test_fit_code_phase_equals_oracle():
    rng = np.random.default_rng(0)
    for _ in range(50):
        corr = rng.uniform(0.1, 1.0, 64)
        mx = int(np.argmax(corr))
        n = len(corr)
        a = fit_code_phase(corr[(mx - 1) % n], corr[mx], corr[(mx + 1) % n], mx)
        assert a == orc.fit_code_phase(corr, mx)
