"""Pin the oracle (oracle/gps_oracle.py) to the real reference's frozen outputs
(tests/golden/ref_*.npz, made by oracle/make_golden.py from
gpsrecv.sweepAllSats / gpslib.SatStream.process).  CPU only."""
import hashlib

import numpy as np
import pytest

import gps_oracle as orc
from conftest import scene_blocks, scene_for

CFG = {'default': dict(code_samples=2048, n_cyc=32),
       'hirate': dict(code_samples=16368, n_cyc=8)}


def _golden(cfg, golden_default, golden_hirate):
    return golden_default if cfg == 'default' else golden_hirate


@pytest.mark.parametrize('cfg', ['default', 'hirate'])
def test_scene_generator_is_stable(cfg, golden_default, golden_hirate):
    g = _golden(cfg, golden_default, golden_hirate)
    sc = scene_for(cfg)
    n = 5 + g['trk_delay'].shape[1]
    h = hashlib.sha256()
    for b in range(n):
        h.update(sc.block_raw(b).tobytes())
    assert h.hexdigest() == str(g['iq_sha256'])


@pytest.mark.parametrize('cfg', ['default', 'hirate'])
def test_sweep_all_sats_first_hit_loop(cfg, golden_default, golden_hirate):
    """gpsrecv.py:241-274 over 5 blocks with reference defaults."""
    g = _golden(cfg, golden_default, golden_hirate)
    p = orc.Params(**CFG[cfg])
    t = orc.sec_time(p)
    spectra = {s: orc.fft_cacode(s, p.code_samples) for s in range(2, 33)}
    sat_lst, found, freq = list(range(2, 33)), [], p.min_freq
    blocks = scene_blocks(cfg, 0, 5)
    for b in range(5):
        ready, freq, found = orc.sweep_all_sats(blocks[b], freq, sat_lst, found,
                                                p.it_sweep_all, p, spectra, t)
        assert (float(ready), freq, len(found)) == tuple(g['sweep_calls'][b])
    ref = g['sweep_found']
    mine = np.array(found, dtype=np.float64)
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, 1:], ref[:, 1:])          # sv, freq, delay
    assert np.array_equal(mine[:, 0], ref[:, 0])            # same expressions


def _check_table(t, g, prefix):
    assert np.array_equal(t['argmax'], g[prefix + 'argmax'])
    for k in ('peak', 'mean', 'std'):
        assert np.array_equal(t[k], g[prefix + k]), k


def test_acq_surface_reference_grid(golden_default):
    """50 bins x 31 SV x 4 ms, bins 10b..10b+9 on block b."""
    p = orc.Params()
    prns = list(range(2, 33))
    spectra = {s: orc.fft_cacode(s) for s in prns}
    blocks = scene_blocks('default', 0, 5)
    f50 = [p.min_freq + p.step_freq * i for i in range(50)]
    for b in range(5):
        t = orc.acq_table(blocks[b], f50[10 * b:10 * b + 10], prns, 4, p,
                          spectra=spectra)
        for k in ('argmax', 'peak', 'mean', 'std'):
            assert np.array_equal(t[k], golden_default[f'ref50_{k}_{b}'])


@pytest.mark.parametrize('cfg', ['default', 'hirate'])
def test_acq_surface_cfg2(cfg, golden_default, golden_hirate):
    """32 SV x 41 bins x 1 ms (BASELINE config 2)."""
    g = _golden(cfg, golden_default, golden_hirate)
    p = orc.Params(**CFG[cfg])
    f41 = [-5000.0 + 250.0 * i for i in range(41)]
    t = orc.acq_table(scene_blocks(cfg, 0, 1)[0], f41, list(range(1, 33)), 1, p)
    _check_table(t, g, 'cfg2_')


def test_acq_surface_cfg4(golden_default):
    """32 SV x 201 bins x 10 ms (BASELINE config 4)."""
    f201 = [-5000.0 + 50.0 * i for i in range(201)]
    t = orc.acq_table(scene_blocks('default', 0, 1)[0], f201,
                      list(range(1, 33)), 10, orc.Params())
    _check_table(t, golden_default, 'cfg4_')


@pytest.mark.parametrize('cfg', ['default', 'hirate'])
def test_satstream_process(cfg, golden_default, golden_hirate):
    """gpslib.py:1141-1210 closed loop, every recorded quantity of every block."""
    g = _golden(cfg, golden_default, golden_hirate)
    p = orc.Params(**CFG[cfg])
    nch, nb = g['trk_delay'].shape
    blocks = scene_blocks(cfg, 5, nb)
    for c in range(nch):
        sv, f0, d0 = g['trk_init'][c]
        ss = orc.SatStream(int(sv), float(f0), p, delay=int(d0))
        for i in range(nb):
            smp = np.int64((5 + i + 1) * p.ngps)
            sw, frames, co_ph, (cq, cl) = ss.process(blocks[i], smp)
            L = ss.last
            nd = int(g['trk_n_dumps'][c, i])
            assert len(L['dumps']) == nd
            assert np.array_equal(L['dumps'], g['trk_dumps'][c, i, :nd])
            assert np.array_equal(L['epl'], g['trk_epl'][c, i])
            exact = dict(delay=ss.delay, code_phase=co_ph, norm=ss.max_corr,
                         freq=ss.freq, phase=ss.phase, locked=ss.phase_locked,
                         nps=len(ss.prev_samples), std_dev=ss.std_dev,
                         amplitude=ss.amplitude, corr_q=cq, corr_l=cl,
                         mx=L['mx'], corr_mean=L['corr_mean'],
                         corr_std=L['corr_std'], sweep=sw, ms_time=ss.ms_time,
                         n_edges=len(ss.edges))
            for k, v in exact.items():
                assert float(v) == g['trk_' + k][c, i], (c, i, k)


RESWEEP_CASES = ('two_calls', 'one_call', 'no_signal')


@pytest.mark.parametrize('case', RESWEEP_CASES)
def test_resweep_matches_reference(case):
    """Per-channel re-acquisition (gpslib.py:1110-1120 initSweep / restoreFreq, :1153-1173
    the sweep branch of process, :1350-1380 getCorrMax / sweepFrequency) against what the
    real SatStream returned (tests/golden/ref_resweep.npz): a hit in the second 40-bin
    call, a hit in the first, and the no-signal run-off to +10800 Hz with restoreFreq."""
    from conftest import load_golden
    g = load_golden('ref_resweep.npz')
    sv, f0, d0, n_before, n_after = g[case + '_init']
    p = orc.Params()
    nb = int(n_before + n_after)
    first = int(g['first_block'])
    blocks = scene_blocks('default', first, nb)
    ss = orc.SatStream(int(sv), float(f0), p, delay=int(d0))
    for i in range(nb):
        smp = np.int64((first + i + 1) * p.ngps)
        sw, fl, co_ph, (cq, cl) = ss.process(blocks[i], smp, sweep=(i == int(n_before)))
        got = dict(sweep=sw, freq=float(ss.freq), freq_is_f32=isinstance(ss.freq, np.float32),
                   max_corr=ss.max_corr, delay=ss.delay, code_phase=co_ph, corr_q=cq,
                   corr_l=cl, phase=float(ss.phase), locked=ss.phase_locked,
                   df_len=len(ss.df), n_frames=len(fl),
                   swp_reported=(fl[0]['SWP'] if fl else -1))
        for k, v in got.items():
            assert float(v) == g[f'{case}_{k}'][i], (case, i, k)
    # the cases really are what their names say
    sweeps = g[case + '_sweep'].astype(int).tolist()
    assert sum(sweeps) == (0 if case == 'one_call' else 1)


def test_fit_code_phase_wraps():
    """gpslib.py:1268-1290 circular neighbours at both ends."""
    corr = np.full(16, 1.0)
    corr[0], corr[15], corr[1] = 5.0, 3.0, 2.0
    v = orc.fit_code_phase(corr, 0)
    assert -0.5 < v < 0
    corr = np.full(16, 1.0)
    corr[15], corr[14], corr[0] = 5.0, 2.0, 3.0
    v = orc.fit_code_phase(corr, 15)
    assert 15 < v < 15.5


def test_get_new_sats():
    """gpsrecv.py:423-440"""
    found = [(20.0, 5, 0.0, 1), (19.0, 7, 0.0, 2), (18.0, 9, 0.0, 3)]
    dele, new = orc.get_new_sats(set(), found, {}, 2)
    assert dele == set() and new == {5, 7}
    dele, new = orc.get_new_sats({5, 11}, found, {11: (0.5, 1.0), 5: (-1, -1)},
                                 2)
    assert dele == set() and new == set()      # 11 kept, 5 is the one refill
    dele, new = orc.get_new_sats({3}, found, {3: (-1.0, -1.0)}, 11)
    assert dele == {3} and new == {5, 7, 9}


def test_get_new_sats_matches_reference():
    """tests/golden/ref_newsats.json: gpsrecv.getNewSats of the reference itself
    (gpsrecv.py:423-440) on 60 seeded (active set, found list, quality table) inputs; the
    oracle's and the product's restatements must return the same sets."""
    import json
    import os
    from conftest import GOLDEN
    from gpsmi.acquisition import getNewSats
    with open(os.path.join(GOLDEN, 'ref_newsats.json')) as f:
        fix = json.load(f)
    assert fix['max_sat'] == 11 and len(fix['cases']) == 60
    n_new = 0
    for c in fix['cases']:
        found = [tuple(e) for e in c['found']]
        cpq = {int(s): tuple(v) for s, v in c['cpq'].items()}
        for fn in (orc.get_new_sats, getNewSats):
            dele, new = fn(set(c['act']), list(found), dict(cpq), fix['max_sat'])
            assert sorted(dele) == c['delete'] and sorted(new) == c['new']
        n_new += len(c['new'])
    assert n_new > 100
