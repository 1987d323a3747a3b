"""GPU parity: acquisition through the C ABI vs the reference fixtures and the
oracle.  Bars: argmax bit-exact, acquired (SV, delay, bin)
bit-exact, normMaxCorr / peak / mean / std within rtol 1e-4."""
import numpy as np
import pytest

import gps_oracle as orc
from conftest import scene_blocks

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope='module')
def acq():
    from gpsmi.acquisition import Acquisition
    a = Acquisition()
    yield a
    a.engine.close()


def _check(table, g, prefix, rtol=RTOL):
    assert np.array_equal(table['argmax'], g[prefix + 'argmax'])
    for k in ('peak', 'mean', 'std'):
        np.testing.assert_allclose(table[k], g[prefix + k], rtol=rtol)
    nmc = (table['peak'].astype(np.float64) - table['mean']) / table['std']
    ref = (g[prefix + 'peak'] - g[prefix + 'mean']) / g[prefix + 'std']
    np.testing.assert_allclose(nmc, ref, rtol=rtol)
    assert np.array_equal(nmc > 8, ref > 8)          # no threshold flips


def test_cfg2_surface(acq, golden_default):
    """32 SV x 41 bins x 1 ms on block 0 (BASELINE config 2)."""
    f41 = [-5000.0 + 250.0 * i for i in range(41)]
    t = acq.search_table(scene_blocks('default', 0, 1)[0], list(range(1, 33)),
                         f41, 1)
    _check(t, golden_default, 'cfg2_')


def test_cfg4_surface(acq, golden_default):
    """32 SV x 201 bins x 10 ms coherent (BASELINE config 4, one GPU)."""
    f201 = [-5000.0 + 50.0 * i for i in range(201)]
    t = acq.search_table(scene_blocks('default', 0, 1)[0], list(range(1, 33)),
                         f201, 10)
    _check(t, golden_default, 'cfg4_')


def test_reference_grid_surface(acq, golden_default):
    """Reference defaults: 50 bins x 31 SV x 4 ms, 10 bins per 32-ms block."""
    f50 = [-5000.0 + 200 * i for i in range(50)]
    blocks = scene_blocks('default', 0, 5)
    for b in range(5):
        t = acq.search_table(blocks[b], list(range(2, 33)),
                             f50[10 * b:10 * b + 10], 4)
        assert np.array_equal(t['argmax'], golden_default[f'ref50_argmax_{b}'])
        for k in ('peak', 'mean', 'std'):
            np.testing.assert_allclose(t[k], golden_default[f'ref50_{k}_{b}'],
                                       rtol=RTOL)


def test_sweep_all_sats_loop(acq, golden_default):
    """The reference's own first-hit loop over 5 blocks (gpsrecv.py:241-274):
    identical (SV, Doppler, delay) list in identical order."""
    g = golden_default
    sat_lst, found, freq = list(range(2, 33)), [], acq.cfg.min_freq
    blocks = scene_blocks('default', 0, 5)
    for b in range(5):
        ready, freq, found = acq.sweepAllSats(blocks[b], freq, sat_lst, found,
                                              itSweep=acq.cfg.it_sweep_all)
        assert (float(ready), freq, len(found)) == tuple(g['sweep_calls'][b])
    mine = np.array(found, dtype=np.float64)
    ref = g['sweep_found']
    assert np.array_equal(mine[:, 1:], ref[:, 1:])
    np.testing.assert_allclose(mine[:, 0], ref[:, 0], rtol=RTOL)
    assert len(sat_lst) == 31 - len(found)


def test_matches_oracle_on_fresh_scene(acq):
    """A scene no fixture holds, GPU vs oracle directly."""
    from gpsmi import synth
    sc = synth.default_scene(6, seed=123)
    data = sc.block(0)
    freqs = [-5000.0 + 500.0 * i for i in range(21)]
    prns = [1, 3, 8, 13, 22, 31] + [s.prn for s in sc.sats]
    o = orc.acq_table(data, freqs, prns, 2, orc.Params())
    t = acq.search_table(data, prns, freqs, 2)
    assert np.array_equal(t['argmax'], o['argmax'])
    for k in ('peak', 'mean', 'std'):
        np.testing.assert_allclose(t[k], o[k], rtol=RTOL)


def test_noise_free_delta_peak(acq):
    """A clean, full-scale replica at a known delay and Doppler 0: the peak
    sits exactly there, for every delay edge case (0, 1, 2047)."""
    from gpsmi import codes
    rep = codes.code_replica(5)
    for d in (0, 1, 1024, 2047):
        x = (0.5 * np.roll(rep, d)).astype(np.complex64)
        t = acq.search_table(np.tile(x, 4), [5, 6], [0.0], 4)
        assert t['argmax'][0, 0] == d
        n5 = (t['peak'][0, 0] - t['mean'][0, 0]) / t['std'][0, 0]
        n6 = (t['peak'][0, 1] - t['mean'][0, 1]) / t['std'][0, 1]
        assert n5 > 20 and n6 < 8


def test_empty_and_invalid_inputs(acq):
    from gpsmi.engine import EngineError
    data = scene_blocks('default', 0, 1)[0]
    assert acq.search_table(data, [], [0.0], 1).shape == (1, 0)
    assert acq.search_table(data, [3], [], 1).shape == (0, 1)
    with pytest.raises(EngineError):
        acq.search_table(data, [0], [0.0], 1)            # PRN 0 does not exist
    with pytest.raises(EngineError):
        acq.search_table(data, [38], [0.0], 1)
    with pytest.raises(EngineError):
        acq.search_table(data[:2048], [3], [0.0], 2)     # ragged: too short
    with pytest.raises(EngineError):
        acq.search_table(data, [3], [0.0], 0)
    with pytest.raises(EngineError):
        acq.search_table(data, [3], [0.0], 33)           # n_avg > N_CYC
    t = acq.search_table(data, [33, 34, 35, 36, 37], [100.0], 1)   # max PRN
    assert t.shape == (1, 5)


def test_device_resident_input_equals_host_input(acq):
    from gpsmi.engine import DeviceBuffer
    data = scene_blocks('default', 0, 1)[0]
    buf = DeviceBuffer(data.nbytes)
    buf.upload(data)
    f = [-1000.0, 0.0, 1000.0]
    a = acq.engine.search(data, [4, 5, 6], f, 4)
    b = acq.engine.search((buf.ptr, data.size), [4, 5, 6], f, 4)
    assert a.tobytes() == b.tobytes()
    buf.free()


@pytest.mark.parametrize('config, cs, n_cyc', [('default', 2048, 32), ('hirate', 16368, 8)])
def test_raw_u8_search_equals_complex64_search(config, cs, n_cyc):
    """gpsmi_acq_set_input_format(GPSMI_IQ_U8): the search reads the recorder's uint16 samples
    (host and device-resident), n_avg = 1 and 4 (the one- and four-group spectrum kernels, the
    general fold kernel): every peak record equals the complex64 search's, byte for byte."""
    from conftest import scene_for
    from gpsmi.acquisition import Acquisition
    from gpsmi.engine import Config, DeviceBuffer
    sc = scene_for(config)
    raw, c64 = sc.block_raw(0), sc.block(0)
    cfg = Config(code_samples=cs, n_cyc=n_cyc)
    a, b = Acquisition(cfg), Acquisition(cfg, raw_u8=True)
    prns = [s.prn for s in sc.sats][:6] + [1]
    f = [-2000.0, -400.0, 0.0, 1800.0]
    buf = DeviceBuffer(raw.nbytes)
    buf.upload(raw)
    for n_avg in (1, 4):
        want = a.engine.search(c64, prns, f, n_avg)
        assert b.engine.search(raw, prns, f, n_avg).tobytes() == want.tobytes(), n_avg
        assert b.engine.search((buf.ptr, raw.size), prns, f, n_avg).tobytes() == want.tobytes()
    t1, n1 = a.engine.search_ex(c64, prns, f, 4)
    t2, n2 = b.engine.search_ex(raw, prns, f, 4)
    assert t1.tobytes() == t2.tobytes() and n1.tobytes() == n2.tobytes()
    with pytest.raises(TypeError):
        b.engine.search(c64, prns, f, 1)
    buf.free()
    a.engine.close()
    b.engine.close()


# ---- BASELINE config 5: CODE_SAMPLES = 16368, N_CYC = 8 (time-domain correlation) ----

@pytest.fixture(scope='module')
def acq_hirate():
    from gpsmi.acquisition import Acquisition
    from gpsmi.engine import Config
    a = Acquisition(Config(code_samples=16368, n_cyc=8))
    yield a
    a.engine.close()


def test_hirate_cfg2_surface(acq_hirate, golden_hirate):
    """32 SV x 41 bins x 1 ms at 16.368 Msps against the reference's own surface."""
    f41 = [-5000.0 + 250.0 * i for i in range(41)]
    t = acq_hirate.search_table(scene_blocks('hirate', 0, 1)[0], list(range(1, 33)),
                                f41, 1)
    _check(t, golden_hirate, 'cfg2_')


def test_hirate_sweep_all_sats_loop(acq_hirate, golden_hirate):
    g = golden_hirate
    acq = acq_hirate
    sat_lst, found, freq = list(range(2, 33)), [], acq.cfg.min_freq
    blocks = scene_blocks('hirate', 0, 5)
    for b in range(5):
        ready, freq, found = acq.sweepAllSats(blocks[b], freq, sat_lst, found,
                                              itSweep=acq.cfg.it_sweep_all)
        assert (float(ready), freq, len(found)) == tuple(g['sweep_calls'][b])
    mine = np.array(found, dtype=np.float64)
    ref = g['sweep_found']
    assert np.array_equal(mine[:, 1:], ref[:, 1:])
    np.testing.assert_allclose(mine[:, 0], ref[:, 0], rtol=RTOL)


def test_rccl_gather_of_peak_records_single_rank(acq, golden_default):
    """The multi-GPU exchange with a world of one: unique id -> communicator -> a search that
    leaves its peak records in device memory -> gpsmi_comm_allgather_peaks (RCCL all-gather
    on the device, copy to the host).  What arrives must be the search's own table."""
    import ctypes as C
    from gpsmi import _lib
    from gpsmi.engine import DeviceBuffer, PEAK_DTYPE, check, ptr
    lib = _lib.load()
    idb = np.zeros(128, dtype=np.uint8)
    check(lib.gpsmi_comm_unique_id(ptr(idb)), 'gpsmi_comm_unique_id')
    comm = C.c_void_p()
    check(lib.gpsmi_comm_create(ptr(idb), 1, 0, 0, C.byref(comm)), 'gpsmi_comm_create')
    prns = list(range(1, 33))
    freqs = [-5000.0 + 250.0 * i for i in range(41)]
    blk = scene_blocks('default', 0, 1)[0]
    d_iq = DeviceBuffer(blk.nbytes)
    d_iq.upload(blk, 0)
    cells = len(prns) * len(freqs)
    d_send = DeviceBuffer(cells * PEAK_DTYPE.itemsize)
    d_recv = DeviceBuffer(cells * PEAK_DTYPE.itemsize)
    table = acq.engine.search((d_iq.ptr, 2048), prns, freqs, 1, out_dev=d_send.ptr)
    got = np.zeros((len(freqs), len(prns)), dtype=PEAK_DTYPE)
    check(lib.gpsmi_comm_allgather_peaks(comm, d_send.ptr, d_recv.ptr, cells, ptr(got)),
          'gpsmi_comm_allgather_peaks')
    check(lib.gpsmi_comm_destroy(comm), 'gpsmi_comm_destroy')
    for b in (d_iq, d_send, d_recv):
        b.free()
    assert got.tobytes() == table.tobytes()
    assert np.array_equal(table['argmax'], golden_default['cfg2_argmax'])
