"""GPU parity: tracking through the C ABI vs the reference fixtures
(gpslib.SatStream.process over 48 consecutive blocks x 12 channels).

Bars: argmax / DELAY / dump count / carry length bit-exact;
prompt dumps rtol 1e-3 + atol 1e-5; the three correlator taps around the peak
rtol 1e-3; codePhase atol 2e-3 samples; FREQ atol 0.05 Hz; lock flag equal
after the same number of blocks."""
import numpy as np
import pytest

from conftest import load_golden, scene_blocks


def load_golden_default():
    return load_golden('ref_default.npz')

pytestmark = pytest.mark.gpu


def _open_all(eng, g):
    for c, (sv, f0, d0) in enumerate(g['trk_init']):
        eng.open(c, int(sv), float(f0), int(d0))
    return len(g['trk_init'])


def _run_closed_loop(g, config, cfg=None):
    from gpsmi.engine import TrkEngine, STATE_DTYPE
    nch, nb = g['trk_delay'].shape
    eng = TrkEngine(cfg, max_ch=nch)
    _open_all(eng, g)
    blocks = scene_blocks(config, 5, nb)
    outs, states = [], []
    for i in range(nb):
        st = np.zeros(nch, dtype=STATE_DTYPE)
        for c in range(nch):
            st[c] = eng.get_state(c)
        states.append(st)
        outs.append(eng.process(blocks[i]))
    return eng, np.array(outs), np.array(states), blocks


@pytest.fixture(scope='module')
def closed_loop(golden_default):
    """Run the closed loop once over the fixture blocks; keep every output and
    the state at the start of every block."""
    r = _run_closed_loop(golden_default, 'default')
    yield r
    r[0].close()


@pytest.fixture(scope='module')
def closed_loop_hirate(golden_hirate):
    """BASELINE config 5: CODE_SAMPLES = 16368, N_CYC = 8 (time-domain correlation,
    chunked correlator)."""
    from gpsmi.engine import Config
    r = _run_closed_loop(golden_hirate, 'hirate', Config(code_samples=16368, n_cyc=8))
    yield r
    r[0].close()


def test_closed_loop_matches_reference(closed_loop, golden_default):
    _check_closed_loop(closed_loop[1], golden_default)


def test_hirate_closed_loop_matches_reference(closed_loop_hirate, golden_hirate):
    _check_closed_loop(closed_loop_hirate[1], golden_hirate)


def test_hirate_replay_reproduces_closed_loop(closed_loop_hirate):
    from gpsmi.engine import DeviceBuffer
    eng, outs, states, blocks = closed_loop_hirate
    nb, nch = outs.shape
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    rep = eng.replay(buf.ptr, nb, states, outs['delay_used'])
    rep2 = eng.replay(buf.ptr, 6, states[:6], None)
    buf.free()
    assert rep.tobytes() == outs.tobytes()
    assert rep2.tobytes() == outs[:6].tobytes()


def _check_closed_loop(outs, g):
    from gpsmi.engine import dumps_of
    nb, nch = outs.shape
    for c in range(nch):
        for i in range(nb):
            o = outs[i, c]
            where = f'channel {c} block {i}'
            assert o['prn'] == int(g['trk_init'][c, 0])
            assert o['mx'] == int(g['trk_mx'][c, i]), where
            assert o['delay_used'] == int(g['trk_delay'][c, i]), where
            assert o['n_dumps'] == int(g['trk_n_dumps'][c, i]), where
            assert o['nps'] == int(g['trk_nps'][c, i]), where
            nd = int(o['n_dumps'])
            np.testing.assert_allclose(dumps_of(o), g['trk_dumps'][c, i, :nd],
                                       rtol=1e-3, atol=1e-5, err_msg=where)
            np.testing.assert_allclose(o['epl'], g['trk_epl'][c, i], rtol=1e-3,
                                       err_msg=where)
            np.testing.assert_allclose(o['corr_mean'], g['trk_corr_mean'][c, i],
                                       rtol=1e-4)
            np.testing.assert_allclose(o['corr_std'], g['trk_corr_std'][c, i],
                                       rtol=1e-4)
            np.testing.assert_allclose(o['norm_max_corr'], g['trk_norm'][c, i],
                                       rtol=1e-3)
            cp = g['trk_code_phase'][c, i]
            if cp < 0:
                assert o['code_phase'] == -1.0 and o['delay'] == -1, where
            else:
                assert abs(o['code_phase'] - cp) < 2e-3, where
            assert abs(o['freq'] - g['trk_freq'][c, i]) < 0.05, where
            assert bool(o['phase_locked']) == bool(g['trk_locked'][c, i]), where
            np.testing.assert_allclose(o['std_dev'], g['trk_std_dev'][c, i],
                                       rtol=2e-3, err_msg=where)
            np.testing.assert_allclose(o['amplitude'], g['trk_amplitude'][c, i],
                                       rtol=2e-3, err_msg=where)
            # phase is only defined modulo 2 pi in the reference too
            dph = (o['phase'] - g['trk_phase'][c, i] + np.pi) % (2 * np.pi) - np.pi
            assert abs(dph) < 2e-3, where


def test_replay_reproduces_closed_loop(closed_loop):
    """Replay of the recorded trajectory gives the closed loop's outputs, and
    its end-of-block states equal the next row of the table."""
    from gpsmi.engine import DeviceBuffer
    eng, outs, states, blocks = closed_loop
    nb, nch = outs.shape
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    rep = eng.replay(buf.ptr, nb, states, outs['delay_used'])
    nxt = eng.replay_states(nb)
    buf.free()
    assert rep.tobytes() == outs.tobytes()
    for k in ('prn', 'delay', 'freq', 'phase', 'phase_locked', 'nps',
              'prev_sum_re', 'prev_sum_im', 'df_len', 'omega0'):
        assert np.array_equal(nxt[:-1][k], states[1:][k]), k
    for i in range(nb - 1):
        for c in range(nch):
            n = int(states[i + 1, c]['df_len'])
            assert np.array_equal(nxt[i, c]['df'][:n], states[i + 1, c]['df'][:n])


@pytest.mark.parametrize('form', [1, 0])
def test_batch_epilogue_forms_reproduce_closed_loop(closed_loop, form):
    """The batch form of the correlator with each form of the batch epilogue (option
    "epilogue_form": 1 = eight lanes per job, trk_epilogue8_kernel; 0 = a wave per job) against
    the closed loop, whose single-block epilogue is the wave form: outputs bytewise, states field
    by field (the edge-scan words included)."""
    from gpsmi.engine import TrkEngine, DeviceBuffer
    _, outs, states, blocks = closed_loop
    nb, nch = outs.shape
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    eng = TrkEngine(max_ch=nch)
    eng.set_option('span_single_max', 1)             # the batch kernels whatever the launch size
    eng.set_option('epilogue_form', form)
    assert eng.get_option('epilogue_form') == form
    rep = eng.replay(buf.ptr, nb, states, outs['delay_used'])
    nxt = eng.replay_states(nb)
    buf.free()
    eng.close()
    assert rep.tobytes() == outs.tobytes()
    for k in ('prn', 'delay', 'freq', 'phase', 'phase_locked', 'nps', 'prev_sum_re', 'prev_sum_im',
              'df_len', 'omega0', 'edge_state', 'prev_signal', 'std_dev'):
        assert nxt[:-1][k].tobytes() == states[1:][k].tobytes(), k
    for i in range(nb - 1):
        for c in range(nch):
            n = int(states[i + 1, c]['df_len'])
            assert nxt[i, c]['df'][:n].tobytes() == states[i + 1, c]['df'][:n].tobytes()


@pytest.mark.parametrize('n_cyc', [32, 16, 8])
def test_batch_epilogue_forms_agree_on_random_states(closed_loop, n_cyc):
    """Differential test of the two batch epilogues (option "epilogue_form") on state rows the closed
    loop does not visit often: forced delay 0 with and without a carry (33 dumps / the rows are the
    windows), every edge_state, locked and unlocked, drift lists of every length 1 .. 32, both signs
    of PREV_SIGNAL, FREQ at and beyond the clamp.  Same IQ, same table: records bytewise, next states
    field by field.  The other block lengths (five / three / two dumps per lane) take the first N_CYC
    code periods of the same blocks."""
    from gpsmi.engine import TrkEngine, DeviceBuffer, STATE_DTYPE, Config
    _, outs, states, blocks = closed_loop
    nb, nch = outs.shape
    blocks = [np.ascontiguousarray(b[:n_cyc * 2048]) for b in blocks]
    df_no = 1024 // n_cyc
    rng = np.random.default_rng(20260405 + n_cyc)
    table = states.copy()
    forced = outs['delay_used'].copy()
    for i in range(nb):
        for c in range(nch):
            st = table[i, c]
            kind = rng.integers(0, 6)
            if kind == 0:                     # no carry, delay 0: the rows are the windows
                st['nps'] = 0
                forced[i, c] = 0
            elif kind == 1:                   # a carry and delay 0: N_CYC + 1 dumps
                st['nps'] = int(rng.integers(1, 2048))
                forced[i, c] = 0
            elif kind == 2:
                st['nps'] = int(rng.integers(0, 2049))
                forced[i, c] = int(rng.integers(1, 2048))
            st['prev_sum_re'] = np.float32(rng.normal() * 3)
            st['prev_sum_im'] = np.float32(rng.normal() * 3)
            st['phase_locked'] = int(rng.integers(0, 2))
            st['edge_state'] = int(rng.integers(-1, 3))
            st['prev_signal'] = np.float32(rng.normal() * 0.05) if rng.integers(0, 4) else np.float32(0)
            st['std_dev'] = np.float32(abs(rng.normal()) * 0.01)
            n = int(rng.integers(1, df_no + 1))
            st['df_len'] = n
            st['df'][:n] = (rng.normal(size=n) * 0.3).astype(np.float32)
            if rng.integers(0, 8) == 0:       # FREQ about to be clamped
                st['freq'] = np.float32(4999.9 if rng.integers(0, 2) else -4999.9)
                st['omega0'] = 0.0
            table[i, c] = st
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    got = []
    for form in (0, 1):
        eng = TrkEngine(Config(n_cyc=n_cyc), max_ch=nch)
        eng.set_option('span_single_max', 1)
        eng.set_option('epilogue_form', form)
        rep = eng.replay(buf.ptr, nb, table, forced)
        nxt = eng.replay_states(nb)
        eng.close()
        got.append((rep, nxt))
    buf.free()
    (r0, n0), (r1, n1) = got
    assert set(np.unique(r0['n_dumps'])) >= {n_cyc, n_cyc + 1}
    assert r0.tobytes() == r1.tobytes()
    for k in STATE_DTYPE.names:
        if k == 'df':
            continue
        assert n0[k].tobytes() == n1[k].tobytes(), k
    for i in range(nb):
        for c in range(nch):
            n = int(n0[i, c]['df_len'])
            assert n0[i, c]['df'][:n].tobytes() == n1[i, c]['df'][:n].tobytes(), (i, c)


def test_replay_without_forced_delay(closed_loop):
    """With no recorded DELAY given, replay derives it from each block's own
    correlation, exactly as the closed loop does."""
    from gpsmi.engine import DeviceBuffer
    eng, outs, states, blocks = closed_loop
    nb = 6
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i in range(nb):
        buf.upload(blocks[i], i * blocks[i].nbytes)
    rep = eng.replay(buf.ptr, nb, states[:nb], None)
    buf.free()
    assert rep.tobytes() == outs[:nb].tobytes()


def test_raw_u8_input_equals_complex64_input(closed_loop, golden_default):
    """Fused ingest (gpsrecv.py:162-173): the kernels that read IQ decode the recorder's
    uint16 (Q << 8 | I) samples themselves.  Closed loop (host blocks), replay of the recorded rows
    (single-block form of the correlator) and replay of 96 blocks in one launch (its batch form) on
    the raw blocks must equal the complex64 path byte for byte."""
    from conftest import scene_for
    from gpsmi.engine import TrkEngine, DeviceBuffer
    _, outs, states, _ = closed_loop
    nb, nch = outs.shape
    sc = scene_for('default')
    raw = [sc.block_raw(5 + i) for i in range(nb)]
    eng = TrkEngine(max_ch=nch)
    eng.set_input_format(True)
    _open_all(eng, golden_default)
    n_cl = 6
    got = np.array([eng.process(raw[i]) for i in range(n_cl)])
    assert got.tobytes() == outs[:n_cl].tobytes()
    buf = DeviceBuffer(nb * raw[0].nbytes)
    for i, b in enumerate(raw):
        buf.upload(b, i * b.nbytes)
    rep = eng.replay(buf.ptr, nb, states, outs['delay_used'])
    buf.free()
    assert rep.tobytes() == outs.tobytes()
    # the same rows tiled to 96 blocks in one launch: past the point where the matrix correlator
    # switches from its single-block form to its batch form (the raw-sample variant of that one)
    rows = np.arange(96) % nb
    big = DeviceBuffer(96 * raw[0].nbytes)
    for i in range(96):
        big.upload(raw[rows[i]], i * raw[0].nbytes)
    rep96 = eng.replay(big.ptr, 96, states[rows], outs['delay_used'][rows])
    big.free()
    eng.close()
    assert rep96.tobytes() == outs[rows].tobytes()
    # other block / code lengths have no fused path: refused, not silently converted
    from gpsmi.engine import Config, EngineError
    e2 = TrkEngine(Config(code_samples=16368, n_cyc=8), max_ch=2)
    with pytest.raises(EngineError):
        e2.set_input_format(True)
    e2.close()


def test_block_from_device_memory(golden_default):
    """process() on a device-resident block equals process() on a host block."""
    from gpsmi.engine import TrkEngine, DeviceBuffer
    g = golden_default
    blocks = scene_blocks('default', 5, 2)
    res = []
    for dev in (False, True):
        eng = TrkEngine(max_ch=4)
        for c in range(4):
            sv, f0, d0 = g['trk_init'][c]
            eng.open(c, int(sv), float(f0), int(d0))
        outs = []
        for b in blocks:
            if dev:
                buf = DeviceBuffer(b.nbytes)
                buf.upload(b)
                outs.append(eng.process(buf.ptr))
                buf.free()
            else:
                outs.append(eng.process(b))
        res.append(np.array(outs).tobytes())
        eng.close()
    assert res[0] == res[1]


def test_first_window_rules():
    """decodeData's window edge cases (gpslib.py:1408-1419, :1440) on a clean
    signal: empty carry with delay 0, a partial first window, and the
    2047 -> 0 wrap that yields N_CYC + 1 dumps."""
    from gpsmi import codes
    from gpsmi.engine import TrkEngine, dumps_of
    rep = codes.code_replica(9)
    cs, ncyc = 2048, 32

    def block(delay):
        return (0.4 * np.tile(np.roll(rep, delay), ncyc)).astype(np.complex64)

    eng = TrkEngine(max_ch=1)
    eng.open(0, 9, 0.0, 0)
    o = eng.process(block(0))[0]                  # NPS = 0, delay 0
    assert (o['n_dumps'], o['first_len'], o['nps'], o['delay_used']) == (32, cs, 0, 0)
    eng.close()

    eng = TrkEngine(max_ch=1)
    eng.open(0, 9, 0.0, 2047)
    o = eng.process(block(2047))[0]               # partial first window
    assert (o['n_dumps'], o['first_len'], o['nps']) == (32, 2047, 1)
    d = dumps_of(o)
    assert np.allclose(d.real[1:], d.real[1], rtol=1e-5)
    o = eng.process(block(0))[0]                  # carry of 1 sample, delay 0
    assert o['delay_used'] == 0
    assert (o['n_dumps'], o['first_len'], o['nps']) == (33, 1, 0)
    o = eng.process(block(0))[0]
    assert (o['n_dumps'], o['first_len'], o['nps']) == (32, cs, 0)
    eng.close()


def test_channel_management_and_errors():
    from gpsmi.engine import TrkEngine, EngineError
    eng = TrkEngine(max_ch=3, prns=[3, 4])
    with pytest.raises(EngineError):
        eng.open(3, 3, 0.0, 0)                    # channel out of range
    with pytest.raises(EngineError):
        eng.open(0, 5, 0.0, 0)                    # no replica for PRN 5
    with pytest.raises(EngineError):
        eng.open(0, 3, 0.0, 2048)                 # delay out of range
    with pytest.raises(EngineError):
        eng.close_channel(1)                      # not open
    with pytest.raises(EngineError):
        eng.process(np.zeros(100, np.complex64))  # ragged block
    eng.open(1, 4, 250.0, 17)
    st = eng.get_state(1)
    assert (st['prn'], st['delay'], st['freq'], st['df_len']) == (4, 17, 250.0, 1)
    out = eng.process(np.zeros(65536, np.complex64))
    assert out[0]['prn'] == 0 and out[2]['prn'] == 0 and out[1]['prn'] == 4
    eng.close_channel(1)
    assert eng.get_state(1)['prn'] == 0
    eng.close()


def test_streams_and_streaming_errors_and_fallbacks(golden_default):
    """Edges of the round-3 entry points: stream counts out of range, replay on a multi-stream
    handle, ragged multi-stream input, set_streams resets the channels, and
    gpsmi_trk_process_stream from PAGEABLE memory (no kernel may read that: the runtime's copy is
    taken instead) and with batched streams, against the blocking call."""
    import ctypes as C
    from gpsmi.engine import TrkEngine, EngineError, STATE_DTYPE, OUT_DTYPE, check
    g = golden_default
    eng = TrkEngine(max_ch=4)
    for bad in (0, -1, 70000):
        with pytest.raises(EngineError):
            check(eng.lib.gpsmi_trk_set_streams(eng.h, bad), 'set_streams')
    eng.open(0, 5, 0.0, 3)
    check(eng.lib.gpsmi_trk_set_streams(eng.h, 3), 'set_streams')
    eng.streams = 3
    assert all(eng.get_state(c, stream=r)['prn'] == 0 for r in range(3) for c in range(4))
    with pytest.raises(EngineError):
        eng.process(np.zeros((2, 65536), np.complex64))          # one block short
    with pytest.raises(EngineError):
        eng.replay_load(2, np.zeros((2, 4), dtype=STATE_DTYPE))   # replay keeps to one stream
    with pytest.raises(ValueError):
        eng.open(0, 5, 0.0, 0, stream=3)
    eng.close()
    # pageable input and batched streams through the streaming call
    blocks = scene_blocks('default', 5, 8)
    R, nch = 2, 3
    ref, got = TrkEngine(max_ch=nch, streams=R), TrkEngine(max_ch=nch, streams=R)
    for e in (ref, got):
        for r in range(R):
            for c, (sv, f0, d0) in enumerate(g['trk_init'][:nch]):
                e.open(c, int(sv), float(f0), int(d0), stream=r)
    outs = []
    for i in range(3):
        slab = np.ascontiguousarray(np.stack([blocks[2 * i], blocks[2 * i + 1]]))   # ordinary numpy memory
        want = ref.process(slab)
        o = np.zeros((R, nch), dtype=OUT_DTYPE)
        got.process_stream(slab, o)
        got.wait()                                      # (pageable source and sink: wait per block)
        assert o.tobytes() == want.tobytes(), i
    ref.close()
    got.close()


def test_pipelined_replay_runs_and_readbacks(closed_loop):
    """run_async / fetch_async / wait_prev: three runs in flight two at a time, every
    read-back equal to the blocking replay (two result slots, copy stream)."""
    from gpsmi.engine import DeviceBuffer, PinnedArray, OUT_DTYPE
    eng, outs, states, blocks = closed_loop
    nb, nch = outs.shape
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    eng.replay_load(nb, states, outs['delay_used'])
    pins = [PinnedArray((nb, nch), OUT_DTYPE) for _ in range(2)]
    got = []
    for k in range(3):
        pins[k & 1].array.view(np.uint8)[:] = 0xAB        # poison before it is refilled
        eng.replay_run_async(buf.ptr, nb)
        eng.replay_fetch_async(pins[k & 1].array)
        eng.wait_prev()
        if k > 0:
            got.append(pins[(k - 1) & 1].array.tobytes())
            assert eng.last_ms()[0] > 0
    eng.wait()
    got.append(pins[2 & 1].array.tobytes())
    buf.free()
    for p in pins:
        p.free()
    assert all(g == outs.tobytes() for g in got)


@pytest.mark.parametrize('done_by_dispatch', ['1', '0'])
@pytest.mark.parametrize('nb', [8, 24])
def test_async_batches_with_different_input_do_not_share_buffers(closed_loop, monkeypatch, nb,
                                                                 done_by_dispatch):
    """Consecutive asynchronous batches on DIFFERENT IQ (same state table): a dropped
    cross-stream dependency or a scratch buffer shared by the two result slots returns the
    other batch's sums.  nb = 8 takes the single-block form of the span correlator (raw
    span records + its own epilogue), nb = 24 the batch form; GPSMI_DONE_BY_DISPATCH=0 swaps
    the dispatch's completion signal for an event record.  Every read-back must equal the
    blocking replay of the same input on a fresh handle."""
    from gpsmi.engine import TrkEngine, DeviceBuffer, PinnedArray, OUT_DTYPE
    _, outs, states, blocks = closed_loop
    nch = outs.shape[1]
    bufs = []
    for first in (0, nb):                      # two different stretches of the recording
        buf = DeviceBuffer(nb * blocks[0].nbytes)
        for i in range(nb):
            buf.upload(blocks[first + i], i * blocks[0].nbytes)
        bufs.append(buf)
    table, forced = states[:nb], outs['delay_used'][:nb]
    ref_eng = TrkEngine(max_ch=nch)
    want = [ref_eng.replay(b.ptr, nb, table, forced).tobytes() for b in bufs]
    ref_eng.close()
    assert want[0] != want[1]
    eng = TrkEngine(max_ch=nch)
    eng.set_option('done_by_dispatch', int(done_by_dispatch))        # (an option of the ABI, gpsmi.h)
    assert eng.get_option('done_by_dispatch') == int(done_by_dispatch)
    eng.replay_load(nb, table, forced)
    pins = [PinnedArray((nb, nch), OUT_DTYPE) for _ in range(2)]
    order = [0, 1, 1, 0, 1, 0, 0]
    for k, which in enumerate(order):
        pins[k & 1].array.view(np.uint8)[:] = 0xAB
        eng.set_timing(k % 3 == 0)
        eng.replay_run_async(bufs[which].ptr, nb)
        eng.replay_fetch_async(pins[k & 1].array)
        eng.wait_prev()
        if k > 0:
            assert pins[(k - 1) & 1].array.tobytes() == want[order[k - 1]], (k - 1, nb)
    eng.wait()
    assert pins[(len(order) - 1) & 1].array.tobytes() == want[order[-1]]
    # a new table while a run is in flight: replay_load waits for it (gpsmi.h)
    eng.replay_run_async(bufs[0].ptr, nb)
    eng.replay_fetch_async(pins[0].array)
    eng.replay_load(nb, table, forced)
    assert pins[0].array.tobytes() == want[0]
    eng.close()
    for b in bufs:
        b.free()
    for p_ in pins:
        p_.free()


@pytest.mark.parametrize('mode', [1])
def test_overlapped_pipelines_return_the_isolated_results(closed_loop, mode):
    """Option "corr_overlap" = 1: consecutive runs alternate between two streams.  Batches on different
    IQ in a mixed order, the batch
    kernels forced (span_single_max = 1), timing modes mixed: every read-back equals the blocking
    replay of the same input on a fresh handle; then back to the isolated pipeline on the same handle."""
    from gpsmi.engine import TrkEngine, DeviceBuffer, PinnedArray, OUT_DTYPE
    _, outs, states, blocks = closed_loop
    nch = outs.shape[1]
    nb = 20
    bufs = []
    for first in (0, nb):
        buf = DeviceBuffer(nb * blocks[0].nbytes)
        for i in range(nb):
            buf.upload(blocks[first + i], i * blocks[0].nbytes)
        bufs.append(buf)
    table, forced = states[:nb], outs['delay_used'][:nb]
    ref_eng = TrkEngine(max_ch=nch)
    ref_eng.set_option('span_single_max', 1)
    want = [ref_eng.replay(b.ptr, nb, table, forced).tobytes() for b in bufs]
    ref_eng.close()
    assert want[0] != want[1]
    eng = TrkEngine(max_ch=nch)
    eng.set_option('span_single_max', 1)
    eng.replay_load(nb, table, forced)
    pins = [PinnedArray((nb, nch), OUT_DTYPE) for _ in range(2)]

    def run(order):
        for k, which in enumerate(order):
            pins[k & 1].array.view(np.uint8)[:] = 0xAB
            eng.set_timing(1 if k % 4 == 0 else 2)
            eng.replay_run_async(bufs[which].ptr, nb)
            eng.replay_fetch_async(pins[k & 1].array)
            eng.wait_prev()
            if k > 0:
                assert pins[(k - 1) & 1].array.tobytes() == want[order[k - 1]], (mode, k - 1)
        eng.wait()
        assert pins[(len(order) - 1) & 1].array.tobytes() == want[order[-1]]

    eng.set_option('corr_overlap', mode)
    assert eng.get_option('corr_overlap') == mode
    run([0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1])
    eng.set_option('corr_overlap', 0)
    run([1, 0, 0, 1])
    eng.set_option('corr_overlap', mode)             # ... and on again
    run([0, 0, 1])
    eng.close()
    for b in bufs:
        b.free()
    for p_ in pins:
        p_.free()


def test_search_ordered_behind_replay_batches(closed_loop):
    """The step of bench.py: a search on its own handle ordered on the device behind the
    previous batch's correlator (gpsmi_acq_after_trk: the dispatch's own completion event),
    the batch's epilogue and read-back on their streams.  Every batch and every search must
    return what the blocking calls return."""
    from gpsmi.engine import AcqEngine, DeviceBuffer, PinnedArray, OUT_DTYPE, PEAK_DTYPE
    eng, outs, states, blocks = closed_loop
    nb, nch = outs.shape
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    eng.replay_load(nb, states, outs['delay_used'])
    acq = AcqEngine()
    prns = list(range(1, 33))
    freqs = [-5000.0 + 250.0 * i for i in range(41)]
    n = 2048
    want = acq.search((buf.ptr, n), prns, freqs, 1)
    pins = [PinnedArray((nb, nch), OUT_DTYPE) for _ in range(2)]
    apins = [PinnedArray((len(freqs), len(prns)), PEAK_DTYPE) for _ in range(2)]
    for k in range(5):
        if k > 0:
            acq.wait()
            assert apins[(k - 1) & 1].array.tobytes() == want.tobytes(), k
        pins[k & 1].array.view(np.uint8)[:] = 0xAB
        apins[k & 1].array.view(np.uint8)[:] = 0xCD
        acq.after(eng)
        eng.set_timing(k % 2 == 0)
        eng.replay_run_async(buf.ptr, nb)
        acq.search_async(buf.ptr, n, prns, freqs, 1, apins[k & 1].array)
        eng.replay_fetch_async(pins[k & 1].array)
        eng.wait_prev()
        if k > 0:
            assert pins[(k - 1) & 1].array.tobytes() == outs.tobytes(), k
    acq.wait()
    eng.wait()
    assert apins[4 & 1].array.tobytes() == want.tobytes()
    assert pins[4 & 1].array.tobytes() == outs.tobytes()
    eng.set_timing(True)
    buf.free()
    for p_ in pins + apins:
        p_.free()
    acq.close()


@pytest.mark.parametrize('n_cyc', [16, 8])
def test_other_block_lengths_match_the_oracle(n_cyc):
    """N_CYC = 16 and 8 at CODE_SAMPLES = 2048 (gpsglob.py:122 allows 8/16/32): no
    fixture from the reference holds these, so the GPU is compared with the oracle
    (itself bit-exact against the reference at N_CYC = 32 and 8) on a fresh scene.  Round 4: these
    block lengths run the span correlator too (template parameter NC of gpsmi_trk_span.h); the
    closed loop takes its one-wave-per-span form, the replay of the recorded trajectory its batch
    form: bytewise equal, complex64 and raw uint16 input alike."""
    import gps_oracle as orc
    from gpsmi import synth
    from gpsmi.engine import Config, TrkEngine, DeviceBuffer, STATE_DTYPE, dumps_of
    p = orc.Params(n_cyc=n_cyc)
    sc = synth.default_scene(5, seed=31 + n_cyc, n_cyc=n_cyc, amp=0.09)
    nb = 12 * 32 // n_cyc
    blocks = [sc.block(b) for b in range(nb)]
    raws = [sc.block_raw(b) for b in range(nb)]
    eng = TrkEngine(Config(n_cyc=n_cyc), max_ch=len(sc.sats))
    assert eng.get_option('correlator') == 1                  # the matrix-pipe (span) form
    eng8 = TrkEngine(Config(n_cyc=n_cyc), max_ch=len(sc.sats))
    eng8.set_input_format(True)
    streams = []
    for c, s in enumerate(sc.sats):
        f0 = round(s.doppler / 200.0) * 200.0
        d0 = int(round(s.delay)) % 2048
        eng.open(c, s.prn, f0, d0)
        eng8.open(c, s.prn, f0, d0)
        streams.append(orc.SatStream(s.prn, f0, p, delay=d0))
    nch = len(sc.sats)
    states = np.zeros((nb, nch), dtype=STATE_DTYPE)
    outs = []
    for i, blk in enumerate(blocks):
        for c in range(nch):
            states[i, c] = eng.get_state(c)
        out = eng.process(blk)
        outs.append(out)
        assert eng8.process(raws[i]).tobytes() == out.tobytes(), f'raw u8 input, block {i}'
        for c, ss in enumerate(streams):
            ss.process(blk, np.int64((i + 1) * p.ngps))
            where = f'n_cyc {n_cyc} channel {c} block {i}'
            assert out[c]['mx'] == ss.last['mx'], where
            assert out[c]['delay_used'] == ss.delay, where
            assert out[c]['n_dumps'] == len(ss.last['dumps']), where
            np.testing.assert_allclose(dumps_of(out[c]), ss.last['dumps'], rtol=1e-3,
                                       atol=1e-5, err_msg=where)
            assert abs(out[c]['freq'] - ss.freq) < 0.05, where
            assert bool(out[c]['phase_locked']) == bool(ss.phase_locked), where
    outs = np.stack(outs)
    # replay of the trajectory: enough blocks that the launch takes the batch form of the correlator
    reps = 4
    nbig = nb * reps
    buf = DeviceBuffer(nbig * blocks[0].nbytes)
    buf8 = DeviceBuffer(nbig * raws[0].nbytes)
    for r in range(reps):
        for i in range(nb):
            buf.upload(blocks[i], (r * nb + i) * blocks[0].nbytes)
            buf8.upload(raws[i], (r * nb + i) * raws[0].nbytes)
    table = np.concatenate([states] * reps)
    forced = np.concatenate([outs['delay_used']] * reps)
    assert nbig * 1 > eng.get_option('span_single_max')
    rep = eng.replay(buf.ptr, nbig, table, forced)
    rep8 = eng8.replay(buf8.ptr, nbig, table, forced)
    for r in range(reps):
        assert rep[r * nb:(r + 1) * nb].tobytes() == outs.tobytes(), r
        assert rep8[r * nb:(r + 1) * nb].tobytes() == outs.tobytes(), r
    buf.free()
    buf8.free()
    eng.close()
    eng8.close()


@pytest.mark.parametrize('n_cyc', [16, 8])
def test_other_block_lengths_forced_delays_agree_with_the_vector_correlator(n_cyc):
    """Every residue mod 4 and every tile / quarter / period edge of the delay through the span
    correlator at N_CYC = 16 / 8 (batch form == single form bytewise), and against the vector
    correlator (an independent kernel) within the reference tolerance."""
    from gpsmi import engine as E
    from gpsmi import synth
    from gpsmi.engine import Config, TrkEngine, DeviceBuffer, STATE_DTYPE
    sc = synth.default_scene(12, seed=5 + n_cyc, n_cyc=n_cyc, amp=0.09)
    nch, nb = 12, 120
    blocks = [sc.block(b % 6) for b in range(nb)]
    edges = [0, 1, 2, 3, 4, 5, 6, 7, 61, 62, 63, 64, 65, 66, 67, 127, 128, 129, 130, 131, 255, 256, 257,
             509, 510, 511, 512, 513, 514, 515, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 1535, 1536, 1537,
             2040, 2041, 2042, 2043, 2044, 2045, 2046, 2047]
    eng = TrkEngine(Config(n_cyc=n_cyc), max_ch=nch)
    for c, s in enumerate(sc.sats):
        eng.open(c, s.prn, round(s.doppler / 200.0) * 200.0, int(round(s.delay)) % 2048)
    st0 = np.array([eng.get_state(c) for c in range(nch)], dtype=STATE_DTYPE)
    table = np.broadcast_to(st0, (nb, nch)).copy()
    rng = np.random.default_rng(3)
    table['phase'] = rng.uniform(0, 6.28, (nb, nch)).astype(np.float32)
    forced = np.empty((nb, nch), dtype=np.int32)
    for i in range(nb):
        for c in range(nch):
            k = i * nch + c
            forced[i, c] = (int(round(sc.sats[c].delay)) + int(rng.integers(-3, 4))) % 2048 if k % 3 == 0 \
                else edges[(k // 3 + 7 * c) % len(edges)]
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i in range(nb):
        buf.upload(blocks[i], i * blocks[0].nbytes)
    whole = eng.replay(buf.ptr, nb, table, forced)                       # batch form
    pieces = np.concatenate([eng.replay(buf.at(i * blocks[0].nbytes), 4, table[i:i + 4], forced[i:i + 4])
                             for i in range(0, nb, 4)])                  # single form
    assert pieces.tobytes() == whole.tobytes()
    eng.close()
    E.set_default('correlator', 0)
    try:
        veng = TrkEngine(Config(n_cyc=n_cyc), max_ch=nch)
    finally:
        E.clear_default('correlator')
    assert veng.get_option('correlator') == 0
    for c, s in enumerate(sc.sats):
        veng.open(c, s.prn, 0.0, 0)
    vec = veng.replay(buf.ptr, nb, table, forced)
    veng.close()
    buf.free()
    for k in ('mx', 'delay', 'delay_used', 'n_dumps', 'nps', 'first_len'):
        assert np.array_equal(vec[k], whole[k]), k
    np.testing.assert_allclose(vec['dumps'], whole['dumps'], rtol=1e-3, atol=5e-4)


@pytest.mark.parametrize('corr_avg', [4, 5, 12])
def test_other_corr_avg_matches_the_oracle(corr_avg):
    """CORR_AVG other than the reference's 8 (gpsglob.py:63): the code-phase correlation
    then folds its rows one at a time instead of through the eight-row pipeline; compared
    with the oracle on a fresh scene, closed loop and replay."""
    import gps_oracle as orc
    from gpsmi import synth
    from gpsmi.engine import Config, TrkEngine, DeviceBuffer, STATE_DTYPE, dumps_of
    p = orc.Params(corr_avg=corr_avg)
    sc = synth.default_scene(5, seed=77 + corr_avg, amp=0.09)
    nb = 6
    blocks = [sc.block(b) for b in range(nb)]
    nch = len(sc.sats)
    eng = TrkEngine(Config(corr_avg=corr_avg), max_ch=nch)
    streams = []
    for c, s in enumerate(sc.sats):
        f0 = round(s.doppler / 200.0) * 200.0
        d0 = int(round(s.delay)) % 2048
        eng.open(c, s.prn, f0, d0)
        streams.append(orc.SatStream(s.prn, f0, p, delay=d0))
    outs, states = [], []
    for i, blk in enumerate(blocks):
        st = np.zeros(nch, dtype=STATE_DTYPE)
        for c in range(nch):
            st[c] = eng.get_state(c)
        states.append(st)
        out = eng.process(blk)
        outs.append(out)
        for c, ss in enumerate(streams):
            ss.process(blk, np.int64((i + 1) * p.ngps))
            where = f'corr_avg {corr_avg} channel {c} block {i}'
            assert out[c]['mx'] == ss.last['mx'], where
            assert out[c]['delay_used'] == ss.delay, where
            np.testing.assert_allclose(out[c]['epl'], ss.last['epl'], rtol=1e-3, err_msg=where)
            np.testing.assert_allclose(dumps_of(out[c]), ss.last['dumps'], rtol=1e-3,
                                       atol=1e-5, err_msg=where)
            assert abs(out[c]['freq'] - ss.freq) < 0.05, where
    outs, states = np.array(outs), np.array(states)
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    rep = eng.replay(buf.ptr, nb, states, outs['delay_used'])      # four-channel form of the kernel
    buf.free()
    eng.close()
    assert rep.tobytes() == outs.tobytes()


def test_twenty_channels_four_groups():
    """More channels than the reference's MAX_SAT: 20 channels = four workgroup groups of
    the correlator, five groups of the correlation kernel; every channel equals the same
    channel run alone, and replay equals the closed loop."""
    from gpsmi.engine import DeviceBuffer, STATE_DTYPE, TrkEngine
    g = load_golden_default()
    blocks = scene_blocks('default', 5, 4)
    init = [tuple(g['trk_init'][c % len(g['trk_init'])]) for c in range(20)]
    eng = TrkEngine(max_ch=20)
    for c, (sv, f0, d0) in enumerate(init):
        eng.open(c, int(sv), float(f0) + 10.0 * (c // 12), int(d0))
    states, outs = [], []
    for blk in blocks:
        st = np.zeros(20, dtype=STATE_DTYPE)
        for c in range(20):
            st[c] = eng.get_state(c)
        states.append(st)
        outs.append(eng.process(blk))
    outs, states = np.array(outs), np.array(states)
    buf = DeviceBuffer(len(blocks) * blocks[0].nbytes)
    for i, b in enumerate(blocks):
        buf.upload(b, i * b.nbytes)
    rep = eng.replay(buf.ptr, len(blocks), states, outs['delay_used'])
    buf.free()
    assert rep.tobytes() == outs.tobytes()
    # the same rows as one launch of 64 blocks: 128 (block, channel group) units = the BATCH form of
    # the matrix correlator with two channel groups, the second one with eight of its twelve columns
    rows = np.arange(64) % len(blocks)
    big = DeviceBuffer(64 * blocks[0].nbytes)
    for i in range(64):
        big.upload(blocks[rows[i]], i * blocks[0].nbytes)
    rep64 = eng.replay(big.ptr, 64, states[rows], outs['delay_used'][rows])
    big.free()
    eng.close()
    assert rep64.tobytes() == outs[rows].tobytes()
    for c in (0, 7, 13, 19):                          # one channel of each group, alone
        solo = TrkEngine(max_ch=1)
        sv, f0, d0 = init[c]
        solo.open(0, int(sv), float(f0) + 10.0 * (c // 12), int(d0))
        for i, blk in enumerate(blocks):
            assert solo.process(blk)[0].tobytes() == outs[i, c].tobytes(), (c, i)
        solo.close()


def test_code_length_4096_matches_the_oracle():
    """CODE_SAMPLES = 4096 (4.096 Msps), N_CYC = 16: neither the 2048 fast path nor a
    fixture; acquisition surface and tracking against the oracle on a fresh scene."""
    import gps_oracle as orc
    from gpsmi import synth
    from gpsmi.acquisition import Acquisition
    from gpsmi.engine import Config, TrkEngine, dumps_of
    cs, n_cyc = 4096, 16
    p = orc.Params(code_samples=cs, n_cyc=n_cyc)
    cfg = Config(code_samples=cs, n_cyc=n_cyc)
    sc = synth.default_scene(4, seed=57, code_samples=cs, n_cyc=n_cyc, amp=0.09)
    blocks = [sc.block(b) for b in range(8)]
    prns = [s.prn for s in sc.sats] + [1]
    freqs = [round(s.doppler / 200.0) * 200.0 for s in sc.sats]
    acq = Acquisition(cfg)
    tab = acq.search_table(blocks[0], prns, freqs, 4)
    ref = orc.acq_table(blocks[0], freqs, prns, 4, p)
    acq.engine.close()
    assert np.array_equal(tab['argmax'], ref['argmax'])
    for k in ('peak', 'mean', 'std'):
        np.testing.assert_allclose(tab[k], ref[k], rtol=1e-4)
    eng = TrkEngine(cfg, max_ch=len(sc.sats) + 3)        # three channels stay closed: the fold's
    streams = []                                           # second group of four is 1 open + 2 closed + padding
    for c, s in enumerate(sc.sats):
        d0 = int(tab['argmax'][c, c])
        eng.open(c, s.prn, freqs[c], d0)
        streams.append(orc.SatStream(s.prn, freqs[c], p, delay=d0))
    for i, blk in enumerate(blocks[1:], start=1):
        out = eng.process(blk)
        assert all(out[c]['prn'] == 0 and out[c]['n_dumps'] == 0 for c in range(len(sc.sats), len(out)))
        for c, ss in enumerate(streams):
            ss.process(blk, np.int64((i + 1) * p.ngps))
            where = f'channel {c} block {i}'
            assert out[c]['mx'] == ss.last['mx'], where
            assert out[c]['delay_used'] == ss.delay, where
            np.testing.assert_allclose(dumps_of(out[c]), ss.last['dumps'], rtol=1e-3,
                                       atol=1e-5, err_msg=where)
            assert abs(out[c]['freq'] - ss.freq) < 0.05, where
    eng.close()


def test_vector_correlator_agrees_with_the_matrix_one(closed_loop, golden_default, monkeypatch):
    """GPSMI_STREAM_MFMA=0: the packed-FMA correlator (the default wherever the matrix-pipe
    kernel does not apply: other block lengths, other code lengths) on the CS = 2048,
    N_CYC = 32 fixture: same reference parity, and against the default kernel only the
    order of the float32 sums differs."""
    from gpsmi import engine as E
    E.set_default('correlator', 0)                      # (gpsmi_set_default: for handles created from now on)
    try:
        r = _run_closed_loop(golden_default, 'default')
    finally:
        E.clear_default('correlator')
    assert r[0].get_option('correlator') == 0
    r[0].close()
    _check_closed_loop(r[1], golden_default)
    outs = closed_loop[1]
    for k in ('mx', 'delay', 'delay_used', 'n_dumps', 'nps', 'phase_locked'):
        assert np.array_equal(r[1][k], outs[k]), k
    # two closed loops: the PLL feeds the last-bit differences back, the trajectories drift
    # apart within the reference tolerance
    np.testing.assert_allclose(r[1]['dumps'], outs['dumps'], rtol=1e-3, atol=5e-5)


def test_forced_delays_at_every_edge_agree_across_correlators(closed_loop, monkeypatch):
    """The window boundary DELAY at every place the matrix correlator treats specially: all four
    residues mod 4 (a K-step of four positions is issued in pieces around it), the edges of a
    64-position tile, of a 512-position quarter and of the code period, 0 and 2047 -- forced
    through replay on the recorded states.  (a) The batch form of the kernel (160 blocks in one
    launch) and its single-block form (the same rows eight at a time) give the same bytes;
    (b) the vector correlator (GPSMI_STREAM_MFMA=0: an independent kernel, float32 sums in
    another order) agrees within the reference tolerance, integer fields exactly."""
    from gpsmi.engine import TrkEngine, DeviceBuffer
    eng, outs, states, blocks = closed_loop
    nb0, nch = outs.shape
    edges = [0, 1, 2, 3, 4, 5, 6, 7, 61, 62, 63, 64, 65, 66, 67, 127, 128, 129, 130, 131, 255, 256, 257,
             509, 510, 511, 512, 513, 514, 515, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 1535, 1536, 1537,
             2040, 2041, 2042, 2043, 2044, 2045, 2046, 2047]
    nb = 160
    rows = np.arange(nb) % nb0
    table = states[rows].copy()
    rng = np.random.default_rng(11)
    forced = np.empty((nb, nch), dtype=np.int32)
    for i in range(nb):
        for c in range(nch):
            k = (i * nch + c)
            if k % 3 == 0:        # near the true delay: the sums carry signal
                forced[i, c] = (int(outs['delay_used'][rows[i], c]) + int(rng.integers(-5, 6))) % 2048
            else:
                forced[i, c] = edges[(k // 3 + 7 * c) % len(edges)]
    buf = DeviceBuffer(nb * blocks[0].nbytes)
    for i in range(nb):
        buf.upload(blocks[rows[i]], i * blocks[0].nbytes)
    whole = eng.replay(buf.ptr, nb, table, forced)
    pieces = np.concatenate([eng.replay(buf.at(i * blocks[0].nbytes), 8, table[i:i + 8], forced[i:i + 8])
                             for i in range(0, nb, 8)])
    assert pieces.tobytes() == whole.tobytes()
    assert np.array_equal(whole['delay_used'], forced)
    monkeypatch.setenv('GPSMI_STREAM_MFMA', '0')
    veng = TrkEngine(None, max_ch=nch)
    monkeypatch.delenv('GPSMI_STREAM_MFMA')
    for c in range(nch):
        veng.open(c, int(states[0, c]['prn']), 0.0, 0)
    vec = veng.replay(buf.ptr, nb, table, forced)
    veng.close()
    buf.free()
    for k in ('mx', 'delay', 'delay_used', 'n_dumps', 'nps', 'first_len'):
        assert np.array_equal(vec[k], whole[k]), k
    np.testing.assert_allclose(vec['dumps'], whole['dumps'], rtol=1e-3, atol=5e-4)
    np.testing.assert_allclose(vec['epl'], whole['epl'], rtol=1e-5)


def test_time_domain_correlation_variant_agrees(closed_loop_hirate, golden_hirate, monkeypatch):
    """GPSMI_DIRECT_CORR=1: the exact time-domain correlation kernel (the fall-back for
    code periods beyond 16384 samples) against the same reference fixture as the
    32768-point FFT path."""
    from gpsmi.engine import Config, TrkEngine
    monkeypatch.setenv('GPSMI_DIRECT_CORR', '1')
    r = _run_closed_loop(golden_hirate, 'hirate', Config(code_samples=16368, n_cyc=8))
    monkeypatch.delenv('GPSMI_DIRECT_CORR')
    r[0].close()
    _check_closed_loop(r[1], golden_hirate)
    fft = closed_loop_hirate[1]
    for k in ('mx', 'delay', 'delay_used', 'n_dumps', 'nps', 'phase_locked'):
        assert np.array_equal(r[1][k], fft[k]), k


@pytest.mark.parametrize('env', [{'GPSMI_STREAM_MFMA': '0'}, {'GPSMI_DIRECT_CORR': '2'}])
def test_hirate_fallback_kernels_agree(closed_loop_hirate, golden_hirate, monkeypatch, env):
    """The kernels other code lengths still run, selected at 16368 samples: GPSMI_STREAM_MFMA=0 the
    chunked vector correlator instead of the matrix one, GPSMI_DIRECT_CORR=2 the zero-padded
    32768-point FFT pair instead of the native-length correlation.  Same reference fixture, same
    integer results as the default path."""
    from gpsmi.engine import Config
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = _run_closed_loop(golden_hirate, 'hirate', Config(code_samples=16368, n_cyc=8))
    for k in env:
        monkeypatch.delenv(k)
    r[0].close()
    _check_closed_loop(r[1], golden_hirate)
    ref = closed_loop_hirate[1]
    for k in ('mx', 'delay', 'delay_used', 'n_dumps', 'nps', 'phase_locked'):
        assert np.array_equal(r[1][k], ref[k]), k


def test_result_slots_carry_nothing_over(closed_loop):
    """The result buffers are not cleared between launches: every byte of an open
    channel's record is rewritten and a closed channel's record is zeroed by the
    epilogue.  A slot that held a full batch is reused for a batch with a closed
    channel and other data; the read-back equals a fresh engine's."""
    from gpsmi.engine import TrkEngine, DeviceBuffer
    eng0, outs, states, blocks = closed_loop
    nb, nch = 6, outs.shape[1]
    buf = DeviceBuffer(2 * nb * blocks[0].nbytes)
    for i in range(2 * nb):
        buf.upload(blocks[i], i * blocks[i].nbytes)
    t1, d1 = states[:nb].copy(), outs['delay_used'][:nb].copy()
    t2, d2 = states[nb:2 * nb].copy(), outs['delay_used'][nb:2 * nb].copy()
    t2[:, 1] = np.zeros((), dtype=t2.dtype)                     # channel 1 closed in batch 2
    t2['df_len'][:, 1] = 1
    second = buf.at(nb * blocks[0].nbytes)
    fresh = TrkEngine(max_ch=nch)
    ref = fresh.replay(second, nb, t2, d2).copy()
    fresh.close()
    eng = TrkEngine(max_ch=nch)
    eng.replay(buf.ptr, nb, t1, d1)                             # slot A <- batch 1
    eng.replay(second, nb, t2, d2)                              # slot B <- batch 2
    again = eng.replay(second, nb, t2, d2).copy()               # slot A again: held batch 1
    eng.close()
    buf.free()
    assert again.tobytes() == ref.tobytes()
    assert not np.frombuffer(again[:, 1].tobytes(), dtype=np.uint8).any()   # closed channel: zeros
    assert (again['prn'][:, 0] > 0).all()


@pytest.mark.parametrize('R', [1, 8, 32, 96])        # (96: past the correlator's switch to its batch form)
def test_batched_receivers_equal_their_solo_closed_loops(golden_default, R):
    """gpsmi_trk_set_streams: R independent IQ streams tracked by one handle, all channels of
    all streams in every launch trio.  Stream r must produce, byte for byte, the records and
    the state of a closed loop that runs alone on a handle of its own."""
    from gpsmi.engine import TrkEngine, DeviceBuffer, OUT_DTYPE
    g = golden_default
    nch, steps = 12, 6
    blocks = scene_blocks('default', 5, R + steps)
    # stream r = the recording from block r on: different samples, phases and delays per stream
    solo = np.zeros((R, steps, nch), dtype=OUT_DTYPE)
    solo_state = []
    for r in range(R):
        eng = TrkEngine(max_ch=nch)
        _open_all(eng, g)
        if r % 3 == 1:
            eng.close_channel(r % nch)             # (a stream with a closed channel among the others)
        for i in range(steps):
            solo[r, i] = eng.process(blocks[r + i])
        solo_state.append([eng.get_state(c).tobytes() for c in range(nch)])
        eng.close()
    eng = TrkEngine(max_ch=nch, streams=R)
    for r in range(R):
        for c, (sv, f0, d0) in enumerate(g['trk_init']):
            eng.open(c, int(sv), float(f0), int(d0), stream=r)
        if r % 3 == 1:
            eng.close_channel(r % nch, stream=r)
    buf = DeviceBuffer(R * blocks[0].nbytes)
    for i in range(steps):
        slab = np.stack([blocks[r + i] for r in range(R)])
        if i % 2 == 0:                               # host input ...
            out = eng.process(slab)
        else:                                        # ... and device-resident input
            buf.upload(slab)
            out = eng.process(buf.ptr)
        out = out.reshape(R, nch)
        for r in range(R):
            assert out[r].tobytes() == solo[r, i].tobytes(), (r, i)
    for r in range(R):
        assert [eng.get_state(c, stream=r).tobytes() for c in range(nch)] == solo_state[r], r
    buf.free()
    eng.close()
