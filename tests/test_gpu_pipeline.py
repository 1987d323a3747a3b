"""GPU: the block loop + hand-off (gpsmi.pipeline.Receiver, mirror of
gpsrecv.processData's data path, reference gpsrecv.py:445-548).  The datagrams of the
reference's own process are tests/test_handoff.py (ref_handoff.npz, 8 satellites); here the
12-satellite default scene (MAX_SAT = 11 of 12 selected) against oracle.process_data, which
that fixture pins bit for bit: same datagrams at the same blocks, same structure after
unpickling, numbers within the tracking tolerances.  Then the commands, skips, raw input and
the streamed forms."""
import pickle

import numpy as np
import pytest

import gps_oracle as orc
from conftest import scene_blocks

pytestmark = pytest.mark.gpu
N_BLOCKS = 45


def test_block_loop_and_hand_off():
    from gpsmi.pipeline import Receiver
    blocks = scene_blocks('default', 0, N_BLOCKS)
    ref = list(orc.process_data(blocks))      # (pinned to gpsrecv.main()'s own datagrams: test_handoff.py)
    rx = Receiver()
    got = []
    for i, b in enumerate(blocks):
        res = rx.feed(b)
        if res is not None:
            got.append((i, pickle.loads(res)))
    assert len(rx.act_sat_set) == 11                       # MAX_SAT of 12 acquired
    assert [i for i, _ in got] == [i for i, _ in ref] and len(got) >= 1
    for (_, (sk_a, fr_a, cp_a)), (_, (sk_b, fr_b, cp_b)) in zip(got, ref):
        assert sk_a == sk_b == 0
        assert [f['SAT'] for f in fr_a] == [f['SAT'] for f in fr_b]
        for fa, fb in zip(fr_a, fr_b):
            assert list(fa.keys()) == list(fb.keys())
            assert fa['SWP'] == fb['SWP']
            assert abs(float(fa['FRQ']) - float(fb['FRQ'])) < 0.05
            assert abs(float(fa['AMP']) - float(fb['AMP'])) < 2e-2
            assert abs(float(fa['CRM']) - float(fb['CRM'])) < 2e-2
        assert sorted(cp_a) == sorted(cp_b)
        for s in cp_a:
            assert [n for n, _ in cp_a[s]] == [n for n, _ in cp_b[s]]
            np.testing.assert_allclose([c for _, c in cp_a[s]], [c for _, c in cp_b[s]],
                                       atol=2e-3)
    # a SWEEP command restarts the cold search, STOP ends the run
    rx.command(b'SWEEP')
    assert rx.sweep_all_freq and rx.sat_lst == list(range(2, 33)) and rx.found_sats == []
    assert rx.feed(blocks[0]) is None
    rx.command(b'STOP')
    assert rx.running is False
    rx.close()


def test_skipped_streams_are_reported():
    from gpsmi.pipeline import Receiver
    blocks = scene_blocks('default', 0, 40)
    rx = Receiver()
    out = None
    for i, b in enumerate(blocks):
        res = rx.feed(b, skip=2 if i == 20 else 0)         # ring buffer lost two streams
        out = res or out
    skipped, frames, co_ph = pickle.loads(out)
    assert skipped == 2 * 65536
    assert int(rx.smp_time) == (40 + 2) * 65536
    rx.close()


def test_raw_u8_flow_equals_complex64_flow():
    """The reference's own path is file -> streamData's u8 decode -> processData (gpsrecv.py:153-186,
    :445-548).  Receiver(raw_u8=True) takes the recorder's uint16 blocks as they come off the
    file; cold search, channel selection, tracking, the per-channel re-sweep and the hand-off
    must produce byte-identical datagrams to the complex64 path (the decode inside the kernels
    is the bits of gpsmi_dev_unpack_u8iq / numpy's expression)."""
    from conftest import scene_for
    from gpsmi.pipeline import Receiver
    sc = scene_for('default')
    n = 40
    raw = [sc.block_raw(i) for i in range(n)]
    c64 = scene_blocks('default', 0, n)
    rx_a, rx_b = Receiver(), Receiver(raw_u8=True)
    n_out = 0
    for i in range(n):
        a, b = rx_a.feed(c64[i]), rx_b.feed(raw[i])
        assert (a is None) == (b is None), i
        if a is not None:
            assert a == b, i                              # the pickled datagrams, byte for byte
            n_out += 1
        if i == 30:                                       # a channel re-sweeps on the raw blocks too
            from gpsmi import receiver as R
            for rx in (rx_a, rx_b):
                sno = sorted(rx.act_sat_set)[0]
                R.initSweep(rx.pool, rx.pool_worker.index(sno))
    assert n_out >= 1 and rx_a.act_sat_set == rx_b.act_sat_set
    assert rx_a.found_sats == rx_b.found_sats
    with pytest.raises(TypeError):
        rx_b.feed(c64[0])                                 # the wrong format is refused, not cast
    rx_a.close()
    rx_b.close()


@pytest.mark.parametrize('inline_max', [None, '0'])
def test_streamed_blocks_equal_blocking_calls(monkeypatch, inline_max):
    """gpsmi_trk_process_stream: blocks from pinned host memory, no host wait per block; the
    upload in front of the block's own kernels on the main stream (the default for steps up to
    8 MiB) or, GPSMI_STREAM_INLINE_MAX=0, on the upload stream under the previous block's kernels
    (two staging blocks, events both ways).  Raw uint16 and complex64 input; the records of every
    block must equal those of the blocking gpsmi_trk_process, and a call returns only once the step
    before last is complete (the host never runs more than two steps ahead: three buffers suffice)."""
    if inline_max is not None:
        monkeypatch.setenv('GPSMI_STREAM_INLINE_MAX', inline_max)
    from conftest import load_golden, scene_for
    from gpsmi.engine import TrkEngine, PinnedArray, OUT_DTYPE
    g = load_golden('ref_default.npz')
    sc = scene_for('default')
    nch, n = len(g['trk_init']), 10
    for raw_u8 in (False, True):
        blocks = [sc.block_raw(5 + i) if raw_u8 else sc.block(5 + i) for i in range(n)]
        engs = [TrkEngine(max_ch=nch) for _ in range(2)]
        for e in engs:
            if raw_u8:
                e.set_input_format(True)
            for c, (sv, f0, d0) in enumerate(g['trk_init']):
                e.open(c, int(sv), float(f0), int(d0))
        want = [engs[0].process(b).tobytes() for b in blocks]
        pins = [PinnedArray(blocks[0].shape, blocks[0].dtype) for _ in range(3)]
        outs = [PinnedArray((nch,), OUT_DTYPE) for _ in range(n)]
        for i, b in enumerate(blocks):
            pins[i % 3].array[:] = b                      # (three buffers: one is never rewritten
            engs[1].process_stream(pins[i % 3].array, outs[i].array)   # while it may still be read)
            if i >= 2:      # the contract: this call returned, so the step before last is complete
                assert outs[i - 2].array.tobytes() == want[i - 2], (raw_u8, i, 'back-pressure')
        engs[1].wait()
        for i in range(n):
            assert outs[i].array.tobytes() == want[i], (raw_u8, i)
        for e in engs:
            e.close()
        for p_ in pins + outs:
            p_.free()


def test_streamed_input_ring_equals_blocking_calls():
    """gpsmi.ingest.StreamedInput: the GPU end of the reference's ring buffer -- raw blocks pulled
    from a RingBuffer, fed through three page-locked buffers without a host wait per block -- gives
    the records of the blocking calls, block for block; feed() hands back the records of the block
    before last, drain() the rest."""
    from conftest import load_golden, scene_for
    from gpsmi.engine import TrkEngine
    from gpsmi.ingest import RingBuffer, StreamedInput
    g = load_golden('ref_default.npz')
    sc = scene_for('default')
    nch, n = 4, 9
    raw = [sc.block_raw(5 + i) for i in range(n)]
    engs = [TrkEngine(max_ch=nch) for _ in range(2)]
    for e in engs:
        e.set_input_format(True)
        for c, (sv, f0, d0) in enumerate(g['trk_init'][:nch]):
            e.open(c, int(sv), float(f0), int(d0))
    want = np.stack([engs[0].process(b) for b in raw])
    rb = RingBuffer()
    for b in raw:
        rb.push(b)
    si = StreamedInput(engs[1], keep_outputs=True)
    early = []
    while True:
        data, skip = rb.pull()
        if len(data) == 0:
            break
        assert skip == 0
        r = si.feed(data)
        if r is not None:                      # (the records of the block fed two calls earlier)
            early.append(r)
    assert len(early) == n - 2
    got = np.concatenate([np.stack(early), si.drain()])
    assert si.drain() is None
    assert got.shape == (n, 1, nch) and got[:, 0].tobytes() == want.tobytes()
    si.free()
    for e in engs:
        e.close()
