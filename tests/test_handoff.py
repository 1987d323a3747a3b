"""The hand-off datagrams of the reference's own receiver process
(tests/golden/ref_handoff.npz: gpsrecv.main() -- streamData reading a u8 recording,
processData with the spawn worker pool, pickle.dumps((skippedData, frameLst, coPhLst)),
SAVE_PICKLE -- run by oracle/make_golden.py handoff on the scene
gpsmi.synth_nav.handoff_scene()) against

  * the oracle's block loop (CPU): every datagram, key, type and value equal;
  * the product's gpsmi.pipeline.Receiver on the GPU (complex64 and raw u8 input, per-block
    feed and the pipelined run): same datagrams at the same blocks, same keys in the same
    order, integers and subframe fields exact, floats within the tracking tolerances.
"""
import hashlib
import json
import pickle

import numpy as np
import pytest

from conftest import load_golden

_CACHE = {}


def handoff_fixture():
    if 'fix' not in _CACHE:
        g = load_golden('ref_handoff.npz')
        _CACHE['fix'] = (json.loads(str(g['datagrams'])), int(g['n_blocks']), str(g['iq_sha256']))
    return _CACHE['fix']


def handoff_raw_blocks():
    """The recording the fixture was made from, regenerated from its seed (uint16 blocks)."""
    if 'raw' not in _CACHE:
        from gpsmi import synth_nav
        _, n_blocks, sha = handoff_fixture()
        scene, info, n = synth_nav.handoff_scene()
        assert n == n_blocks
        raw = [scene.block_raw(b) for b in range(n)]
        h = hashlib.sha256()
        for r in raw:
            h.update(r.tobytes())
        assert h.hexdigest() == sha, 'the scene generator drifted from the fixture'
        _CACHE['raw'] = raw
    return _CACHE['raw']


def decode(dg):
    """fixture entry -> (skipped, [[(key, typename, value)]], {sat: [(n, c)]}) """
    return (dg['skipped'][1],
            [[(k, t, v) for k, (t, v) in f] for f in dg['frames']],
            [(s, [(n[1], c[1]) for n, c in lst]) for s, lst in dg['coph']])


FLOAT_TOL = {'AMP': 2e-2, 'CRM': 2e-2, 'FRQ': 0.05}


def compare(got, ref_dgs, exact):
    """got: [(skipped, frameLst, coPhLst)] unpickled; ref_dgs: fixture entries."""
    assert len(got) == len(ref_dgs) and len(got) >= 10
    n_sub = 0
    for (sk, frames, coph), dg in zip(got, ref_dgs):
        r_sk, r_frames, r_coph = decode(dg)
        assert sk == r_sk
        assert len(frames) == len(r_frames)
        for f, rf in zip(frames, r_frames):
            assert list(f.keys()) == [k for k, _, _ in rf]          # same keys, same order
            n_sub += 'ID' in f
            for k, tname, v in rf:
                if exact:
                    assert type(f[k]).__name__ == tname, (k, type(f[k]), tname)
                    assert f[k] == v, (k, f[k], v)
                elif k in FLOAT_TOL:
                    assert abs(float(f[k]) - v) < FLOAT_TOL[k], (k, f[k], v)
                else:                                               # SAT, SWP, ID, tow, ST, ephemeris
                    assert f[k] == v, (k, f[k], v)
        assert list(coph.keys()) == [s for s, _ in r_coph]          # satellites in the same order
        for (s, r_lst) in r_coph:
            assert [n for n, _ in coph[s]] == [n for n, _ in r_lst]
            a, b = np.array([c for _, c in coph[s]]), np.array([c for _, c in r_lst])
            if exact:
                assert np.array_equal(a, b)
            else:
                np.testing.assert_allclose(a, b, atol=2e-3)
    assert n_sub >= 8                                               # real subframes went through
    return n_sub


def test_oracle_block_loop_equals_the_reference_process():
    """CPU: oracle.process_data (sweep, getNewSats, pool bookkeeping, SatStream.process,
    hand-off) reproduces every datagram of gpsrecv.main() bit for bit."""
    import gps_oracle as orc
    from gpsmi import synth
    ref_dgs, n_blocks, _ = handoff_fixture()
    raw = handoff_raw_blocks()
    got = list(orc.process_data(synth.raw_to_c64(r) for r in raw))
    assert [i for i, _ in got] == [32 * (k + 1) - 1 for k in range(len(ref_dgs))]
    compare([dg for _, dg in got], ref_dgs, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize('raw_u8', [False, True])
def test_receiver_feed_equals_the_reference_process(raw_u8):
    """GPU: gpsmi.pipeline.Receiver.feed block by block."""
    from gpsmi import synth
    from gpsmi.pipeline import Receiver
    ref_dgs, n_blocks, _ = handoff_fixture()
    raw = handoff_raw_blocks()
    rx = Receiver(raw_u8=raw_u8)
    got, at = [], []
    for i, r in enumerate(raw):
        res = rx.feed(r if raw_u8 else synth.raw_to_c64(r))
        if res is not None:
            got.append(pickle.loads(res))
            at.append(i)
    rx.close()
    assert at == [32 * (k + 1) - 1 for k in range(len(ref_dgs))]
    compare(got, ref_dgs, exact=False)


@pytest.mark.gpu
@pytest.mark.parametrize('lag', [3, 16])
def test_a_lagged_report_sends_the_same_datagrams_later(lag):
    """Receiver(report_lag=L): the once-a-second host work is done L blocks behind the report block,
    while the GPU works on the blocks queued meanwhile (L = 16: the caller runs up to 16 blocks ahead
    of the device, "stream_depth"); same batches, same datagrams."""
    from gpsmi.pipeline import Receiver
    ref_dgs, n_blocks, _ = handoff_fixture()
    raw = handoff_raw_blocks()
    rx = Receiver(raw_u8=True, report_lag=lag)
    at = []
    for i, r in enumerate(raw):
        if rx.feed(r) is not None:
            at.append(i)
    rx.drain()
    got = [pickle.loads(d) for d in rx.result_list]
    rx.close()
    steady = [i for i in at if i % 32 != 31]               # (a report inside the channels' sweeps is not lagged)
    assert steady and all(i % 32 == (31 + lag) % 32 for i in steady)
    assert len(got) == len(ref_dgs)
    compare(got, ref_dgs, exact=False)


@pytest.mark.gpu
def test_a_command_flushes_a_lagged_report():
    """A SWEEP command (or anything else that needs the host state) while a report is still outstanding:
    the pool absorbs the report's batch first, then the blocks behind it; the datagrams are those of the
    unlagged receiver given the same command at the same block."""
    from gpsmi.pipeline import Receiver
    raw = handoff_raw_blocks()[:80]

    def run(lag):
        rx = Receiver(raw_u8=True, report_lag=lag)
        for i, r in enumerate(raw):
            rx.feed(r)
            if i == 36:                                    # five blocks behind the report block 31
                rx.command(b'SWEEP')
        rx.drain()
        out = list(rx.result_list)                         # (the pickled datagrams themselves)
        rx.close()
        return out

    a, b = run(0), run(16)
    assert len(a) == len(b) >= 2
    assert a == b
