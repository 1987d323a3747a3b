"""The reference's worker message protocol (gpsrecv.runProc, gpsrecv.py:300-337: initPool /
initInst / delInst / runInst / done) served by gpsmi.workers.WorkerLoop, driven through
queue.Queue by the parent-side functions of gpsrecv.py:340-417 and compared with the direct
calls of gpsmi.receiver on the same engine outputs.  No GPU: the engine is a stand-in that
hands out the reference's own per-block numbers from the golden fixture (as
tests/test_receiver_host.py does) and counts its calls."""
import numpy as np

from gpsmi import receiver as R
from gpsmi import workers as W
from gpsmi._lib import STATE_DTYPE
from gpsmi.engine import Config
from test_receiver_host import _record


class FixtureEngine:
    """TrkEngine stand-in with per-channel state: process() hands every open channel the fixture
    block ITS position points at and advances all of them, as the real engine advances every
    open channel's state; get_state / set_state carry the position (in `reserved`)."""

    def __init__(self, g, nch):
        self.g, self.nch = g, nch
        self.fix = {}                    # worker slot -> fixture channel
        self.pos = {}                    # worker slot -> next fixture block
        self.block = 0
        self.calls = 0

    def open(self, ch, prn, freq, delay, stream=0):
        init = self.g['trk_init']
        self.fix[ch] = [c for c in range(len(init)) if int(init[c, 0]) == prn][0]
        self.pos[ch] = self.block

    def close_channel(self, ch, stream=0):
        self.fix.pop(ch)
        self.pos.pop(ch)

    def erase_prev(self, ch, stream=0):
        pass

    def get_state(self, ch, stream=0):
        st = np.zeros(1, dtype=STATE_DTYPE)[0]
        st['reserved'] = self.pos.get(ch, 0)
        return st

    def set_state(self, ch, st, stream=0):
        self.pos[ch] = int(st['reserved'])

    def process(self, data, want_out=True):
        from gpsmi._lib import OUT_DTYPE
        out = np.zeros(self.nch, dtype=OUT_DTYPE)
        for ch, c in self.fix.items():
            out[ch] = _record(self.g, c, self.pos[ch])
            self.pos[ch] += 1
        self.block += 1
        self.calls += 1
        return out

    def close(self):
        pass


def _drive(g, api, make_pool, nb):
    cfg = Config()
    init = g['trk_init']
    found = [(20.0 - c, int(init[c, 0]), float(init[c, 1]), int(init[c, 2])) for c in range(len(init))]
    pool, pool_no, worker = make_pool(cfg)
    act = set()
    worker, act = api['init'](pool, pool_no, worker, act, {e[1] for e in found[:8]}, found)
    data = np.zeros(cfg.ngps, np.complex64)
    res = []
    for i in range(nb):
        smp = np.int64((5 + i + 1) * cfg.ngps)
        if i == 10:                      # one satellite leaves, two others join mid-run
            worker, act = api['dele'](pool, pool_no, worker, act, {found[2][1]})
            worker, act = api['init'](pool, pool_no, worker, act, {found[8][1], found[9][1]}, found)
        res.append(api['calc'](act, pool, worker, data, smp))
    return pool, worker, act, res


def test_worker_loop_matches_direct_calls(golden_default):
    g = golden_default
    nb = 40
    eng_a, eng_b = FixtureEngine(g, 11), FixtureEngine(g, 11)
    direct = dict(init=R.initPoolStreams, dele=R.delPoolStreams, calc=R.satCalc)
    pool_a, worker_a, act_a, res_a = _drive(
        g, direct, lambda cfg: (R.GpuPool(11, cfg, trk=eng_a), 11, [0] * 11), nb)
    queued = dict(init=W.q_initPoolStreams, dele=W.q_delPoolStreams, calc=W.q_satCalc)
    pool_b, worker_b, act_b, res_b = _drive(
        g, queued, lambda cfg: W.q_initMultiProcPool(11, cfg, pool=R.GpuPool(11, cfg, trk=eng_b)), nb)
    assert worker_a == worker_b and act_a == act_b
    assert len(res_a) == len(res_b) == nb
    n_frames = 0
    for ra, rb in zip(res_a, res_b):
        assert len(ra) == len(rb)
        for (sw_a, s_a, f_a, cp_a, q_a), (sw_b, s_b, f_b, cp_b, q_b) in zip(ra, rb):
            assert (sw_a, s_a, cp_a) == (sw_b, s_b, cp_b)
            assert tuple(map(float, q_a)) == tuple(map(float, q_b))
            assert [sorted(d.items(), key=str) for d in f_a] == [sorted(d.items(), key=str) for d in f_b]
            n_frames += len(f_a)
    assert n_frames > 0
    # all runInst of a block went into ONE engine call
    assert eng_b.calls == nb == eng_a.calls
    W.q_closeMultiProcPool(pool_b)
    assert not pool_b[0][2].thread.is_alive()


def test_messages_and_answers_have_the_reference_shapes(golden_default):
    import queue
    g = golden_default
    eng = FixtureEngine(g, 2)
    pairs = [(queue.Queue(), queue.Queue()) for _ in range(2)]
    loop = W.WorkerLoop(pairs, Config(), pool=R.GpuPool(2, Config(), trk=eng), grace=0.01)
    import threading
    th = threading.Thread(target=loop.run, daemon=True)
    th.start()
    (i0, o0), (i1, o1) = pairs
    i0.put(('initPool', 0))
    name, no = o0.get(timeout=5)
    assert isinstance(name, str) and no == 0
    i1.put(('delInst', None))
    assert o1.get(timeout=5) is False               # nothing to delete yet (gpsrecv.py:324-328)
    sv, f0, d0 = g['trk_init'][0]
    i0.put(('initInst', (int(sv), float(f0), int(d0))))
    assert o0.get(timeout=5) == int(sv)
    # a runInst burst that covers the only instance is answered at once ...
    i0.put(('runInst', (np.zeros(65536, np.complex64), np.int64(6 * 65536))))
    sw, sat, frames, co_ph, cp_q = o0.get(timeout=5)
    assert sat == int(sv) and sw is False and isinstance(frames, list) and len(cp_q) == 2
    # ... and a partial burst (two instances, one message) after the grace period
    sv1, f1, d1 = g['trk_init'][1]
    i1.put(('initInst', (int(sv1), float(f1), int(d1))))
    assert o1.get(timeout=5) == int(sv1)
    i1.put(('runInst', (np.zeros(65536, np.complex64), np.int64(7 * 65536))))
    late_first = o1.get(timeout=5)
    assert late_first[1] == int(sv1)
    # the other worker's message for the SAME block arrives late: its channel must not have been
    # advanced by the partial burst (and worker 1's not again by this one)
    i0.put(('runInst', (np.zeros(65536, np.complex64), np.int64(7 * 65536))))
    late = o0.get(timeout=5)
    assert late[1] == int(sv)
    assert eng.pos == {0: 2, 1: 2}                    # each channel saw two blocks, once each
    assert late[3] == float(g['trk_code_phase'][eng.fix[0], 1])
    assert late_first[3] == float(g['trk_code_phase'][eng.fix[1], 1])
    i0.put(('delInst', None))
    assert o0.get(timeout=5) is True
    for q in (i0, i1):
        q.put(('done', None))
    th.join(timeout=5)
    assert not th.is_alive()


def test_a_lagged_report_gives_the_same_batches_later(golden_default):
    """satCalcLazy with GpuPool.report_lag = L: the batch that ends with a report block is absorbed L
    blocks later (pipeline.Receiver(report_lag=L)); its content -- result list, code phases, frames --
    and everything after it is what the unlagged run gives."""
    g = golden_default
    cfg = Config()
    init = g['trk_init']
    found = [(20.0 - c, int(init[c, 0]), float(init[c, 1]), int(init[c, 2])) for c in range(len(init))]
    data = np.zeros(cfg.ngps, np.complex64)
    nb, first = 44, 20                       # stream numbers 21 .. 64: report blocks 32 and 64

    def run(lag):
        pool = R.GpuPool(11, cfg, trk=FixtureEngine(g, 11))
        pool.report_lag = lag
        worker, act = R.initPoolStreams(pool, 11, [0] * 11, set(), {e[1] for e in found[:8]}, found)
        out = []
        for i in range(nb):
            smp = np.int64((first + i + 1) * cfg.ngps)
            out += [(i, b) for b in R.satCalcLazy(act, pool, worker, data, smp)]
        pool.absorb_pending()
        out += [(nb, b) for b in pool.take_done()]
        return out

    plain, lagged = run(0), run(5)
    with_frames = lambda runs: [(i, b) for i, b in runs if any(r[2] for r in b.res)]
    fa, fb = with_frames(plain), with_frames(lagged)
    assert len(fa) == len(fb) >= 1
    assert [i for i, _ in fb][0] == [i for i, _ in fa][0] + 5          # the first report: five blocks later
    for (_, a), (_, b) in zip(fa, fb):
        assert a.sats == b.sats and list(a.smp_times) == list(b.smp_times)
        np.testing.assert_array_equal(a.code_phase, b.code_phase)
        assert len(a.res) == len(b.res)
        for (sw_a, s_a, f_a, cp_a, q_a), (sw_b, s_b, f_b, cp_b, q_b) in zip(a.res, b.res):
            assert (sw_a, s_a, cp_a) == (sw_b, s_b, cp_b)
            assert tuple(map(float, q_a)) == tuple(map(float, q_b))
            assert [sorted(d.items(), key=str) for d in f_a] == [sorted(d.items(), key=str) for d in f_b]
    # block for block the same code phases overall
    cat = lambda runs: np.concatenate([b.code_phase for _, b in runs])
    np.testing.assert_array_equal(cat(plain), cat(lagged))
