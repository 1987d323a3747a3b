"""GPU: the worker-pool mirror (gpsmi.receiver: initMultiProcPool /
initPoolStreams / satCalc / delPoolStreams, reference gpsrecv.py:340-417) driven
like processData drives the reference, against the reference fixture."""
import numpy as np
import pytest

from conftest import scene_blocks

pytestmark = pytest.mark.gpu


def test_pool_flow_matches_reference(golden_default):
    from gpsmi import receiver as R
    g = golden_default
    nch, nb = g['trk_delay'].shape
    found = [tuple(r) for r in g['sweep_found']]
    found = [(n, int(s), f, int(d)) for n, s, f, d in found]
    pool, pool_no, worker = R.initMultiProcPool(nch)
    act = set()
    new = {s for _, s, _, _ in found[:nch]}
    worker, act = R.initPoolStreams(pool, pool_no, worker, act, set(new), found)
    assert act == new and sorted(worker) == sorted(new)
    init_sv = [int(s) for s in g['trk_init'][:, 0]]
    blocks = scene_blocks('default', 5, nb)
    co_ph = {s: [] for s in act}
    for i in range(nb):
        smp = np.int64((5 + i + 1) * 65536)
        res = R.satCalc(act, pool, worker, blocks[i], smp)
        assert len(res) == nch
        for sw, sat, frames, cp, (cq, cl) in res:
            c = init_sv.index(sat)
            assert not sw
            ref_cp = g['trk_code_phase'][c, i]
            if ref_cp < 0:
                assert cp == -1.0
            else:
                assert abs(cp - ref_cp) < 2e-3
            assert float(cq) == g['trk_corr_q'][c, i]
            assert float(cl) == g['trk_corr_l'][c, i]
            hc = pool.chan[worker.index(sat)]
            assert hc.MS_TIME == g['trk_ms_time'][c, i], (c, i)
            assert len(hc.EDGES) == g['trk_n_edges'][c, i], (c, i)
            if (smp // 65536) % 32 == 0:
                assert len(frames) == 1 and frames[0]['SAT'] == sat
                assert abs(frames[0]['FRQ'] - g['trk_freq'][c, i - 1]) < 0.05
                assert abs(frames[0]['AMP'] - g['trk_amplitude'][c, i]) < 2e-2
            else:
                assert frames == []
            co_ph[sat].append(cp)
    # drop two satellites, as getNewSats would ask
    drop = set(list(act)[:2])
    worker, act = R.delPoolStreams(pool, pool_no, worker, act, drop)
    assert len(act) == nch - 2 and worker.count(0) == 2
    res = R.satCalc(act, pool, worker, blocks[0], np.int64((5 + nb + 1) * 65536))
    assert {r[1] for r in res} == act
    R.closeMultiProcPool(pool)


def test_stream_gap_resets_carry(golden_default):
    """gpslib.py:1143-1146: a skipped stream erases PREV_SAMPLES and the edges."""
    from gpsmi import receiver as R
    g = golden_default
    sv, f0, d0 = g['trk_init'][0]
    found = [(20.0, int(sv), float(f0), int(d0))]
    pool, n, worker = R.initMultiProcPool(1)
    worker, act = R.initPoolStreams(pool, n, worker, set(), {int(sv)}, found)
    blocks = scene_blocks('default', 5, 3)
    R.satCalc(act, pool, worker, blocks[0], np.int64(6 * 65536))
    assert pool.trk.get_state(0)['nps'] > 0
    # next call skips one stream number
    R.satCalc(act, pool, worker, blocks[2], np.int64(8 * 65536))
    hc = pool.chan[0]
    assert hc.PREV_STREAM_NO == 8
    rec_first_len = pool.trk.get_state(0)['nps']
    assert rec_first_len == 2048 - pool.trk.get_state(0)['delay']
    R.closeMultiProcPool(pool)


@pytest.mark.parametrize('case', ['two_calls', 'one_call', 'no_signal'])
def test_resweep_matches_reference(case):
    """Per-channel re-acquisition against the real SatStream (tests/golden/ref_resweep.npz,
    gpslib.py:1110-1120, :1153-1173, :1350-1380): tracking, a sweep triggered the way
    process(..., sweep=True) does, the sweep blocks through the acquisition engine, tracking
    again.  SWEEP flag, FREQ of the sweep blocks and DELAY bit-exact; codePhase atol 2e-3,
    MAX_CORR rtol 1e-3, FREQ of the tracking blocks atol 0.05 Hz."""
    from conftest import load_golden
    from gpsmi import receiver as R
    g = load_golden('ref_resweep.npz')
    sv, f0, d0, n_before, n_after = g[case + '_init']
    sv, d0, n_before, nb = int(sv), int(d0), int(n_before), int(n_before + n_after)
    first = int(g['first_block'])
    found = [(20.0, sv, float(f0), d0)]
    pool, n, worker = R.initMultiProcPool(1)
    worker, act = R.initPoolStreams(pool, n, worker, set(), {sv}, found)
    blocks = scene_blocks('default', first, nb)
    hc = pool.chan[0]
    for i in range(nb):
        smp = np.int64((first + i + 1) * 65536)
        res = R.satCalc(act, pool, worker, blocks[i], smp, sweep=({sv} if i == n_before else ()))
        sw, sno, frames, co_ph, (cq, cl) = res[0]
        where = f'{case} block {i}'
        assert bool(sw) == bool(g[case + '_sweep'][i]), where
        assert sno == sv
        assert int(hc.DELAY) == int(g[case + '_delay'][i]), where
        assert cq == g[case + '_corr_q'][i] and cl == g[case + '_corr_l'][i], where
        assert len(frames) == int(g[case + '_n_frames'][i]), where
        if frames:
            assert float(frames[0]['SWP']) == g[case + '_swp_reported'][i], where
        ref_cp = g[case + '_code_phase'][i]
        if ref_cp < 0:
            assert co_ph == -1, where
        else:
            assert abs(co_ph - ref_cp) < 2e-3, where
        # FREQ: a Python float while it is a bin frequency of the sweep (exact), float32 once
        # the PLL has run or restoreFreq has put FREQ_SAVE back (tracking tolerance)
        assert isinstance(hc.FREQ, np.float32) == bool(g[case + '_freq_is_f32'][i]), where
        if not g[case + '_freq_is_f32'][i]:
            assert float(hc.FREQ) == g[case + '_freq'][i], where
        else:
            assert abs(float(hc.FREQ) - g[case + '_freq'][i]) < 0.05, where
        np.testing.assert_allclose(hc.MAX_CORR, g[case + '_max_corr'][i], rtol=1e-3, err_msg=where)
        assert bool(hc.PHASE_LOCKED) == bool(g[case + '_locked'][i]), where
    R.closeMultiProcPool(pool)


def test_worker_message_loop_equals_direct_calls(golden_default):
    """gpsmi.workers: the reference's five worker messages (gpsrecv.runProc, gpsrecv.py:300-337)
    over queue.Queue, all runInst of a block batched into one gpsmi_trk_process, against the
    direct calls of gpsmi.receiver on the same blocks -- same tuples, including a channel that is
    put into its re-sweep (gpslib.py:1153-1173) and comes back."""
    from gpsmi import receiver as R
    from gpsmi import workers as W
    g = golden_default
    found = [(n, int(s), f, int(d)) for n, s, f, d in (tuple(r) for r in g['sweep_found'])]
    nch, nb = 6, 14
    blocks = scene_blocks('default', 5, nb)
    runs = []
    for api in ('direct', 'queues'):
        if api == 'direct':
            pool, pool_no, worker = R.initMultiProcPool(nch)
            init, calc, close = R.initPoolStreams, R.satCalc, R.closeMultiProcPool
            gpu_pool = pool
        else:
            gpu_pool = R.GpuPool(nch)
            pool, pool_no, worker = W.q_initMultiProcPool(nch, pool=gpu_pool)
            init, calc, close = W.q_initPoolStreams, W.q_satCalc, W.q_closeMultiProcPool
        worker, act = init(pool, pool_no, worker, set(), {s for _, s, _, _ in found[:nch]}, found)
        out = []
        for i in range(nb):
            if i == 4:                               # (between two blocks: the loop is idle)
                R.initSweep(gpu_pool, 0)
            out.append(calc(act, pool, worker, blocks[i], np.int64((5 + i + 1) * 65536)))
        runs.append((worker, act, out))
        close(pool)
    (w_a, act_a, out_a), (w_b, act_b, out_b) = runs
    assert w_a == w_b and act_a == act_b
    for ra, rb in zip(out_a, out_b):
        assert len(ra) == len(rb) == nch
        for (sw_a, s_a, f_a, cp_a, q_a), (sw_b, s_b, f_b, cp_b, q_b) in zip(ra, rb):
            assert (sw_a, s_a, cp_a) == (sw_b, s_b, cp_b)
            assert tuple(map(float, q_a)) == tuple(map(float, q_b))
            assert [sorted(d.items(), key=str) for d in f_a] == [sorted(d.items(), key=str) for d in f_b]
    # (the swept channel found its satellite again: a hit inside the first call ends the sweep at once)
    assert not any(r[0] for r in out_a[-1])
    assert out_a[4][[r[1] for r in out_a[4]].index(w_a[0])][3] != out_a[3][[r[1] for r in out_a[3]].index(w_a[0])][3]


def test_lazy_blocks_equal_per_block_calls(golden_default):
    """receiver.satCalcLazy (blocks enqueued on the device, the host absorbing a second of records
    at a time) against receiver.satCalc block by block on the same blocks: the result list of every
    report block, the code phases of every block, the edge lists and counters of every channel and
    the engine state at the end are the same."""
    from gpsmi import receiver as R
    g = golden_default
    nch, nb = 12, 80
    found = [(n, int(s), f, int(d)) for n, s, f, d in (tuple(r) for r in g['sweep_found'])]
    blocks = scene_blocks('default', 5, 48)
    runs = []
    for lazy in (False, True):
        pool, pool_no, worker = R.initMultiProcPool(nch)
        worker, act = R.initPoolStreams(pool, pool_no, worker, set(), {s for _, s, _, _ in found[:nch]}, found)
        res, cps = [], []
        for i in range(nb):
            smp = np.int64((5 + i + 1) * 65536)
            blk = blocks[i % 48]
            if lazy:
                for b in R.satCalcLazy(act, pool, worker, blk, smp):
                    if any(x[2] for x in b.res):
                        res.append(b.res)
                    cps.append(b.code_phase)
            else:
                r = R.satCalc(act, pool, worker, blk, smp)
                cps.append(np.array([[x[3] for x in r]]))
                if any(x[2] for x in r):
                    res.append(r)
        pool.absorb_pending()
        for b in pool.take_done():
            cps.append(b.code_phase)
        chans = [(hc.MS_TIME, list(hc.EDGES), hc.nps, hc.DELAY, bool(hc.PHASE_LOCKED), float(hc.FREQ),
                  list(hc.CORRLST)) for hc in pool.chan]
        states = [pool.trk.get_state(w).tobytes() for w in range(nch)]
        runs.append((res, np.concatenate(cps), chans, states, worker))
        R.closeMultiProcPool(pool)
    (res_a, cp_a, ch_a, st_a, w_a), (res_b, cp_b, ch_b, st_b, w_b) = runs
    assert w_a == w_b and len(res_a) == len(res_b) == 2          # blocks 32 and 64 of the stream
    assert cp_a.shape == cp_b.shape == (nb, nch) and np.array_equal(cp_a, cp_b)
    for ra, rb in zip(res_a, res_b):
        assert [(x[0], x[1], x[3], tuple(map(float, x[4]))) for x in ra] == \
               [(x[0], x[1], x[3], tuple(map(float, x[4]))) for x in rb]
        assert [[sorted(d.items(), key=str) for d in x[2]] for x in ra] == \
               [[sorted(d.items(), key=str) for d in x[2]] for x in rb]
    assert ch_a == ch_b
    assert st_a == st_b
    assert any(len(c[1]) > 3 for c in ch_a)                     # edges were found


def test_a_subset_call_leaves_the_other_channels_alone(golden_default):
    """The reference's workers are independent processes: running some of them on a block and the
    others later (a partial runInst burst) gives what running all of them at once gives.  Here the
    engine advances every open channel per call, so a call for a subset puts the others' state rows
    back (receiver._sat_calc)."""
    from gpsmi import receiver as R
    g = golden_default
    nch = 6
    found = [(n, int(s), f, int(d)) for n, s, f, d in (tuple(r) for r in g['sweep_found'])]
    blocks = scene_blocks('default', 5, 8)
    outs = []
    for split in (False, True):
        pool, pool_no, worker = R.initMultiProcPool(nch)
        worker, act = R.initPoolStreams(pool, pool_no, worker, set(), {s for _, s, _, _ in found[:nch]}, found)
        order = list(act)
        run = []
        for i in range(8):
            smp = np.int64((5 + i + 1) * 65536)
            if split and i in (3, 6):
                first = R.satCalc(order[:2], pool, worker, blocks[i], smp)
                late = R.satCalc(order[2:], pool, worker, blocks[i], smp)
                run.append(first + late)
            else:
                run.append(R.satCalc(order, pool, worker, blocks[i], smp))
        outs.append((run, [pool.trk.get_state(w).tobytes() for w in range(nch)]))
        R.closeMultiProcPool(pool)
    (run_a, st_a), (run_b, st_b) = outs
    for ra, rb in zip(run_a, run_b):
        assert [(x[0], x[1], x[3], tuple(map(float, x[4]))) for x in ra] == \
               [(x[0], x[1], x[3], tuple(map(float, x[4]))) for x in rb]
    assert st_a == st_b


@pytest.mark.parametrize('lazy', [False, True])
def test_bad_correlation_history_triggers_the_sweep(lazy):
    """checkCorrQuality (gpslib.py:1134-1138): at a report block, with a minute of correlation
    history whose mean is below -0.9, process() calls initSweep INSTEAD of the PLL -- FREQ_SAVE /
    DF_SAVE are the values before that block's update, PHASE_LOCKED / CORRLST / MS_TIME are reset,
    and the next blocks run sweepFrequency.  The history is planted (a minute of failed
    correlations is 1,920 blocks), the channel tracks a satellite that is not in the scene, and the
    same is done to the oracle's SatStream; per-block calls and the lazy path (whose steady state
    must notice that a trigger is possible and take the per-block route for that block)."""
    from collections import deque
    import gps_oracle as orc
    from gpsmi import receiver as R
    p = orc.Params()
    sv, f0, d0 = 3, 1234.5, 100                       # PRN 3 is not in the default scene
    first, nb = 58, 12                                # streams 59 .. 70: report block at 64
    blocks = scene_blocks('default', first, nb)
    pool, n, worker = R.initMultiProcPool(1)
    worker, act = R.initPoolStreams(pool, n, worker, set(), {sv}, [(20.0, sv, f0, d0)])
    ss = orc.SatStream(sv, f0, p, delay=d0)
    hc = pool.chan[0]
    planted = [-1] * (hc.CORRLST_NO - 3)
    hc.CORRLST = deque([0] + planted, maxlen=hc.CORRLST_NO)
    hc._corr_sum = -len(planted)
    ss.corrlst = [0] + planted
    got = []
    for i in range(nb):
        smp = np.int64((first + i + 1) * 65536)
        sw_ref, fl_ref, cp_ref, (cq_ref, cl_ref) = ss.process(blocks[i], smp)
        if lazy:
            batches = R.satCalcLazy(act, pool, worker, blocks[i], smp)
            if (first + i + 1) % 32 != 0 and not hc.SWEEP and i > 0:
                assert batches == [] or all(len(b.smp_times) == 1 for b in batches)
            pool.absorb_pending()
            batches += pool.take_done()
            res = batches[-1].res
        else:
            res = R.satCalc(act, pool, worker, blocks[i], smp)
        sw, sno, frames, co_ph, (cq, cl) = res[0]
        where = f'block {i} (stream {first + i + 1})'
        assert bool(sw) == bool(sw_ref), where
        assert float(cq) == float(cq_ref) and float(cl) == float(cl_ref), where
        assert len(frames) == len(fl_ref), where
        assert bool(hc.PHASE_LOCKED) == bool(ss.phase_locked), where
        assert hc.MS_TIME == ss.ms_time and len(hc.CORRLST) == len(ss.corrlst), where
        if sw_ref:
            assert float(hc.FREQ) == float(ss.freq), where          # a bin frequency of the sweep: exact
        got.append(bool(sw))
        if bool(sw) and not any(got[:-1]):                          # the trigger block itself
            assert (first + i + 1) % 32 == 0, where
            assert abs(float(hc.FREQ_SAVE) - float(ss.freq_save)) < 0.05, where
            assert len(hc.DF_SAVE) == len(ss.df_save), where
            np.testing.assert_allclose(hc.DF_SAVE, ss.df_save, atol=2e-3)
            assert frames and frames[0]['SWP'] is False             # reported before the trigger (:1190-1203)
    assert got.count(True) >= 2 and not got[0]                       # the sweep ran over several blocks
    R.closeMultiProcPool(pool)
