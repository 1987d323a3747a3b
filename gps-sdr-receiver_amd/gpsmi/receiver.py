"""Host side of the tracking path: the reference's per-channel worker API
(``gpsrecv.runProc`` / ``initMultiProcPool`` / ``initPoolStreams`` /
``delPoolStreams`` / ``satCalc``, reference src/gpsrecv.py:300-417) and the
part of ``gpslib.SatStream`` that is control flow rather than array arithmetic
(reference src/gpslib.py:1095-1210, :1331-1339, :1394-1398, :1421-1436,
:1451-1492), on top of one ``TrkEngine`` per GPU.

The reference spawns one OS process per satellite and pickles each 512 KiB
block to every one of them.  Here all channels are slots of one engine handle
and ``satCalc`` is one ``gpsmi_trk_process`` call per block; the functions keep
the reference's names, arguments and return shapes so that ``processData``
(gpsrecv.py:445-548) can call them unchanged.

``HostChannel`` consumes one engine output record per block and reproduces
what ``SatStream.process`` returns: ``(SWEEP, frameLst, codePhase,
(CORR_Q, CORR_L))``.  The engine has already done demodDoppler, cacodeCorr,
fitCodePhase, decodeData's sums, the statistics and the PLL.
"""
import numpy as np

from . import navbits
from .acquisition import norm_max_corr
from .engine import AcqEngine, Config, TrkEngine, dumps_of

MIN_CORR_Q = -0.9                      # gpslib.py:1048


def fit_code_phase(lo, pk, hi, mx):
    """fitCodePhase (gpslib.py:1268-1290) on the three taps around the peak."""
    lo, pk, hi = float(lo), float(pk), float(hi)
    tri = 0.5 * (hi - lo) / (pk - hi) if lo > hi else 0.5 * (hi - lo) / (pk - lo)
    par = 0.5 * (hi - lo) / (2 * pk - hi - lo)
    return mx + 0.5 * (tri + par)


class HostChannel:
    """Control state of one SatStream (gpslib.py:1050-1091) outside the GPU."""

    def __init__(self, sat_no, freq, delay, cfg):
        self.cfg = cfg
        self.SAT_NO = sat_no
        self.EDGES = [0]
        self.GPSBITS = np.array([], dtype=np.int8)
        self.GPSBITS_ST = np.array([], dtype=np.int64)
        self.PHASE_LOCKED = False
        self.FREQ = freq
        self.DELAY = delay
        self.MS_TIME = 0
        self.SMP_TIME = 0
        self.NO_SEC = 1024 // cfg.n_cyc
        self.STD_DEV = 0.005
        self.AMPLITUDE = 0.0
        self.MAX_CORR = 0.0
        self.SWEEP = False
        self.PREV_STREAM_NO = 0
        self.PREV_SIGNAL = 0
        self.CORR_Q = 0
        self.CORR_L = 0
        self.CORRLST_NO = 60 * self.NO_SEC
        self.CORRLST = [0]
        self.REP_SWEEP = False
        self.nps = 0                        # len(PREV_SAMPLES) before the block
        self.FREQ_SAVE = freq
        self.DF_SAVE = [0.0]

    # ---- gpslib.py:1095-1120
    def erasePrevData(self):
        self.EDGES = [0]
        self.GPSBITS = np.array([], dtype=np.int8)
        self.GPSBITS_ST = np.array([], dtype=np.int64)
        self.nps = 0

    def setPhaseUnlocked(self):
        self.PHASE_LOCKED = False
        self.CORRLST = [0]
        self.MS_TIME = 0
        self.erasePrevData()

    # ---- gpslib.py:1124-1138
    def reportValues(self, frame_lst):
        for dct in frame_lst:
            dct['SAT'] = self.SAT_NO
            dct['AMP'] = self.AMPLITUDE
            dct['CRM'] = self.MAX_CORR
            dct['FRQ'] = self.FREQ
            dct['SWP'] = self.REP_SWEEP
        self.REP_SWEEP = False

    def checkCorrQuality(self):
        return len(self.CORRLST) >= self.CORRLST_NO and self.CORR_Q < MIN_CORR_Q

    def corrQuality(self, code_phase):      # gpslib.py:1331-1339
        self.CORRLST.append(-1 if code_phase < 0 else 1)
        if len(self.CORRLST) > self.CORRLST_NO:
            del self.CORRLST[0]
        return np.mean(self.CORRLST), np.mean(self.CORRLST[-self.NO_SEC:])

    # ---- edge detection of decodeData (gpslib.py:1394-1398, :1408-1436)
    def detect_edges(self, dumps, delay):
        cs = self.cfg.code_samples
        min_edge_amp = 3 * self.STD_DEV
        prev_sign = (2 * (len(self.EDGES) % 2) - 1) * self.EDGES[0]
        n1 = self.nps + delay
        if n1 == 0:
            n1 = cs
            st = self.SMP_TIME
        else:
            st = self.SMP_TIME + delay - cs
        n0 = 0
        for m in dumps:
            if self.PHASE_LOCKED:
                re = np.float32(m.real)
                sgn = np.sign(re)
                if self.EDGES[0] == 0:
                    self.EDGES[0] = sgn
                    prev_sign = sgn
                elif (sgn != prev_sign and prev_sign * self.PREV_SIGNAL > 0
                      and abs(re - self.PREV_SIGNAL) > min_edge_amp):
                    self.EDGES.append((self.MS_TIME, st + n0))
                    prev_sign = sgn
                self.PREV_SIGNAL = re
                self.MS_TIME += 1
            n0 = n1
            n1 += cs

    # ---- edge list -> 20-ms bits (gpslib.py:1451-1492)
    def logicalBits(self):
        bits, stamps = [], []
        sign = self.EDGES[0]
        if len(self.EDGES) > 2:
            t1, st1 = self.EDGES[1]
            for t2, st2 in self.EDGES[2:]:
                m, r = np.divmod(t2 - t1, 20)
                if r > 17:
                    m += 1
                if m > 0:
                    bits += [sign] * m
                    stamps += [st1] + [0] * (m - 1)
                t1, st1 = t2, st2
                sign = -sign
            self.EDGES = [sign, self.EDGES[-1]]
        return (np.asarray(bits, dtype=np.int8), np.asarray(stamps, dtype=np.int64))

    def evalEdges(self):
        frames = []
        if len(self.EDGES) > 2:
            bits, stamps = self.logicalBits()
            self.GPSBITS = np.append(self.GPSBITS, bits)
            self.GPSBITS_ST = np.append(self.GPSBITS_ST, stamps)
            frames, self.GPSBITS, self.GPSBITS_ST = navbits.eval_gps_bits(
                self.GPSBITS, self.GPSBITS_ST)              # gpslib.py:1504-1580
        return frames

    # ---- the tracking branch of process() after the GPU work (gpslib.py:1178-1208)
    def absorb(self, rec, smp_time):
        """rec: one gpsmi_trk_out record of this channel for the block."""
        self.SMP_TIME = smp_time
        stream_no = smp_time // self.cfg.ngps
        code_phase = float(rec['code_phase'])
        self.CORR_Q, self.CORR_L = self.corrQuality(code_phase)
        self.DELAY = int(rec['delay_used'])
        self.detect_edges(dumps_of(rec), self.DELAY)
        self.STD_DEV = rec['std_dev']
        self.AMPLITUDE = rec['amplitude']
        self.MAX_CORR = rec['norm_max_corr']
        frames, sweep = [], False
        if stream_no % self.NO_SEC == 0:
            if self.PHASE_LOCKED:
                frames = self.evalEdges()
            if len(frames) == 0:
                frames = [{}]
            self.reportValues(frames)
            sweep = self.checkCorrQuality()
        self.nps = int(rec['nps'])
        if not sweep:
            self.PHASE_LOCKED = bool(rec['phase_locked'])
            self.FREQ = rec['freq']
        return sweep, frames, code_phase


class GpuPool:
    """What ``initMultiProcPool`` returns as `pool`: one tracking engine and one
    acquisition engine (for per-channel re-sweeps) instead of a process list."""

    def __init__(self, pool_no, cfg=None, raw_u8=False, trk=None):
        self.cfg = cfg or Config()
        self.raw_u8 = bool(raw_u8)          # blocks are the recorder's uint16 samples (fused ingest)
        self.trk = trk if trk is not None else TrkEngine(self.cfg, max_ch=pool_no)   # (trk: tests)
        if self.raw_u8:
            self.trk.set_input_format(True)
        self.acq = None
        self.chan = [None] * pool_no        # HostChannel per worker slot

    def acq_engine(self):
        if self.acq is None:
            self.acq = AcqEngine(self.cfg)
            if self.raw_u8:
                self.acq.set_input_format(True)
        return self.acq

    def close(self):
        self.trk.close()
        if self.acq is not None:
            self.acq.close()


def initMultiProcPool(poolNo, cfg=None, raw_u8=False):
    """gpsrecv.py:340-360 -> (pool, poolNo, poolWorker); poolWorker[w] is 0 when
    slot w is free, else the PRN it tracks.  raw_u8: the blocks given to satCalc are the
    recorder's uint16 samples, decoded inside the kernels (2 B/sample over PCIe)."""
    return GpuPool(poolNo, cfg, raw_u8), poolNo, [0] * poolNo


def closeMultiProcPool(pool):               # gpsrecv.py:363-367
    pool.close()


def open_worker(pool, wno, sat_no, freq, delay):
    """('initInst',(satNo,freq,delay)) for worker slot wno (gpsrecv.py:312-321)."""
    pool.trk.open(wno, sat_no, freq, delay)
    pool.chan[wno] = HostChannel(sat_no, freq, delay, pool.cfg)
    return sat_no


def close_worker(pool, wno):
    """('delInst',None) (gpsrecv.py:323-328) -> whether an instance existed."""
    had = pool.chan[wno] is not None
    if had and not pool.chan[wno].SWEEP:
        pool.trk.close_channel(wno)
    pool.chan[wno] = None
    return had


def delPoolStreams(pool, poolNo, poolWorker, actSatSet, delSatSet):
    """gpsrecv.py:370-382"""
    for sat_no in delSatSet:
        wno = poolWorker.index(sat_no)
        close_worker(pool, wno)
        poolWorker[wno] = 0
    return poolWorker, actSatSet - delSatSet


def initPoolStreams(pool, poolNo, poolWorker, actSatSet, newSatSet, foundSats):
    """gpsrecv.py:385-401: ('initInst',(satNo,freq,delay)) for every new SV."""
    if len(newSatSet) > 0:
        for wno, sno in enumerate(poolWorker):
            if sno == 0:
                new_sat = newSatSet.pop()
                poolWorker[wno] = new_sat
                _, _, freq, delay = [e for e in foundSats if e[1] == new_sat][0]
                actSatSet.add(open_worker(pool, wno, new_sat, freq, delay))
                if len(newSatSet) == 0:
                    break
    return poolWorker, actSatSet


def _sweep_frequency(pool, hc, data):
    """sweepFrequency / getCorrMax (gpslib.py:1350-1380) through the acquisition
    engine: up to IT_SWEEP bins from hc.FREQ, stop at the first hit.  The
    reference's missing upper-frequency check inside the loop is kept."""
    cfg = pool.cfg
    freqs = [hc.FREQ + cfg.step_freq * j for j in range(cfg.it_sweep)]
    table, nbr = pool.acq_engine().search_ex(data, [hc.SAT_NO], freqs,
                                             cfg.sweep_corr_avg)
    sweeping, delay, co_ph, freq, nmc = True, -1, -1, hc.FREQ, 0.0
    for j, f in enumerate(freqs):
        cell = table[j, 0]
        nmc = norm_max_corr(cell)
        freq = f
        if nmc > cfg.corr_min:
            delay = int(cell['argmax'])
            co_ph = fit_code_phase(nbr[j, 0, 0], cell['peak'], nbr[j, 0, 1], delay)
            break
        freq = f + cfg.step_freq
    if delay >= 0:
        sweeping = False
    elif freq > cfg.max_freq:
        freq = cfg.min_freq
        sweeping = False
    return sweeping, freq, nmc, delay, co_ph


def initSweep(pool, wno, st=None):
    """SatStream.initSweep (gpslib.py:1110-1116) for the channel in worker slot wno:
    remember FREQ and DF (st: the engine state to fall back to, default the current
    one), unlock, restart at MIN_FREQ; the engine channel is closed until the sweep ends."""
    hc = pool.chan[wno]
    if st is None:
        st = pool.trk.get_state(wno)
    hc.setPhaseUnlocked()
    # FREQ is float32 once the PLL has run, a Python float right after initInst or a clamp
    hc.FREQ_SAVE = st['freq'] if st['omega0'] == 0 else float(st['freq'])
    hc.DF_SAVE = list(st['df'][:int(st['df_len'])])
    hc.FREQ = pool.cfg.min_freq
    hc.SWEEP = True
    pool.trk.close_channel(wno)


def satCalc(actSatSet, pool, poolWorker, data, smpTime, sweep=()):
    """gpsrecv.py:404-417 -> [(swFq, satNo, frameData, coPh, cpQ), ...] in the
    iteration order of actSatSet, one engine call for all tracking channels.
    sweep: satellites whose channel is to start a sweep with this block, the
    ``sweep=True`` argument of SatStream.process (gpslib.py:1141, :1147-1151)."""
    cfg = pool.cfg
    stream_no = smpTime // cfg.ngps
    order = [(sno, poolWorker.index(sno)) for sno in actSatSet]
    saved = {}
    for sno, wno in order:
        hc = pool.chan[wno]
        if stream_no - 1 != hc.PREV_STREAM_NO:          # a stream was skipped
            hc.erasePrevData()                          # gpslib.py:1143-1146
            if not hc.SWEEP:
                pool.trk.erase_prev(wno)
        hc.PREV_STREAM_NO = stream_no
        if sno in sweep and not hc.SWEEP:               # ignore the trigger if one is running
            initSweep(pool, wno)
        # the state a sweep trigger must fall back to (gpslib.py:1110-1116)
        if (not hc.SWEEP and stream_no % hc.NO_SEC == 0
                and len(hc.CORRLST) + 1 >= hc.CORRLST_NO):
            saved[wno] = pool.trk.get_state(wno)
    out = pool.trk.process(data) if any(not pool.chan[w].SWEEP for _, w in order) \
        else None
    res = []
    for sno, wno in order:
        hc = pool.chan[wno]
        if hc.SWEEP:                                    # gpslib.py:1153-1173
            hc.SMP_TIME = smpTime
            hc.REP_SWEEP = True
            hc.SWEEP, hc.FREQ, hc.MAX_CORR, delay, code_phase = \
                _sweep_frequency(pool, hc, data)
            hc.CORR_Q, hc.CORR_L = hc.corrQuality(code_phase)
            if delay >= 0:
                hc.DELAY = delay
            elif not hc.SWEEP:
                hc.FREQ = hc.FREQ_SAVE                  # restoreFreq
            if not hc.SWEEP:                            # back to tracking next block
                pool.trk.open(wno, sno, hc.FREQ, hc.DELAY)
                if delay < 0:                           # restoreFreq: FREQ_SAVE and DF_SAVE
                    st = pool.trk.get_state(wno)
                    n = len(hc.DF_SAVE)
                    st['df_len'] = n
                    st['df'][:n] = hc.DF_SAVE
                    if isinstance(hc.FREQ_SAVE, np.float32):
                        # FREQ is float32 again: the reference's next demodDoppler forms
                        # float32(2 pi) * FREQ in float32, which the kernel does for omega0 == 0
                        st['omega0'] = 0.0
                    pool.trk.set_state(wno, st)
            frames = []
            if stream_no % hc.NO_SEC == 0:
                frames = [{}]
                hc.reportValues(frames)
        else:
            trig, frames, code_phase = hc.absorb(out[wno], smpTime)
            if trig:                                    # initSweep (gpslib.py:1110-1116) with the
                initSweep(pool, wno, saved[wno])        # state before this block's PLL update
        res.append((hc.SWEEP, sno, frames, code_phase, (hc.CORR_Q, hc.CORR_L)))
    return res
