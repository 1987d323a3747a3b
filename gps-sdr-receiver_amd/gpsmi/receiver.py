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

``HostChannel`` consumes the engine's output records and reproduces what
``SatStream.process`` returns: ``(SWEEP, frameLst, codePhase, (CORR_Q, CORR_L))``.
The engine has already done demodDoppler, cacodeCorr, fitCodePhase, decodeData's
sums AND its edge scan (the sign / hysteresis rule of gpslib.py:1421-1436 runs in
the device epilogue, the record carries a bit mask of the dumps that are edges),
the statistics and the PLL; what is left here is list bookkeeping.

Nothing the host does between two report blocks (``streamNo % NO_SEC == 0``,
once a second) feeds back into the tracking loop -- the loop is closed on the
device -- so the records of up to a second of blocks can be absorbed in one
vectorised pass: ``satCalcLazy`` enqueues a block without waiting for it
(``gpsmi_trk_process_stream``: page-locked input ring, page-locked record ring)
and absorbs everything that is pending when a report block, a sweep, a stream gap
or any other pool operation needs the host state; ``satCalc`` is the same with a
batch of one (the reference's per-block call).
"""
from collections import deque
from itertools import islice

import numpy as np

from . import navbits
from .acquisition import norm_max_corr
from .engine import OUT_DTYPE, AcqEngine, Config, PinnedArray, TrkEngine, ptr

MIN_CORR_Q = -0.9                      # gpslib.py:1048
GPSMI_MAX_DUMPS = 33                   # N_CYC + 1 (gpsmi.h)


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
        self._corr_reset()
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

    def _corr_reset(self):
        # CORRLST (gpslib.py:1085-1087) as a bounded deque: append drops the oldest entry, which is
        # ``del CORRLST[0]`` of corrQuality; the sum of its entries is carried along
        self.CORRLST = deque([0], maxlen=self.CORRLST_NO)
        self._corr_sum = 0

    def setPhaseUnlocked(self):
        self.PHASE_LOCKED = False
        self._corr_reset()
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
        return self.corrQualityMany([-1 if code_phase < 0 else 1])

    def corrQualityMany(self, vals):
        """corrQuality for consecutive blocks (vals: +1 / -1 per block) -> (CORR_Q, CORR_L) after
        the last one: means over the last CORRLST_NO / NO_SEC entries.  Sums of +-1 are exact, so
        sum / len is np.mean's value bit for bit."""
        d = self.CORRLST
        n_drop = len(d) + len(vals) - d.maxlen
        if n_drop > 0:
            self._corr_sum -= sum(islice(d, 0, n_drop))
        d.extend(vals)
        self._corr_sum += sum(vals)
        m = min(len(d), self.NO_SEC)
        return self._corr_sum / len(d), sum(islice(reversed(d), 0, m)) / m

    # ---- edge list -> 20-ms bits (gpslib.py:1451-1492)
    def logicalBits(self):
        bits, stamps = [], []
        sign = self.EDGES[0]
        if len(self.EDGES) > 2:
            t1, st1 = self.EDGES[1]
            for t2, st2 in self.EDGES[2:]:
                m, r = divmod(t2 - t1, 20)
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
        recs = np.empty((1, 1), dtype=rec.dtype)
        recs[0, 0] = rec
        trig, frames, cps = self.absorb_many(recs[:, 0], [smp_time])
        return trig, frames, cps[0]

    def absorb_many(self, recs, smp_times):
        """recs: the records of this channel for K consecutive tracking blocks (a 1-D structured
        array), smp_times their SMP_TIMEs.  Only the last block may be a report block (the caller
        absorbs at every one).  -> (sweep trigger, frameLst of the last block, [codePhase] * K)."""
        x = extract_records(recs.reshape(len(recs), 1), [self], smp_times, self.cfg.code_samples)
        return self.absorb_extracted(x, 0, smp_times)

    def absorb_extracted(self, x, col, smp_times):
        """The same from the lists extract_records pulled out of a [K, channels] record array in
        one vectorised pass; col = this channel's column there."""
        cfg = self.cfg
        cs = cfg.code_samples
        cps = x.code_phase[col]
        self.CORR_Q, self.CORR_L = self.corrQualityMany(x.corr[col])
        # decodeData's edges (gpslib.py:1421-1436) from the device's scan: bit i of a record's mask
        # = dump i is an edge, at (MS_TIME before the block + i, ST + n0 of window i)
        lo, hi = x.ev_lo[col], x.ev_lo[col + 1]
        if self.EDGES[0] == 0 and x.sg_lo[col + 1] > x.sg_lo[col]:
            self.EDGES[0] = np.float32(x.sign0[x.sg_lo[col]])       # np.sign(m.real), :1424-1426
        if hi > lo:
            self.EDGES.extend(x.edges[lo:hi])
        self.MS_TIME += x.ms_total[col]
        self.SMP_TIME = smp_times[-1]
        stream_no = self.SMP_TIME // cfg.ngps
        self.DELAY = x.last_delay[col]
        self.STD_DEV = x.last_std[col]
        self.AMPLITUDE = x.last_amp[col]
        self.MAX_CORR = x.last_norm[col]
        if x.prev_locked is not None:                   # the lock flag and FREQ the report sees are
            self.PHASE_LOCKED = x.prev_locked[col]      # those before the last block's PLL (:1192-1208)
            self.FREQ = x.prev_freq[col]
        frames, sweep = [], False
        if stream_no % self.NO_SEC == 0:
            if self.PHASE_LOCKED:
                frames = self.evalEdges()
            if len(frames) == 0:
                frames = [{}]
            self.reportValues(frames)
            sweep = self.checkCorrQuality()
        self.nps = x.last_nps[col]
        if not sweep:
            self.PHASE_LOCKED = x.last_locked[col]
            self.FREQ = x.last_freq[col]
        return sweep, frames, cps


class Extracted:
    """Plain Python data of a [K blocks, W channels] record array (see extract_records)."""
    __slots__ = ('code_phase', 'corr', 'ms_total', 'ev_lo', 'edges', 'sg_lo', 'sign0', 'last_delay',
                 'last_std', 'last_amp', 'last_norm', 'last_nps', 'last_locked', 'last_freq',
                 'prev_locked', 'prev_freq', 'cp_array')


def extract_records(recs, chans, smp_times, cs):
    """One vectorised pass over recs [K, W] (K consecutive blocks, W channels; chans[w] the
    HostChannel of column w -- None for a free slot, whose records are all-zero -- for MS_TIME
    and len(PREV_SAMPLES) before the first block) -> the
    per-channel lists HostChannel.absorb_extracted consumes.  The numpy calls are per batch, not
    per channel: a dozen channels cost what one costs."""
    x = Extracted()
    k, w = recs.shape
    cp = recs['code_phase']
    x.cp_array = cp
    x.code_phase = cp.T.tolist()
    x.corr = np.where(cp < 0, -1, 1).T.tolist()        # corrQuality's +-1 per block (:1331-1333)
    msc = recs['ms_count']
    x.ms_total = msc.sum(axis=0).tolist()
    em, eh, s0 = recs['edge_mask'], recs['edge_mask_hi'], recs['edge_sign0']
    # every edge of the batch at once: (channel, block) pairs with a non-empty mask, then their set bits
    masks = em.T.astype(np.uint64) | (eh.T.astype(np.uint64) << np.uint64(32))          # [W, K]
    ev_w, ev_j = np.nonzero(masks)                      # by channel, then block
    x.edges, x.ev_lo = [], [0] * (w + 1)
    if len(ev_w):
        ms0 = np.array([int(hc.MS_TIME) if hc else 0 for hc in chans], dtype=np.int64)
        nps0 = np.array([hc.nps if hc else 0 for hc in chans], dtype=np.int64)
        bits = (masks[ev_w, ev_j][:, None] >> np.arange(GPSMI_MAX_DUMPS, dtype=np.uint64)[None, :]) & np.uint64(1)
        e_n, e_i = np.nonzero(bits)                     # event n, dump i (ascending within an event)
        bw, bj = ev_w[e_n], ev_j[e_n]
        ms = (ms0[None, :] + np.cumsum(msc, axis=0) - msc)[bj, bw] + e_i
        delay = recs['delay_used'][bj, bw].astype(np.int64)
        nps_before = np.where(bj > 0, recs['nps'][np.maximum(bj - 1, 0), bw], nps0[bw])
        smp = np.asarray(smp_times, dtype=np.int64)[bj]
        st = np.where(nps_before + delay == 0, smp, smp + delay - cs)        # ST of decodeData (:1408-1416)
        first = recs['first_len'][bj, bw].astype(np.int64)                   # n1 of the first window
        n0 = np.where(e_i > 0, first + (e_i - 1) * cs, 0)                    # n0 of window i
        x.edges = list(zip(ms.tolist(), (st + n0).tolist()))
        x.ev_lo = np.searchsorted(bw, np.arange(w + 1)).tolist()
    sg_w, sg_j = np.nonzero(s0.T)                       # blocks that stored a sign into EDGES[0]
    x.sg_lo = np.searchsorted(sg_w, np.arange(w + 1)).tolist()
    x.sign0 = s0.T[sg_w, sg_j].tolist()
    last = recs[k - 1]
    x.last_delay = last['delay_used'].tolist()
    x.last_nps = last['nps'].tolist()
    x.last_locked = (last['phase_locked'] != 0).tolist()
    x.last_std, x.last_amp = last['std_dev'].copy(), last['amplitude'].copy()   # (np.float32 elements)
    x.last_norm, x.last_freq = last['norm_max_corr'].copy(), last['freq'].copy()
    x.prev_locked = x.prev_freq = None
    if k > 1:
        prev = recs[k - 2]
        x.prev_locked = (prev['phase_locked'] != 0).tolist()
        x.prev_freq = prev['freq'].copy()
    return x


class Batch:
    """What absorbing K pending blocks yields: `res` = the reference's result list of the LAST
    block ([(swFq, satNo, frameData, coPh, cpQ)] in actSatSet order), and for the hand-off the
    code phases of all K blocks: `sats` (that order), `smp_times` [K], `code_phase` [K][len(sats)]."""

    def __init__(self, res, sats, smp_times, code_phase):
        self.res, self.sats, self.smp_times, self.code_phase = res, sats, smp_times, code_phase


class GpuPool:
    """What ``initMultiProcPool`` returns as `pool`: one tracking engine and one
    acquisition engine (for per-channel re-sweeps) instead of a process list."""

    RING = 64                               # record rows (blocks that may be pending at most)

    def __init__(self, pool_no, cfg=None, raw_u8=False, trk=None):
        self.cfg = cfg or Config()
        self.raw_u8 = bool(raw_u8)          # blocks are the recorder's uint16 samples (fused ingest)
        self.trk = trk if trk is not None else TrkEngine(self.cfg, max_ch=pool_no)   # (trk: tests)
        if self.raw_u8:
            self.trk.set_input_format(True)
        self.acq = None
        self.pool_no = pool_no
        self.chan = [None] * pool_no        # HostChannel per worker slot
        # blocks enqueued on the engine and not yet absorbed: [(smpTime, record row)], all for the
        # same channel order `pending_order` [(satNo, slot)]
        self.pending, self.pending_order = [], None
        self.done = []                      # batches absorbed on the side (by a pool operation
                                            # that needed the host state), for the next satCalcLazy
        self.streamed = hasattr(self.trk, 'process_stream_ptr')
        # the steady state of satCalcLazy: the stream number of the last block enqueued for exactly
        # `steady_act` / `steady_worker` with no channel in its sweep (None: look at every channel)
        self.steady_no, self.steady_act, self.steady_worker, self.steady_order = None, None, None, None
        self.corr_len_max = 0               # longest CORRLST among the channels of the last batch absorbed
        self.in_ring, self.out_ring, self.n_fed = None, None, 0
        self.rows = [None] * self.RING      # (a stand-in engine's records, tests)
        # report_lag = L > 0 (satCalcLazy): the host work of a report block is done L blocks later, when
        # the device has L newer blocks queued to work on meanwhile and (L >= the stream depth of 3) the
        # report block's records are complete without a wait.  Same batches, same datagrams, L blocks late.
        self.report_lag, self.depth = 0, 3
        self.report_at, self.report_age = None, 0    # pending entries up to a report block still to absorb

    def acq_engine(self):
        if self.acq is None:
            self.acq = AcqEngine(self.cfg)
            if self.raw_u8:
                self.acq.set_input_format(True)
        return self.acq

    # ---- enqueue one block for all open channels, no wait
    def submit(self, data, smp_time, order):
        if len(self.pending) == self.RING:
            self.absorb_pending()
        row = (self.pending[-1][1] + 1) % self.RING if self.pending else 0
        if self.streamed:
            want = np.uint16 if self.raw_u8 else np.complex64
            if self.in_ring is None:
                # D + 1 page-locked input buffers: with "stream_depth" = D gpsmi_trk_process_stream
                # returns once the step D calls back is done, so the buffer of the call D + 1 back
                # is free (gpsmi.h); the host prepares block k + 1 while block k is being enqueued
                # (D = 3), and with a lagged report it runs up to D blocks ahead of the device, which
                # then has work queued for the time the host spends on the report
                self.depth = min(max(3, self.report_lag), self.RING // 2)
                self.trk.set_option('stream_depth', self.depth)
                self.in_ring = [PinnedArray((self.cfg.ngps,), want) for _ in range(self.depth + 1)]
                self.out_ring = PinnedArray((self.RING, self.pool_no), OUT_DTYPE)
                # (the addresses the library is called with, made once: ctypes conversions cost
                # microseconds apiece)
                self.in_ptr = [ptr(p.array) for p in self.in_ring]
                self.out_ptr = [ptr(self.out_ring.array[r]) for r in range(self.RING)]
            data = np.asarray(data)
            if data.dtype != want:          # a silent cast would turn one format into garbage of the other
                raise TypeError(f'block dtype {data.dtype} does not match the input format '
                                f'({np.dtype(want).name})')
            j = self.n_fed % (self.depth + 1)
            buf = self.in_ring[j].array
            np.copyto(buf, data.reshape(buf.shape))
            self.trk.process_stream_ptr(self.in_ptr[j], self.out_ptr[row])
            self.n_fed += 1
        else:
            self.rows[row] = self.trk.process(data)
        self.pending.append((smp_time, row))
        self.pending_order = order
        self.last_stream_no = smp_time // self.cfg.ngps

    # ---- everything that is pending -> host state; the batch (two, if a report block is among them:
    # the one that ends with it, then the blocks behind it) goes to `done`
    def absorb_pending(self):
        if self.report_at:
            self._absorb(self.report_at, True)
        self.report_at = None
        if self.pending:
            self._absorb(len(self.pending), True)

    def report_tick(self, report):
        """satCalcLazy, behind a block enqueued the plain way: absorb what is due."""
        if report:
            if self.report_lag:
                if self.report_at:                    # (a lag of a second or more: the older one first)
                    self._absorb(self.report_at, True)
                self.report_at, self.report_age = len(self.pending), 0    # due `report_lag` blocks on
            else:
                self.absorb_pending()
        elif self.report_at:
            self.report_age += 1
            if self.report_age >= self.report_lag:
                # (lag >= the stream depth D: the library returned from the call D blocks behind the report
                # block's when that block's step was complete -- gpsmi.h, "stream_depth" -- nothing to wait for)
                self._absorb(self.report_at, not self.streamed or self.report_lag < self.depth)
                self.report_at = None

    def _absorb(self, k, wait):
        pend, self.pending = self.pending[:k], self.pending[k:]
        smp = [t for t, _ in pend]
        rows = [r for _, r in pend]
        if self.streamed:
            if wait:
                self.trk.wait()
            lo = rows[0]
            recs = (self.out_ring.array[lo:lo + k] if lo + k <= self.RING
                    else self.out_ring.array[rows])
        else:
            recs = np.stack([self.rows[r] for r in rows])
        order = self.pending_order
        if not self.pending:
            self.pending_order = None
        res, trig = [], []
        x = extract_records(recs, self.chan, smp, self.cfg.code_samples)   # (all worker slots: column = slot)
        cp = x.cp_array[:, [wno for _, wno in order]]
        stream_no = smp[-1] // self.cfg.ngps
        self.corr_len_max = 0
        for sno, wno in order:
            hc = self.chan[wno]
            hc.PREV_STREAM_NO = stream_no
            t, frames, cps = hc.absorb_extracted(x, wno, smp)
            if t:
                trig.append(wno)
            if len(hc.CORRLST) > self.corr_len_max:
                self.corr_len_max = len(hc.CORRLST)
            res.append([hc, sno, frames, cps[-1]])
        for wno in trig:                                # initSweep (gpslib.py:1110-1116) with the
            initSweep(self, wno, self.saved.pop(wno))   # state before this block's PLL update
        self.done.append(Batch([(hc.SWEEP, sno, fr, c, (hc.CORR_Q, hc.CORR_L)) for hc, sno, fr, c in res],
                               [sno for sno, _ in order], smp, cp))

    saved = None

    def take_done(self):
        out, self.done = self.done, []
        return out

    def close(self):
        self.pending = []
        if self.streamed and self.in_ring is not None:
            self.trk.wait()
            for p in self.in_ring + [self.out_ring]:
                p.free()
            self.in_ring = self.out_ring = None
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
    pool.absorb_pending()
    pool.steady_no = None
    pool.trk.open(wno, sat_no, freq, delay)
    pool.chan[wno] = HostChannel(sat_no, freq, delay, pool.cfg)
    return sat_no


def close_worker(pool, wno):
    """('delInst',None) (gpsrecv.py:323-328) -> whether an instance existed."""
    pool.absorb_pending()
    pool.steady_no = None
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
    pool.absorb_pending()
    pool.steady_no = None
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


def _sat_calc(actSatSet, pool, poolWorker, data, smpTime, sweep=()):
    """One block for every active satellite, waited for: tracking channels through the engine,
    sweeping ones through the acquisition engine -> Batch of one block."""
    pool.absorb_pending()                               # (earlier lazy blocks go to pool.done)
    pool.steady_no = None
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
    tracking = [(sno, wno) for sno, wno in order if not pool.chan[wno].SWEEP]
    tracked = {}
    if tracking:
        # The engine advances every open channel.  A call for a SUBSET of the instances (the
        # reference's workers are independent: a parent may run some of them on a block and
        # the others later) must leave the rest untouched: their state rows are put back.
        asked = {wno for _, wno in order}
        others = {w: pool.trk.get_state(w) for w, hc in enumerate(pool.chan)
                  if hc is not None and not hc.SWEEP and w not in asked}
        n_done = len(pool.done)
        pool.saved = saved
        pool.submit(data, smpTime, tracking)
        pool.absorb_pending()
        tracked = {r[1]: r for r in pool.done.pop(n_done).res}
        for w, st in others.items():
            pool.trk.set_state(w, st)
    res = []
    for sno, wno in order:
        hc = pool.chan[wno]
        if sno in tracked:
            res.append(tracked[sno])
            continue
        hc.SMP_TIME = smpTime                           # gpslib.py:1153-1173
        hc.REP_SWEEP = True
        hc.SWEEP, hc.FREQ, hc.MAX_CORR, delay, code_phase = \
            _sweep_frequency(pool, hc, data)
        hc.CORR_Q, hc.CORR_L = hc.corrQuality(code_phase)
        if delay >= 0:
            hc.DELAY = delay
        elif not hc.SWEEP:
            hc.FREQ = hc.FREQ_SAVE                      # restoreFreq
        if not hc.SWEEP:                                # back to tracking next block
            pool.trk.open(wno, sno, hc.FREQ, hc.DELAY)
            if delay < 0:                               # restoreFreq: FREQ_SAVE and DF_SAVE
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
        res.append((hc.SWEEP, sno, frames, code_phase, (hc.CORR_Q, hc.CORR_L)))
    return Batch(res, [sno for sno, _ in order], [smpTime], np.array([[r[3] for r in res]], dtype=np.float64))


def satCalc(actSatSet, pool, poolWorker, data, smpTime, sweep=()):
    """gpsrecv.py:404-417 -> [(swFq, satNo, frameData, coPh, cpQ), ...] in the
    iteration order of actSatSet, one engine call for all tracking channels.
    sweep: satellites whose channel is to start a sweep with this block, the
    ``sweep=True`` argument of SatStream.process (gpslib.py:1141, :1147-1151)."""
    return _sat_calc(actSatSet, pool, poolWorker, data, smpTime, sweep).res


def satCalcLazy(actSatSet, pool, poolWorker, data, smpTime):
    """satCalc without the wait: the block is enqueued behind the ones before it and the call
    returns the batches that became complete -- [] for most blocks, at a report block
    (``streamNo % NO_SEC == 0``, where the reference sends its datagram) one Batch holding that
    block's result list and the code phases of every block since the last one.  A block that needs
    a host decision (a channel in its sweep, a stream gap, a possible sweep trigger) takes the
    per-block path of satCalc; the results are the same either way."""
    cfg = pool.cfg
    stream_no = smpTime // cfg.ngps
    no_sec = 1024 // cfg.n_cyc
    report = stream_no % no_sec == 0
    # steady state: the block before went the plain way for the same satellites in the same worker
    # slots, so no channel is in its sweep and none has a gap; only a report block that could trigger
    # a sweep (a minute of correlation history, gpslib.py:1134-1138) needs a look at the channels
    if (pool.steady_no is not None and stream_no - 1 == pool.steady_no and actSatSet == pool.steady_act
            and poolWorker == pool.steady_worker
            and not (report and pool.corr_len_max + len(pool.pending) + 1 >= 60 * no_sec)):
        pool.saved = {}
        pool.submit(data, smpTime, pool.pending_order or pool.steady_order)
        pool.steady_no = stream_no
        pool.report_tick(report)
        return pool.take_done()
    order = [(sno, poolWorker.index(sno)) for sno in actSatSet]
    plain = len(order) > 0 and (pool.pending_order is None or pool.pending_order == order)
    if plain:
        n_pend = len(pool.pending)
        for _, wno in order:
            hc = pool.chan[wno]
            if (hc.SWEEP or stream_no - 1 != hc.PREV_STREAM_NO + n_pend
                    or (report and len(hc.CORRLST) + n_pend + 1 >= hc.CORRLST_NO)):
                plain = False
                break
    if not plain:
        b = _sat_calc(actSatSet, pool, poolWorker, data, smpTime)
        return pool.take_done() + [b]
    pool.saved = {}
    pool.submit(data, smpTime, order)
    pool.steady_no, pool.steady_act, pool.steady_worker = stream_no, set(actSatSet), list(poolWorker)
    pool.steady_order = order
    pool.corr_len_max = max(len(pool.chan[w].CORRLST) for _, w in order)
    pool.report_tick(report)
    return pool.take_done()
