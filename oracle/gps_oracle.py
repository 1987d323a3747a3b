"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (numpy/scipy) of the one hot path of annappo/GPS-SDR-Receiver
that this repository accelerates: cold acquisition (``gpsrecv.sweepAllSats``)
and per-channel tracking (``gpslib.SatStream.process``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
path (``gps-sdr-receiver_amd/``) never imports it and has no CPU fallback.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference
from /root/reference (possible in the build container only) and freezes its
outputs on seeded synthetic IQ under ``tests/golden/``; ``tests/test_oracle.py``
checks every function below against those fixtures (bit-for-bit where the
expression is the same numpy expression, to 1e-12 otherwise).  The reference
holds no tests or golden vectors of its own.

All file:line citations are into /root/reference/src/.  The dtype chain is kept
on purpose: complex64 samples, float32 phase argument, complex64 carrier wipe
and forward FFT, complex128 replica spectrum / inverse FFT, float64
correlation statistics.  Scalars follow numpy >= 2 promotion
(NEP 50), the numpy of this image.
"""
from dataclasses import dataclass

import numpy as np
from scipy.fft import fft, ifft

F32 = np.float32
C64 = np.complex64


# --------------------------------------------------------------------------
# configuration (gpsglob.py:35-131) as a plain object instead of module globals
# --------------------------------------------------------------------------
@dataclass
class Params:
    code_samples: int = 2048        # gpsglob.py:119
    n_cyc: int = 32                 # gpsglob.py:122
    corr_avg: int = 8               # gpsglob.py:63
    corr_min: float = 8             # gpsglob.py:65
    sweep_corr_avg: int = 4         # gpsglob.py:67
    min_freq: float = -5000.0       # gpsglob.py:72
    max_freq: float = +5000.0       # gpsglob.py:73
    step_freq: float = 200          # gpsglob.py:74
    it_sweep: int = 40              # gpsglob.py:41
    it_sweep_all: int = 10          # gpsglob.py:42
    max_sat: int = 11               # gpsglob.py:38

    @property
    def sample_rate(self):          # gpsglob.py:121
        return 1000 * self.code_samples

    @property
    def ngps(self):                 # gpsglob.py:125
        return self.n_cyc * self.code_samples


def sec_time(p):
    """t[k] = (k+1)/fs as float32 (gpsrecv.py:32-33, gpslib.py:1053-1054)."""
    n = p.ngps
    return np.linspace(1, n, n, endpoint=True, dtype=F32) / p.sample_rate


# --------------------------------------------------------------------------
# C/A chips and the interpolated replica (cacodes.py:5-80, gpslib.py:62-87)
# --------------------------------------------------------------------------
_G2_SEL = [(2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9),
           (3, 10), (2, 3), (3, 4), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10),
           (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (1, 3), (4, 6),
           (5, 7), (6, 8), (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9),
           (5, 10), (4, 10), (1, 7), (2, 8), (4, 10)]


def chips(prn):
    """+-1 chips; integer-register form of the IS-GPS-200 generators.  The
    reference stores the same sequences as literals (cacodes.py:5-80)."""
    a, b = _G2_SEL[prn - 1]
    r1 = r2 = 0x3FF                       # bit s-1 holds stage s
    seq = np.empty(1023, dtype=np.int8)
    for n in range(1023):
        o = ((r1 >> 9) ^ (r2 >> (a - 1)) ^ (r2 >> (b - 1))) & 1
        seq[n] = 2 * o - 1
        f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1
        f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^ (r2 >> 8)
              ^ (r2 >> 9)) & 1
        r1 = ((r1 << 1) | f1) & 0x3FF
        r2 = ((r2 << 1) | f2) & 0x3FF
    return seq


def gps_cacode(prn, code_samples=2048):
    """GPSCacode (gpslib.py:73-77) on doubledCacode (gpslib.py:62-69)."""
    y = np.repeat(chips(prn), 2).astype(F32)
    x = np.arange(y.size, dtype=F32)
    xp = np.linspace(x[0], x[-1], code_samples, endpoint=True, dtype=F32)
    return np.interp(xp, x, y)


def gps_cacode_rep(prn, copies, delay, code_samples=2048):
    """GPSCacodeRep (gpslib.py:81-87)."""
    return np.roll(np.tile(gps_cacode(prn, code_samples), copies), delay)


def fft_cacode(prn, code_samples=2048):
    """gpsrecv.py:574-577, gpslib.py:1065."""
    return fft(gps_cacode(prn, code_samples))


# --------------------------------------------------------------------------
# acquisition (gpsrecv.py:217-274)
# --------------------------------------------------------------------------
def demod_doppler(data, freq, phase, n, t):
    """Carrier wipe-off over the first n samples (gpsrecv.py:232-235, identical
    in gpslib.py:1343-1346).  Returns (wiped complex64[n], new phase)."""
    factor = np.exp(-1j * (phase + 2 * np.pi * freq * t[:n]))
    phase = phase + 2 * np.pi * freq * t[n - 1]
    return factor * data[:n], np.remainder(phase, 2 * np.pi)


def peak_stats(corr):
    """mean, population std, first-index argmax (gpsrecv.py:218-223)."""
    mx = int(np.argmax(corr))
    return mx, corr[mx], np.mean(corr), np.std(corr)


def find_code_phase(corr, corr_min):
    """gpsrecv.py:217-227 -> (delay or -1, normMaxCorr)."""
    mx, peak, mean, std = peak_stats(corr)
    norm = (peak - mean) / std
    return (mx if norm > corr_min else -1), norm


def folded_spectrum(wiped, first, count, cs):
    """Mean of `count` consecutive code-period FFTs starting at period `first`
    (gpsrecv.py:250-254, gpslib.py:1316-1323)."""
    acc = 0
    for i in range(first, first + count):
        acc = acc + fft(wiped[i * cs:(i + 1) * cs])
    return acc / count


def circ_corr(spec, replica_spec):
    """|ifft(X * conj(R))| (gpsrecv.py:258, gpslib.py:1324-1325)."""
    return np.abs(ifft(spec * np.conjugate(replica_spec)))


def acq_table(data, freqs, prns, n_avg, p, t=None, spectra=None):
    """Every (Doppler bin, SV) cell of the search surface, without the
    first-hit bookkeeping: the array expressions of gpsrecv.py:249-259 for an
    arbitrary frequency list.  Returns dict of arrays [nbins, nsv]:
    argmax (int32), peak, mean, std (float64)."""
    cs = p.code_samples
    t = sec_time(p) if t is None else t
    if spectra is None:
        spectra = {s: fft_cacode(s, cs) for s in prns}
    nb, ns = len(freqs), len(prns)
    out = dict(argmax=np.zeros((nb, ns), np.int32), peak=np.zeros((nb, ns)),
               mean=np.zeros((nb, ns)), std=np.zeros((nb, ns)))
    for b, f in enumerate(freqs):
        wiped, _ = demod_doppler(data, f, 0, n_avg * cs, t)
        spec = folded_spectrum(wiped, 0, n_avg, cs)
        for j, s in enumerate(prns):
            mx, peak, mean, std = peak_stats(circ_corr(spec, spectra[s]))
            out['argmax'][b, j] = mx
            out['peak'][b, j] = peak
            out['mean'][b, j] = mean
            out['std'][b, j] = std
    return out


def sweep_all_sats(data, freq, sat_lst, sat_found, it_sweep, p, spectra, t):
    """sweepAllSats (gpsrecv.py:241-274): up to it_sweep bins per call, first
    bin over threshold claims the SV (first-hit), lists mutated in place."""
    ready = False
    avg = min(p.sweep_corr_avg, p.n_cyc)
    cs = p.code_samples
    it = 0
    while freq < p.max_freq and it < it_sweep:
        wiped, _ = demod_doppler(data, freq, 0, avg * cs, t)
        spec = folded_spectrum(wiped, 0, avg, cs)
        hit = []
        for s in sat_lst:
            delay, norm = find_code_phase(circ_corr(spec, spectra[s]),
                                          p.corr_min)
            if delay > -1:
                sat_found.append((norm, s, freq, delay))
                hit.append(s)
        for s in hit:
            sat_lst.remove(s)
        freq += p.step_freq
        if freq >= p.max_freq:
            ready = True
            freq -= p.max_freq - p.min_freq
        it += 1
    return ready, freq, sorted(sat_found, reverse=True)


def get_new_sats(act, found, cpq, max_sat):
    """getNewSats (gpsrecv.py:423-440)."""
    good = {s for s, (q, l) in cpq.items() if q > 0 or l > 0}
    rest = [e for e in found if e[1] not in good]
    want = good | {e[1] for e in rest[:max_sat - len(good)]}
    common = act & want
    return act - common, want - common


# --------------------------------------------------------------------------
# navigation bits: subframe check / extraction (gpslib.py:96-419)
# --------------------------------------------------------------------------
GPS_PI = 3.1415926535898                  # gpslib.py:16
PREAMBLE = np.array([1, 0, 0, 0, 1, 0, 1, 1], dtype=np.int8)      # gpslib.py:109


def bin_to_int(bits, signed=False):
    """BinToInt (gpslib.py:408-419)."""
    neg = signed and bits[0] == 1
    if neg:
        bits = 1 - bits
    z, f = 0, 1
    for b in reversed(bits):
        z += int(b) * f
        f *= 2
    return -(z + 1) if neg else z


def check_parity(words):
    """CheckParity (gpslib.py:379-405): index of the first failing word (1..9)
    or 0.  Words whose predecessor ends in D30* = 1 are complemented in place;
    word 0 is never checked."""
    taps = ((28, (0, 1, 2, 4, 5, 9, 10, 11, 12, 13, 16, 17, 19, 22)),
            (29, (1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23)),
            (28, (0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21)),
            (29, (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22)),
            (29, (0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23)),
            (28, (2, 4, 5, 7, 8, 9, 10, 12, 14, 18, 21, 22, 23)))
    for i in range(1, 10):
        if words[i - 1, 29] == 1:
            words[i, :24] = 1 - words[i, :24]
        d = words[i, :24]
        for k, (prev, idx) in enumerate(taps):
            bit = int(words[i - 1, prev])
            for j in idx:
                bit ^= int(d[j])
            if bit != words[i, 24 + k]:
                return i
    return 0


def extract_subframe(bits):
    """Subframe.Extract with getDataSub1..3 (gpslib.py:282-371) ->
    (status, dict); status codes as gpslib.py:97-108."""
    if len(bits) != 300:
        return 1, {}
    data = np.copy(bits)
    if not (data[:8] == PREAMBLE).all():
        data = 1 - data
        if not (data[:8] == PREAMBLE).all():
            return 2, {}
    w = np.reshape(data, (10, 30))
    if check_parity(w) > 0:
        return 3, {}
    f = {'tow': bin_to_int(w[1, :17]), 'ID': bin_to_int(w[1, 19:22])}
    if f['ID'] < 1 or f['ID'] > 5:
        return 4, {}
    cat = np.append
    if f['ID'] == 1:
        f['weekNum'] = bin_to_int(w[2, :10])
        f['satAcc'] = bin_to_int(w[2, 12:16])
        f['satHealth'] = bin_to_int(w[2, 16:22])
        f['IODC'] = bin_to_int(cat(w[2, 22:24], w[7, :8]))
        f['Tgd'] = bin_to_int(w[6, 16:24], True) * 2 ** (-31)
        f['Toc'] = bin_to_int(w[7, 8:24]) * 16
        f['af2'] = bin_to_int(w[8, 0:8], True) * 2.0 ** (-55)
        f['af1'] = bin_to_int(w[8, 8:24], True) * 2.0 ** (-43)
        f['af0'] = bin_to_int(w[9, 0:22], True) * 2.0 ** (-31)
    elif f['ID'] == 2:
        f['IODE2'] = bin_to_int(w[2, 0:8])
        f['Crs'] = bin_to_int(w[2, 8:24], True) * 2.0 ** (-5)
        f['deltaN'] = bin_to_int(w[3, 0:16], True) * 2.0 ** (-43) * GPS_PI
        f['M0'] = bin_to_int(cat(w[3, 16:24], w[4, 0:24]), True) * 2.0 ** (-31) * GPS_PI
        f['Cuc'] = bin_to_int(w[5, 0:16], True) * 2.0 ** (-29)
        f['e'] = bin_to_int(cat(w[5, 16:24], w[6, 0:24])) * 2 ** (-33)
        f['Cus'] = bin_to_int(w[7, 0:16], True) * 2.0 ** (-29)
        f['sqrtA'] = bin_to_int(cat(w[7, 16:24], w[8, 0:24])) * 2.0 ** (-19)
        f['Toe'] = bin_to_int(w[9, 0:16]) * 16
    elif f['ID'] == 3:
        f['Cic'] = bin_to_int(w[2, 0:16], True) * 2.0 ** (-29)
        f['omegaBig'] = bin_to_int(cat(w[2, 16:24], w[3, 0:24]), True) * 2.0 ** (-31) * GPS_PI
        f['Cis'] = bin_to_int(w[4, 0:16], True) * 2.0 ** (-29)
        f['i0'] = bin_to_int(cat(w[4, 16:24], w[5, 0:24]), True) * 2.0 ** (-31) * GPS_PI
        f['Crc'] = bin_to_int(w[6, 0:16], True) * 2.0 ** (-5)
        f['omegaSmall'] = bin_to_int(cat(w[6, 16:24], w[7, 0:24]), True) * 2.0 ** (-31) * GPS_PI
        f['omegaDot'] = bin_to_int(w[8, 0:24], True) * 2.0 ** (-43) * GPS_PI
        f['IDOT'] = bin_to_int(w[9, 8:22], True) * 2.0 ** (-43) * GPS_PI
        f['IODE3'] = bin_to_int(w[9, 0:8])
    return 0, f


_FRAME_KEYS = {
    1: ('ID', 'tow', 'weekNum', 'satAcc', 'satHealth', 'Tgd', 'IODC', 'Toc', 'af2', 'af1',
        'af0'),
    2: ('ID', 'tow', 'Crs', 'deltaN', 'M0', 'Cuc', 'IODE2', 'e', 'Cus', 'sqrtA', 'Toe'),
    3: ('ID', 'tow', 'Cic', 'omegaBig', 'Cis', 'i0', 'IODE3', 'Crc', 'omegaSmall',
        'omegaDot', 'IDOT'),
    4: ('ID', 'tow'), 5: ('ID', 'tow'),
}


def eval_gps_bits(gps_bits, stamps):
    """evalGpsBits (gpslib.py:1504-1580)."""
    result = []
    if len(gps_bits) < 300:
        return result, gps_bits, stamps
    gb = np.copy(gps_bits)
    pre = np.array([1, -1, -1, -1, 1, -1, 1, 1], dtype=np.int8)       # gpslib.py:1045
    corr = np.correlate(gb, pre, mode='same')
    loc = [i - 4 for i in range(len(corr)) if abs(corr[i]) == 8]
    start = 0
    if len(loc) > 0:
        gb[gb == -1] = 0
        lp = 0
        start = loc[lp]
        ok = True
        while ok and start + 300 < len(gb):
            status, f = extract_subframe(gb[start:start + 300])
            if status == 0:
                res = {k: f[k] for k in _FRAME_KEYS[f['ID']]}
                res['ST'] = stamps[start]
                result.append(res)
                start += 300
            else:
                ok = False
                while not ok and lp < len(loc) - 1:
                    lp += 1
                    s = loc[lp]
                    ok = s > start
                if ok:
                    start = s
    return result, gps_bits[start:], stamps[start:]


# --------------------------------------------------------------------------
# tracking (gpslib.py:1044-1446, signal part)
# --------------------------------------------------------------------------
def fit_code_phase(corr, mx):
    """fitCodePhase (gpslib.py:1268-1290): mean of a triangle and a parabola
    fit through the peak and its circular neighbours."""
    n = len(corr)
    lo = corr[mx - 1 if mx > 0 else n - 1]
    hi = corr[mx + 1 if mx < n - 1 else 0]
    pk = corr[mx]
    tri = 0.5 * (hi - lo) / (pk - (hi if lo > hi else lo))
    par = 0.5 * (hi - lo) / (2 * pk - hi - lo)
    return mx + 0.5 * (tri + par)


class SatStream:
    """Numeric and control restatement of gpslib.SatStream (gpslib.py:1044-1446)
    including the edge list, its slicing into 20-ms bits and subframe
    extraction (evalGpsBits / Subframe)."""
    DF_GAIN1 = 10                        # gpslib.py:1046
    DF_GAIN2 = 1                         # gpslib.py:1047
    MIN_CORR_Q = -0.9                    # gpslib.py:1048

    def __init__(self, sat_no, freq, p=None, delay=0):
        p = Params() if p is None else p
        self.p = p
        self.sat_no = sat_no
        self.t = sec_time(p)
        self.edges = [0]
        self.gpsbits = np.array([], dtype=np.int8)
        self.gpsbits_st = np.array([], dtype=np.int64)
        self.phase_locked = False
        self.phase = 0.0
        self.freq = freq
        self.prev_samples = []
        self.ms_time = 0
        self.smp_time = 0
        self.spectrum = fft_cacode(sat_no, p.code_samples)
        self.delay = delay
        self.no_sec = 1024 // p.n_cyc
        self.corr_avg = min(p.corr_avg, p.n_cyc)
        self.code_rep = gps_cacode_rep(sat_no, p.n_cyc, 0, p.code_samples)
        self.std_dev = 0.005
        self.amplitude = 0.0
        self.max_corr = 0.0
        self.sweep = False
        self.prev_stream_no = 0
        self.prev_signal = 0
        self.df = [0]
        self.corr_q = 0
        self.corr_l = 0
        self.corrlst_no = 60 * self.no_sec
        self.corrlst = [0]
        self.rep_sweep = False
        self.last = {}                   # per-block intermediates for tests

    # -- state resets (gpslib.py:1095-1120)
    def erase_prev_data(self):
        self.edges = [0]
        self.gpsbits = np.array([], dtype=np.int8)
        self.gpsbits_st = np.array([], dtype=np.int64)
        self.prev_samples = []

    def set_phase_unlocked(self):
        self.phase_locked = False
        self.corrlst = [0]
        self.ms_time = 0
        self.phase = 0.0
        self.erase_prev_data()

    def init_sweep(self):
        self.set_phase_unlocked()
        self.freq_save = self.freq
        self.df_save = self.df.copy()
        self.freq = self.p.min_freq
        self.df = [0]
        self.sweep = True

    def restore_freq(self):
        self.freq = self.freq_save
        self.df = self.df_save.copy()

    def report_values(self, frames):     # gpslib.py:1124-1131
        for d in frames:
            d['SAT'] = self.sat_no
            d['AMP'] = self.amplitude
            d['CRM'] = self.max_corr
            d['FRQ'] = self.freq
            d['SWP'] = self.rep_sweep
        self.rep_sweep = False

    def check_corr_quality(self):        # gpslib.py:1134-1138
        return (len(self.corrlst) >= self.corrlst_no
                and self.corr_q < self.MIN_CORR_Q)

    # -- correlation (gpslib.py:1293-1327)
    def find_code_phase(self, corr):
        mx, peak, mean, std = peak_stats(corr)
        norm = (peak - mean) / std
        if norm > self.p.corr_min:
            return mx, fit_code_phase(corr, mx), norm
        return -1, -1.0, norm

    def cacode_corr(self, wiped, corr_avg):
        cs = self.p.code_samples
        first = (len(wiped) // cs - corr_avg) // 2
        spec = folded_spectrum(wiped, first, corr_avg, cs)
        corr = circ_corr(spec, self.spectrum)
        return (corr,) + self.find_code_phase(corr)

    def corr_quality(self, code_phase):  # gpslib.py:1331-1339
        self.corrlst.append(-1 if code_phase < 0 else 1)
        if len(self.corrlst) > self.corrlst_no:
            del self.corrlst[0]
        return np.mean(self.corrlst), np.mean(self.corrlst[-self.no_sec:])

    # -- per-channel re-acquisition (gpslib.py:1350-1380)
    def get_corr_max(self, data, corr_avg, freq):
        wiped, _ = demod_doppler(data, freq, 0, corr_avg * self.p.code_samples,
                                 self.t)
        corr, delay, co_ph, norm = self.cacode_corr(wiped, corr_avg)
        return delay, (corr[delay] if delay > -1 else 0), co_ph, norm

    def sweep_frequency(self, data, freq):
        sweeping, j, delay, co_ph = True, 0, -1, -1
        while delay < 0 and j < self.p.it_sweep:
            delay, _, co_ph, norm = self.get_corr_max(
                data, self.p.sweep_corr_avg, freq)
            if delay < 0:
                freq = freq + self.p.step_freq
            j += 1
        if delay >= 0:
            sweeping = False
        elif freq > self.p.max_freq:
            freq = self.p.min_freq
            sweeping = False
        return sweeping, freq, norm, delay, co_ph

    # -- prompt integrate-and-dump (gpslib.py:1394-1446)
    def decode_data(self, wiped, delay):
        p = self.p
        cs, ngps = p.code_samples, p.ngps
        min_edge_amp = 3 * self.std_dev
        prev_sign = (2 * (len(self.edges) % 2) - 1) * self.edges[0]
        y = np.roll(self.code_rep, delay) * wiped
        nps = len(self.prev_samples)
        if nps > 0:
            y = np.append(self.prev_samples, y)
        ns = ngps + nps
        n0, n1 = 0, nps + delay
        if n1 == 0:
            n1 = cs
            st = self.smp_time
        else:
            st = self.smp_time + delay - cs
        dumps = []
        while n1 <= ns:
            m = np.mean(y[n0:n1])
            dumps.append(m)
            if self.phase_locked:
                sgn = np.sign(m.real)
                if self.edges[0] == 0:
                    self.edges[0] = sgn
                    prev_sign = sgn
                elif (sgn != prev_sign and prev_sign * self.prev_signal > 0
                      and abs(m.real - self.prev_signal) > min_edge_amp):
                    self.edges.append((self.ms_time, st + n0))
                    prev_sign = sgn
                self.prev_signal = m.real
                self.ms_time += 1
            n0 = n1
            n1 += cs
        self.prev_samples = y[n0:ns]
        return np.asarray(dumps, dtype=C64)

    # -- edge list -> 20-ms bits (gpslib.py:1451-1492)
    def logical_bits(self):
        bits, stamps = [], []
        sign = self.edges[0]
        if len(self.edges) > 2:
            t1, st1 = self.edges[1]
            for t2, st2 in self.edges[2:]:
                m, r = np.divmod(t2 - t1, 20)
                if r > 17:
                    m += 1
                if m > 0:
                    bits += [sign] * m
                    stamps += [st1] + [0] * (m - 1)
                t1, st1 = t2, st2
                sign = -sign
            self.edges = [sign, self.edges[-1]]
        return (np.asarray(bits, dtype=np.int8),
                np.asarray(stamps, dtype=np.int64))

    def eval_edges(self):
        frames = []
        if len(self.edges) > 2:
            bits, stamps = self.logical_bits()
            self.gpsbits = np.append(self.gpsbits, bits)
            self.gpsbits_st = np.append(self.gpsbits_st, stamps)
            frames, self.gpsbits, self.gpsbits_st = self.eval_gps_bits(
                self.gpsbits, self.gpsbits_st)
        return frames

    def eval_gps_bits(self, bits, stamps):
        return eval_gps_bits(bits, stamps)

    # -- PLL (gpslib.py:1215-1262)
    def phase_locked_loop(self, dumps):
        max_df = 20 / self.no_sec
        locked = self.phase_locked
        ph = np.arctan(dumps.imag / dumps.real)
        dp = 0
        real = np.copy(ph)
        for i in range(1, len(dumps)):
            d = ph[i] - ph[i - 1]
            if abs(d) > 2.0:
                dp -= np.sign(d)
            real[i] += dp * np.pi
        offset = np.mean(real[-4:])
        dev = np.mean(real)
        if locked:
            df = self.DF_GAIN2 * dev + np.mean(self.df)
            if abs(df) > max_df:
                df = np.sign(df) * max_df
            if len(self.df) >= self.no_sec:
                del self.df[0]
            self.df.append(df)
        else:
            df = self.DF_GAIN1 * dev
            self.df = [df]
        if abs(dev) < 0.1:
            locked = True
        return df, offset, locked, real

    def confine(self, f):                # gpslib.py:1626-1631
        if f > self.p.max_freq:
            return self.p.max_freq
        if f < self.p.min_freq:
            return self.p.min_freq
        return f

    # -- one block (gpslib.py:1141-1210)
    def process(self, data, smp_time, sweep=False):
        p = self.p
        self.smp_time = smp_time
        stream_no = smp_time // p.ngps
        if stream_no - 1 != self.prev_stream_no:
            self.erase_prev_data()
        self.prev_stream_no = stream_no
        sweep = sweep and not self.sweep
        if sweep:
            self.init_sweep()
        if self.sweep:
            self.rep_sweep = self.sweep
            self.sweep, self.freq, self.max_corr, delay, code_phase = \
                self.sweep_frequency(data, self.freq)
            self.corr_q, self.corr_l = self.corr_quality(code_phase)
            if delay >= 0:
                self.delay = delay
            elif not self.sweep:
                self.restore_freq()
            frames = []
            if stream_no % self.no_sec == 0:
                frames = [{}]
                self.report_values(frames)
            self.last = dict(mode='sweep', delay=delay, code_phase=code_phase)
        else:
            wiped, self.phase = demod_doppler(data, self.freq, self.phase,
                                              p.ngps, self.t)
            corr, delay, code_phase, norm = self.cacode_corr(wiped,
                                                             self.corr_avg)
            self.corr_q, self.corr_l = self.corr_quality(code_phase)
            if delay >= 0:
                self.delay = delay
            dumps = self.decode_data(wiped, self.delay)
            self.std_dev = np.std(np.abs(dumps))
            self.amplitude = np.mean(np.abs(dumps)) / self.std_dev
            self.max_corr = norm
            frames = []
            if stream_no % self.no_sec == 0:
                if self.phase_locked:
                    frames = self.eval_edges()
                if len(frames) == 0:
                    frames = [{}]
                self.report_values(frames)
                sweep = self.check_corr_quality()
            mx = int(np.argmax(corr))
            n = len(corr)
            self.last = dict(mode='track', dumps=dumps, mx=mx,
                             epl=np.array([corr[(mx - 1) % n], corr[mx],
                                           corr[(mx + 1) % n]]),
                             corr_mean=np.mean(corr), corr_std=np.std(corr),
                             delay=delay, code_phase=code_phase, norm=norm,
                             nps=len(self.prev_samples))
            if sweep:
                self.init_sweep()
            else:
                df, shift, self.phase_locked, _ = self.phase_locked_loop(dumps)
                self.phase += shift
                self.freq = self.confine(self.freq + df)
                self.last.update(df=df, shift=shift)
        return self.sweep, frames, code_phase, (self.corr_q, self.corr_l)


# --------------------------------------------------------------------------
# the block loop and the hand-off (gpsrecv.py:370-417, :445-519)
# --------------------------------------------------------------------------
def process_data(blocks, p=None, sat_all=None):
    """processData's data path (gpsrecv.py:466-519) over an iterable of complex64 blocks,
    with the pool bookkeeping of delPoolStreams / initPoolStreams / satCalc (:370-417) kept
    as the reference has it (worker slots filled in slot order from ``newSatSet.pop()``,
    results in the iteration order of ``actSatSet``): yields ``(block_index, (skippedData,
    frameLst, coPhLst))`` for every block on which the reference sends a datagram."""
    p = p or Params()
    t = sec_time(p)
    sat_all = list(range(2, 33)) if sat_all is None else list(sat_all)   # gpsrecv.py:36
    spectra = {s: fft_cacode(s, p.code_samples) for s in sat_all}
    sat_lst, found, freq = sat_all.copy(), [], p.min_freq
    sweeping, smp_time = True, np.int64(0)
    pool_worker = [0] * p.max_sat                       # initMultiProcPool (:358)
    streams = [None] * p.max_sat
    act = set()
    co_ph_lst, cp_q_lst, skipped = {}, {}, 0
    for i, data in enumerate(blocks):
        smp_time += p.ngps                              # :471 (no skips in file mode)
        if sweeping:
            ready, freq, found = sweep_all_sats(data, freq, sat_lst, found,
                                                p.it_sweep_all, p, spectra, t)
            if ready:
                sweeping = False
                dele, new = get_new_sats(act, found, cp_q_lst, p.max_sat)
                for s in dele:                          # delPoolStreams (:370-382)
                    w = pool_worker.index(s)
                    streams[w] = None
                    pool_worker[w] = 0
                act = act - dele
                if len(new) > 0:                        # initPoolStreams (:385-401)
                    for w, sno in enumerate(pool_worker):
                        if sno == 0:
                            s = new.pop()
                            pool_worker[w] = s
                            _, _, f0, d0 = [e for e in found if e[1] == s][0]
                            streams[w] = SatStream(s, f0, p, delay=d0)
                            act.add(s)
                            if len(new) == 0:
                                break
            continue
        res_lst = []                                    # satCalc (:404-417)
        for s in act:
            ss = streams[pool_worker.index(s)]
            sw, f_lst, co_ph, cp_q = ss.process(data, smp_time)
            res_lst.append((sw, ss.sat_no, f_lst, co_ph, cp_q))
        frame_lst = []
        stream_no = smp_time // p.ngps
        for sw, s, f_lst, co_ph, cp_q in res_lst:       # :496-505
            frame_lst += f_lst
            cp_q_lst[s] = cp_q
            if co_ph >= 0:
                co_ph_lst.setdefault(s, []).append((stream_no, co_ph))
        if len(frame_lst) > 0:                          # :509-519
            yield i, (skipped, frame_lst, co_ph_lst)
            co_ph_lst, skipped = {}, 0
