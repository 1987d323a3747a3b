"""Position fix from the receiver's hand-off stream.

Host logic, no GPU: what the evaluation process of the reference does with the
``(skippedData, frameLst, coPhLst)`` datagrams that ``gpsmi.pipeline.Receiver``
emits.  One module, in the order the data flows:

* code-phase clean-up per datagram: roll-over of the code phase inside a list,
  correlated phase jumps = lost streams (``cpOflCorrection`` / ``prepCodePhase``,
  reference src/gpseval.py:377-457),
* ephemeris table and time references from the decoded subframes (``SatData``,
  src/gpslib.py:651-790),
* broadcast-orbit evaluation, IS-GPS-200 user algorithm with the relativistic clock
  term (``SatPos.ecefCoord``, src/gpslib.py:425-642),
* code phases -> (satellite position at transmit time, reception time) tuples
  (``SatOrbit.evalCodePhase``, src/gpslib.py:897-1039),
* Gauss-Newton fix of (c t0, x, y, z) with Sagnac displacement and optional weights
  (``leastSquaresPos4`` / ``rotEarth`` / ``JacobianCalc``, src/gpslib.py:1640-1737),
* grouping of the tuples by (TOW, cycle) and the fix per group (``evalData`` /
  ``ecefPositions``, src/gpseval.py:197-318), ECEF <-> geodetic (``ecefToGeo`` /
  ``geoToEcef``, src/gpslib.py:1799-1890).

Parity: the gpslib-level pieces are pinned by ``tests/golden/ref_position.npz``
(outputs of the reference's own classes and functions on a constructed scene,
``oracle/make_golden.py position``).  The gpseval-level pieces (code-phase clean-up,
grouping) cannot be imported here -- gpseval needs a Qt matplotlib backend, gpxpy and
folium -- so they are restated from the source and their parity is UNPINNED; they are
covered by the end-to-end property instead: a constructed constellation and receiver
position go in as subframes + code phases, the fix comes out within centimetres.
"""
import datetime
import math

import numpy as np

# ---- constants (src/gpslib.py:14-21, :428-432; src/gpsglob.py:35-59)
WEEK_IN_SEC = 604800
GPS_C = 2.99792458e8
OMEGA_EARTH = 7.292115147e-5
ROLLOVER = 2
LEAPSEC = 18
MU_E = 3.986005e14
F_REL = -4.44280763310e-10
MAX_RESIDUAL = 1.0e-7
LSF_MAX_IT = 15
MIN_SAT = 4

EPHEM_SF1 = ('weekNum', 'Tgd', 'Toc', 'af2', 'af1', 'af0', 'IODC', 'satAcc')
EPHEM_SF2 = ('Crs', 'deltaN', 'M0', 'Cuc', 'e', 'Cus', 'sqrtA', 'Toe', 'IODE2')
EPHEM_SF3 = ('Cic', 'omegaBig', 'Cis', 'i0', 'Crc', 'omegaSmall', 'omegaDot', 'IDOT',
             'IODE3')

# status codes shared by EphemerisTable and OrbitTracker (gpslib.py:653-663, :812-821)
NO_ERR, NOT_READY, NEW_EPHEM, FLAWED_FRAME, HEALTH_ERR = range(5)


# ---------------------------------------------------------------- broadcast orbit
def cross_week(t):
    """Fold a time difference into +-half a week (``CrossTime``, gpslib.py:463-470)."""
    half = WEEK_IN_SEC // 2
    while t > half:
        t -= 2 * half
    while t < -half:
        t += 2 * half
    return t


def kepler_E(M, e, it_max=10, eps=1.0e-12):
    """Eccentric anomaly by Newton iteration, started at M (``Ek``, gpslib.py:503-511)."""
    prev, E, it = 0, M, 0
    while abs(E - prev) > eps and it < it_max:
        prev = E
        E = prev - (prev - e * np.sin(prev) - M) / (1 - e * np.cos(prev))
        it += 1
    return E


def sat_clock_offset(t_sv, eph, dtr=0):
    """``dtsv`` (gpslib.py:477-480): af0 + af1 dt + af2 dt^2 + dtr - Tgd."""
    dt = cross_week(t_sv - eph['Toc'])
    return eph['af0'] + eph['af1'] * dt + eph['af2'] * dt ** 2 + dtr - eph['Tgd']


def sat_ecef(tow, eph, DT=0, rel_corr=True):
    """ECEF position of the satellite at the transmit time belonging to subframe
    `tow` plus DT seconds, and the satellite clock correction (``ecefCoord``,
    gpslib.py:590-642).  Returns (x, y, z, dt_sv)."""
    t_sv = (tow - 1) * 6 + DT                       # tsv(): tow counts the NEXT subframe
    dtr = 0
    for it in range(2 if rel_corr else 1):
        dt_sv = sat_clock_offset(t_sv, eph, dtr)
        t_k = cross_week((t_sv - dt_sv) - eph['Toe'])
        M_k = eph['M0'] + (np.sqrt(MU_E) / eph['sqrtA'] ** 3 + eph['deltaN']) * t_k
        E_k = kepler_E(M_k, eph['e'])
        if it == 0:
            dtr = F_REL * eph['e'] * eph['sqrtA'] * np.sin(E_k)
    e = eph['e']
    nu = np.arctan2(np.sqrt(1 - e ** 2) * np.sin(E_k), np.cos(E_k) - e)
    phi = nu + eph['omegaSmall']
    s2, c2 = np.sin(2 * phi), np.cos(2 * phi)
    d_i = eph['Cic'] * c2 + eph['Cis'] * s2
    d_u = eph['Cus'] * s2 + eph['Cuc'] * c2
    d_r = eph['Crc'] * c2 + eph['Crs'] * s2
    inc = eph['i0'] + d_i + eph['IDOT'] * t_k
    u = phi + d_u
    r = eph['sqrtA'] ** 2 * (1 - e * np.cos(E_k)) + d_r
    xp, yp = r * np.cos(u), r * np.sin(u)
    Om = eph['omegaBig'] + (eph['omegaDot'] - OMEGA_EARTH) * t_k - OMEGA_EARTH * eph['Toe']
    x = xp * np.cos(Om) - yp * np.cos(inc) * np.sin(Om)
    y = xp * np.sin(Om) + yp * np.cos(inc) * np.cos(Om)
    z = yp * np.sin(inc)
    return x, y, z, dt_sv


def gps_datetime(tow, week_num):
    """UTC datetime of the subframe that carries `tow` (``gpsTime``, gpslib.py:1946-1955)."""
    tow = getattr(tow, 'tolist', lambda: tow)()
    week_num = getattr(week_num, 'tolist', lambda: week_num)()
    return (datetime.datetime(1980, 1, 6)
            + datetime.timedelta(days=(week_num + ROLLOVER * 1024) * 7)
            + datetime.timedelta(seconds=(tow - 1) * 6 - LEAPSEC))


# ---------------------------------------------------------------- ephemeris table
class EphemerisTable:
    """Ephemeris and (TOW, ST) time references of one satellite, built from its
    subframes; watches IODC/IODE for a change of data set (``SatData``)."""

    def __init__(self, sat_no, ephemeris=None):
        self.sat_no = sat_no
        self.status = NO_ERR
        self.ephem = {}
        self.time_data = []
        self.ephem_ok = False
        self._have = [False, False, False]           # subframes 1..3 seen
        self._last_iodc = -1
        self.loaded = ephemeris is not None
        if self.loaded:
            self.ephem = dict(ephemeris)
            self.ephem['SAT'] = sat_no
            self.ephem_ok = True
            self._have = [True, True, True]
            self._last_iodc = ephemeris['IODC'] & 255

    def _check(self, sf):
        status, iodc = NO_ERR, -1
        if sf['ID'] == 1:
            iodc = sf['IODC'] & 255
            if sf['satHealth'] != 0:
                status = HEALTH_ERR
        elif sf['ID'] == 2:
            iodc = sf['IODE2']
        elif sf['ID'] == 3:
            iodc = sf['IODE3']
        if status == NO_ERR and iodc > -1:
            if self._last_iodc > -1 and iodc != self._last_iodc:
                status = NEW_EPHEM
            self._last_iodc = iodc
        return status

    def read_subframe(self, sf):
        self.status = self._check(sf)
        if self.status != NO_ERR:
            return self.status
        if not self.ephem_ok:
            k = sf['ID'] - 1
            if 0 <= k < 3 and not self._have[k]:
                for key in (EPHEM_SF1, EPHEM_SF2, EPHEM_SF3)[k]:
                    self.ephem[key] = sf[key]
                self._have[k] = True
            self.ephem_ok = all(self._have)
            self.loaded = False
        # a loaded (possibly out-dated) ephemeris accepts time references only from
        # subframes that carry an issue number (gpslib.py:779-786)
        if (self.ephem_ok and not self.loaded) or (self.loaded and sf['ID'] < 4):
            self.time_data.append((sf['tow'], sf['ST']))
        return self.status


# ---------------------------------------------------------------- code phases -> tuples
class OrbitTracker:
    """Per-satellite state of the evaluation side (``SatOrbit``): reads subframes,
    turns cleaned code-phase lists into measurement tuples
    (satNo, TOW, x, y, z, sample time [s], weekNum, cycNo, cophStd [m])."""

    MAX_SLOPE = 6.55e-3                              # samples per ms (gpslib.py:822)

    def __init__(self, sat_no, code_samples=2048, n_cyc=32, eph=None):
        self.sat_no = sat_no
        self.cs, self.n_cyc = code_samples, n_cyc
        self.ngps = code_samples * n_cyc
        self.fs = 1000 * code_samples
        self.status = NO_ERR
        self.data = EphemerisTable(sat_no, eph)
        self.cp_lst = []
        self.last_sno = 0
        self.last_cp = 0
        self.ref_time = None
        self.ref_ephem = None
        self.phase_err = []
        self.slopes = []
        self.max_slopes = 1024 // n_cyc
        self.min_slopes = 4

    def read_frame(self, sf):
        stream_no = sf['ST'] // self.ngps
        if self.phase_err and stream_no < self.phase_err[-1]:
            self.status = FLAWED_FRAME
        else:
            self.status = self.data.read_subframe(sf)
            if self.status == NEW_EPHEM:             # start over with the new data set
                self.data = EphemerisTable(self.sat_no)
                self.data.read_subframe(sf)
        return self.status

    def _std_and_slope(self, snos, cps):
        if len(cps) > 3:
            p = np.polyfit(snos, cps, 1)
            std = np.std(cps - np.poly1d(p)(snos))
            self.slopes.append(p[0] / self.n_cyc)    # per ms
            if len(self.slopes) > self.max_slopes:
                del self.slopes[0]
        else:
            std = 0.5
        std *= GPS_C / self.fs                       # metres
        slope = np.mean(self.slopes) if len(self.slopes) > self.min_slopes else 0
        if abs(slope) > self.MAX_SLOPE:
            slope = np.sign(slope) * self.MAX_SLOPE
        return std, slope

    def _clear_ref(self):
        self.last_sno = 0
        self.cp_lst = []
        self.slopes = []
        self.ref_time = None
        self.ref_ephem = None

    def eval_code_phase(self, cpl, rel_corr=True):
        cs, n_cyc, ngps, fs = self.cs, self.n_cyc, self.ngps, self.fs
        min_gap, max_gap = 1000, 10000               # streams
        min_fit, max_fit, tol = n_cyc // 2, 100, 200
        out = []
        if len(cpl) > 0:
            if cpl[0][1] is None:                    # phase error: drop the time reference
                self.phase_err.append(cpl[0][0])
                self.data.time_data = []
                self._clear_ref()
                return out
            cpl = [item for item in cpl if item[0] > self.last_sno]
        if self.ref_time is not None and self.data.ephem_ok \
                and self.data.ephem['IODC'] != self.ref_ephem['IODC']:
            self._clear_ref()
        if self.ref_time is None and len(self.data.time_data) > 0:
            self.ref_time = self.data.time_data[-1]
            self.ref_ephem = dict(self.data.ephem)
        if len(cpl) == 0 or self.ref_time is None:
            return out

        week_num = self.ref_ephem['weekNum']
        TOW, ST = self.ref_time
        st_del = ST % cs                             # integer code phase of the reference
        ST = (ST // cs) * cs
        st_sno = ST // ngps
        if st_sno > self.last_sno:
            self.last_sno, self.last_cp = st_sno, st_del

        snos, cps = zip(*cpl)
        cps = np.asarray(cps)
        gap = snos[0] - self.last_sno
        if gap > max_gap:
            self._clear_ref()
            return out
        if gap > min_gap:                            # bridge the gap by a line fit
            if len(self.cp_lst) >= min_fit:
                x, y = zip(*self.cp_lst[-max_fit:])
                self.last_cp = np.poly1d(np.polyfit(x, y, 1))(snos[0])
            else:
                self._clear_ref()
                return out
        wraps = self.last_cp // cs                   # roll-overs accumulated so far
        if wraps != 0:
            cps += wraps * cs
        diff = self.last_cp - cps[0]
        if np.isclose(abs(diff), cs, rtol=1e-5, atol=tol):
            cps += np.sign(diff) * cs

        std, slope = self._std_and_slope(snos, cps)
        cpl = list(zip(snos, cps))
        self.cp_lst += cpl
        self.last_sno, self.last_cp = cpl[-1]

        # the subframe starts `offms` code periods into its stream (gpslib.py:987)
        offms = (TOW % 2 ** (n_cyc // 32)) * 16 if n_cyc > 16 else 0
        while (ST + 6 * fs) // ngps < snos[0]:       # advance to the first listed stream
            ST += 6 * fs
            TOW += 1
            offms = (offms + 16) % n_cyc
        CP = cps[0]
        cyc_no = 0
        d_st = offms * cs
        stream_no = (ST + d_st) // ngps
        code_no = (ST + d_st) // cs - stream_no * n_cyc
        idx = 0
        while idx < len(snos):
            if snos[idx] < stream_no:
                idx += 1
            elif snos[idx] > stream_no:
                stream_no += 1
                cyc_no += 1
                d_st += ngps
            else:
                x, y, z, dt_sv = sat_ecef(TOW, self.ref_ephem, DT=d_st / fs, rel_corr=rel_corr)
                CP = cps[idx]
                # the code phase was measured in the middle of the stream
                corr = (code_no + CP // cs - n_cyc // 2) * slope
                t_smp = (ST + d_st + CP + corr) / fs + dt_sv
                out.append((self.sat_no, TOW, x, y, z, t_smp, week_num, cyc_no, std))
                stream_no += 1
                cyc_no += 1
                d_st += ngps
                idx += 1
            if d_st >= 6 * fs:                       # next subframe
                TOW += 1
                cyc_no = 0
                ST += 6 * fs
                offms = (offms + 16) % n_cyc
                d_st = offms * cs
                if stream_no < snos[-1]:
                    self.ref_time = (TOW, ST + CP % cs)
        return out


# ---------------------------------------------------------------- least squares
def sagnac_shift(rec, ranges):
    """Displacement of the receiver during the signal's flight (``rotEarth``)."""
    v = [-rec[2] * OMEGA_EARTH, rec[1] * OMEGA_EARTH, 0]
    return np.tensordot(v, ranges / GPS_C, 0)


def least_squares_pos4(sat_pos, t_rx, rec=None, max_residual=1.0e-8, max_it=10,
                       t0_guess=0.07, std_dev=None):
    """Gauss-Newton solution of (c t0, x, y, z) from >= 4 satellites
    (``leastSquaresPos4``): sat_pos [3, n] ECEF, t_rx [n] reception times in s.
    Returns (rec, residuals, ranges, measured delays in m)."""
    rec = np.zeros(4) if rec is None else np.asarray(rec, dtype=np.float64).copy()
    cdt = GPS_C * (t_rx - t_rx[0])
    rec[0] = GPS_C * t0_guess
    n = len(cdt)
    W = np.eye(n) if std_dev is None else np.linalg.inv(np.diag(std_dev) ** 2)
    dp = np.zeros((3, n))
    resid, residual, it = [], 1, 0
    ranges = None
    while it < max_it and residual > max_residual:
        ranges = np.sqrt((sat_pos[0] - rec[1] - dp[0]) ** 2
                         + (sat_pos[1] - rec[2] - dp[1]) ** 2
                         + (sat_pos[2] - rec[3] - dp[2]) ** 2)
        dp = sagnac_shift(rec, ranges)
        f = ranges - rec[0] - cdt
        J = np.empty((n, 4))
        J[:, 0] = -1.0
        J[:, 1] = (rec[1] - sat_pos[0]) / ranges
        J[:, 2] = (rec[2] - sat_pos[1]) / ranges
        J[:, 3] = (rec[3] - sat_pos[2]) / ranges
        step = -np.linalg.pinv(J.T.dot(W).dot(J)).dot(J.T).dot(W) @ f
        rec = rec + step
        residual = np.linalg.norm(step)
        resid.append(residual)
        it += 1
    return rec, resid, ranges, cdt + rec[0]


# ---------------------------------------------------------------- ECEF <-> geodetic
# K. Osen, "Accurate Conversion of Earth-Fixed Earth-Centered Coordinates to Geodetic
# Coordinates", NTNU 2017 (the method and constants the reference uses, gpslib.py:1782-1890)
_INVAA = 2.45817225764733181057e-14
_AADC = 7.79540464078689228919e+7
_BBDCC = 1.48379031586596594555e+2
_L = 3.34718999507065852867e-3
_P1MEE = 9.93305620009858682943e-1
_P1MEEDAA = 2.44171631847341700642e-14
_HMIN = 2.25010182030430273673e-14
_LL4 = 4.48147234524044602618e-5
_LL = 1.12036808631011150655e-5
_INVCBRT2 = 7.93700525984099737380e-1
_INV3 = 3.33333333333333333333e-1
_INV6 = 1.66666666666666666667e-1
_D2R = 1.74532925199432957691e-2
_R2D = 5.72957795130823208766e+1


def geo_to_ecef(lat, lon, alt):
    lat, lon = _D2R * lat, _D2R * lon
    cl, sl = np.cos(lat), np.sin(lat)
    N = _AADC / np.sqrt(cl * cl + _BBDCC)
    d = (N + alt) * cl
    return d * np.cos(lon), d * np.sin(lon), (_P1MEE * N + alt) * sl


def ecef_to_geo(xyz):
    x, y, z = xyz
    ww = x * x + y * y
    m = ww * _INVAA
    n = z * z * _P1MEEDAA
    mpn = m + n
    p = _INV6 * (mpn - _LL4)
    G = m * n * _LL
    H = 2 * p * p * p + G
    if H < _HMIN:
        return None
    C = math.pow(H + G + 2 * np.sqrt(H * G), _INV3) * _INVCBRT2
    i = -_LL - 0.5 * mpn
    beta = _INV3 * i - C - p * p / C
    k = _LL * (_LL - mpn)
    t2 = np.sqrt(beta * beta - k)
    t4 = np.sqrt(t2 - 0.5 * (beta + i))
    t6 = np.sqrt(np.abs(0.5 * (beta - i)))
    t = t4 + (t6 if m < n else -t6)
    # one Newton step on the quartic
    g = 2 * _L * (m - n)
    tt = t * t
    F = tt * tt + 2 * i * tt + g * t + k
    dF = 4 * tt * t + 4 * i * t + g
    t = t - F / dF
    u, v = t + _L, t - _L
    w = np.sqrt(ww)
    zu, wv = z * u, w * v
    lat = np.arctan2(zu, wv)
    inv_uv = 1 / (u * v)
    dw = w - wv * inv_uv
    dz = z - zu * _P1MEE * inv_uv
    da = np.sqrt(dw * dw + dz * dz)
    alt = -da if u < 1 else da
    return _R2D * lat, _R2D * np.arctan2(y, x), alt


# ---------------------------------------------------------------- datagram level
class PositionSolver:
    """The evaluation side's data path for one receiver: feed it the hand-off tuples
    ``(skippedData, frameLst, coPhLst)``; it returns the fixes of each datagram as
    ``[(posix time, x, y, z)]`` (ECEF, metres).  Restates ``prepCodePhase``,
    ``cpOflCorrection``, ``evalData`` and ``ecefPositions`` (parity unpinned, see the
    module docstring)."""

    def __init__(self, code_samples=2048, n_cyc=32, ephemerides=None, lsf_weight=True,
                 min_sat=MIN_SAT):
        self.cs, self.n_cyc = code_samples, n_cyc
        self.orbits = {}
        self.ephemerides = ephemerides or {}
        self.coph_hist = {}                          # COPH_LIST
        self.n_phase_err = 0
        self.lsf_weight = lsf_weight
        self.min_sat = max(min_sat, 4)
        self.mean_pos = None                         # start value of the next fit
        self.fail_lst = []
        self.sat_results = []

    def orbit(self, sat_no):
        if sat_no not in self.orbits:
            self.orbits[sat_no] = OrbitTracker(sat_no, self.cs, self.n_cyc,
                                               self.ephemerides.get(sat_no))
        return self.orbits[sat_no]

    # -- code-phase clean-up (gpseval.py:377-457)
    def _unwrap(self, sat_no, lst, err_stream):
        cs, tol = self.cs, 200
        max_gap = self.n_cyc // 4
        out = list(lst)
        wraps = 0
        pno, pcp = out[0]

        def flag(no, pno):                           # every stream in (pno, no] is suspect
            for j in range(no - pno):
                err_stream[no - j] = err_stream.get(no - j, 0) + 1

        for i in range(1, len(out)):
            no, cp = out[i]
            cp += wraps * cs
            diff = pcp - cp
            if np.isclose(abs(diff), cs, rtol=1e-5, atol=tol):
                cp += np.sign(diff) * cs
                wraps += np.sign(diff)
            if abs(cp - pcp) > (1 + (no - pno - 1) * 0.2):
                flag(no, pno)
            out[i] = (no, cp)
            pno, pcp = no, cp
        prev = self.coph_hist.get(sat_no)
        if out and prev:
            no, cp = out[0]
            pno, pcp = prev[-1]
            if no - pno <= max_gap:
                diff = pcp - cp
                if np.isclose(abs(diff), cs, rtol=1e-5, atol=tol):
                    cp += np.sign(diff) * cs
                if abs(cp - pcp) > (1 + (no - pno - 1) * 0.2):
                    flag(no, pno)
        return out

    def prep_code_phase(self, coph_lst):
        min_entries, min_sat_err = self.n_cyc // 4, 3
        cpl, err_stream = {}, {}
        for sat_no, lst in coph_lst.items():
            if len(lst) >= min_entries:
                cpl[sat_no] = self._unwrap(sat_no, lst, err_stream)
        if err_stream and max(err_stream.values()) >= min_sat_err:
            key = max(err_stream, key=err_stream.get)
            for sat_no in self.orbits:               # every known satellite drops its reference
                cpl[sat_no] = [(key, None)]
            self.n_phase_err += 1
        return cpl

    # -- tuples per datagram (gpseval.py:197-232)
    def eval_data(self, frame_lst, cpl):
        for sf in frame_lst:
            orb = self.orbit(sf['SAT'])
            if 'ID' in sf:
                orb.read_frame(sf)
        res = []
        for sat_no in cpl:
            res += self.orbit(sat_no).eval_code_phase(cpl[sat_no], rel_corr=True)
        return res

    # -- fixes (gpseval.py:235-318)
    def ecef_positions(self, sat_res):
        fixes = []
        start = np.zeros(4)
        if self.mean_pos is not None:
            start[1:] = self.mean_pos
        sat_res = sorted(sat_res, key=lambda e: (e[1], e[7], e[0]))
        r = 0
        while r < len(sat_res):
            tow, cyc = sat_res[r][1], sat_res[r][7]
            grp = []
            while r < len(sat_res) and (sat_res[r][1], sat_res[r][7]) == (tow, cyc):
                grp.append(sat_res[r])
                r += 1
            if len(grp) < self.min_sat:
                continue
            pos = np.array([g[2:5] for g in grp], dtype=np.float64).T
            t_rx = np.array([g[5] for g in grp], dtype=np.float64)
            std = np.array([g[8] for g in grp], dtype=np.float64)
            try:
                rec, resid, _, _ = least_squares_pos4(
                    pos, t_rx, rec=start, max_residual=MAX_RESIDUAL, max_it=LSF_MAX_IT,
                    std_dev=std if self.lsf_weight else None)
            except (np.linalg.LinAlgError, FloatingPointError, ValueError):
                self.fail_lst.append((tow, cyc, 'EXCEPTION'))
                continue
            if resid[-1] <= MAX_RESIDUAL:
                t = gps_datetime(tow, grp[0][6]) \
                    + datetime.timedelta(seconds=cyc * self.n_cyc / 1000)
                fixes.append((t.timestamp(), rec[1], rec[2], rec[3]))
            else:
                self.fail_lst.append((tow, cyc, 'MAX_RESIDUAL'))
        return fixes

    def feed(self, datagram):
        """One hand-off tuple (already unpickled).  Returns the list of fixes."""
        _, frame_lst, coph_lst = datagram
        cpl = self.prep_code_phase(coph_lst)
        res = self.eval_data(frame_lst, cpl)
        self.sat_results += res
        fixes = self.ecef_positions(res)
        if fixes:
            self.mean_pos = np.mean(np.array([f[1:] for f in fixes]), axis=0)
        for sat_no, lst in coph_lst.items():
            self.coph_hist[sat_no] = self.coph_hist.get(sat_no, []) + list(lst)
        return fixes
