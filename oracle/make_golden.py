#!/usr/bin/env python3
"""Freeze outputs of the REAL reference on seeded synthetic IQ as fixtures.

Build-container only (needs /root/reference, which never travels to the GPU
box).  Imports ``gpslib`` / ``gpsrecv`` from the reference tree exactly as they
are (a two-line ``rtlsdr`` stand-in module is put on sys.path because
``gpsrecv.py:11`` imports pyrtlsdr, which is only used for live SDR input),
drives them with scenes from ``gpsmi.synth`` and writes small ``.npz`` files
under ``tests/golden/``.  Inputs are NOT stored: the scene is regenerated from
its seed, and a sha256 of the raw IQ is stored so that generator drift is
detected.

    python oracle/make_golden.py            # all fixtures (two sub-processes)
    python oracle/make_golden.py default    # CODE_SAMPLES=2048,  N_CYC=32
    python oracle/make_golden.py hirate     # CODE_SAMPLES=16368, N_CYC=8
    python oracle/make_golden.py navbits    # Subframe / evalGpsBits on constructed frames
    python oracle/make_golden.py position   # SatOrbit / SatPos / leastSquaresPos4 / ecefToGeo
    python oracle/make_golden.py resweep    # SatStream.initSweep / sweepFrequency / restoreFreq
    python oracle/make_golden.py handoff    # gpsrecv.main(): file -> streamData -> processData -> datagrams

The reference binds its configuration at import time (``from gpsglob import``),
so each configuration runs in its own interpreter.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/src'
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.dont_write_bytecode = True


def _import_reference(code_samples, n_cyc):
    stub = tempfile.mkdtemp(prefix='rtlsdr_stub_')
    with open(os.path.join(stub, 'rtlsdr.py'), 'w') as f:
        f.write('class RtlSdr:\n    pass\n')
    sys.path[:0] = [stub, REF, os.path.join(ROOT, 'gps-sdr-receiver_amd')]
    import gpsglob
    gpsglob.CODE_SAMPLES = code_samples
    gpsglob.SAMPLE_RATE = 1000 * code_samples
    gpsglob.N_CYC = n_cyc
    gpsglob.NGPS = n_cyc * code_samples
    import gpslib
    import gpsrecv
    from scipy.fft import fft
    # gpsrecv builds this table only under __main__ (gpsrecv.py:574-577)
    gpsrecv.FFT_CACODE = [0] + [fft(gpslib.GPSCacode(s)) for s in range(1, 33)]
    return gpsglob, gpslib, gpsrecv


def scene_for(config):
    """The fixture scenes; tests rebuild them with the same call."""
    from gpsmi import synth
    if config == 'default':
        return synth.default_scene(12, seed=7, code_samples=2048, n_cyc=32)
    return synth.default_scene(12, seed=11, code_samples=16368, n_cyc=8)


def ref_table(gpsrecv, np, data, freqs, prns, n_avg, cs):
    """Search surface from the reference's own expressions
    (gpsrecv.py:249-259) without its first-hit pruning."""
    from scipy.fft import fft, ifft
    nb, ns = len(freqs), len(prns)
    am = np.zeros((nb, ns), np.int32)
    pk = np.zeros((nb, ns))
    mean = np.zeros((nb, ns))
    std = np.zeros((nb, ns))
    for b, f in enumerate(freqs):
        new, _ = gpsrecv.demodDoppler(data, f, 0, n_avg * cs)
        df = 0
        for i in range(n_avg):
            df += fft(new[i * cs:(i + 1) * cs])
        spec = df / n_avg
        for j, s in enumerate(prns):
            corr = np.abs(ifft(spec * np.conjugate(gpsrecv.FFT_CACODE[s])))
            mx = np.argmax(corr)
            am[b, j], pk[b, j] = mx, corr[mx]
            mean[b, j], std[b, j] = np.mean(corr), np.std(corr)
            # the reference's own threshold function must agree
            d, nmc = gpsrecv.findCodePhase(corr)
            assert nmc == (corr[mx] - mean[b, j]) / std[b, j]
    return dict(argmax=am, peak=pk, mean=mean, std=std)


def run(config):
    import numpy as np
    cs, n_cyc = (2048, 32) if config == 'default' else (16368, 8)
    gpsglob, gpslib, gpsrecv = _import_reference(cs, n_cyc)
    ngps = cs * n_cyc
    scene = scene_for(config)
    out = {'numpy': np.__version__, 'code_samples': cs, 'n_cyc': n_cyc}
    n_acq_blocks = 5
    n_trk_blocks = 48 if config == 'default' else 40
    blocks = [scene.block(b) for b in range(n_acq_blocks + n_trk_blocks)]
    h = hashlib.sha256()
    for b in range(n_acq_blocks + n_trk_blocks):
        h.update(scene.block_raw(b).tobytes())
    out['iq_sha256'] = h.hexdigest()

    # ---- replica known answers
    rep = np.array([gpslib.GPSCacode(p) for p in range(1, 33)])
    out['replica_sha256'] = hashlib.sha256(rep.tobytes()).hexdigest()
    out['replica_sum'] = rep.sum(axis=1)

    # ---- (ii) acquisition through the real sweepAllSats loop (reference defaults)
    sat_lst = list(gpsrecv.SAT_ALL)
    found, freq = [], gpsglob.MIN_FREQ
    calls = []
    for b in range(n_acq_blocks):
        ready, freq, found = gpsrecv.sweepAllSats(
            blocks[b], freq, sat_lst, found, itSweep=gpsglob.IT_SWEEP_ALL)
        calls.append((ready, freq, len(found)))
    assert ready
    out['sweep_calls'] = np.array(calls, dtype=np.float64)
    out['sweep_found'] = np.array(found, dtype=np.float64)   # norm, sv, f, delay

    # full surfaces
    if config == 'default':
        prn31 = list(range(2, 33))
        f50 = [gpsglob.MIN_FREQ + gpsglob.STEP_FREQ * i for i in range(50)]
        for b in range(n_acq_blocks):            # bins 10b..10b+9 on block b
            t = ref_table(gpsrecv, np, blocks[b], f50[10 * b:10 * b + 10],
                          prn31, 4, cs)
            for k, v in t.items():
                out[f'ref50_{k}_{b}'] = v
        prn32 = list(range(1, 33))
        f41 = [-5000.0 + 250.0 * i for i in range(41)]
        for k, v in ref_table(gpsrecv, np, blocks[0], f41, prn32, 1, cs).items():
            out[f'cfg2_{k}'] = v
        f201 = [-5000.0 + 50.0 * i for i in range(201)]
        for k, v in ref_table(gpsrecv, np, blocks[0], f201, prn32, 10,
                              cs).items():
            out[f'cfg4_{k}'] = v
    else:
        prn32 = list(range(1, 33))
        f41 = [-5000.0 + 250.0 * i for i in range(41)]
        for k, v in ref_table(gpsrecv, np, blocks[0], f41, prn32, 1, cs).items():
            out[f'cfg2_{k}'] = v

    # ---- (iii) tracking through the real SatStream.process
    chans = [(int(s), float(f), int(d)) for _, s, f, d in found][:12]
    out['trk_init'] = np.array(chans, dtype=np.float64)
    nch = len(chans)
    rec = {k: np.zeros((nch, n_trk_blocks)) for k in (
        'delay', 'code_phase', 'norm', 'freq', 'phase', 'locked', 'nps',
        'std_dev', 'amplitude', 'corr_q', 'corr_l', 'mx', 'corr_mean',
        'corr_std', 'n_dumps', 'sweep', 'ms_time', 'n_edges')}
    dumps = np.zeros((nch, n_trk_blocks, n_cyc + 1), np.complex64)
    epl = np.zeros((nch, n_trk_blocks, 3))
    frames = []
    for c, (sv, f0, d0) in enumerate(chans):
        ss = gpslib.SatStream(sv, f0, delay=d0, itSweep=gpsglob.IT_SWEEP,
                              corrMin=gpsglob.CORR_MIN,
                              corrAvg=gpsglob.CORR_AVG,
                              sweepCorrAvg=gpsglob.SWEEP_CORR_AVG)
        cap = {}
        orig_corr, orig_dec = ss.cacodeCorr, ss.decodeData

        def corr_spy(data, avg, _o=orig_corr, _c=cap):
            r = _o(data, avg)
            _c['corr'] = r[0]
            return r

        def dec_spy(data, delay, _o=orig_dec, _c=cap):
            r = _o(data, delay)
            _c['dumps'] = r
            return r
        ss.cacodeCorr, ss.decodeData = corr_spy, dec_spy
        for i in range(n_trk_blocks):
            b = n_acq_blocks + i
            smp_time = np.int64((b + 1) * ngps)       # gpsrecv.py:471
            sw, fl, co_ph, (cq, cl) = ss.process(blocks[b], smp_time)
            corr = cap['corr']
            mx = int(np.argmax(corr))
            g = cap['dumps']
            dumps[c, i, :len(g)] = g
            epl[c, i] = [corr[mx - 1], corr[mx], corr[(mx + 1) % len(corr)]]
            vals = dict(delay=ss.DELAY, code_phase=co_ph, norm=ss.MAX_CORR,
                        freq=ss.FREQ, phase=ss.PHASE, locked=ss.PHASE_LOCKED,
                        nps=len(ss.PREV_SAMPLES), std_dev=ss.STD_DEV,
                        amplitude=ss.AMPLITUDE, corr_q=cq, corr_l=cl, mx=mx,
                        corr_mean=np.mean(corr), corr_std=np.std(corr),
                        n_dumps=len(g), sweep=sw, ms_time=ss.MS_TIME,
                        n_edges=len(ss.EDGES))
            for k, v in vals.items():
                rec[k][c, i] = v
            for d in fl:
                frames.append((c, i, sorted(d.items())))
    for k, v in rec.items():
        out[f'trk_{k}'] = v
    out['trk_dumps'] = dumps
    out['trk_epl'] = epl
    out['trk_frames_repr'] = np.array(repr(frames))

    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, f'ref_{config}.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {os.path.getsize(path)} bytes, {nch} channels, '
          f'{len(found)} SVs found')


def run_navbits():
    """Subframe.Extract and SatStream.evalGpsBits of the reference on
    constructed 300-bit frames (gpsmi.navbits.encode_subframe builds valid
    IS-GPS-200 parity; the reference has no encoder) -> ref_navbits.npz."""
    import json
    import numpy as np
    gpsglob, gpslib, gpsrecv = _import_reference(2048, 32)
    from gpsmi import navbits as nb
    rng = np.random.default_rng(20240)
    frames, status, fields = [], [], []
    ds29 = ds30 = 0
    chain = []
    for k in range(60):
        w = rng.integers(0, 2, (10, 24)).astype(np.int8)
        w[0, :8] = nb.PREAMBLE_BITS
        sid = (k % 5) + 1
        kind = 'ok'
        if k % 12 == 7:
            sid, kind = (0, 6, 7)[(k // 12) % 3], 'bad_id'
        w[1, 19:22] = [(sid >> 2) & 1, (sid >> 1) & 1, sid & 1]
        f = nb.encode_subframe(w, ds29, ds30)
        ds29, ds30 = int(f[298]), int(f[299])
        chain.append(f.copy())
        if k % 12 == 3:
            f = 1 - f; kind = 'inverted'
        elif k % 12 == 5:
            f[30 * (1 + k % 9) + k % 24] ^= 1; kind = 'parity'
        elif k % 12 == 9:
            f[3] ^= 1; kind = 'preamble'
        elif k % 12 == 11:
            f = f[:299]; kind = 'short'
        sf = gpslib.Subframe()
        st = sf.Extract(f)
        d = {}
        if st == 0:
            d = {'ID': sf.ID, 'tow': sf.tow}
            for name in ('weekNum satAcc satHealth Tgd IODC Toc af2 af1 af0 '
                         'Crs deltaN M0 Cuc IODE2 e Cus sqrtA Toe '
                         'Cic omegaBig Cis i0 IODE3 Crc omegaSmall omegaDot IDOT').split():
                v = getattr(sf, name)
                if {1: 'weekNum satAcc satHealth Tgd IODC Toc af2 af1 af0',
                    2: 'Crs deltaN M0 Cuc IODE2 e Cus sqrtA Toe',
                    3: 'Cic omegaBig Cis i0 IODE3 Crc omegaSmall omegaDot IDOT'}.get(
                        sf.ID, '').split().count(name):
                    d[name] = v
        pad = np.zeros(300, np.int8)
        pad[:len(f)] = f
        frames.append(pad)
        status.append((st, len(f)))
        fields.append(d)
    # streams through evalGpsBits: noise + a run of consecutive valid frames + noise,
    # normal and inverted polarity, and one with a corrupted frame in the middle
    streams = []
    ss = gpslib.SatStream(5, 0.0)
    for variant in range(3):
        run = [c.copy() for c in chain[10 * variant:10 * variant + 6]]
        if variant == 2:
            run[2][77] ^= 1
        bits01 = np.concatenate([rng.integers(0, 2, 41 + 13 * variant).astype(np.int8)]
                                + run + [rng.integers(0, 2, 120).astype(np.int8)])
        pm = (2 * bits01 - 1).astype(np.int8)
        if variant == 1:
            pm = (-pm).astype(np.int8)
        stamps = (np.arange(len(pm), dtype=np.int64) * 40960 + 777)
        res, rest, rest_st = ss.evalGpsBits(pm, stamps)
        streams.append(dict(bits=pm.tolist(), stamps=stamps.tolist(),
                            frames=[{k: (int(v) if isinstance(v, (int, np.integer))
                                         else float(v)) for k, v in r.items()} for r in res],
                            keys=[list(r.keys()) for r in res],
                            rest=len(rest), rest_first_stamp=int(rest_st[0])))
    path = os.path.join(GOLD, 'ref_navbits.npz')
    np.savez_compressed(path, frames=np.array(frames), status=np.array(status),
                        fields=np.array(json.dumps(fields)),
                        streams=np.array(json.dumps(streams)))
    print(path, os.path.getsize(path), 'bytes;', sum(1 for s, _ in status if s == 0),
          'valid frames;', [len(s['frames']) for s in streams], 'frames per stream')


def run_position():
    """The reference's own orbit / code-phase / least-squares functions on a
    constructed scene -> ref_position.npz.

    Scene: eight satellites with constructed ephemerides above a receiver at
    (49.082961 N, 8.307581 E, 160 m); the forward model uses the reference's
    SatPos.ecefCoord for the satellite positions and clocks and the same first-order
    Sagnac term as its solver, so that the observations (subframe sample times and one
    code phase per 32-ms stream, noise-free) are consistent with what the fix inverts.
    Stored: the datagrams, the cleaned code-phase lists that went into
    SatOrbit.evalCodePhase, every tuple it returned, every leastSquaresPos4 solution,
    ecefToGeo / geoToEcef samples, the truth."""
    import json
    import numpy as np
    gpsglob, gpslib, gpsrecv = _import_reference(2048, 32)
    from gpsmi import position as P
    CS, N_CYC = 2048, 32
    NGPS, FS = CS * N_CYC, 1000 * CS
    rng = np.random.default_rng(424242)
    truth = np.array(gpslib.geoToEcef(49.082961, 8.307581, 160.0))
    tow0 = 50001                       # first subframe carries tow0: its start is (tow0-1)*6 s
    T0 = (tow0 - 1) * 6 - 3.123456789  # GPS time of local sample 0
    sp = gpslib.SatPos()

    def make_eph():
        return {'weekNum': 290, 'Tgd': float(rng.uniform(-2e-8, 2e-8)), 'Toc': 302400,
                'af2': 0.0, 'af1': float(rng.uniform(-5e-12, 5e-12)),
                'af0': float(rng.uniform(-4e-4, 4e-4)), 'IODC': 77, 'satAcc': 0,
                'Crs': float(rng.uniform(-60, 60)), 'deltaN': float(rng.uniform(3e-9, 6e-9)),
                'M0': float(rng.uniform(-np.pi, np.pi)), 'Cuc': float(rng.uniform(-3e-6, 3e-6)),
                'e': float(rng.uniform(0.002, 0.02)), 'Cus': float(rng.uniform(-9e-6, 9e-6)),
                'sqrtA': float(5153.6 + rng.uniform(-0.3, 0.3)), 'Toe': 302400, 'IODE2': 77,
                'Cic': float(rng.uniform(-2e-7, 2e-7)),
                'omegaBig': float(rng.uniform(-np.pi, np.pi)),
                'Cis': float(rng.uniform(-2e-7, 2e-7)),
                'i0': float(0.96 + rng.uniform(-0.03, 0.03)),
                'Crc': float(rng.uniform(150, 350)),
                'omegaSmall': float(rng.uniform(-np.pi, np.pi)),
                'omegaDot': float(rng.uniform(-8.6e-9, -7.6e-9)),
                'IDOT': float(rng.uniform(-5e-10, 5e-10)), 'IODE3': 77}

    up = truth / np.linalg.norm(truth)
    ephs = {}
    prn = 1
    while len(ephs) < 8:               # keep satellites above 15 degrees elevation
        e = make_eph()
        x, y, z, _ = sp.ecefCoord(tow0, e)
        los = np.array([x, y, z]) - truth
        if los.dot(up) / np.linalg.norm(los) > np.sin(np.radians(15)):
            prn += int(rng.integers(1, 4))
            ephs[prn] = e

    om = gpslib.OMEGA_EARTH
    vrot = np.array([-truth[1] * om, truth[0] * om, 0.0])

    def arrive(eph, m):
        """local sample time at which the code epoch sent m ms after the first
        subframe start arrives"""
        tow, DT = tow0 + m // 6000, (m % 6000) / 1000.0
        x, y, z, dt_sv = sp.ecefCoord(tow, eph, DT=DT)
        t_tx = (tow - 1) * 6 + DT - dt_sv
        X = np.array([x, y, z])
        tau = 0.07
        for _ in range(4):
            tau = np.linalg.norm(X - truth - vrot * tau) / gpslib.GPS_C
        return (t_tx + tau - T0) * FS

    seconds = 48
    n_streams = int(seconds / 0.032)
    n_sub = seconds // 6
    frames = {}                         # second -> list of frame dicts
    cps = {s: [] for s in ephs}
    for s, e in ephs.items():
        for n in range(n_sub):
            ST = int(np.floor(arrive(e, 6000 * n)))
            sid = n % 5 + 1
            f = {'SAT': s, 'ID': sid, 'tow': tow0 + n, 'ST': ST}
            if sid == 1:
                f.update({k: e[k] for k in gpslib.ephemSF1})
                f['satHealth'] = 0
            elif sid == 2:
                f.update({k: e[k] for k in gpslib.ephemSF2})
            elif sid == 3:
                f.update({k: e[k] for k in gpslib.ephemSF3})
            frames.setdefault(int((ST + 6 * FS) // FS), []).append(f)
        k0 = arrive(e, 0)
        for sno in range(1, n_streams):
            mid = sno * NGPS + (N_CYC // 2) * CS
            m = int(round((mid - k0) / CS))
            k = arrive(e, m)
            while k < mid:
                m += 1
                k = arrive(e, m)
            while k >= mid + CS:
                m -= 1
                k = arrive(e, m)
            cps[s].append((sno, float(k - np.floor(k / CS) * CS)))
    datagrams = []
    for sec in range(seconds):
        lo, hi = sec / 0.032, (sec + 1) / 0.032
        coph = {s: [(n, c) for n, c in cps[s] if lo <= n < hi] for s in ephs}
        datagrams.append((0, sorted(frames.get(sec, []), key=lambda f: f['SAT']), coph))

    # the reference's classes on the same stream; the lists that go into evalCodePhase are
    # cleaned by the build's own prep_code_phase (gpseval cannot be imported here)
    solver = P.PositionSolver(CS, N_CYC)
    orbits = {s: gpslib.SatOrbit(s) for s in ephs}
    rec_tuples, rec_cpl, rec_fix = [], [], []
    mean_pos = None
    for skipped, frame_lst, coph in datagrams:
        cpl = solver.prep_code_phase(coph)
        for sf in frame_lst:
            solver.orbit(sf['SAT'])                 # keeps solver.orbits in step (phase errors)
            orbits[sf['SAT']].readFrame(dict(sf))
        res = []
        for s in cpl:
            res += orbits[s].evalCodePhase(list(cpl[s]), relCorr=True)
        rec_cpl.append({str(s): [[int(n), float(c)] for n, c in cpl[s]] for s in cpl})
        rec_tuples.append([[float(v) for v in t] for t in res])
        # fixes exactly as ecefPositions calls the solver (gpseval.py:235-318)
        fixes = []
        res_sorted = sorted(res, key=lambda t: (t[1], t[7], t[0]))
        r = 0
        while r < len(res_sorted):
            key = (res_sorted[r][1], res_sorted[r][7])
            grp = []
            while r < len(res_sorted) and (res_sorted[r][1], res_sorted[r][7]) == key:
                grp.append(res_sorted[r])
                r += 1
            if len(grp) < 4:
                continue
            pos = np.array([g[2:5] for g in grp], dtype=np.float64).T
            t_rx = np.array([g[5] for g in grp])
            std = np.array([g[8] for g in grp])
            start = [0, 0, 0, 0]
            if mean_pos is not None:
                start[1:] = mean_pos
            recp, resid, rng_est, meas = gpslib.leastSquaresPos(
                4, pos, t_rx, maxResidual=gpsglob.MAX_RESIDUAL, maxIt=gpsglob.LSF_MAX_IT,
                recPos=start, height=gpsglob.HEIGHT, hDev=gpsglob.HEIGHT_DEV, stdDev=std)
            if resid[-1] <= gpsglob.MAX_RESIDUAL:
                fixes.append([float(key[0]), float(key[1])] + [float(v) for v in recp]
                             + [float(len(resid))])
        if fixes:
            mean_pos = list(np.mean(np.array([f[3:6] for f in fixes]), axis=0))
        rec_fix.append(fixes)
        for s, lst in coph.items():
            solver.coph_hist[s] = solver.coph_hist.get(s, []) + list(lst)

    geo = []
    for v in ([truth[0], truth[1], truth[2]], [4.1e6, 6.2e5, 4.8e6], [-2.7e6, -4.3e6, 3.85e6],
              [1.2e6, -5.0e6, -3.7e6]):
        geo.append(list(v) + list(gpslib.ecefToGeo(v)))
    ecf = [[la, lo, al] + list(gpslib.geoToEcef(la, lo, al))
           for la, lo, al in ((49.082961, 8.307581, 160.0), (-33.9, 151.2, 30.0), (0.0, 0.0, 0.0))]
    orb = []
    for s, e in ephs.items():
        for tow, DT in ((tow0, 0.0), (tow0 + 3, 2.272), (tow0 + 700, 5.999)):
            orb.append([s, tow, DT] + [float(v) for v in sp.ecefCoord(tow, e, DT=DT)]
                       + [float(v) for v in sp.ecefCoord(tow, e, DT=DT, relCorr=False)])
    nfix = sum(len(f) for f in rec_fix)
    last = np.array(rec_fix[-1][-1][3:6])
    path = os.path.join(GOLD, 'ref_position.npz')
    np.savez_compressed(
        path, truth=truth, ephs=np.array(json.dumps({str(k): v for k, v in ephs.items()})),
        datagrams=np.array(json.dumps(
            [[sk, fl, {str(k): v for k, v in co.items()}] for sk, fl, co in datagrams])),
        cpl=np.array(json.dumps(rec_cpl)), tuples=np.array(json.dumps(rec_tuples)),
        fixes=np.array(json.dumps(rec_fix)), geo=np.array(geo), ecf=np.array(ecf),
        orb=np.array(orb))
    print(path, os.path.getsize(path), 'bytes;', nfix, 'fixes; last fix off truth by',
          float(np.linalg.norm(last - truth)), 'm')


RESWEEP_CASES = (
    # name, PRN, Doppler and delay the channel is opened with, tracking blocks before the
    # trigger, blocks after it.  PRN 21 sits at +3200 Hz: the 40 bins of the first call
    # (-5000 .. +2800 Hz) miss it and the second call finds it; PRN 4 (-2000 Hz) is found in
    # the first call; PRN 3 is not in the scene: two calls run off to +10800 Hz, restoreFreq.
    ('two_calls', 21, 3200.0, 1458, 3, 7),
    ('one_call', 4, -2000.0, 143, 3, 6),
    ('no_signal', 3, 1234.5, 100, 3, 6),
)


def run_resweep():
    """Per-channel re-acquisition of the real gpslib.SatStream (initSweep :1110-1116,
    the sweep branch of process :1153-1173, sweepFrequency / getCorrMax :1350-1380,
    restoreFreq :1118-1120) on the default fixture scene -> ref_resweep.npz.  The sweep is
    triggered through process(..., sweep=True), the reference's own argument."""
    import numpy as np
    gpsglob, gpslib, gpsrecv = _import_reference(2048, 32)
    ngps = 2048 * 32
    scene = scene_for('default')
    first = 5
    out = {'numpy': np.__version__, 'first_block': first}
    keys = ('sweep', 'freq', 'freq_is_f32', 'max_corr', 'delay', 'code_phase', 'corr_q',
            'corr_l', 'phase', 'locked', 'df_len', 'n_frames', 'swp_reported')
    for name, sv, f0, d0, n_before, n_after in RESWEEP_CASES:
        nb = n_before + n_after
        ss = gpslib.SatStream(sv, f0, delay=d0, itSweep=gpsglob.IT_SWEEP,
                              corrMin=gpsglob.CORR_MIN, corrAvg=gpsglob.CORR_AVG,
                              sweepCorrAvg=gpsglob.SWEEP_CORR_AVG)
        rec = {k: np.zeros(nb) for k in keys}
        for i in range(nb):
            b = first + i
            blk = scene.block(b)
            smp_time = np.int64((b + 1) * ngps)
            sw, fl, co_ph, (cq, cl) = ss.process(blk, smp_time, sweep=(i == n_before))
            vals = dict(sweep=sw, freq=float(ss.FREQ),
                        freq_is_f32=isinstance(ss.FREQ, np.float32), max_corr=ss.MAX_CORR,
                        delay=ss.DELAY, code_phase=co_ph, corr_q=cq, corr_l=cl,
                        phase=float(ss.PHASE), locked=ss.PHASE_LOCKED, df_len=len(ss.DF),
                        n_frames=len(fl),
                        swp_reported=(fl[0]['SWP'] if fl else -1))
            for k, v in vals.items():
                rec[k][i] = v
        for k, v in rec.items():
            out[f'{name}_{k}'] = v
        out[f'{name}_init'] = np.array([sv, f0, d0, n_before, n_after], dtype=np.float64)
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, 'ref_resweep.npz')
    np.savez_compressed(path, **out)
    for name, *_ in RESWEEP_CASES:
        print(name, 'sweep', out[f'{name}_sweep'].astype(int).tolist(), 'freq',
              out[f'{name}_freq'].tolist(), 'delay', out[f'{name}_delay'].astype(int).tolist())
    print(path, os.path.getsize(path), 'bytes')


def run_newsats():
    """gpsrecv.getNewSats (gpsrecv.py:423-440) of the reference itself on seeded inputs ->
    ref_newsats.json: active set, found list, correlation-quality table -> (delete, new)."""
    import json
    import numpy as np
    gpsglob, gpslib, gpsrecv = _import_reference(2048, 32)
    rng = np.random.default_rng(423)
    cases = []
    for k in range(60):
        n_found = int(rng.integers(0, 18))
        sats = [int(s) for s in rng.permutation(np.arange(2, 33))[:n_found]]
        found = sorted(((float(np.round(rng.uniform(8, 40), 3)), s, float(rng.integers(-25, 25) * 200),
                         int(rng.integers(0, 2048))) for s in sats), reverse=True)
        n_act = int(rng.integers(0, 12))
        act = {int(s) for s in rng.permutation(np.arange(2, 33))[:n_act]}
        cpq = {}
        for s in act:
            if rng.random() < 0.85:          # (a channel may not have reported yet)
                cpq[s] = (float(rng.choice([-1.0, -0.2, 0.0, 0.4, 1.0])),
                          float(rng.choice([-1.0, 0.0, 0.25, 1.0])))
        dele, new = gpsrecv.getNewSats(set(act), [tuple(e) for e in found], dict(cpq))
        cases.append({'act': sorted(act), 'found': [list(e) for e in found],
                      'cpq': {str(s): list(v) for s, v in cpq.items()},
                      'delete': sorted(dele), 'new': sorted(new)})
    path = os.path.join(GOLD, 'ref_newsats.json')
    with open(path, 'w') as f:
        json.dump({'max_sat': int(gpsglob.MAX_SAT), 'cases': cases}, f)
    print(path, os.path.getsize(path), 'bytes', sum(len(c['new']) for c in cases), 'new in all')


def handoff_scene():
    """The hand-off fixture scene; tests rebuild it with gpsmi.synth_nav.handoff_scene()."""
    from gpsmi import synth_nav
    return synth_nav.handoff_scene()


def run_handoff():
    """The receiver process of the reference itself, end to end, on a recording
    (gpsrecv.py:553-579): the scene is written as the recorder's u8 .bin file, gpsglob's
    file-mode settings point at it (LIVE_MEAS = False is the default, BIN_DATA / DATA_PATH),
    SAVE_PICKLE = True, and the real ``main()`` runs -- streamData (:153-186) reading and
    decoding the file, processData (:445-548) with the spawn worker pool (initMultiProcPool,
    initPoolStreams, satCalc, :340-417), sweepAllSats, getNewSats, the hand-off
    ``pickle.dumps((skippedData, frameLst, coPhLst))`` and saveResults (:205-212).  The
    pickle file that run writes (our own run's output) is unpickled here and frozen as
    JSON in ref_handoff.npz: every datagram with every key in order, value and type name.
    SEND_OVER_UDP is switched off (gpsglob.py:25 allows that with SAVE_PICKLE)."""
    import asyncio
    import json
    import multiprocessing as mp
    import pickle
    import shutil
    import numpy as np
    gpsglob, gpslib, gpsrecv = _import_reference(2048, 32)
    scene, info, n_blocks = handoff_scene()
    tmp = tempfile.mkdtemp(prefix='gps_handoff_')
    h = hashlib.sha256()
    with open(os.path.join(tmp, 'handoff.bin'), 'wb') as f:
        for b in range(n_blocks):
            raw = scene.block_raw(b)
            h.update(raw.tobytes())
            raw.astype('<u2').tofile(f)
    # the spawned workers re-import gpsrecv: they need the same module path
    os.environ['PYTHONPATH'] = os.pathsep.join(sys.path[:3] + [os.environ.get('PYTHONPATH', '')])
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    gpsrecv.FFT_CACODE = [0, 0] + gpsrecv.FFT_CACODE[2:]       # as __main__ builds it (:574-577)
    gpsrecv.DATA_PATH = tmp + os.sep
    gpsrecv.BIN_DATA = 'handoff.bin'
    gpsrecv.SAVE_PICKLE = True
    gpsrecv.SEND_OVER_UDP = False
    gpsrecv.SAVE_DATE = 'handoff'
    assert gpsrecv.LIVE_MEAS is False and gpsrecv.START_STREAM == 0
    mp.set_start_method('spawn')                               # gpsrecv.py:572
    asyncio.run(gpsrecv.main())
    with open(os.path.join(tmp, 'handoff_gpsResult.pickle'), 'rb') as f:
        result_list = pickle.load(f)                           # written by the run above
    shutil.rmtree(tmp)

    def enc(v):
        if isinstance(v, (bool, np.bool_)):
            return [type(v).__name__, bool(v)]
        if isinstance(v, (int, np.integer)):
            return [type(v).__name__, int(v)]
        if isinstance(v, (float, np.floating)):
            return [type(v).__name__, float(v)]                # (float32 -> double: exact)
        raise TypeError(type(v))
    dgs = []
    for res in result_list:
        skipped, frame_lst, co_ph = pickle.loads(res)
        dgs.append({'skipped': enc(skipped),
                    'frames': [[[k, enc(v)] for k, v in d.items()] for d in frame_lst],
                    'coph': [[int(s), [[enc(n), enc(c)] for n, c in lst]]
                             for s, lst in co_ph.items()]})
    path = os.path.join(GOLD, 'ref_handoff.npz')
    np.savez_compressed(path, numpy=np.__version__, n_blocks=n_blocks,
                        iq_sha256=h.hexdigest(), datagrams=np.array(json.dumps(dgs)))
    n_sub = sum(1 for d in dgs for f in d['frames'] if any(k == 'ID' for k, _ in f))
    print(path, os.path.getsize(path), 'bytes;', len(dgs), 'datagrams,', n_sub,
          'decoded subframes, satellites',
          sorted({v[1] for d in dgs for f in d['frames'] for k, v in f if k == 'SAT'}))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        if sys.argv[1] == 'newsats':
            run_newsats()
        elif sys.argv[1] == 'navbits':
            run_navbits()
        elif sys.argv[1] == 'position':
            run_position()
        elif sys.argv[1] == 'resweep':
            run_resweep()
        elif sys.argv[1] == 'handoff':
            run_handoff()
        else:
            run(sys.argv[1])
    else:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'newsats'])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'navbits'])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'position'])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'resweep'])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'handoff'])
        for cfg in ('default', 'hirate'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__),
                                   cfg])
