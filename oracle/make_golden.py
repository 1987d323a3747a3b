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


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run_navbits() if sys.argv[1] == 'navbits' else run(sys.argv[1])
    else:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'navbits'])
        for cfg in ('default', 'hirate'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__),
                                   cfg])
