"""Synthetic scene whose code delays follow a geometric range (for the IQ -> position
fix test).

A constellation with constructed ephemerides, a receiver at a known ECEF position with
an arbitrary clock offset, and for every satellite: the 50 bit/s message = IS-GPS-200
encoded subframes carrying that ephemeris and increasing TOW counts, the code delay as
a polynomial fitted to the exact arrival times (satellite position and clock at
transmit time from ``gpsmi.position.sat_ecef``, light time with the first-order Sagnac
term), and the matching carrier Doppler.  Test/simulation helper only.
"""
import numpy as np

from . import navbits, position, synth

F_L1 = 1575.42e6


def quantised_ephemeris(eph):
    """The ephemeris as a receiver decodes it from subframes 1-3 (LSB-quantised)."""
    out = {}
    for sid in (1, 2, 3):
        _, f = navbits.encode_nav_subframe(sid, 1, eph)
        out.update({k: v for k, v in f.items() if k not in ('ID', 'tow')})
    return out


def constellation(truth, tow0, n_sats=8, seed=4242, min_elev_deg=15.0):
    """{prn: quantised ephemeris} of satellites above `min_elev_deg` at the receiver
    `truth` (ECEF) at the time of subframe `tow0`."""
    rng = np.random.default_rng(seed)
    toe = int(round(((tow0 - 1) * 6 + 1800) / 16.0)) * 16
    up = np.asarray(truth) / np.linalg.norm(truth)
    ephs, prn = {}, 1
    while len(ephs) < n_sats:
        e = {'weekNum': 290, 'satAcc': 0, 'satHealth': 0, 'IODC': 77, 'IODE2': 77, 'IODE3': 77,
             'Tgd': rng.uniform(-2e-8, 2e-8), 'Toc': toe, 'af2': 0.0,
             'af1': rng.uniform(-5e-12, 5e-12), 'af0': rng.uniform(-4e-4, 4e-4),
             'Crs': rng.uniform(-60, 60), 'deltaN': rng.uniform(3e-9, 6e-9),
             'M0': rng.uniform(-3.1, 3.1), 'Cuc': rng.uniform(-3e-6, 3e-6),
             'e': rng.uniform(0.002, 0.02), 'Cus': rng.uniform(-9e-6, 9e-6),
             'sqrtA': 5153.6 + rng.uniform(-0.3, 0.3), 'Toe': toe,
             'Cic': rng.uniform(-2e-7, 2e-7), 'omegaBig': rng.uniform(-3.1, 3.1),
             'Cis': rng.uniform(-2e-7, 2e-7), 'i0': 0.96 + rng.uniform(-0.03, 0.03),
             'Crc': rng.uniform(150, 350), 'omegaSmall': rng.uniform(-3.1, 3.1),
             'omegaDot': rng.uniform(-8.6e-9, -7.6e-9), 'IDOT': rng.uniform(-5e-10, 5e-10)}
        q = quantised_ephemeris(e)
        x, y, z, _ = position.sat_ecef(tow0, q)
        los = np.array([x, y, z]) - truth
        if los.dot(up) / np.linalg.norm(los) > np.sin(np.radians(min_elev_deg)):
            prn += int(rng.integers(1, 4))
            ephs[prn] = q
    return ephs


def arrival_sample(eph, truth, tow0, t0_gps, m, fs):
    """Local sample time at which the code epoch transmitted `m` ms after the start of
    subframe `tow0` arrives at `truth`; local sample 0 is GPS time `t0_gps`."""
    tow, DT = tow0 + m // 6000, (m % 6000) / 1000.0
    x, y, z, dt_sv = position.sat_ecef(tow, eph, DT=DT)
    t_tx = (tow - 1) * 6 + DT - dt_sv
    X = np.array([x, y, z])
    v = np.array([-truth[1], truth[0], 0.0]) * position.OMEGA_EARTH
    tau = 0.07
    for _ in range(4):
        tau = np.linalg.norm(X - truth - v * tau) / position.GPS_C
    return (t_tx + tau - t0_gps) * fs


def nav_message(eph, tow0, n_subframes, seed, first_sid=1):
    """0/1 bits of `n_subframes` subframes, IDs first_sid, first_sid+1, ... cycling
    through 1..5, TOW counts tow0, tow0+1, ... (a TOW count names the start of the
    NEXT subframe)."""
    rng = np.random.default_rng(seed)
    out = []
    for n in range(n_subframes):
        sid = (n + first_sid - 1) % 5 + 1
        bits, _ = navbits.encode_nav_subframe(sid, tow0 + n, eph,
                                              fill=rng.integers(0, 2, (10, 24)))
        out.append(bits)
    return np.concatenate(out)


def geometric_scene(truth, seconds, tow0=50001, lead_s=0.25, n_sats=8, amp=0.11,
                    noise_sigma=0.35, seed=77, code_samples=2048, n_cyc=32, first_sid=1):
    """(Scene, info): `seconds` of IQ in which subframe `tow0` of every satellite starts
    arriving `lead_s` (+ light-time differences) after sample 0."""
    cs, fs = code_samples, 1000.0 * code_samples
    truth = np.asarray(truth, dtype=np.float64)
    ephs = constellation(truth, tow0, n_sats, seed)
    t0_gps = (tow0 - 1) * 6 + 0.07 - lead_s
    n_sub = int(seconds // 6) + 2
    ms_nodes = np.arange(-1500, int(seconds * 1000) + 1501, 250)
    sats, fit_err = [], 0.0
    for i, (prn, eph) in enumerate(ephs.items()):
        k_nodes = np.array([arrival_sample(eph, truth, tow0, t0_gps, int(m), fs)
                            for m in ms_nodes])
        k0, ks = 0.5 * seconds * fs, 0.5 * seconds * fs + 2 * fs
        coefs = np.polyfit((k_nodes - k0) / ks, ms_nodes * float(cs), 5)
        fit_err = max(fit_err, float(np.max(np.abs(
            np.polyval(coefs, (k_nodes - k0) / ks) - ms_nodes * float(cs)))))
        d1 = np.polyder(coefs)
        slope0 = np.polyval(d1, (0 - k0) / ks) / ks                   # d pos / d k at k = 0
        slope1 = np.polyval(d1, (seconds * fs - k0) / ks) / ks
        dop0, dop1 = (slope0 - 1.0) * F_L1, (slope1 - 1.0) * F_L1
        sats.append(synth.Sat(prn=prn, doppler=float(dop0), delay=0.0, amp=amp,
                              phase0=0.7 * i, doppler_rate=float((dop1 - dop0) / seconds),
                              nav_bits=nav_message(eph, tow0, n_sub, seed * 100 + prn, first_sid),
                              pos_poly=(coefs, k0, ks)))
    scene = synth.Scene(sats=sats, seed=seed, noise_sigma=noise_sigma, code_samples=cs,
                        n_cyc=n_cyc)
    return scene, {'ephs': ephs, 'truth': truth, 'tow0': tow0, 't0_gps': t0_gps,
                   'fit_err_samples': fit_err}


HANDOFF_SECONDS = 14.0


def handoff_scene():
    """(scene, info, n_blocks) of the hand-off fixture (tests/golden/ref_handoff.npz: what the
    reference's own gpsrecv.main() sent for this recording): 14 s, eight satellites with real
    subframes, at the README's position."""
    truth = np.array(position.geo_to_ecef(49.082961, 8.307581, 160.0))
    scene, info = geometric_scene(truth, HANDOFF_SECONDS, seed=79)
    return scene, info, int(HANDOFF_SECONDS / 0.032)
