"""C/A code chips and the receiver's sampled code replica.

Host-side, run once at start-up; the results are uploaded to the device by
``gpsmi.engine``.  Nothing here is copied from the reference's literal table
(reference ``src/cacodes.py:5-80``): the chips are generated from the IS-GPS-200
G1/G2 shift registers and checked against the sha256 of the reference table in
``tests/test_codes.py``.

The *replica* deliberately follows the reference's recipe
(``src/gpslib.py:62-77``), which is not a nearest-chip sampler: every chip is
doubled (2046 points) and that sequence is linearly interpolated onto
``code_samples`` points, giving a float64 array in which about a quarter of the
points are not +-1.  Peak positions depend on it, so it is restated exactly.
"""
import numpy as np

N_CHIPS = 1023

# G2 output taps (1-based register stages), IS-GPS-200 table 3-Ia, PRN 1..37
_G2_TAPS = (
    (2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10),
    (2, 3), (3, 4), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (1, 4), (2, 5),
    (3, 6), (4, 7), (5, 8), (6, 9), (1, 3), (4, 6), (5, 7), (6, 8), (7, 9),
    (8, 10), (1, 6), (2, 7), (3, 8), (4, 9), (5, 10), (4, 10), (1, 7), (2, 8),
    (4, 10),
)


def ca_chips(prn):
    """1023 chips of PRN ``prn`` (1..37) as int8 +1/-1 (logic 1 -> +1, 0 -> -1,
    the sign convention of reference ``src/cacodes.py``)."""
    if not 1 <= prn <= len(_G2_TAPS):
        raise ValueError(f"PRN {prn} out of range 1..{len(_G2_TAPS)}")
    t1, t2 = _G2_TAPS[prn - 1]
    g1 = [1] * 10
    g2 = [1] * 10
    out = np.empty(N_CHIPS, dtype=np.int8)
    for i in range(N_CHIPS):
        bit = g1[9] ^ g2[t1 - 1] ^ g2[t2 - 1]
        out[i] = 1 if bit else -1
        f1 = g1[2] ^ g1[9]
        f2 = g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]
        g1 = [f1] + g1[:9]
        g2 = [f2] + g2[:9]
    return out


def _replica_grid(n_knots, code_samples, numpy1_promotion=False):
    """Abscissae ``np.linspace(x[0], x[-1], code_samples, dtype=float32)`` of
    reference ``src/gpslib.py:75`` with ``x[0] = 0``, ``x[-1] = n_knots-1`` as
    float32 scalars.

    Under numpy >= 2 (NEP 50; the numpy of this image and of the golden
    fixtures) both end points are float32, so linspace works in float32:
    ``fl32(fl32(i) * fl32((n_knots-1)/(code_samples-1)))``, last point forced to
    ``n_knots-1``.  Under the reference's pinned numpy 1.26 the same call
    promotes to float64 and rounds once at the end; ``numpy1_promotion=True``
    selects that variant (8 of 2048 points differ, by <= 1.3e-4).
    """
    if numpy1_promotion:
        step = float(n_knots - 1) / (code_samples - 1)
        y = np.arange(code_samples, dtype=np.float64) * step
        y[-1] = n_knots - 1
        return y.astype(np.float32)
    step = np.float32(n_knots - 1) / np.float32(code_samples - 1)
    y = np.arange(code_samples, dtype=np.float32) * step
    y[-1] = np.float32(n_knots - 1)
    return y


def code_replica(prn, code_samples=2048, numpy1_promotion=False):
    """Sampled replica of one code period, float64[code_samples].

    Restates ``GPSCacode`` (reference ``src/gpslib.py:62-77``): chips doubled to
    2046 points at abscissae 0..2045, linearly interpolated (float64
    arithmetic, as ``np.interp`` does: ``slope*(x - x_j) + y_j``) at the float32
    grid of ``_replica_grid``.
    """
    y = np.repeat(ca_chips(prn).astype(np.float64), 2)      # 2046 points
    n = y.size
    xp = _replica_grid(n, code_samples, numpy1_promotion).astype(np.float64)
    j = np.minimum(np.floor(xp).astype(np.int64), n - 2)
    out = (y[j + 1] - y[j]) * (xp - j) + y[j]
    # np.interp returns the right-hand ordinate exactly at the last knot
    out[xp >= n - 1] = y[-1]
    return out


def replica_table(prns, code_samples=2048):
    """float64[len(prns), code_samples]"""
    return np.stack([code_replica(p, code_samples) for p in prns])


def replica_spectrum(prn, code_samples=2048):
    """``fft(GPSCacode(prn))`` (reference ``src/gpsrecv.py:574-577``,
    ``src/gpslib.py:1065``) as complex128[code_samples]."""
    return np.fft.fft(code_replica(prn, code_samples))
