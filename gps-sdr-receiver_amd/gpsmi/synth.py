"""Seeded synthetic GPS L1 C/A baseband, in the reference's raw file format.

The reference ships no usable recording (``data/test.bin`` is absent from the
checkout), so every test and benchmark input is generated here.
The generator is counter-based (a 64-bit integer hash of the absolute sample
index), so any block of a scene can be produced on its own, on any machine,
with the same bits; only *outputs* of the reference are committed as golden
fixtures.

File format = what ``streamData`` reads (reference ``src/gpsrecv.py:162-173``)
and ``gpsbin.py`` writes: little-endian uint16 per sample, low byte I, high
byte Q, offset-binary 8 bit; ``sample = (I + jQ)/127.5 - (1+1j)`` as complex64.

Signal model: for each satellite
``A * replica[(k - delay(k)) mod CS] * d(k) * exp(j(2 pi f (k+1)/fs + phi0))``
with the receiver's own interpolated replica (fractional delays by linear
interpolation between replica samples), optional +-1 data bits every 20 code
periods, plus complex white Gaussian noise, then 8-bit quantisation.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import codes

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays (wraps modulo 2**64)."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def _uniform(seed, stream, idx):
    """float64 in (0,1), a pure function of (seed, stream, idx)."""
    key = _mix64(np.uint64(seed) * np.uint64(0x1000003) + np.uint64(stream))
    h = _mix64(idx.astype(np.uint64) ^ key)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def gaussian_pair(seed, idx):
    """Two independent N(0,1) float64 arrays for absolute sample indices idx
    (Box-Muller on two hashed uniforms)."""
    u1 = _uniform(seed, 1, idx)
    u2 = _uniform(seed, 2, idx)
    r = np.sqrt(-2.0 * np.log(u1))
    a = 2.0 * np.pi * u2
    return r * np.cos(a), r * np.sin(a)


@dataclass
class Sat:
    prn: int
    doppler: float            # Hz
    delay: float              # samples, position of the code start in block 0
    amp: float = 0.06
    phase0: float = 0.0       # rad
    delay_rate: float = None  # samples per sample; None -> -doppler/1575.42e6
    data_bits: bool = True
    doppler_rate: float = 0.0  # Hz/s
    nav_bits: object = None   # optional 0/1 array: the 50 bit/s message, repeated
    pos_poly: object = None   # optional (coefs, k0, ks): samples since code epoch 0 as a
                              # polynomial in (k - k0)/ks (delay following a geometric range)


@dataclass
class Scene:
    sats: List[Sat]
    seed: int = 1
    noise_sigma: float = 0.35        # per complex sample
    code_samples: int = 2048
    n_cyc: int = 32
    _rep: dict = field(default_factory=dict, repr=False)

    @property
    def ngps(self):
        return self.code_samples * self.n_cyc

    @property
    def sample_rate(self):
        return 1000.0 * self.code_samples

    def _replica(self, prn):
        if prn not in self._rep:
            self._rep[prn] = codes.code_replica(prn, self.code_samples)
        return self._rep[prn]

    def block_float(self, block_no, n=None):
        """complex128 samples of block `block_no` before quantisation."""
        cs = self.code_samples
        n = self.ngps if n is None else n
        k = np.arange(n, dtype=np.float64) + float(block_no) * self.ngps
        fs = self.sample_rate
        x = np.zeros(n, dtype=np.complex128)
        for s in self.sats:
            rate = (-s.doppler / 1575.42e6) if s.delay_rate is None \
                else s.delay_rate
            if s.pos_poly is not None:
                coefs, k0, ks = s.pos_poly
                pos = np.polyval(coefs, (k - k0) / ks)
            else:
                pos = k - (s.delay + rate * k)          # samples since code start
            period = np.floor(pos / cs)
            off = pos - period * cs
            j = np.floor(off).astype(np.int64)
            a = off - j
            rep = self._replica(s.prn)
            code = (1.0 - a) * rep[j % cs] + a * rep[(j + 1) % cs]
            if s.nav_bits is not None:
                bit_no = np.floor(period / 20.0).astype(np.int64)
                nav = np.asarray(s.nav_bits)
                code = code * (2.0 * nav[bit_no % len(nav)] - 1.0)
            elif s.data_bits:
                bit_no = np.floor(period / 20.0).astype(np.int64)
                h = _mix64((bit_no + (1 << 40)).astype(np.uint64)
                           ^ np.uint64(self.seed * 1000 + s.prn))
                code = code * (1.0 - 2.0 * (h & np.uint64(1)).astype(np.float64))
            t = (k + 1.0) / fs
            ph = 2.0 * np.pi * (s.doppler * t + 0.5 * s.doppler_rate * t * t) \
                + s.phase0
            x += s.amp * code * np.exp(1j * ph)
        if self.noise_sigma > 0:
            g1, g2 = gaussian_pair(self.seed, k.astype(np.int64))
            x += (self.noise_sigma / np.sqrt(2.0)) * (g1 + 1j * g2)
        return x

    def block_raw(self, block_no, n=None):
        """uint16 raw samples (Q<<8 | I), the on-disk format."""
        x = self.block_float(block_no, n)
        i = np.clip(np.rint((x.real + 1.0) * 127.5), 0, 255).astype(np.uint16)
        q = np.clip(np.rint((x.imag + 1.0) * 127.5), 0, 255).astype(np.uint16)
        return (q << 8) | i

    def block(self, block_no, n=None):
        """complex64 samples exactly as streamData hands them on."""
        return raw_to_c64(self.block_raw(block_no, n))


def raw_to_c64(raw):
    """uint16 (Q<<8|I) -> complex64, the arithmetic of reference
    ``src/gpsrecv.py:170-172`` (complex128 sum cast to complex64, complex64
    divide, complex64 subtract)."""
    im, re = np.divmod(raw, 256)
    return np.asarray(re + 1j * im, dtype=np.complex64) / 127.5 - (1 + 1j)


def default_scene(n_sats=8, seed=7, code_samples=2048, n_cyc=32, amp=0.06,
                  noise_sigma=0.35, prns=None):
    """A reproducible scene: `n_sats` satellites with spread Dopplers and
    delays, hashed from `seed`."""
    if prns is None:
        order = np.argsort(_uniform(seed, 11, np.arange(2, 33)))
        prns = [int(p) for p in (np.arange(2, 33)[order][:n_sats])]
    idx = np.arange(len(prns))
    dop = -4000.0 + 8400.0 * _uniform(seed, 12, idx)
    dly = np.floor(code_samples * _uniform(seed, 13, idx)) \
        + np.round(_uniform(seed, 15, idx), 2)
    ph0 = 2.0 * np.pi * _uniform(seed, 14, idx)
    sats = [Sat(prn=p, doppler=float(np.round(dop[i], 1)), delay=float(dly[i]),
                amp=amp, phase0=float(ph0[i])) for i, p in enumerate(prns)]
    return Scene(sats=sats, seed=seed, noise_sigma=noise_sigma,
                 code_samples=code_samples, n_cyc=n_cyc)
