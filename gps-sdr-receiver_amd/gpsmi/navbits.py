"""Navigation-bit path of the receiver's host side: 300-bit subframe check and
field extraction, and the preamble search over the 20-ms bit stream.

Mirrors reference src/gpslib.py ``Subframe`` (:96-419: ``Extract`` :282-314,
``CheckParity`` :379-405, ``BinToInt`` :408-419, ``getDataSub1..3`` :316-371)
and ``SatStream.evalGpsBits`` (:1504-1580), including their quirks: the TLM
word's own parity is never checked, a word whose preceding D30* is set is
stored with its data bits complemented, and ``evalGpsBits`` only ever advances
to a *later* preamble candidate after a failed extraction.

Written table-driven (IS-GPS-200 parity as six 24-bit masks, fields as
(word, first, last) slices); nothing here touches the GPU.
"""
import numpy as np

GPS_PI = 3.1415926535898                     # gpslib.py:16

NO_ERR, LENGTH_ERR, PREAMBLE_ERR, PARITY_ERR, ID_ERR, NO_DATA = range(6)
PREAMBLE_BITS = np.array([1, 0, 0, 0, 1, 0, 1, 1], dtype=np.int8)    # gpslib.py:109
PREAMBLE_PM = np.array([1, -1, -1, -1, 1, -1, 1, 1], dtype=np.int8)   # gpslib.py:1045

# IS-GPS-200 table 20-XIV.  For parity bit D25..D30: which of (D29*, D30*) enters
# and which source data bits d1..d24 (1-based here, as in the ICD).
_PARITY = (
    (29, (1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23)),
    (30, (2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24)),
    (29, (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22)),
    (30, (2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23)),
    (30, (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24)),
    (29, (3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24)),
)
_PARITY_ROWS = np.zeros((6, 24), dtype=np.int64)
for _k, (_star, _taps) in enumerate(_PARITY):
    _PARITY_ROWS[_k, np.array(_taps) - 1] = 1
_PARITY_STAR = np.array([0 if s == 29 else 1 for s, _ in _PARITY])    # index into (D29*, D30*)


def word_parity(d, ds29, ds30):
    """The six parity bits of one word from its 24 source data bits."""
    star = np.array([ds29, ds30], dtype=np.int64)
    return ((_PARITY_ROWS @ np.asarray(d, dtype=np.int64)) + star[_PARITY_STAR]) % 2


def bits_to_int(bits, signed=False):
    """MSB-first bit list to integer; two's complement if signed (gpslib.py:408-419)."""
    v = 0
    for b in bits:
        v = (v << 1) | int(b)
    if signed and int(bits[0]) == 1:
        v -= 1 << len(bits)
    return v


# fields: name -> ([(word, first, last), ...], signed, scale)
_F1 = {
    'weekNum': ([(2, 0, 10)], False, 1), 'satAcc': ([(2, 12, 16)], False, 1),
    'satHealth': ([(2, 16, 22)], False, 1), 'IODC': ([(2, 22, 24), (7, 0, 8)], False, 1),
    'Tgd': ([(6, 16, 24)], True, 2 ** (-31)), 'Toc': ([(7, 8, 24)], False, 16),
    'af2': ([(8, 0, 8)], True, 2.0 ** (-55)), 'af1': ([(8, 8, 24)], True, 2.0 ** (-43)),
    'af0': ([(9, 0, 22)], True, 2.0 ** (-31)),
}
_F2 = {
    'IODE2': ([(2, 0, 8)], False, 1), 'Crs': ([(2, 8, 24)], True, 2.0 ** (-5)),
    'deltaN': ([(3, 0, 16)], True, (2.0 ** (-43), GPS_PI)),
    'M0': ([(3, 16, 24), (4, 0, 24)], True, (2.0 ** (-31), GPS_PI)),
    'Cuc': ([(5, 0, 16)], True, 2.0 ** (-29)),
    'e': ([(5, 16, 24), (6, 0, 24)], False, 2 ** (-33)),
    'Cus': ([(7, 0, 16)], True, 2.0 ** (-29)),
    'sqrtA': ([(7, 16, 24), (8, 0, 24)], False, 2.0 ** (-19)),
    'Toe': ([(9, 0, 16)], False, 16),
}
_F3 = {
    'Cic': ([(2, 0, 16)], True, 2.0 ** (-29)),
    'omegaBig': ([(2, 16, 24), (3, 0, 24)], True, (2.0 ** (-31), GPS_PI)),
    'Cis': ([(4, 0, 16)], True, 2.0 ** (-29)),
    'i0': ([(4, 16, 24), (5, 0, 24)], True, (2.0 ** (-31), GPS_PI)),
    'Crc': ([(6, 0, 16)], True, 2.0 ** (-5)),
    'omegaSmall': ([(6, 16, 24), (7, 0, 24)], True, (2.0 ** (-31), GPS_PI)),
    'omegaDot': ([(8, 0, 24)], True, (2.0 ** (-43), GPS_PI)),
    'IDOT': ([(9, 8, 22)], True, (2.0 ** (-43), GPS_PI)),
    'IODE3': ([(9, 0, 8)], False, 1),
}
FIELDS = {1: _F1, 2: _F2, 3: _F3, 4: {}, 5: {}}
# key order of the frame dictionaries evalGpsBits builds (gpslib.py:1527-1568)
FRAME_KEYS = {
    1: ['ID', 'tow', 'weekNum', 'satAcc', 'satHealth', 'Tgd', 'IODC', 'Toc', 'af2',
        'af1', 'af0', 'ST'],
    2: ['ID', 'tow', 'Crs', 'deltaN', 'M0', 'Cuc', 'IODE2', 'e', 'Cus', 'sqrtA', 'Toe',
        'ST'],
    3: ['ID', 'tow', 'Cic', 'omegaBig', 'Cis', 'i0', 'IODE3', 'Crc', 'omegaSmall',
        'omegaDot', 'IDOT', 'ST'],
    4: ['ID', 'tow', 'ST'], 5: ['ID', 'tow', 'ST'],
}


def extract_subframe(bits):
    """Subframe.Extract (gpslib.py:282-314) -> (status, fields).  `bits` is a
    0/1 array starting at a (possibly inverted) preamble."""
    if len(bits) != 300:
        return LENGTH_ERR, {}
    data = np.array(bits, dtype=np.int8)
    if not (data[:8] == PREAMBLE_BITS).all():
        data = 1 - data
        if not (data[:8] == PREAMBLE_BITS).all():
            return PREAMBLE_ERR, {}
    words = data.reshape(10, 30)
    for i in range(1, 10):                       # the TLM word itself is not checked
        ds29, ds30 = int(words[i - 1, 28]), int(words[i - 1, 29])
        if ds30 == 1:
            words[i, :24] = 1 - words[i, :24]
        if (word_parity(words[i, :24], ds29, ds30) != words[i, 24:]).any():
            return PARITY_ERR, {}
    out = {'tow': bits_to_int(words[1, :17]), 'ID': bits_to_int(words[1, 19:22])}
    if not 1 <= out['ID'] <= 5:
        return ID_ERR, {}
    for name, (parts, signed, scale) in FIELDS[out['ID']].items():
        raw = bits_to_int(np.concatenate([words[w, a:b] for w, a, b in parts]), signed)
        if isinstance(scale, tuple):             # same operation order as the reference
            out[name] = raw * scale[0] * scale[1]
        else:
            out[name] = raw * scale
    return NO_ERR, out


def eval_gps_bits(gps_bits, smp_times):
    """SatStream.evalGpsBits (gpslib.py:1504-1580): find preambles in the +-1 bit
    stream, extract consecutive subframes.  Returns (frames, rest_bits,
    rest_times) exactly as the reference does."""
    result = []
    if len(gps_bits) < 300:
        return result, gps_bits, smp_times
    gb = np.copy(gps_bits)
    corr = np.correlate(gb, PREAMBLE_PM, mode='same')
    loc = (np.flatnonzero(np.abs(corr) == 8) - 4).tolist()
    start = 0
    if loc:
        gb[gb == -1] = 0
        idx = 0
        start = loc[0]
        ok = True
        while ok and start + 300 < len(gb):
            status, f = extract_subframe(gb[start:start + 300])
            if status == NO_ERR:
                f['ST'] = smp_times[start]
                result.append({k: f[k] for k in FRAME_KEYS[f['ID']]})
                start += 300
            else:
                ok = False
                while not ok and idx < len(loc) - 1:
                    idx += 1
                    s = loc[idx]
                    ok = s > start
                if ok:
                    start = s
    return result, gps_bits[start:], smp_times[start:]


def encode_subframe(words24, ds29=0, ds30=0):
    """Test/simulation helper (the reference has no encoder): ten 24-bit source
    words -> 300 transmitted bits with IS-GPS-200 parity, given the last two
    parity bits of the previous subframe."""
    out = np.zeros(300, dtype=np.int8)
    for i in range(10):
        d = np.asarray(words24[i], dtype=np.int8)
        p = word_parity(d, ds29, ds30)
        out[30 * i:30 * i + 24] = d ^ ds30
        out[30 * i + 24:30 * i + 30] = p
        ds29, ds30 = int(p[4]), int(p[5])
    return out


def _with_zero_tail_parity(word24, ds29, ds30):
    """Choose the two non-information bits (23, 24) of a word so that its parity bits
    29 and 30 come out 0 (IS-GPS-200 does this in words 2 and 10: the next word,
    and the next subframe's preamble, are then never inverted)."""
    for t in range(4):
        w = np.array(word24, dtype=np.int8)
        w[22], w[23] = t >> 1, t & 1
        p = word_parity(w, ds29, ds30)
        if p[4] == 0 and p[5] == 0:
            return w
    raise ValueError('no solution for the non-information bits')


def encode_nav_subframe(sid, tow, eph=None, fill=None):
    """Test/simulation helper (the reference has no encoder): subframe `sid` (1..5)
    carrying TOW count `tow` and, for sid 1..3, the fields of `eph` quantised to
    their IS-GPS-200 LSBs (the inverse of ``extract_subframe``) -> 300 bits whose
    last two parity bits are 0.  `fill`: optional 10 x 24 array for the bits that
    carry no field.  Returns (bits, fields_as_decoded)."""
    w = np.zeros((10, 24), dtype=np.int8) if fill is None else np.array(fill, dtype=np.int8)
    w[0, :8] = PREAMBLE_BITS
    w[1, :17] = [(tow >> (16 - i)) & 1 for i in range(17)]
    w[1, 17:19] = 0
    w[1, 19:22] = [(sid >> 2) & 1, (sid >> 1) & 1, sid & 1]
    for name, (parts, signed, scale) in FIELDS[sid].items():
        nbits = sum(b - a for _, a, b in parts)
        sc = scale[0] * scale[1] if isinstance(scale, tuple) else scale
        raw = int(round(eph[name] / sc))
        lo, hi = (-(1 << (nbits - 1)), (1 << (nbits - 1)) - 1) if signed else (0, (1 << nbits) - 1)
        if not lo <= raw <= hi:
            raise ValueError(f'{name} = {eph[name]} does not fit {nbits} bits')
        raw &= (1 << nbits) - 1
        bits = [(raw >> (nbits - 1 - i)) & 1 for i in range(nbits)]
        k = 0
        for word, a, b in parts:
            w[word, a:b] = bits[k:k + b - a]
            k += b - a
    out = np.zeros(300, dtype=np.int8)
    ds29 = ds30 = 0
    for i in range(10):
        d = _with_zero_tail_parity(w[i], ds29, ds30) if i in (1, 9) else w[i]
        par = word_parity(d, ds29, ds30)
        out[30 * i:30 * i + 24] = d ^ ds30
        out[30 * i + 24:30 * i + 30] = par
        ds29, ds30 = int(par[4]), int(par[5])
    status, fields = extract_subframe(out)
    assert status == NO_ERR
    return out, fields
