"""Host side of cold acquisition: the reference's ``sweepAllSats`` /
``findCodePhase`` / ``getNewSats`` interface (reference src/gpsrecv.py:217-274,
:423-440) over the GPU search surface.

The GPU computes every (Doppler bin, SV) cell of a call; the first-hit rule
(scan bins upward, the first bin over CORR_MIN claims the SV and removes it
from the list, gpsrecv.py:256-265) is applied here in the reference's order, so
the result list is the one the reference builds.
"""
import numpy as np

from .engine import AcqEngine, Config

SAT_ALL = list(range(2, 33))          # gpsrecv.py:36


def norm_max_corr(cell):
    """(peak - mean)/std of findCodePhase (gpsrecv.py:223) from a peak record,
    evaluated in float64 like the reference."""
    return (np.float64(cell['peak']) - np.float64(cell['mean'])) \
        / np.float64(cell['std'])


class Acquisition:
    """Drop-in for the module-level search functions of gpsrecv.py."""

    def __init__(self, cfg=None, engine=None, raw_u8=False):
        """raw_u8: `data` of the calls below is the recorder's uint16 (Q << 8 | I) block as
        streamData reads it (gpsrecv.py:162-173) instead of complex64; it is decoded on the GPU."""
        self.cfg = cfg or Config()
        self.engine = engine or AcqEngine(self.cfg)
        if raw_u8:
            self.engine.set_input_format(True)

    def bin_frequencies(self, freq, it_sweep):
        """Frequencies one sweepAllSats call visits (gpsrecv.py:248, :267-272),
        and the (sweepReady, next freq) it returns."""
        c = self.cfg
        freqs, ready, it = [], False, 0
        while freq < c.max_freq and it < it_sweep:
            freqs.append(freq)
            freq += c.step_freq
            if freq >= c.max_freq:
                ready = True
                freq -= c.max_freq - c.min_freq
            it += 1
        return freqs, ready, freq

    def sweepAllSats(self, data, freq, satLst, satFound, itSweep=2):
        """Same contract as gpsrecv.sweepAllSats (gpsrecv.py:241-274):
        ``satLst`` and ``satFound`` are mutated in place; returns
        ``(sweepReady, freq, sorted(satFound, reverse=True))`` with entries
        ``(normMaxCorr, satNo, freq, delay)``."""
        c = self.cfg
        avg = min(c.sweep_corr_avg, c.n_cyc)
        freqs, ready, freq_next = self.bin_frequencies(freq, itSweep)
        if freqs and satLst:
            prns = list(satLst)
            table = self.engine.search(data, prns, freqs, avg)
            for b, f in enumerate(freqs):
                hit = []
                for j, s in enumerate(prns):
                    if s not in satLst:
                        continue
                    nmc = norm_max_corr(table[b, j])
                    if nmc > c.corr_min:
                        satFound.append((nmc, s, f, int(table[b, j]['argmax'])))
                        hit.append(s)
                for s in hit:
                    satLst.remove(s)
        return ready, freq_next, sorted(satFound, reverse=True)

    def search_table(self, data, prns, freqs, n_avg):
        """The whole surface, no pruning (BASELINE configs 2 and 4)."""
        return self.engine.search(data, prns, freqs, n_avg)


def getNewSats(actSatSet, foundSats, cpQLst, max_sat=11):
    """gpsrecv.getNewSats (gpsrecv.py:423-440): keep satellites whose channels
    still correlate, fill up to MAX_SAT with the strongest new ones."""
    good = {s for s, (q, l) in cpQLst.items() if q > 0 or l > 0}
    fs = [e for e in foundSats if e[1] not in good]
    found = good | {e[1] for e in fs[:max_sat - len(good)]}
    common = actSatSet & found
    return actSatSet - common, found - common
