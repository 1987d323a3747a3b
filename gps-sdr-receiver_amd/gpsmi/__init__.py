"""gpsmi -- MI355X-native GPS L1 C/A acquisition and tracking engine.

Host-side Python mirror of the reference receiver's hot-path interface
(``gpsrecv.sweepAllSats``, ``gpslib.SatStream.process``, the per-channel worker
message API) on top of ``libgpsmi.so`` (hand-written HIP kernels behind a C ABI,
see ``include/gpsmi.h``).  There is no CPU fallback: anything that computes
raises ``gpsmi.EngineError`` if the library is missing.
"""
from . import codes, synth  # noqa: F401

__all__ = ["codes", "synth"]
