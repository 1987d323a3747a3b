"""Thin Python objects over the C ABI: device buffers, the acquisition engine and
the tracking engine.  All arithmetic happens in libgpsmi.so on the GPU; these
classes only marshal numpy arrays and keep handles alive."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib, codes
from ._lib import (EngineError, OUT_DTYPE, PEAK_DTYPE, STATE_DTYPE, check,  # noqa: F401
                   ptr)


@dataclass
class Config:
    """The gpsglob.py constants the path depends on (reference
    src/gpsglob.py:35-131)."""
    code_samples: int = 2048
    n_cyc: int = 32
    corr_avg: int = 8
    sweep_corr_avg: int = 4
    corr_min: float = 8.0
    min_freq: float = -5000.0
    max_freq: float = 5000.0
    step_freq: float = 200
    it_sweep: int = 40
    it_sweep_all: int = 10
    max_sat: int = 11
    device: int = 0

    @property
    def ngps(self):
        return self.code_samples * self.n_cyc

    @property
    def sample_rate(self):
        return 1000 * self.code_samples

    def c_struct(self):
        return _lib.Cfg(self.code_samples, self.n_cyc, self.corr_avg,
                        self.sweep_corr_avg, self.corr_min, self.min_freq,
                        self.max_freq, self.device)


def set_default(key, value):
    """Process-wide default of a create-time / tuning option (gpsmi_set_default, gpsmi.h)."""
    check(_lib.load().gpsmi_set_default(key.encode(), int(value)), 'gpsmi_set_default')


def clear_default(key):
    check(_lib.load().gpsmi_clear_default(key.encode()), 'gpsmi_clear_default')


def device_count():
    n = C.c_int(0)
    check(_lib.load().gpsmi_device_count(C.byref(n)), 'gpsmi_device_count')
    return n.value


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(_lib.load().gpsmi_device_name(device, buf, 256), 'gpsmi_device_name')
    return buf.value.decode()


class DeviceBuffer:
    """A block of HBM owned by Python (IQ kept resident between calls)."""

    def __init__(self, nbytes, device=0):
        self.lib = _lib.load()
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self.lib.gpsmi_dev_alloc(device, self.nbytes, C.byref(p)),
              'gpsmi_dev_alloc')
        self.ptr = p

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        if offset + arr.nbytes > self.nbytes:
            raise ValueError('upload exceeds the buffer')
        dst = C.c_void_p(self.ptr.value + offset)
        check(self.lib.gpsmi_dev_upload(self.device, dst, ptr(arr), arr.nbytes),
              'gpsmi_dev_upload')

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        if offset + out.nbytes > self.nbytes:
            raise ValueError('download exceeds the buffer')
        src = C.c_void_p(self.ptr.value + offset)
        check(self.lib.gpsmi_dev_download(self.device, ptr(out), src,
                                          out.nbytes), 'gpsmi_dev_download')
        return out

    def at(self, offset):
        return C.c_void_p(self.ptr.value + int(offset))

    def free(self):
        if self.ptr:
            check(self.lib.gpsmi_dev_free(self.device, self.ptr),
                  'gpsmi_dev_free')
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """numpy array over page-locked host memory (hipHostMalloc)."""

    def __init__(self, shape, dtype):
        self.lib = _lib.load()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        check(self.lib.gpsmi_host_alloc(max(n, 1), C.byref(p)),
              'gpsmi_host_alloc')
        self._p = p
        buf = (C.c_char * n).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            check(self.lib.gpsmi_host_free(self._p), 'gpsmi_host_free')
            self._p = None


def unpack_u8iq(d_out, d_raw, n, device=0):
    """raw uint16 (Q<<8|I) -> complex64 on the device (gpsrecv.py:170-172)."""
    check(_lib.load().gpsmi_dev_unpack_u8iq(device, d_out, d_raw, n),
          'gpsmi_dev_unpack_u8iq')


def sync(device=0):
    check(_lib.load().gpsmi_dev_sync(device), 'gpsmi_dev_sync')


def _spectrum_c64(prn, cs):
    return np.ascontiguousarray(codes.replica_spectrum(prn, cs)
                                .astype(np.complex64))


class AcqEngine:
    """Search surface of gpsrecv.sweepAllSats (reference gpsrecv.py:241-274)."""

    def __init__(self, cfg=None, prns=range(1, 38)):
        self.cfg = cfg or Config()
        self.lib = _lib.load()
        h = C.c_void_p()
        cs = self.cfg.c_struct()
        check(self.lib.gpsmi_acq_create(C.byref(cs), C.byref(h)),
              'gpsmi_acq_create')
        self.h = h
        for p in prns:
            if self.cfg.code_samples == 2048:
                spec = _spectrum_c64(p, self.cfg.code_samples)
                check(self.lib.gpsmi_acq_set_replica(self.h, p, ptr(spec)),
                      'gpsmi_acq_set_replica')
            else:                      # time-domain correlation for other code lengths
                rep = np.ascontiguousarray(
                    codes.code_replica(p, self.cfg.code_samples).astype(np.float32))
                check(self.lib.gpsmi_acq_set_replica_time(self.h, p, ptr(rep)),
                      'gpsmi_acq_set_replica_time')

    raw_u8 = False

    def set_input_format(self, raw_u8):
        """raw_u8 = True: the iq of the searches that follow is the recorder's uint16
        (Q << 8 | I) samples (gpsrecv.py:162-173), decoded inside the wipe-off kernel."""
        check(self.lib.gpsmi_acq_set_input_format(self.h, 1 if raw_u8 else 0),
              'gpsmi_acq_set_input_format')
        self.raw_u8 = bool(raw_u8)

    def _host_iq(self, iq):
        want = np.uint16 if self.raw_u8 else np.complex64
        iq = np.asarray(iq)
        if iq.dtype != want:
            raise TypeError(f'iq dtype {iq.dtype} does not match the input format '
                            f'({np.dtype(want).name}; see set_input_format)')
        return np.ascontiguousarray(iq)

    def search(self, iq, prns, freqs, n_avg, out_dev=None):
        """iq: complex64 numpy array (host; uint16 after set_input_format(True)) or a
        (c_void_p, n) device pair.
        Returns a structured array [nbins, nsv] of (argmax, peak, mean, std)."""
        prn_a = np.ascontiguousarray(prns, dtype=np.int32)
        f_a = np.ascontiguousarray(freqs, dtype=np.float64)
        out = np.zeros((len(f_a), len(prn_a)), dtype=PEAK_DTYPE)
        if isinstance(iq, tuple):
            d_iq, n = iq
            check(self.lib.gpsmi_acq_search_dev(
                self.h, d_iq, n, ptr(prn_a), len(prn_a), ptr(f_a), len(f_a),
                n_avg, ptr(out), out_dev), 'gpsmi_acq_search_dev')
        else:
            iq = self._host_iq(iq)
            check(self.lib.gpsmi_acq_search(
                self.h, ptr(iq), iq.size, ptr(prn_a), len(prn_a), ptr(f_a),
                len(f_a), n_avg, ptr(out)), 'gpsmi_acq_search')
        return out

    def search_async(self, d_iq, n, prns, freqs, n_avg, out, out_dev=None):
        """Enqueue a search on device-resident iq; `out` is a pinned PEAK_DTYPE
        array [nbins, nsv] filled when wait() returns."""
        prn_a = np.ascontiguousarray(prns, dtype=np.int32)
        f_a = np.ascontiguousarray(freqs, dtype=np.float64)
        check(self.lib.gpsmi_acq_search_dev_async(
            self.h, d_iq, n, ptr(prn_a), len(prn_a), ptr(f_a), len(f_a), n_avg,
            ptr(out), out_dev), 'gpsmi_acq_search_dev_async')

    def wait(self):
        check(self.lib.gpsmi_acq_wait(self.h), 'gpsmi_acq_wait')

    def after(self, trk_engine):
        """Later searches start when the TrkEngine's enqueued work is done."""
        check(self.lib.gpsmi_acq_after_trk(self.h, trk_engine.h), 'gpsmi_acq_after_trk')

    def search_ex(self, iq, prns, freqs, n_avg):
        """search() plus corr[argmax-1], corr[argmax+1] per cell: (table, nbr)."""
        prn_a = np.ascontiguousarray(prns, dtype=np.int32)
        f_a = np.ascontiguousarray(freqs, dtype=np.float64)
        out = np.zeros((len(f_a), len(prn_a)), dtype=PEAK_DTYPE)
        nbr = np.zeros((len(f_a), len(prn_a), 2), dtype=np.float32)
        iq = self._host_iq(iq)
        check(self.lib.gpsmi_acq_search_ex(
            self.h, ptr(iq), iq.size, ptr(prn_a), len(prn_a), ptr(f_a), len(f_a),
            n_avg, ptr(out), ptr(nbr)), 'gpsmi_acq_search_ex')
        return out, nbr

    def last_ms(self):
        ms = C.c_float(0)
        check(self.lib.gpsmi_acq_last_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.lib.gpsmi_acq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrkEngine:
    """All tracking channels of one GPU (numeric part of SatStream.process,
    reference gpslib.py:1141-1210)."""

    def __init__(self, cfg=None, max_ch=12, prns=range(1, 38), streams=1):
        """streams > 1: that many independent receivers (IQ streams) share the handle and
        every launch (gpsmi_trk_set_streams); a channel is then addressed as (ch, stream)."""
        self.cfg = cfg or Config()
        self.max_ch = max_ch
        self.streams = 1
        self.lib = _lib.load()
        h = C.c_void_p()
        cs = self.cfg.c_struct()
        check(self.lib.gpsmi_trk_create(C.byref(cs), max_ch, C.byref(h)),
              'gpsmi_trk_create')
        self.h = h
        if streams != 1:
            check(self.lib.gpsmi_trk_set_streams(self.h, int(streams)), 'gpsmi_trk_set_streams')
            self.streams = int(streams)
        for p in prns:
            rep = np.ascontiguousarray(
                codes.code_replica(p, self.cfg.code_samples).astype(np.float32))
            spec = _spectrum_c64(p, self.cfg.code_samples)
            check(self.lib.gpsmi_trk_set_replica(self.h, p, ptr(rep), ptr(spec)),
                  'gpsmi_trk_set_replica')

    def _row(self, ch, stream):
        if self.streams == 1 and stream == 0:
            return ch                         # (the library checks the range itself)
        if not (0 <= ch < self.max_ch and 0 <= stream < self.streams):
            raise ValueError('channel / stream out of range')
        return stream * self.max_ch + ch

    def open(self, ch, prn, freq, delay, stream=0):
        check(self.lib.gpsmi_trk_open(self.h, self._row(ch, stream), prn, float(freq), int(delay)),
              'gpsmi_trk_open')

    def close_channel(self, ch, stream=0):
        check(self.lib.gpsmi_trk_close(self.h, self._row(ch, stream)), 'gpsmi_trk_close')

    def get_state(self, ch, stream=0):
        st = np.zeros(1, dtype=STATE_DTYPE)
        check(self.lib.gpsmi_trk_get_state(self.h, self._row(ch, stream), ptr(st)),
              'gpsmi_trk_get_state')
        return st[0]

    def set_state(self, ch, st, stream=0):
        a = np.zeros(1, dtype=STATE_DTYPE)
        a[0] = st
        check(self.lib.gpsmi_trk_set_state(self.h, self._row(ch, stream), ptr(a)),
              'gpsmi_trk_set_state')

    def erase_prev(self, ch, stream=0):
        check(self.lib.gpsmi_trk_erase_prev(self.h, self._row(ch, stream)), 'gpsmi_trk_erase_prev')

    def process(self, iq, want_out=True):
        """One closed-loop block for every open channel.  iq: complex64[NGPS]
        on the host, or a c_void_p to a device-resident block.  With streams > 1:
        [streams, NGPS] (one block per stream, back to back); out is [streams, max_ch]."""
        shape = self.max_ch if self.streams == 1 else (self.streams, self.max_ch)
        out = np.zeros(shape, dtype=OUT_DTYPE) if want_out else None
        if isinstance(iq, C.c_void_p):
            check(self.lib.gpsmi_trk_process_dev(self.h, iq, self.streams * self.cfg.ngps,
                                                 ptr(out)),
                  'gpsmi_trk_process_dev')
        else:
            want = np.uint16 if getattr(self, 'raw_u8', False) else np.complex64
            iq = np.asarray(iq)
            if iq.dtype != want:       # a silent cast would turn one format into garbage of the other
                raise TypeError(f'block dtype {iq.dtype} does not match the input format '
                                f'({np.dtype(want).name}; see set_input_format)')
            iq = np.ascontiguousarray(iq)
            if out is None:
                out = np.zeros(shape, dtype=OUT_DTYPE)
            check(self.lib.gpsmi_trk_process(self.h, ptr(iq), iq.size, ptr(out)),
                  'gpsmi_trk_process')
        return out

    def process_stream(self, iq, out=None):
        """One closed-loop block (per stream) from host memory without waiting for it
        (gpsmi_trk_process_stream): the call returns once the step of the call before last is
        complete, i.e. the host runs at most two steps ahead.  iq: a C-contiguous array in the
        handle's input format, ideally a PinnedArray's, untouched until two later calls (or
        wait()) have returned; out: optional pinned OUT_DTYPE array, valid from the same moment."""
        n = self.streams * self.cfg.ngps
        if iq.size != n or iq.dtype != (np.uint16 if getattr(self, 'raw_u8', False) else np.complex64):
            raise TypeError('block size or dtype does not match the handle')
        if not iq.flags.c_contiguous:
            raise ValueError('the block must be C-contiguous (its address is handed to the device)')
        check(self.lib.gpsmi_trk_process_stream(self.h, ptr(iq), n, ptr(out)),
              'gpsmi_trk_process_stream')

    def process_stream_ptr(self, iq_ptr, out_ptr):
        """process_stream with the two addresses already as ctypes pointers (a caller that cycles
        through a fixed ring of page-locked buffers converts them once): no checks here, the
        library checks the sample count and refuses what it cannot read."""
        rc = self.lib.gpsmi_trk_process_stream(self.h, iq_ptr, self.streams * self.cfg.ngps, out_ptr)
        if rc:
            check(rc, 'gpsmi_trk_process_stream')

    def replay(self, d_iq, nb, table, delay_used=None):
        """nb device-resident blocks, states at block start [nb, max_ch]."""
        table = np.ascontiguousarray(table, dtype=STATE_DTYPE)
        if table.shape != (nb, self.max_ch):
            raise ValueError('table must be [nb, max_ch]')
        if delay_used is not None:
            delay_used = np.ascontiguousarray(delay_used, dtype=np.int32)
            if delay_used.shape != (nb, self.max_ch):
                raise ValueError('delay_used must be [nb, max_ch]')
        out = np.zeros((nb, self.max_ch), dtype=OUT_DTYPE)
        check(self.lib.gpsmi_trk_replay(self.h, d_iq, nb, ptr(table),
                                        ptr(delay_used), ptr(out)),
              'gpsmi_trk_replay')
        return out

    def replay_load(self, nb, table, delay_used=None):
        table = np.ascontiguousarray(table, dtype=STATE_DTYPE)
        if table.shape != (nb, self.max_ch):
            raise ValueError('table must be [nb, max_ch]')
        if delay_used is not None:
            delay_used = np.ascontiguousarray(delay_used, dtype=np.int32)
            if delay_used.shape != (nb, self.max_ch):
                raise ValueError('delay_used must be [nb, max_ch]')
        check(self.lib.gpsmi_trk_replay_load(self.h, nb, ptr(table),
                                             ptr(delay_used)),
              'gpsmi_trk_replay_load')

    def replay_run(self, d_iq, nb):
        check(self.lib.gpsmi_trk_replay_run(self.h, d_iq, nb),
              'gpsmi_trk_replay_run')

    def replay_run_async(self, d_iq, nb):
        check(self.lib.gpsmi_trk_replay_run_async(self.h, d_iq, nb),
              'gpsmi_trk_replay_run_async')

    def replay_fetch_async(self, out):
        check(self.lib.gpsmi_trk_replay_fetch_async(self.h, ptr(out), out.size),
              'gpsmi_trk_replay_fetch_async')

    def wait(self):
        check(self.lib.gpsmi_trk_wait(self.h), 'gpsmi_trk_wait')

    def after(self, acq_engine):
        """Later work of this handle starts when the AcqEngine's enqueued work is done."""
        check(self.lib.gpsmi_trk_after_acq(self.h, acq_engine.h), 'gpsmi_trk_after_acq')

    def set_input_format(self, raw_u8):
        """raw_u8 = True: the blocks handed to process / replay are the recorder's uint16
        (Q << 8 | I) samples (gpsrecv.py:162-173), decoded inside the kernels that read IQ."""
        check(self.lib.gpsmi_trk_set_input_format(self.h, 1 if raw_u8 else 0),
              'gpsmi_trk_set_input_format')
        self.raw_u8 = bool(raw_u8)

    def set_option(self, key, value):
        """A tuning option of this handle (gpsmi_trk_set_option; the keys are listed in gpsmi.h)."""
        check(self.lib.gpsmi_trk_set_option(self.h, key.encode(), int(value)), 'gpsmi_trk_set_option')

    def get_option(self, key):
        v = C.c_longlong(0)
        check(self.lib.gpsmi_trk_get_option(self.h, key.encode(), C.byref(v)), 'gpsmi_trk_get_option')
        return v.value

    def set_timing(self, on):
        """Kernel-timing events for the launches that follow (see gpsmi.h)."""
        check(self.lib.gpsmi_trk_set_timing(self.h, 2 if on == 2 else int(bool(on))), 'gpsmi_trk_set_timing')

    def wait_prev(self):
        """The run before the latest one, and its read-back, are done."""
        check(self.lib.gpsmi_trk_wait_prev(self.h), 'gpsmi_trk_wait_prev')

    def replay_fetch(self, out):
        """out: C-contiguous OUT_DTYPE array (ideally from pinned_array)."""
        check(self.lib.gpsmi_trk_replay_fetch(self.h, ptr(out), out.size),
              'gpsmi_trk_replay_fetch')
        return out

    def replay_states(self, nb):
        st = np.zeros((nb, self.max_ch), dtype=STATE_DTYPE)
        check(self.lib.gpsmi_trk_replay_states(self.h, ptr(st), st.size),
              'gpsmi_trk_replay_states')
        return st

    def last_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        check(self.lib.gpsmi_trk_last_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_codephase_ms(self):
        a = C.c_float(0)
        check(self.lib.gpsmi_trk_last_codephase_ms(self.h, C.byref(a)))
        return a.value

    def close(self):
        if self.h:
            self.lib.gpsmi_trk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dumps_of(rec):
    """complex64 prompt dumps of one gpsmi_trk_out record."""
    n = int(rec['n_dumps'])
    d = rec['dumps']
    return (d[0:2 * n:2] + 1j * d[1:2 * n:2]).astype(np.complex64)
