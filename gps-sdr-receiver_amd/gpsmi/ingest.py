"""Ingest side of the receiver: the raw-file reader and the overflow-dropping
ring buffer of reference src/gpsrecv.py:47-104, :153-186, with the sample decode
(``(I + jQ)/127.5 - (1+1j)``, gpsrecv.py:170-172) done on the GPU so that only
2 bytes per sample cross PCIe instead of 8.  ``StreamedInput`` is the consumer end
of that ring buffer on the GPU: blocks go from a small ring of page-locked buffers
into the tracking engine without a host wait per block
(``gpsmi_trk_process_stream``)."""
from collections import deque

import numpy as np

from . import engine as E
from .synth import raw_to_c64

MAXBUFSIZE = 16                       # gpsrecv.py:47


class RingBuffer:
    """pushToBuffer / pullFromBuffer (gpsrecv.py:76-104): when all slots are
    taken the whole buffer is dropped and the loss is reported with the next
    block that is pulled."""

    def __init__(self, maxsize=MAXBUFSIZE):
        self.maxsize = maxsize
        self.buf = deque([], maxlen=maxsize)
        self.nbuf = 0
        self.bufskip = 0

    def push(self, data):
        if self.nbuf >= self.maxsize:
            self.buf.clear()
            self.nbuf = 0
            self.bufskip += self.maxsize
        self.buf.append(data)
        self.nbuf += 1

    def pull(self):
        """-> (data, skipped streams); ([], 0) when empty."""
        try:
            data = self.buf.popleft()
            self.nbuf -= 1
            skip, self.bufskip = self.bufskip, 0
        except IndexError:
            data, skip = [], 0
        return data, skip


def read_raw_blocks(path, ngps=65536, start_stream=0):
    """streamData's file loop (gpsrecv.py:162-176): little-endian uint16
    (Q<<8 | I) blocks of NGPS samples; a short last block ends the stream."""
    with open(path, 'rb') as f:
        for _ in range(start_stream):
            np.fromfile(f, dtype=np.uint16, count=ngps)
        while True:
            raw = np.fromfile(f, dtype=np.uint16, count=ngps)
            if len(raw) != ngps:
                return
            yield raw


def decode_host(raw):
    """The reference's own decode on the host (numpy)."""
    return raw_to_c64(raw)


class DeviceIngest:
    """Raw blocks -> complex64 blocks resident in HBM: upload 2 B/sample, unpack
    there (gpsmi_dev_unpack_u8iq, bit-identical to the numpy expression)."""

    def __init__(self, n_blocks, ngps=65536, device=0):
        self.ngps, self.device = ngps, device
        self.n_blocks = n_blocks
        self.d_raw = E.DeviceBuffer(ngps * 2, device)
        self.d_iq = E.DeviceBuffer(n_blocks * ngps * 8, device)

    def put(self, slot, raw):
        """Decode one raw block into slot `slot`; returns its device pointer."""
        raw = np.ascontiguousarray(raw, dtype=np.uint16)
        if raw.size != self.ngps or not 0 <= slot < self.n_blocks:
            raise ValueError('bad block size or slot')
        self.d_raw.upload(raw)
        dst = self.d_iq.at(slot * self.ngps * 8)
        E.unpack_u8iq(dst, self.d_raw.ptr, self.ngps, self.device)
        return dst

    def get(self, slot):
        return self.d_iq.download(np.complex64, self.ngps, slot * self.ngps * 8)

    def free(self):
        self.d_raw.free()
        self.d_iq.free()


class StreamedInput:
    """streamData -> pushToBuffer -> processData (gpsrecv.py:153-186, :76-104, :445-548) with the
    consumer on the GPU: ``feed(block)`` copies the block (one per stream of the engine, raw uint16
    after ``trk.set_input_format(True)``, complex64 otherwise) into the next of `depth` page-locked
    buffers and enqueues upload + tracking kernels without waiting for them.  The library lets the
    host run two steps ahead (gpsmi_trk_process_stream returns once the step before last is done),
    so three buffers are enough and the memory in use is bounded however long the stream is.
    With `keep_outputs`, ``feed`` returns the records [streams, ch] of the block fed two calls
    earlier (None for the first two) and ``drain()`` the ones still outstanding."""

    def __init__(self, trk, depth=3, keep_outputs=False):
        if depth < 3:
            raise ValueError('three buffers at least: one being filled, two with the device')
        self.trk = trk
        dt = np.uint16 if getattr(trk, 'raw_u8', False) else np.complex64
        shape = (trk.streams, trk.cfg.ngps)
        self.ring = [E.PinnedArray(shape, dt) for _ in range(depth)]
        self.keep = keep_outputs
        self.outs = ([E.PinnedArray((trk.streams, trk.max_ch), E.OUT_DTYPE) for _ in range(depth)]
                     if keep_outputs else [])
        self.k = 0           # blocks fed
        self.taken = 0       # records handed back

    def feed(self, block):
        # (buffer k % depth was last handed over `depth` >= 3 calls ago; the calls since have returned,
        # so that step is complete -- see the contract in include/gpsmi.h)
        j = self.k % len(self.ring)
        slot = self.ring[j]
        block = np.asarray(block)
        if block.dtype != slot.array.dtype:    # a silent cast would turn one format into garbage of the other
            raise TypeError(f'block dtype {block.dtype} does not match the handle\'s input format '
                            f'({slot.array.dtype.name}; see TrkEngine.set_input_format)')
        slot.array[...] = block.reshape(slot.array.shape)
        self.trk.process_stream(slot.array, self.outs[j].array if self.keep else None)
        self.k += 1
        if self.keep and self.k - self.taken > 2:
            return self._take()
        return None

    def _take(self):
        rec = self.outs[self.taken % len(self.outs)].array.copy()
        self.taken += 1
        return rec

    def drain(self):
        """Wait for everything enqueued; returns the records not yet handed back [blocks, streams, ch]
        (None without keep_outputs or when there are none)."""
        self.trk.wait()
        if not self.keep or self.taken == self.k:
            return None
        return np.stack([self._take() for _ in range(self.k - self.taken)])

    def free(self):
        self.trk.wait()
        for p in self.ring + self.outs:
            p.free()
        self.ring, self.outs = [], []
