"""Ingest: ring-buffer overflow semantics and the raw-file
reader of reference gpsrecv.py:76-104, :162-176; the device unpack is bit-exact
against the numpy decode for every possible raw value."""
import numpy as np
import pytest

from gpsmi import ingest
from gpsmi.synth import raw_to_c64


def test_ring_buffer_overflow_drops_everything_and_reports_it():
    rb = ingest.RingBuffer()
    assert rb.pull() == ([], 0)
    for i in range(16):
        rb.push(i)
    assert rb.nbuf == 16 and rb.bufskip == 0
    rb.push(16)                                   # 17th: the 16 buffered streams are lost
    assert rb.nbuf == 1 and rb.bufskip == 16
    assert rb.pull() == (16, 16)                  # loss reported once, with the next block
    assert rb.pull() == ([], 0)
    rb.push('a'); rb.push('b')
    assert rb.pull() == ('a', 0) and rb.pull() == ('b', 0)


def test_read_raw_blocks(tmp_path):
    raw = (np.arange(3 * 1000 + 17) % 65536).astype('<u2')
    p = tmp_path / 'x.bin'
    raw.tofile(p)
    blocks = list(ingest.read_raw_blocks(str(p), ngps=1000))
    assert len(blocks) == 3 and np.array_equal(blocks[2], raw[2000:3000])
    assert len(list(ingest.read_raw_blocks(str(p), ngps=1000, start_stream=2))) == 1
    assert np.array_equal(ingest.decode_host(blocks[0]), raw_to_c64(raw[:1000]))


@pytest.mark.gpu
def test_device_unpack_is_bit_exact_for_every_raw_value():
    """fl32(fl32(v) * fl32(1/127.5)) - 1 per component: what numpy's portable
    complex64-by-real division loop computes (the loop the fixtures were made
    with).  numpy's AVX-512 loop on some hosts differs from it in the last ulp, so
    numpy itself is only required to agree within 1 ulp."""
    raw = np.arange(65536, dtype=np.uint16)       # every (I, Q) byte pair once
    di = ingest.DeviceIngest(2)
    di.put(1, raw)
    got = di.get(1)
    di.free()
    assert got.dtype == np.complex64
    scl = np.float32(1.0) / np.float32(127.5)
    re = (raw & 255).astype(np.float32) * scl - np.float32(1)
    im = (raw >> 8).astype(np.float32) * scl - np.float32(1)
    assert np.array_equal(got.real, re) and np.array_equal(got.imag, im)
    ref = raw_to_c64(raw)
    ulp = np.spacing(np.maximum(np.abs(ref.real), np.abs(ref.imag)).astype(np.float32))
    assert np.all(np.abs(got.real - ref.real) <= ulp)
    assert np.all(np.abs(got.imag - ref.imag) <= ulp)
