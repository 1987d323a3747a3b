"""The C ABI without a GPU: the library loads, exports every symbol that
include/gpsmi.h declares, struct layouts agree, and compute entry points fail
loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'gpsmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gpsmi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from gpsmi import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f'{n} declared in gpsmi.h but not exported'
    assert sorted(_lib.EXPORTS) == names, 'gpsmi/_lib.py and gpsmi.h disagree'


def test_struct_layouts():
    from gpsmi import _lib
    lib = _lib.load()
    assert lib.gpsmi_abi_sizeof(0) == C.sizeof(_lib.Cfg) == 32
    assert lib.gpsmi_abi_sizeof(1) == _lib.PEAK_DTYPE.itemsize == 16
    assert lib.gpsmi_abi_sizeof(2) == _lib.STATE_DTYPE.itemsize
    assert lib.gpsmi_abi_sizeof(3) == _lib.OUT_DTYPE.itemsize
    assert lib.gpsmi_abi_sizeof(4) == _lib.OUT_DTYPE.fields['code_phase'][1]
    assert lib.gpsmi_abi_sizeof(99) == -1
    assert b'gfx950' in lib.gpsmi_version()


def test_argument_errors_do_not_need_a_gpu():
    from gpsmi import _lib
    lib = _lib.load()
    assert lib.gpsmi_device_count(None) == -1          # GPSMI_E_ARG
    assert b'null' in lib.gpsmi_last_error()
    assert lib.gpsmi_acq_create(None, None) == -1
    assert lib.gpsmi_trk_open(None, 0, 1, 0.0, 0) == -1
    assert lib.gpsmi_comm_unique_id(None) == -1
    assert lib.gpsmi_dev_free(0, None) == 0            # freeing NULL is fine


def test_no_cpu_fallback():
    """Without a GPU the engines cannot be created; nothing silently computes
    on the CPU.  (With a GPU present this test is skipped.)"""
    from gpsmi import engine
    try:
        n = engine.device_count()
    except engine.EngineError:
        n = 0
    if n > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(engine.EngineError):
        engine.AcqEngine()
    with pytest.raises(engine.EngineError):
        engine.TrkEngine()


def test_failed_create_leaves_no_handle():
    """Free-on-failure: a create call that fails (bad configuration, or no usable GPU)
    destroys whatever it had built and hands back NULL, never a half-built handle."""
    from gpsmi import _lib
    lib = _lib.load()
    bad = _lib.Cfg(1000, 32, 8, 4, 8.0, -5000.0, 5000.0, 0)          # code_samples not a multiple of 16
    h = C.c_void_p(0xDEAD)
    assert lib.gpsmi_trk_create(C.byref(bad), 12, C.byref(h)) == -1 and h.value is None
    h = C.c_void_p(0xDEAD)
    assert lib.gpsmi_acq_create(C.byref(bad), C.byref(h)) == -1 and h.value is None
    no_dev = _lib.Cfg(2048, 32, 8, 4, 8.0, -5000.0, 5000.0, 4096)    # no such device anywhere
    for create in (lambda p: lib.gpsmi_trk_create(C.byref(no_dev), 12, p),
                   lambda p: lib.gpsmi_acq_create(C.byref(no_dev), p)):
        h = C.c_void_p(0xDEAD)
        assert create(C.byref(h)) == -2 and h.value is None              # GPSMI_E_HIP
        assert len(lib.gpsmi_last_error()) > 0
    idb = (C.c_ubyte * _lib.COMM_ID_BYTES)()
    h = C.c_void_p(0xDEAD)
    assert lib.gpsmi_comm_create(idb, 0, 0, 0, C.byref(h)) == -1         # bad nranks


def test_abi_under_host_sanitizers():
    """the host shim under ASan + UBSan (`make asan-test`: this file's
    other tests against lib/libgpsmi_asan.so with the sanitizer runtime preloaded)."""
    import shutil
    import subprocess
    if os.environ.get('GPSMI_LIB_PATH'):
        pytest.skip('already running against an alternative build')
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc to build the sanitizer variant')
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'gps-sdr-receiver_amd'), 'asan-test'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'passed' in r.stdout and 'ERROR: AddressSanitizer' not in r.stdout + r.stderr


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gps-sdr-receiver_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'gps_oracle' not in src, f
                assert 'import oracle' not in src, f


def test_unpack_formula_matches_numpy_for_all_bytes():
    """gpsrecv.py:170-172: complex64/127.5 is a multiply by fl32(1/127.5)."""
    from gpsmi.synth import raw_to_c64
    v = np.arange(256, dtype=np.uint16)
    ref = raw_to_c64((v[::-1] << 8) | v)
    scl = np.float32(1.0) / np.float32(127.5)
    assert np.array_equal(ref.real, v.astype(np.float32) * scl - np.float32(1))
    assert np.array_equal(ref.imag, v[::-1].astype(np.float32) * scl - np.float32(1))
