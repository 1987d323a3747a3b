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
    # no implicit padding anywhere in the records that cross the ABI: every byte is a field the
    # kernels write, so two runs can be compared bytewise (a padded tail would carry garbage)
    for dt in (_lib.OUT_DTYPE, _lib.STATE_DTYPE, _lib.PEAK_DTYPE):
        covered = sum(dt.fields[n][0].itemsize for n in dt.names)
        assert covered == dt.itemsize, (dt.names, covered, dt.itemsize)


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


@pytest.mark.parametrize('nblocks', [1, 7, 8, 9, 1024])
@pytest.mark.parametrize('ng', [1, 3, 12])
def test_codephase_correlation_workgroup_map(nblocks, ng):
    """csrc/gpsmi_wgmap.h through gpsmi_trk_corr_grid / gpsmi_trk_corr_wg_map (the function the
    host sizes the grid with and the kernel finds its unit with): every (block, group) is served
    exactly once; in a batch (>= 8 blocks) all groups of a block sit on ONE XCD (workgroup w runs
    on XCD w % 8) in consecutive slots, so the block's rows are fetched from HBM once and hit that
    XCD's L2 for the other groups; fewer than 8 blocks launch exactly nblocks * ng workgroups.
    (Round 3 inferred the mode from the grid size inside the kernel and silently lost the XCD
    grouping for every block count that is a multiple of 8, the 1024-block batch included.)"""
    from gpsmi import _lib
    lib = _lib.load()
    grid = lib.gpsmi_trk_corr_grid(nblocks, ng)
    assert grid == (nblocks * ng if nblocks < 8 else -(-nblocks // 8) * 8 * ng)
    seen, xcd_of, slots = set(), {}, {}
    b, g = C.c_int(), C.c_int()
    for wg in range(grid):
        assert lib.gpsmi_trk_corr_wg_map(nblocks, ng, wg, C.byref(b), C.byref(g)) == 0
        assert 0 <= g.value < ng
        if b.value >= nblocks:
            assert nblocks >= 8                        # padding workgroups only in batches
            continue
        assert (b.value, g.value) not in seen
        seen.add((b.value, g.value))
        xcd_of.setdefault(b.value, set()).add(wg % 8)
        slots.setdefault(b.value, []).append(wg // 8)
    assert seen == {(bb, gg) for bb in range(nblocks) for gg in range(ng)}
    if nblocks >= 8:
        for bb in range(nblocks):
            assert len(xcd_of[bb]) == 1, (bb, xcd_of[bb])
            assert slots[bb] == list(range(slots[bb][0], slots[bb][0] + ng))
    assert lib.gpsmi_trk_corr_wg_map(nblocks, ng, grid, C.byref(b), C.byref(g)) == -1


def test_options_are_part_of_the_abi():
    """The kernel-variant / threshold switches are ABI calls (gpsmi_set_default for handles to
    come, gpsmi_trk_set_option for a live one), not only environment variables."""
    from gpsmi import _lib
    lib = _lib.load()
    assert lib.gpsmi_set_default(b'correlator', 0) == 0
    assert lib.gpsmi_clear_default(b'correlator') == 0
    assert lib.gpsmi_set_default(b'no_such_option', 1) == -1
    assert b'no_such_option' in lib.gpsmi_last_error()
    assert lib.gpsmi_set_default(None, 1) == -1
    assert lib.gpsmi_trk_set_option(None, b'corr_cg', 4) == -1       # null handle
    v = C.c_longlong()
    assert lib.gpsmi_trk_get_option(None, b'corr_cg', C.byref(v)) == -1
