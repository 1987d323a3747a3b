"""Shared test setup.

`-m "not gpu"`: oracle vs golden fixtures, host logic, C-ABI symbol check.
`-m gpu`: parity tests proper, HIP path through the C ABI vs the oracle.
Nothing here (or in any test) reads /root/reference: the fixtures under
tests/golden/ were frozen from it by oracle/make_golden.py.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'gps-sdr-receiver_amd'), os.path.join(ROOT, 'oracle'),
          ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_default():
    return load_golden('ref_default.npz')


@pytest.fixture(scope='session')
def golden_hirate():
    return load_golden('ref_hirate.npz')


def scene_for(config):
    """Same scenes as oracle/make_golden.py:scene_for."""
    from gpsmi import synth
    if config == 'default':
        return synth.default_scene(12, seed=7, code_samples=2048, n_cyc=32)
    return synth.default_scene(12, seed=11, code_samples=16368, n_cyc=8)


_BLOCK_CACHE = {}


def scene_blocks(config, first, count):
    """complex64 blocks [first, first+count) of a fixture scene (memoised)."""
    sc = scene_for(config)
    out = []
    for b in range(first, first + count):
        key = (config, b)
        if key not in _BLOCK_CACHE:
            _BLOCK_CACHE[key] = sc.block(b)
        out.append(_BLOCK_CACHE[key])
    return out
