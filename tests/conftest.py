import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """one device context for the session; fails loudly when the HIP library or the GPU is missing"""
    from jxlatte_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def assert_bits_equal(a, b, what="", any_nan=False):
    """bit-exact comparison of float32 / int arrays (NaN payloads included). any_nan=True compares NaNs as one value:
    Java leaves NaN bit patterns unspecified (Float.floatToIntBits collapses them), and an invalid operation such as
    inf * 0 yields 0xFFC00000 on x86 SSE (the oracle's host) but 0x7FC00000 on gfx950."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype == np.float32:
        ai, bi = a.view(np.uint32), b.view(np.uint32)
        if any_nan:
            ai = np.where(np.isnan(a), np.uint32(0x7FC00000), ai)
            bi = np.where(np.isnan(b), np.uint32(0x7FC00000), bi)
    else:
        ai, bi = a, b
    if not np.array_equal(ai, bi):
        bad = np.argwhere(ai != bi)
        first = tuple(bad[0])
        raise AssertionError("%s: %d of %d elements differ; first at %s: got %r expected %r" %
                             (what, bad.shape[0], a.size, first, a[first], b[first]))
