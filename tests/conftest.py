"""Shared fixtures.  `-m "not gpu"`: oracle vs known answers, host logic, C-ABI symbol check.
`-m gpu`: parity of the HIP path (through the C-ABI) against the oracle on an MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    return orc


@pytest.fixture(scope="session")
def fm():
    """The product package (hyphenated directory name → importlib)."""
    import importlib
    return importlib.import_module("finmath-lib-cuda-extensions_amd")


@pytest.fixture(scope="session")
def gpu(fm):
    """Initialised runtime on cuda:0; fails loudly if the HIP library or the device is missing."""
    fm.init(0)
    yield fm
    fm.purge()


def assert_bits_equal(got, want, what=""):
    """Bit-exact fp32 equality, NaN == NaN (payload ignored)."""
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    both_nan = np.isnan(got) & np.isnan(want)
    same = (got.view(np.uint32) == want.view(np.uint32)) | both_nan
    if not same.all():
        idx = np.flatnonzero(~same)
        i = idx[0]
        raise AssertionError(f"{what}: {idx.size} of {got.size} elements differ; first at {i}: "
                             f"got {got[i]!r} (0x{got.view(np.uint32)[i]:08x}) want {want[i]!r} (0x{want.view(np.uint32)[i]:08x})")
