"""Shared fixtures.  `-m "not gpu"` runs the oracle / host-logic / ABI tests on CPU;
`-m gpu` runs the parity tests proper through the C ABI on an MI355X."""

import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ii-vision_amd", "transcoder")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    class G:
        def __getattr__(self, name):
            d = np.load(os.path.join(GOLDEN, name + ".npz"))
            setattr(self, name, d)
            return d
    return G()


@pytest.fixture(scope="session")
def dms(O):
    """Oracle CIE2000 int matrices per palette id."""
    return {pal: O.cie2000_matrix(O.PALETTE_RGB[pal])[1] for pal in (5, 0)}


class _OracleTables:
    def __init__(self, O, dms):
        self.O, self.dms, self.cache = O, dms, {}

    def get(self, mode, pal=5):
        key = (mode, pal)
        if key not in self.cache:
            self.cache[key] = self.O.build_table(mode, self.dms[pal], symmetric=True)
        return self.cache[key]


@pytest.fixture(scope="session")
def oracle_tables(O, dms):
    return _OracleTables(O, dms)


@pytest.fixture(scope="session")
def native():
    import _iiv_native
    _iiv_native.lib()
    return _iiv_native


class _DeviceTables:
    """Tables of the GPU tests, built on the device FROM THE DEVICE'S OWN delta-E matrix (cie2000_kernel):
    RGB -> delta-E -> table -> encode runs on the HIP path end to end; the oracle's matrix is only the
    checker the device's is compared with."""

    def __init__(self, native, O, dms):
        self.native, self.O, self.dms, self.cache, self.dm = native, O, dms, {}, {}

    def get(self, mode, pal=5):
        key = (mode, pal)
        if key not in self.cache:
            f_dev, dm_dev = self.native.cie2000_matrix(self.O.PALETTE_RGB[pal])
            f_orc, dm_orc = self.O.cie2000_matrix(self.O.PALETTE_RGB[pal])
            assert np.array_equal(dm_dev, self.dms[pal]) and np.array_equal(dm_dev, dm_orc)
            assert np.abs(f_dev - f_orc).max() < 1e-5          # north_star: float CIE2000 tables within 1e-5
            self.cache[key] = (self.native.build_table(mode, dm_dev, True),
                               self.native.build_store_table(mode, dm_dev))
            self.dm[key] = dm_dev
        return self.cache[key]


@pytest.fixture(scope="session")
def device_tables(native, O, dms):
    return _DeviceTables(native, O, dms)
