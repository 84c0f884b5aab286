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
    def __init__(self, native, dms):
        self.native, self.dms, self.cache, self.dm = native, dms, {}, {}

    def get(self, mode, pal=5):
        key = (mode, pal)
        if key not in self.cache:
            self.cache[key] = (self.native.build_table(mode, self.dms[pal], True),
                               self.native.build_store_table(mode, self.dms[pal]))
            self.dm[key] = self.dms[pal]
        return self.cache[key]


@pytest.fixture(scope="session")
def device_tables(native, dms):
    return _DeviceTables(native, dms)
