import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP library through the C ABI)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def gist(oracle):
    return oracle.read_mtx(os.path.join(GOLDEN, "GIST.mtx"))


@pytest.fixture(scope="session")
def modsim():
    return np.loadtxt(os.path.join(GOLDEN, "modsimdata.csv"), delimiter=",").astype(np.float32)


@pytest.fixture(scope="session")
def emul_lib():
    """TEST-ONLY build of the kernel sources on the fiber workgroup emulator (tests/emul)."""
    import ctypes
    from cogaps_amd import _capi
    libs = {}

    def get(win=256, extra="", tag=""):
        if (win, tag) not in libs:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "WIN=%d" % win, "EXTRA=" + extra, "TAG=" + tag])
            libs[(win, tag)] = _capi.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libcogaps_emul_TESTONLY_w%d%s.so" % (win, tag))))
        return libs[(win, tag)]
    return get


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  GPU tests fail (not skip) when it is missing: there is no fallback."""
    from cogaps_amd import _capi
    return _capi.load()
