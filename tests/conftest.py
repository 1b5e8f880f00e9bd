import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/vbx_oracle.c."""
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes binding of libvoxbox_hip.so)."""
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def vb(pkg):
    """A live GPU context; GPU tests fail (not skip) when the HIP library cannot run."""
    ctx = pkg.VoxBox(0)
    yield ctx
    ctx.close()


def rel_close(a, b, rtol=1e-6, floor=1e-6):
    """SURVEY 8d parity metric: |a-b| <= rtol * max(|b|, floor * ||b||_inf)."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) if b.size else 0.0
    return np.abs(a - b) <= rtol * np.maximum(np.abs(b), floor * scale) + 1e-300
