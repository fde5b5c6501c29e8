import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hc-spmm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as the driver's
    build step does.  This is a build, not a fallback -- with hipcc absent the session fails right here."""
    import glob
    lib = os.path.join(PKG, "csrc", "libhcspmm.so")
    ext = glob.glob(os.path.join(PKG, "hybrid_kernel", "HCSPMM*.so"))
    if not os.path.exists(lib) or not ext:
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def capi():
    from hcspmm import capi as c
    c.lib()
    return c
