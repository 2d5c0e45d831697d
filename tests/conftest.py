import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge

    mod = ge.load_pkg()
    if not os.path.exists(mod.lacx.LIB_PATH):
        mod.lacx.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracleshim

    oracleshim.lib()
    return oracleshim


@pytest.fixture(scope="session")
def ref():
    import refshim

    if not refshim.available():
        pytest.skip("oracle/_ref/liblac_ref.so not built (reference sources absent)")
    return refshim
