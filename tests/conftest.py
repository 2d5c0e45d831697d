import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, liblacx.so links the system one.  Loaded
    # torch-first the two share a runtime; the other way round torch later finds "No HIP GPUs".  Some GPU tests hand
    # torch tensors to the library, so torch goes first whatever subset of the tests is selected.
    try:
        import torch  # noqa: F401
    except Exception:
        pass


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
