import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces are built in-tree by __graft_entry__.build(); build whatever is missing
    # (same image here and on the GPU box: hipcc cross-compiles gfx950 without a GPU)
    from ropebwt3_amd import _build
    if not (os.path.exists(_build.LIB_GPU) and os.path.exists(_build.LIB_HOST) and os.path.exists(_build.BIN_CLI)):
        _build.build_gpu()
        _build.build_host()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        _build.build_oracle()


@pytest.fixture(scope="session")
def oracle():
    from tests import util
    return util.Oracle()


@pytest.fixture(scope="session")
def engine():
    """One HIP engine handle for the session; fails loudly when the HIP library/device is missing."""
    from ropebwt3_amd import Rb3Gpu
    h = Rb3Gpu(verbose=1)
    yield h
    h.close()
