import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The native pieces, built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__
    return __graft_entry__.build()


@pytest.fixture(scope="session")
def cams(built):
    import fixtures_util as fx
    return fx.golden_cameras()


@pytest.fixture(scope="session")
def masks():
    import fixtures_util as fx
    return fx.golden_masks()


@pytest.fixture(scope="session")
def frames(masks):
    import fixtures_util as fx
    return fx.synthetic_frames(4, *masks[0].shape)
