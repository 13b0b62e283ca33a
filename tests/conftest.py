import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torchmd-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP extension; built on demand (cross-compiles without a GPU)."""
    import __graft_entry__ as ge

    ge.build_hip(verbose=False)
    from torchmdnet_amd import _C

    return _C.lib()
