import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` run (no -m) skips the GPU tests; `-m gpu` selects them and then a missing
    device is a hard failure (tests/test_gpu_parity.py::_need_gpu)."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="GPU test: select with -m gpu")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
