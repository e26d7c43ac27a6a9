import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests; an explicit `-m gpu` run (the GPU box)
    never skips -- there a missing device or library must fail loudly (tests/test_gpu_parity.py `_need_gpu`)."""
    expr = config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        return
    try:
        from opticomlib_amd import _lib
        have = _lib.device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no MI355X visible (run `pytest -m gpu` on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
