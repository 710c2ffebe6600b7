import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from tensornetworkforml_amd import _hip
        return _hip.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly, not silently skip: only skip GPU
    # tests when the user did not ask for them explicitly.
    if 'gpu' in (config.getoption('-m') or ''):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
