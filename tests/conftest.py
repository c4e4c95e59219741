import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")


@pytest.fixture
def tuning():
    """Set library tuning values (frcnn_set_tuning) for one test; restored afterwards."""
    from objectdetection_amd import _lib
    saved = {}

    def set_(key, value):
        if key not in saved:
            saved[key] = _lib.get_tuning(key)
        _lib.set_tuning(key, value)

    yield set_
    for key, value in saved.items():
        _lib.set_tuning(key, value)
