import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


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
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def lpg_options():
    """set kernel-selection options of the loaded library for one test (include/lpg_hip.h: lpg_set_option);
    everything is back at its default afterwards."""
    from lpgnn_amd import _lib

    def set_(**kw):
        for k, v in kw.items():
            _lib.set_option(k, v)
    yield set_
    _lib.check(_lib.load().lpg_reset_options(), "reset_options")
