import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _reload_library_env_after_the_test():
    """a test that flips a PGNN_* knob restores os.environ through monkeypatch; the library caches the values per call site, so it
    is told to look again AFTER monkeypatch's teardown (this fixture is set up first, hence torn down last) -- also when the test
    failed half way, so that one failure does not cascade into the tests behind it"""
    yield
    mod = sys.modules.get("pretrain_gnns_amd._lib")
    lib = getattr(mod, "_lib", None) if mod is not None else None
    if lib is not None:
        lib.pgnn_reload_env()
