import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _tgru_switches():
    """The library caches its TGRU_* environment switches.  Tests that change them call
    timegan_amd.reload_switches() after monkeypatch.setenv; this fixture (set up first, torn down last,
    i.e. after monkeypatch has restored the environment) makes the library read the restored values."""
    yield
    lib_path = os.path.join(ROOT, "time-gan_amd", "libtgru.so")
    if os.path.exists(lib_path):
        import timegan_amd as tg
        tg.reload_switches()
