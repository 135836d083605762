import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


class _Knobs:
    """Launch-path switches of the C library inside one test process: the LWDETR_* environment is read once per process (lwdetr_tuning_set's
    comment in include/lwdetr_hip.h), so a test that switches a kernel choice goes through the setter; everything it set is cleared afterwards."""

    def __init__(self):
        self._touched = set()

    def set(self, name, value):
        from lwdetr_amd import _native
        name = name[7:] if name.startswith("LWDETR_") else name
        _native.tuning_set(name, int(value))
        self._touched.add(name)

    def clear(self):
        from lwdetr_amd import _native
        for name in self._touched:
            _native.tuning_set(name, None)
        self._touched.clear()


@pytest.fixture
def knobs():
    k = _Knobs()
    yield k
    k.clear()
