import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def cpu_backend():
    """Route the C-ABI calls to the TEST-ONLY torch-CPU stand-in (tests/cpu_backend.py)."""
    from tests import cpu_backend as cb
    cb.install()
    yield cb
    cb.uninstall()
