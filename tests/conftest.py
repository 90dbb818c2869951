import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The oracle runs on torch-CPU.  A GPU box exposes far more logical CPUs (256) than the share a job may use (16 per GPU):
    # torch's default of one thread per logical CPU then oversubscribes the share and the oracle crawls.
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 16)))


@pytest.fixture
def cpu_backend():
    """Route the C-ABI calls to the TEST-ONLY torch-CPU stand-in (tests/cpu_backend.py)."""
    from tests import cpu_backend as cb
    cb.install()
    yield cb
    cb.uninstall()
