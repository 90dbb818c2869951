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


# ---- collection order: the cheapest, most diagnostic tests first -------------------------------------------------------------------
# (round-3 review: the suite was collected alphabetically, so ONE failure in the op-level file under `-x` hid 76 tests behind 100+
# model-level ones.)  Order: ABI / oracle pins -> op-level parity -> 16-bit op level -> layout / host logic -> model-level steps ->
# whole iterations -> training loops -> full-size properties last.  Files not named here run at the end.
_ORDER = ['test_native_abi.py', 'test_oracle_kats.py', 'test_lds_layout.py', 'test_ops_parity.py', 'test_act16.py',
          'test_unet_normalise.py', 'test_data_pipeline.py', 'test_graph_capture.py', 'test_dp_rccl.py', 'test_dp_gloo.py',
          'test_mmsdnet_step.py', 'test_dafnet_step.py', 'test_dafnet_auto.py', 'test_evaluate_parity.py', 'test_train_loop.py',
          'test_free_running.py', 'test_experiment_cli.py', 'test_bench_cli.py', 'test_full_size.py']


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(_ORDER)))      # stable: order inside a file is kept


@pytest.fixture(autouse=True, scope='module')
def _fresh_device_state():
    """every test module starts from empty operator caches and an empty allocator cache: results must not depend on which tests ran
    before (the round-3 failure only showed with the driver's allocator history)"""
    import gc
    import torch
    from multimodal_segmentation_amd import ops
    gc.collect()
    ops.release_caches()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    yield
