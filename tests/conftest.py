import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_ORACLE_CACHE_OWNER = False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's results (fp32 and fp64, outputs and gradients) at the larger shapes cost 30 - 60 s each and tests/test_gpu_virt_cs.py
    # re-runs some comparisons of tests/test_gpu_properties.py in child processes under other kernel switches: the CHECKER's results for one
    # (configuration, parameters, inputs) are kept for the length of this pytest session (tests/test_gpu_properties.py: _oracle_results; the
    # key is a hash of every byte that enters the oracle).  The product's results are never cached.
    global _ORACLE_CACHE_OWNER
    if "FASTEGNN_ORACLE_CACHE" not in os.environ:
        import tempfile
        os.environ["FASTEGNN_ORACLE_CACHE"] = tempfile.mkdtemp(prefix="fastegnn_oracle_")
        _ORACLE_CACHE_OWNER = True


def pytest_unconfigure(config):
    if _ORACLE_CACHE_OWNER:
        import shutil
        shutil.rmtree(os.environ.pop("FASTEGNN_ORACLE_CACHE"), ignore_errors=True)


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a device they are skipped, never silently passed.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
