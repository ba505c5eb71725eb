import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")  # as vector_store_amd sets it; here before anything can initialise the HIP runtime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    # /dev/kfd is present only where a ROCm GPU is exposed; avoids initialising HIP at collection.
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _gpu_warmup():
    """On a GPU box: pay the cold-start costs (HIP init, loading the engine's code objects, first allocations) once,
    before any test with a deadline runs -- on a fresh box they take tens of seconds."""
    if _has_gpu():
        try:
            import numpy as np

            import vector_store_amd as vs
            ix = vs.HipUsearchIndex(8, vs.L2SQ)
            ix.reserve(64)
            ix.add_batch(np.arange(32, dtype=np.uint64), np.random.default_rng(0).standard_normal((32, 8)).astype(np.float32))
            ix.search(np.zeros(8, dtype=np.float32), 3)
            ix.exact_search_batch(np.zeros((1, 8), dtype=np.float32), 3)
        except Exception:  # the tests themselves will say what is wrong
            pass
    yield

