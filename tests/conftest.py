import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cv-ssl-mis_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        # the GPU box has 128 cores / 256 threads and torch takes them all by default; the CPU oracle the parity tests
        # evaluate beside the HIP path (conv / linear layers of 1-48 samples) is 2-2.5x FASTER at 32 threads (bench.py's
        # thread sweep, every round) -- two thirds of the suite's wall time is that oracle
        torch.set_num_threads(min(int(os.environ.get("MIS_TEST_THREADS", "32")), os.cpu_count() or 1))
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """GPU tests start with NaNs in every CU's LDS (mis_debug_poison_lds): a kernel that reads a cell it never wrote --
    typically under a zero weight -- then fails every time instead of depending on what ran before it."""
    if "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            from mis_hip import lib
            L = lib.load()
            lib.check(L.mis_debug_poison_lds(None, lib.stream_ptr()), "mis_debug_poison_lds")
    yield
