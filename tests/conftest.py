import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _sane_thread_count():
    """torch-CPU (the oracle) is fastest at ~32 threads on the many-core GPU hosts; the default
    (one thread per logical core, 256 there) is an order of magnitude slower for these op sizes."""
    try:
        import torch
        n = os.cpu_count() or 1
        if n > 32:
            torch.set_num_threads(32)
    except Exception:
        pass


def pytest_configure(config):
    _sane_thread_count()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference, imported through oracle/ref_shim (build container only)."""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    return ref_shim.load_reference()
