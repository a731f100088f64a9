import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference, imported through oracle/ref_shim (build container only)."""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present (GPU box)")
    return ref_shim.load_reference()
