"""CPU-side checks of the drop-in boundary: the shared library builds/loads and exports every
symbol include/p2p_hip.h declares; host-only entry points behave (no GPU compute here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from patch2pix_amd import build
    build.build(verbose=False)
    from patch2pix_amd import _lib
    return _lib


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "p2p_hip.h")).read()
    declared = set(re.findall(r"\b(p2p_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.EXPORTS), (declared ^ set(lib.EXPORTS))
    for name in declared:
        assert hasattr(lib.lib, name)


def test_version_and_workspace_query(lib):
    assert lib.p2p_version() >= 100
    small = lib.p2p_coarse_workspace_bytes(256, 8, 12, 8, 12, 2)
    big = lib.p2p_coarse_workspace_bytes(256, 60, 80, 60, 80, 2)
    assert 0 < small < big
    # 480x640 pair: hidden consensus layer (32 ch x 1200 x 1200 fp32) dominates
    assert big >= 32 * 1200 * 1200 * 4
    assert lib.p2p_coarse_workspace_bytes(0, 8, 8, 8, 8, 2) == 0


def test_argument_errors_are_reported(lib):
    out = ctypes.c_void_p()
    st = lib.p2p_ncn_create(None, None, None, None, ctypes.byref(out))
    assert st == -1 and b"null" in lib.p2p_last_error()
    with pytest.raises(RuntimeError):
        lib.check(st, "p2p_ncn_create")
    st = lib.p2p_coarse_forward(1, 1, 256, 8, 8, 8, 8, 3, 1, 1, None, 1, 1 << 30, None)
    assert st == -3
    with pytest.raises(NotImplementedError):
        lib.check(st, "p2p_coarse_forward")


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of routing through any CPU code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from patch2pix_amd.utils.eval import model_helper
    from patch2pix_amd.utils import synthetic
    with pytest.raises(RuntimeError):
        model_helper.load_model(synthetic.make_checkpoint(0, backbone=False))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "patch2pix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test-oracle", ""), f"{f} mentions the oracle"
