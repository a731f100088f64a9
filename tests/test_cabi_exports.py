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
    # 480x640 pair: two fp16 planes of both feature maps + three [1200 x 1200] fp32 volumes (pooled volume and the two
    # branches of the consensus net); the 16-channel hidden volume never leaves LDS, so nothing near its 184 MB
    assert 2 * 4800 * 256 * 4 + 3 * 1200 * 1200 * 4 <= big < 40 << 20
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


def test_batch_argument_errors(lib):
    """The batch entry points validate before touching the device (status -1 = P2P_EINVAL, -3 = unsupported)."""
    assert lib.p2p_coarse_forward_batch(1, 1, 0, 256, 8, 8, 8, 8, 2, 1, 1, None, 1, 1 << 30, None) == -1     # batch 0
    assert b"batch" in lib.p2p_last_error()
    assert lib.p2p_coarse_forward_batch(1, 1, 2, 256, 8, 8, 8, 8, 3, 1, 1, None, 1, 1 << 30, None) == -3     # ksize 3
    assert lib.p2p_coarse_forward_batch(1, 1, 2, 256, 8, 8, 8, 8, 2, 1, 1, None, 1, 64, None) != 0           # workspace < one pair
    assert b"workspace" in lib.p2p_last_error()
    assert lib.p2p_coarse_matches_batch(1, None, 2, 4, 4, 4, 4, 2, 8, 1, 1, 1, None) == -1                   # delta missing, ksize 2
    assert lib.p2p_coarse_matches_batch(None, None, 2, 4, 4, 4, 4, 1, 8, 1, 1, 1, None) == -1


def test_plain_c_example_builds_and_reports_missing_device(tmp_path):
    """examples/cabi_coarse.c is C11 built by gcc against include/p2p_hip.h and the in-tree library: the header is
    valid C and the library needs nothing from Python.  Without a GPU the program must say so (exit code 3)."""
    import subprocess
    import torch
    from patch2pix_amd import build
    exe = build.build_examples(verbose=False)
    assert os.path.exists(exe)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 2 and "usage" in res.stderr
    if not torch.cuda.is_available():
        res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
        assert res.returncode == 3 and "no HIP device" in res.stderr


def test_library_reads_no_environment_variables():
    """no global mutable state besides the weight handles.  Modes and tiles are per-handle setters
    (p2p_regressor_set_mode, p2p_ncn_set_tile, p2p_conv_set_tile); the environment variables of the tools are mapped onto
    them by the Python host layer, the library itself calls no getenv."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = []
    for f in glob.glob(os.path.join(root, "patch2pix_amd", "csrc", "*.h*")):
        for i, line in enumerate(open(f, errors="replace"), 1):
            if "getenv" in line:
                hits.append(f"{os.path.basename(f)}:{i}")
    assert not hits, hits
