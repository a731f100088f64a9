"""The drop-in route INTEGRATION.md documents for image-matching-toolbox: put <repo>/patch2pix_amd on sys.path (where
the toolbox puts third_party/patch2pix) and import the reference's module paths.  Runs in fresh interpreters so
that nothing imported by the test session helps.  (With a GPU the same route also loads a model and matches a pair:
tests/test_gpu_parity.py::test_documented_import_route_on_gpu.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "patch2pix_amd")


def _run(code, cwd):
    return subprocess.run([sys.executable, "-W", "error::ImportWarning", "-c", code], capture_output=True, text=True, cwd=cwd)


def test_reference_module_paths_import(tmp_path):
    code = f"""
import sys
sys.path.append({PKG!r})
from utils.eval.model_helper import load_model, estimate_matches, refine_matches, init_patch2pix_matcher, init_ncn_matcher
from networks.patch2pix import Patch2Pix
from utils.datasets.preprocess import load_im_flexible, load_im_tensor
from networks.utils import filter_coarse
import inspect
assert list(inspect.signature(load_model).parameters) == ['ckpt_path', 'method', 'lprint']
assert list(inspect.signature(estimate_matches).parameters) == ['net', 'im1', 'im2', 'ksize', 'ncn_thres', 'mutual', 'io_thres', 'eval_type', 'imsize']
assert list(inspect.signature(refine_matches).parameters) == ['im1_path', 'im2_path', 'net', 'coarse_matcher', 'io_thres', 'imsize', 'coarse_only']
assert {ROOT!r} not in sys.path, "the route must not need the repository root on sys.path"
print("IMPORT_OK")
"""
    res = _run(code, str(tmp_path))
    assert res.returncode == 0 and "IMPORT_OK" in res.stdout, res.stderr[-3000:]


def test_bare_and_qualified_names_are_one_module(tmp_path):
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
sys.path.append({PKG!r})
import patch2pix_amd.networks.patch2pix as q
import networks.patch2pix as b
import utils.eval.model_helper as h
import patch2pix_amd.utils.eval.model_helper as hq
import patch2pix_amd._lib as lib
assert q is b and h is hq and b.Patch2Pix is q.Patch2Pix
assert sum(1 for k in sys.modules if k.endswith('_lib')) == 1, "the shared library must be bound once"
print("ALIAS_OK")
"""
    res = _run(code, str(tmp_path))
    assert res.returncode == 0 and "ALIAS_OK" in res.stdout, res.stderr[-3000:]


def test_load_model_without_gpu_fails_loudly(tmp_path):
    """No CPU fallback through this route either."""
    import torch
    if torch.cuda.is_available():
        return
    code = f"""
import sys
sys.path.append({PKG!r})
from utils.eval.model_helper import load_model
try:
    load_model({{'state_dict': {{}}}}, method='nc')
except RuntimeError as e:
    print("LOUD", e)
"""
    res = _run(code, str(tmp_path))
    assert "LOUD" in res.stdout, res.stdout + res.stderr[-2000:]


def test_alias_does_not_capture_foreign_packages(tmp_path):
    """Another project's top-level `utils` package must stay importable in the same process: the alias finder answers
    only for submodules of OUR `utils` / `networks`, and `_alias.uninstall()` drops the aliases altogether."""
    other = tmp_path / "other"
    (other / "utils").mkdir(parents=True)
    (other / "utils" / "__init__.py").write_text("WHO = 'foreign'\n")
    (other / "utils" / "helper.py").write_text("X = 41\n")
    code = f"""
import sys
sys.path.append({PKG!r})
import utils.eval.model_helper as ours
assert ours.__name__ == 'patch2pix_amd.utils.eval.model_helper'
import patch2pix_amd._alias as alias
alias.uninstall()
assert 'utils' not in sys.modules and 'utils.eval' not in sys.modules
sys.path.insert(0, {str(other)!r})
import utils, utils.helper
assert utils.WHO == 'foreign' and utils.helper.X == 41
import patch2pix_amd.utils.eval.model_helper as again
assert again is ours
print("FOREIGN_OK")
"""
    res = _run(code, str(tmp_path))
    assert res.returncode == 0 and "FOREIGN_OK" in res.stdout, res.stderr[-3000:]
