"""The C ABI driven from a plain-C process (examples/cabi_coarse.c: gcc, hipMalloc'd buffers, its own stream, no
Python) against the ctypes front end on the same inputs.  Needs an MI355X:  pytest -m gpu"""
import numpy as np
import pytest
import torch

import cabi_example_io as io
import golden_util as gu
from patch2pix_amd.utils import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (run on the GPU box)")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from patch2pix_amd import ops
    return ops


@pytest.fixture(scope="module")
def weights(dev, ops):
    sd = gu.state_dict(0)
    ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"],
                         sd["ncn.conv.2.bias"], dev)
    return sd, ncn


@pytest.mark.parametrize("ksize", [2, 1])
def test_plain_c_host_matches_python_host(ksize, dev, ops, weights, tmp_path):
    """examples/cabi_coarse.c (gcc, hipMalloc'd buffers, its own stream, no Python in the process) gives the same
    corr4d / delta / matches / scores as the ctypes front end on the same inputs."""
    import subprocess
    from patch2pix_amd import build
    sd, ncn = weights
    exe = build.build_examples(verbose=False, trust_existing=True)
    H, W, B = 96, 128, 3
    pairs = [synthetic.make_correlated_pyramids(700 + i, H, W) for i in range(B)]
    fa = torch.stack([p[0][4] for p in pairs]).contiguous()
    fb = torch.stack([p[1][4] for p in pairs]).contiguous()
    io.write_input(tmp_path / "in.bin", sd, fa, fb, ksize)
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    corr, delta = ops.coarse_forward_batch(fa.to(dev), fb.to(dev), ksize, ncn)
    m, sc = ops.coarse_matches_batch(corr, delta, ksize, 8, True)
    c_corr, c_delta, c_m, c_s = io.read_output(tmp_path / "out.bin", corr.numel(), sc.numel(), ksize)
    np.testing.assert_allclose(c_corr, corr.cpu().numpy().ravel(), rtol=1e-6, atol=1e-9)
    if ksize > 1:
        assert np.array_equal(c_delta, delta.cpu().numpy().ravel())
    assert np.array_equal(c_m, m.cpu().numpy().ravel())
    np.testing.assert_allclose(c_s, sc.cpu().numpy().ravel(), rtol=1e-6)
