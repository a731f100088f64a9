"""The CPU oracle against the committed golden vectors (outputs of the unmodified reference,
tests/golden/*.npz made by oracle/make_golden.py).  Runs anywhere, no reference tree needed."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import p2p_oracle as orc


@pytest.mark.parametrize("name", gu.COARSE_CASES)
def test_coarse_stage(name):
    g = gu.load(name)
    sd = gu.state_dict(int(g["sd_seed"]))
    p1, p2 = gu.coarse_inputs(g)
    ncn, _, _ = orc.split_params(sd)
    ksize = int(g["ksize"])
    corr, delta = orc.coarse_forward(p1[4], p2[4], ksize, ncn)
    np.testing.assert_allclose(corr.numpy(), g["corr4d"], rtol=2e-4, atol=1e-7)
    if ksize > 1:
        assert np.array_equal(torch.stack(delta).numpy().astype(np.int8), g["delta4d"])
    m, s = orc.cal_coarse_matches(corr, delta, ksize, 8)
    assert np.array_equal(m.numpy(), g["all_matches"])
    np.testing.assert_allclose(s.numpy(), g["all_scores"], rtol=1e-4)
    fm, fs = orc.filter_coarse(m, s, 0.0, True)
    assert np.array_equal(fm.numpy(), g["mutual_matches"])
    fu, _ = orc.filter_coarse(m, s, 0.0, False)
    assert np.array_equal(fu.numpy(), g["unique_matches"])


@pytest.mark.parametrize("name", gu.FINE_CASES)
def test_fine_levels(name):
    g = gu.load(name)
    sd = gu.state_dict(int(g["sd_seed"]))
    p1, p2 = gu.fine_inputs(g)
    _, mid_p, fine_p = orc.split_params(sd)
    for tag, params in (("int_mid", mid_p), ("float_fine", fine_p)):
        m_in = torch.from_numpy(g[tag + "_in"])
        m, p, _ = orc.fine_level(p1[:4], p2[:4], m_in, params)
        np.testing.assert_allclose(m.numpy(), g[tag + "_matches"], atol=2e-4)
        np.testing.assert_allclose(p.numpy(), g[tag + "_probs"], atol=1e-5)


@pytest.mark.parametrize("name", gu.PAIR_CASES)
def test_predict_fine(name):
    g = gu.load(name)
    sd = gu.state_dict(int(g["sd_seed"]))
    p1, p2 = gu.pair_inputs(g)
    out = orc.predict_fine(p1, p2, sd)
    assert np.array_equal(out["coarse"].numpy(), g["coarse"])
    np.testing.assert_allclose(out["mid"].numpy(), g["mid"], atol=2e-4)
    np.testing.assert_allclose(out["fine"].numpy(), g["fine"], atol=1e-3)
    np.testing.assert_allclose(out["fine_scores"].numpy(), g["fine_scores"], atol=1e-5)


def test_fp64_mode_agrees():
    """The fp64 oracle (used to adjudicate near-ties) agrees with fp32 to fp32 round-off."""
    g = gu.load("predict_fine_128x160")
    sd = gu.state_dict(int(g["sd_seed"]))
    p1, p2 = gu.pair_inputs(g)
    o32 = orc.predict_fine(p1, p2, sd)
    o64 = orc.predict_fine(p1, p2, sd, dtype=torch.float64)
    assert torch.equal(o32["coarse"], o64["coarse"])
    assert (o32["mid"].double() - o64["mid"]).abs().max() < 2e-4
