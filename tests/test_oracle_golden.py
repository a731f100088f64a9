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


def test_full_size_baseline_configuration():
    """480x640, ksize 2, ptmax 400 (BASELINE.json configs[1]) -- the oracle against the unmodified reference's
    output (tests/golden/full_480x640.npz): all 2400 coarse rows, mutual set, sampled proposals, both regressors."""
    g = gu.load("full_480x640")
    sd = gu.state_dict(int(g["sd_seed"]))
    p1, p2 = gu.pair_inputs(g)
    ncn, mid_p, fine_p = orc.split_params(sd)
    with torch.no_grad():
        corr, delta = orc.coarse_forward(p1[4], p2[4], 2, ncn)
        m, s = orc.cal_coarse_matches(corr, delta, 2, 8)
    assert np.array_equal(m.numpy(), g["all_matches"].astype(np.int64))
    np.testing.assert_allclose(s.numpy(), g["all_scores"], rtol=1e-4)
    np.testing.assert_allclose(corr.reshape(-1)[::int(g["corr_sample_stride"])].numpy(), g["corr_sample"], rtol=2e-4, atol=1e-7)
    code = ((delta[0] * 2 + delta[1]) * 2 + delta[2]) * 2 + delta[3]
    assert np.array_equal(np.bincount(code.numpy().reshape(-1), minlength=16), g["delta_hist"])
    fm, _ = orc.filter_coarse(m, s, 0.0, True)
    assert np.array_equal(fm.numpy(), g["mutual_matches"].astype(np.int64))
    cm, cs = orc.filter_coarse(m, s, 0.0, True, ptmax=int(g["ptmax"]), rng=np.random.RandomState(0))
    assert np.array_equal(cm.numpy(), g["proposals"].astype(np.int64))
    with torch.no_grad():
        mid, midp, _ = orc.fine_level(p1[:4], p2[:4], cm, mid_p)
        fine, finep, _ = orc.fine_level(p1[:4], p2[:4], torch.from_numpy(g["mid"]), fine_p)
    np.testing.assert_allclose(mid.numpy(), g["mid"], atol=2e-4)
    np.testing.assert_allclose(fine.numpy(), g["fine"], atol=2e-4)
    np.testing.assert_allclose(finep.numpy(), g["fine_scores"], atol=1e-5)


@pytest.mark.parametrize("name,imsize", [("real_pair_1", None), ("real_pair_2", 640)])
def test_real_image_pairs(name, imsize):
    """Real photographs (the reference's examples/images): image loading + this repository's backbone on the CPU +
    the oracle against the unmodified reference's estimate_matches output.  The backbone's pyramids are bit-identical
    to the reference's; with the random-init checkpoint the consensus volume of a real pair is nearly flat, so a few
    coarse argmaxes are near-ties that two fp32 evaluations order differently (tests/adjudicate.py) -- rows are
    compared by coarse match and the agreement is asserted as a fraction.  (pair_3 at imsize 1024 takes minutes on the
    CPU; it is covered on the GPU box, tests/test_gpu_parity.py::test_real_image_pairs.)"""
    import os
    from patch2pix_amd.networks import resnet
    from patch2pix_amd.utils.datasets.preprocess import load_im_flexible
    g = gu.load(name)
    sd = gu.state_dict(int(g["sd_seed"]))
    d = os.path.join(gu.GOLDEN, "images", str(g["pair"]))
    t1, s1 = load_im_flexible(os.path.join(d, "1.jpg"), 2, 8, imsize=imsize)
    t2, s2 = load_im_flexible(os.path.join(d, "2.jpg"), 2, 8, imsize=imsize)
    net = resnet.ResNet34()
    net.change_stride("layer3")
    net.load_state_dict({k[len("extract."):]: v for k, v in sd.items() if k.startswith("extract.")}, strict=False)
    net.eval()
    with torch.no_grad():
        pyr1 = [f[0] for f in net.pyramid(t1[None])]
        pyr2 = [f[0] for f in net.pyramid(t2[None])]
        assert abs(gu.checksum([pyr1[4][None]]) - float(g["feat1_checksum"])) <= 1e-5 * float(g["feat1_checksum"])
        out = orc.predict_fine(pyr1, pyr2, sd)
    to_original = np.array([tuple(s1) + tuple(s2)])
    keep = np.flatnonzero(out["fine_scores"].numpy() > 0.25)
    keep = keep if keep.size else np.arange(out["fine"].shape[0])
    coarse = to_original * out["coarse"].numpy()[keep]
    fine = to_original * out["fine"].numpy()[keep]
    ref = {tuple(np.round(r, 4)): i for i, r in enumerate(g["fine_coarse"])}
    hits = [(i, ref[tuple(np.round(r, 4))]) for i, r in enumerate(coarse) if tuple(np.round(r, 4)) in ref]
    assert len(hits) >= 0.85 * len(g["fine_coarse"]), (len(hits), len(g["fine_coarse"]))
    gi, ri = np.array([h[0] for h in hits]), np.array([h[1] for h in hits])
    # rows whose mid match sits within 2e-4 px of an integer may move by one patch pixel (trunc, networks/utils.py:19)
    frac = out["mid"].numpy()[keep][gi] % 1.0
    stable = ~((frac < 2e-4) | (frac > 1 - 2e-4)).any(axis=1)
    err = np.abs(fine[gi] - g["fine_matches"][ri]).max(axis=1)
    assert err[stable].max() < 2e-3 and (~stable).sum() <= 2
    np.testing.assert_allclose(out["fine_scores"].numpy()[keep][gi][stable], g["fine_scores"][ri][stable], atol=1e-5)


@pytest.mark.parametrize("name,imsize", [("real_pair_1_contrast", None), ("real_pair_2_contrast", 640)])
def test_real_image_pairs_contrast_checkpoint(name, imsize):
    """The example photographs with the contrast checkpoint (sparse backbone features, synthetic.contrast_shift; the
    shift the fixture was made with is stored in it): the oracle's coarse rows against ALL rows the unmodified reference
    produced (before filter_coarse).  pair_1: every row equal.  pair_2 (night shot, a third of its cells have an
    identical twin): the rows that differ must be undecidable in fp32 -- the two candidates closer in fp64 than the
    error bound of an fp32 evaluation (tests/adjudicate.py) --, every decidable row of both lists must hold the fp64
    winner, and the oracle's fp32 volume must lie inside the error model (which validates the model on real data)."""
    import os
    from adjudicate import ErrorModel, assert_decidable_rows, differing_rows_are_near_ties
    from patch2pix_amd.networks import resnet
    from patch2pix_amd.utils import synthetic
    from patch2pix_amd.utils.datasets.preprocess import load_im_flexible
    g = gu.load(name)
    sd = synthetic.make_state_dict(int(g["sd_seed"]), contrast=torch.from_numpy(g["contrast_shift"]))
    d = os.path.join(gu.GOLDEN, "images", str(g["pair"]))
    t1, _ = load_im_flexible(os.path.join(d, "1.jpg"), 2, 8, imsize=imsize)
    t2, _ = load_im_flexible(os.path.join(d, "2.jpg"), 2, 8, imsize=imsize)
    net = resnet.ResNet34()
    net.change_stride("layer3")
    net.load_state_dict({k[len("extract."):]: v for k, v in sd.items() if k.startswith("extract.")}, strict=False)
    net.eval()
    ncn, _, _ = orc.split_params(sd)
    with torch.no_grad():
        fa, fb = net.pyramid(t1[None])[4][0], net.pyramid(t2[None])[4][0]
        assert abs(gu.checksum([fa[None]]) - float(g["feat1_checksum"])) <= 1e-5 * float(g["feat1_checksum"])
        corr, delta = orc.coarse_forward(fa, fb, 2, ncn)
        rows, _ = orc.cal_coarse_matches(corr, delta, 2, 8)
        ref_rows = torch.from_numpy(g["all_rows"].astype(np.int64))
        ndiff = int((rows != ref_rows).any(dim=1).sum())
        if name == "real_pair_1_contrast":
            assert ndiff == 0
        em = ErrorModel(fa, fb, sd, 2)
        assert em.check(corr, "oracle fp32 volume") < 0.25        # measured 0.05: the bound is conservative, not vacuous
        nd, worst = differing_rows_are_near_ties(rows, ref_rows, em)
        n_dec, n = assert_decidable_rows(rows, em)
        assert assert_decidable_rows(ref_rows, em)[0] == n_dec
    assert nd == ndiff and nd <= 16 and n_dec >= 0.4 * n
    print(f"\n{name}: {nd} of {n} rows differ between oracle and reference (fp64 gap <= {worst:.3f} of the bound); {n_dec} decidable")


def test_local_error_model_equals_the_full_one():
    """oracle/error_model.py: LocalErrorModel (used at 960x1280, where the full fp64 model costs minutes) evaluates the same
    fp64 volume and the same fp32 bound as ErrorModel -- on crops around the cells asked for.  Corners, edges (where the
    crops meet the zero padding of the convolutions) and interior cells; and the near-tie adjudication of a forced flip."""
    from adjudicate import ErrorModel, LocalErrorModel, differing_rows_are_near_ties_local
    from patch2pix_amd.utils import synthetic
    sd = gu.state_dict(0)
    p1, p2 = synthetic.make_correlated_pyramids(5, 160, 192)
    em = ErrorModel(p1[4], p2[4], sd, 2)
    lm = LocalErrorModel(p1[4], p2[4], sd, 2)
    sh = tuple(em.Z.shape)
    gen = torch.Generator().manual_seed(1)
    cells = [(0, 0, 0, 0), (sh[0] - 1, sh[1] - 1, sh[2] - 1, sh[3] - 1), (0, sh[1] - 1, 1, 0), (1, 1, sh[2] - 1, 2)]
    cells += [tuple(int(torch.randint(0, s, (1,), generator=gen)) for s in sh) for _ in range(6)]
    for cell in cells:
        z, e = lm.cell(*cell)
        assert abs(z - float(em.Z[cell])) <= 1e-12 * max(abs(z), 1e-30) + 1e-300 and abs(e - float(em.E[cell])) <= 1e-9 * e, cell
    # rows: the oracle's, and a copy whose first row points at the runner-up A cell of its column (a genuine difference)
    ncn, _, _ = orc.split_params(sd)
    corr, delta = orc.coarse_forward(p1[4], p2[4], 2, ncn)
    rows, _ = orc.cal_coarse_matches(corr, delta, 2, 8)
    assert differing_rows_are_near_ties_local(rows, rows, p1[4], p2[4], sd, 2) == (0, 0.0)
    col = em.Z.reshape(sh[0] * sh[1], -1)[:, 0]
    second = int(torch.topk(col, 2).indices[1])
    forged = rows.clone()
    forged[0, 0], forged[0, 1] = 8 * (2 * (second % sh[1])) + 4, 8 * (2 * (second // sh[1])) + 4
    with pytest.raises(AssertionError):
        differing_rows_are_near_ties_local(forged, rows, p1[4], p2[4], sd, 2)
