"""Parity of the HIP matching path (through the C ABI) with the CPU oracle and with the golden
vectors of the unmodified reference.  Needs an MI355X:  pytest -m gpu

Tolerances (BASELINE.json north_star): match indices bit-exact, regressed coordinates within
1e-3 px, scores within 1e-5.  Two legitimate sources of discontinuity are handled explicitly and
reported rather than hidden: near-ties in an argmax (top-2 gap below fp32 round-off of the
different summation order) and a mid-level coordinate within 1e-4 px of an integer, where
trunc() (networks/utils.py:19) moves the whole fine-level patch by one pixel.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import p2p_oracle as orc
from patch2pix_amd.utils import synthetic

pytestmark = pytest.mark.gpu

COORD_TOL = 1e-3
SCORE_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (run on the GPU box)")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from patch2pix_amd import ops
    return ops


MODES = ["fp16x2", "fp16x2w", "f32"]   # fp32-equivalent on two fp16 planes (direct conv2 | Winograd conv2), exact fp32 MFMA
DEFAULT_MODE = "fp16x2w"                        # include/p2p_hip.h: P2P_REGRESS_DEFAULT


@pytest.fixture(scope="module")
def cweights(dev, ops):
    """Coarse-stage weights.  The coarse stage has ONE arithmetic whatever the batch: fp32-equivalent fp16x2 (correlation
    GEMM and the fused consensus kernel on the fp16 matrix cores, everything else fp32)."""
    sd = gu.state_dict(0)
    ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"],
                         sd["ncn.conv.2.bias"], dev)
    return sd, ncn, None, None


@pytest.fixture(scope="module", params=MODES)
def weights(dev, ops, request, cweights):
    """All arithmetic modes of the regressor kernels are held to the same bars."""
    sd, ncn = cweights[0], cweights[1]
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid, fine = ops.RegressorWeights(sub("regress_mid."), dev), ops.RegressorWeights(sub("regress_fine."), dev)
    mid.set_mode(request.param)
    fine.set_mode(request.param)
    assert mid.mode == request.param
    return sd, ncn, mid, fine


def _gpu(pyr, dev):
    return [t.to(dev) for t in pyr]


# ------------------------------------------------------------------------------------------ coarse
def _check_coarse(corr, delta, ref_corr, ref_delta, ksize):
    """corr4d within fp32 round-off; delta bit-exact except on near-ties (reported)."""
    np.testing.assert_allclose(corr, ref_corr, rtol=2e-4, atol=1e-7)
    if ksize > 1:
        k = ksize
        ref_s = ((ref_delta[0] * k + ref_delta[1]) * k + ref_delta[2]) * k + ref_delta[3]
        bad = np.argwhere(delta != ref_s)
        return len(bad)
    return 0


@pytest.mark.parametrize("name", gu.COARSE_CASES)
def test_coarse_golden(name, dev, ops, cweights):
    sd, ncn, _, _ = cweights
    g = gu.load(name)
    p1, p2 = gu.coarse_inputs(g)
    ksize = int(g["ksize"])
    corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), ksize, ncn)
    flips = _check_coarse(corr.cpu().numpy(), None if delta is None else delta.cpu().numpy().astype(np.int64),
                          g["corr4d"], g.get("delta4d", None) if ksize > 1 else None, ksize)
    assert flips == 0, f"{flips} relocalisation argmax differ from the reference"
    m, s = ops.coarse_matches(corr, delta, ksize, 8, True)
    assert np.array_equal(m.cpu().numpy(), g["all_matches"])
    np.testing.assert_allclose(s.cpu().numpy(), g["all_scores"], rtol=2e-4)
    if ksize > 1:
        planes = ops.delta_unpack(delta, ksize)
        assert np.array_equal(torch.stack(planes).cpu().numpy().astype(np.int8), g["delta4d"])


@pytest.mark.parametrize("hw", [(64, 96), (80, 48), (112, 176), (240, 320)])
@pytest.mark.parametrize("ksize", [1, 2])
def test_coarse_vs_oracle(hw, ksize, dev, ops, cweights):
    sd, ncn, _, _ = cweights
    H, W = hw
    if ksize == 1 and H * W > 112 * 176:
        pytest.skip("k=1 volume too slow for the CPU oracle at this size")
    p1, p2 = synthetic.make_correlated_pyramids(100 + H + ksize, H, W)
    o_ncn, _, _ = orc.split_params(sd)
    rc, rd = orc.coarse_forward(p1[4], p2[4], ksize, o_ncn)
    corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), ksize, ncn)
    flips = _check_coarse(corr.cpu().numpy(), None if delta is None else delta.cpu().numpy().astype(np.int64),
                          rc.numpy(), None if rd is None else [d.numpy() for d in rd], ksize)
    assert flips <= 2, f"{flips} relocalisation flips"
    rm, rs = orc.cal_coarse_matches(rc, rd, ksize, 8)
    # match extraction on identical inputs must be bit-exact: feed the oracle's volume to the kernel
    k_delta = None
    if ksize > 1:
        k = ksize
        k_delta = (((rd[0] * k + rd[1]) * k + rd[2]) * k + rd[3]).to(torch.uint8).to(dev)
    m, s = ops.coarse_matches(rc.to(dev), k_delta, ksize, 8, True)
    assert torch.equal(m.cpu(), rm)
    assert torch.allclose(s.cpu(), rs, rtol=1e-5)


@pytest.mark.parametrize("dims", [(7, 9, 8, 13), (30, 40, 30, 40), (5, 3, 6, 70)])
def test_fused_consensus_vs_oracle(dims, dev, ops, cweights):
    """NeighConsensus.forward as its own operator (p2p_neigh_consensus_batch: both layers, both branches, one kernel on the
    fp16 matrix cores) against the oracle (ncn/model.py:145-155): a volume no multiple of the tile, the volume of a 480x640
    pair, a B row longer than one column tile; a second batch item with negative values far above 1 (rescaled inside)."""
    sd, ncn = cweights[0], cweights[1]
    g = torch.Generator().manual_seed(sum(dims))
    x = torch.rand(2, *dims, generator=g)
    x[1] = (x[1] - 0.5) * 300.0
    y = ops.neigh_consensus_batch(x.to(dev), ncn).cpu()
    o_ncn, _, _ = orc.split_params(sd)
    for b in range(2):
        ref = orc.neigh_consensus(x[b], o_ncn)
        assert (y[b] - ref).abs().max() <= 3e-6 * ref.abs().max(), (dims, b)
    assert torch.equal(ops.neigh_consensus_batch(x.to(dev), ncn).cpu(), y), "the fused consensus kernel is not deterministic"


def test_fused_consensus_tilings(dev, ops, cweights):
    """The fused consensus kernel marches along the first axis in chunks of `ta` slices over tiles of tb x tc cells; the
    tile is normally picked from the volume and batch size.  Forced (ta,tb,tc) shapes (0 = pick ta), including chunks that
    do not divide the axes, against the oracle -- and BIT-IDENTICAL to each other: every output cell sums the contributions
    of its 3 x 3 hidden strips in one fixed order whatever the tile."""
    sd = cweights[0]
    ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"], sd["ncn.conv.2.bias"], dev)
    x = torch.rand(1, 7, 11, 7, 11, generator=torch.Generator().manual_seed(5))
    o_ncn, _, _ = orc.split_params(sd)
    ref = orc.neigh_consensus(x[0], o_ncn)
    base = ops.neigh_consensus_batch(x.to(dev), ncn).cpu()
    assert (base[0] - ref).abs().max() <= 3e-6 * ref.abs().max()
    for tile in ((2, 3, 2), (4, 6, 6), (0, 2, 5), (3, 4, 3), (30, 6, 6)):
        ncn.set_tile(*tile)
        assert torch.equal(ops.neigh_consensus_batch(x.to(dev), ncn).cpu(), base), tile


@pytest.mark.parametrize("ksize", [1, 2])
def test_coarse_batch_equals_per_pair(ksize, dev, ops, cweights, monkeypatch):
    """p2p_coarse_forward_batch / p2p_coarse_matches_batch over B pairs == B single-pair calls, bit for bit, also
    when the workspace only holds some of the pairs at a time (the library then works through the batch in groups)."""
    sd, ncn, _, _ = cweights
    H, W, B = 96, 128, 5
    pairs = [synthetic.make_correlated_pyramids(500 + i, H, W) for i in range(B)]
    fa = torch.stack([p[0][4] for p in pairs]).to(dev)
    fb = torch.stack([p[1][4] for p in pairs]).to(dev)
    singles = [ops.coarse_forward(fa[i], fb[i], ksize, ncn) for i in range(B)]
    per_pair = ops._lib.p2p_coarse_workspace_bytes(fa.shape[1], fa.shape[2], fa.shape[3], fb.shape[2], fb.shape[3], ksize)
    for limit in (ops.COARSE_WORKSPACE_LIMIT, 2 * per_pair + 1):
        monkeypatch.setattr(ops, "COARSE_WORKSPACE_LIMIT", limit)
        corr, delta = ops.coarse_forward_batch(fa, fb, ksize, ncn)
        m, sc = ops.coarse_matches_batch(corr, delta, ksize, 8, True)
        for i in range(B):
            assert torch.equal(corr[i], singles[i][0]), (limit, i)
            if ksize > 1:
                assert torch.equal(delta[i], singles[i][1]), (limit, i)
            m1, s1 = ops.coarse_matches(singles[i][0], singles[i][1], ksize, 8, True)
            assert torch.equal(m[i], m1) and torch.equal(sc[i], s1), (limit, i)


def test_pair_results_do_not_depend_on_the_batch(dev, ops, cweights):
    """The reference treats batch items independently (networks/patch2pix.py:120-136, ncn/conv4d.py:47-71).  Here too: a
    480x640 pair's corr4d, relocalisation codes, coarse rows and -- through the default regressors -- its mid / fine matches
    are BIT-IDENTICAL whether the pair is launched alone, in a batch of 8 or in a batch of 16, and for every forced tile of
    the consensus kernel (estimate_matches and estimate_matches_stream therefore agree bit for bit)."""
    from patch2pix_amd.networks.utils import filter_coarse
    sd = cweights[0]
    ncn = ops.NcnWeights(sd["ncn.conv.0.weight"], sd["ncn.conv.0.bias"], sd["ncn.conv.2.weight"], sd["ncn.conv.2.bias"], dev)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid_w, fine_w = ops.RegressorWeights(sub("regress_mid."), dev), ops.RegressorWeights(sub("regress_fine."), dev)
    H, W, B = 480, 640, 16
    pairs = [synthetic.make_correlated_pyramids(3000 + i, H, W) for i in range(B)]
    fa = torch.stack([p[0][4] for p in pairs]).to(dev)
    fb = torch.stack([p[1][4] for p in pairs]).to(dev)

    def run(nb, tile=(0, 0, 0)):
        ncn.set_tile(*tile)
        corr, delta = ops.coarse_forward_batch(fa[:nb], fb[:nb], 2, ncn)
        m, sc = ops.coarse_matches_batch(corr, delta, 2, 8, True)
        return corr[0].clone(), delta[0].clone(), m[0].clone(), sc[0].clone()

    base = run(1)
    for nb, tile in ((8, (0, 0, 0)), (16, (0, 0, 0)), (1, (4, 5, 8)), (1, (30, 5, 8)), (16, (7, 3, 6)), (2, (15, 6, 4))):
        got = run(nb, tile)
        for k, (a, b) in enumerate(zip(got, base)):
            assert torch.equal(a, b), f"batch {nb}, tile {tile}: output {k} of pair 0 differs from the single-pair launch"
    ncn.set_tile(0, 0, 0)
    cm, _ = filter_coarse(base[2][None], base[3][None], 0.0, True)
    props = [cm[0]] + [torch.randint(8, 400, (300, 4), generator=torch.Generator().manual_seed(i)).to(dev) for i in range(1, 4)]
    pyr = lambda i, s: _gpu(pairs[i][s][:4], dev)
    alone = ops.regress(mid_w, fine_w, pyr(0, 0), pyr(0, 1), props[0])
    batch = ops.regress_batch(mid_w, fine_w, [pyr(i, 0) for i in range(4)], [pyr(i, 1) for i in range(4)], props)
    for k in ("matches1", "probs1", "matches2", "probs2"):
        assert torch.equal(alone[k], batch[0][k]), k


# ------------------------------------------------------------------------------------------ fine
def _compare_matches(got, ref, tol=COORD_TOL):
    diff = (got - ref).abs().max().item() if got.numel() else 0.0
    assert diff <= tol, f"max coordinate error {diff} px > {tol}"


@pytest.mark.parametrize("name", gu.FINE_CASES)
def test_fine_golden(name, dev, ops, weights):
    _, _, mid_w, fine_w = weights
    g = gu.load(name)
    p1, p2 = gu.fine_inputs(g)
    g1, g2 = _gpu(p1[:4], dev), _gpu(p2[:4], dev)
    for tag, w in (("int_mid", mid_w), ("float_fine", fine_w)):
        props = torch.from_numpy(g[tag + "_in"]).to(dev)
        out = ops.regress(w, None, g1, g2, props)
        _compare_matches(out["matches1"].cpu(), torch.from_numpy(g[tag + "_matches"]))
        assert (out["probs1"].cpu() - torch.from_numpy(g[tag + "_probs"])).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("hw,n", [((48, 64), 33), ((96, 128), 257), ((480, 640), 400)])
def test_fine_vs_oracle(hw, n, dev, ops, weights):
    sd, _, mid_w, fine_w = weights
    H, W = hw
    p1 = synthetic.make_pyramid(7, H, W)
    p2 = synthetic.make_pyramid(8, H, W)
    g = torch.Generator().manual_seed(9)
    props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                         torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
    props[0] = torch.tensor([0, 0, W, H])
    props[1] = torch.tensor([W, H, 0, 0])
    _, mid_p, fine_p = orc.split_params(sd)
    ref_mid, ref_midp, ref_raw = orc.fine_level(p1[:4], p2[:4], props, mid_p)
    g1, g2 = _gpu(p1[:4], dev), _gpu(p2[:4], dev)
    out = ops.regress(mid_w, fine_w, g1, g2, props.to(dev), want_mid=True, want_raw=True)
    assert (out["raw1"].cpu() - ref_raw).abs().max() < 5e-5
    _compare_matches(out["matches1"].cpu(), ref_mid)
    assert (out["probs1"].cpu() - ref_midp).abs().max() <= SCORE_TOL
    # second level: feed the *kernel's* mid matches to the oracle so that a 1e-6 px wobble across an
    # integer boundary (trunc) cannot masquerade as an error of the fine regressor
    ref_fine, ref_finep, _ = orc.fine_level(p1[:4], p2[:4], out["matches1"].cpu(), fine_p)
    _compare_matches(out["matches2"].cpu(), ref_fine)
    assert (out["probs2"].cpu() - ref_finep).abs().max() <= SCORE_TOL
    # single-level call on float proposals == second half of the chained launch
    single = ops.regress(fine_w, None, g1, g2, out["matches1"])
    assert torch.equal(single["matches1"], out["matches2"])


def test_fine_empty_and_single(dev, ops, weights):
    _, _, mid_w, fine_w = weights
    p1 = _gpu(synthetic.make_pyramid(1, 48, 64)[:4], dev)
    p2 = _gpu(synthetic.make_pyramid(2, 48, 64)[:4], dev)
    out = ops.regress(mid_w, fine_w, p1, p2, torch.zeros((0, 4), dtype=torch.int64, device=dev))
    assert out["matches2"].shape == (0, 4) and out["probs2"].shape == (0,)
    out = ops.regress(mid_w, fine_w, p1, p2, torch.tensor([[20, 20, 30, 30]], device=dev))
    assert out["matches2"].shape == (1, 4) and torch.isfinite(out["matches2"]).all()


def test_regress_workspace_is_sized_per_mode(dev, ops, weights):
    """p2p_regress_workspace_bytes_mode: the direct fp16x2 mode parks 4 KB per proposal, the Winograd mode also the
    transformed conv2 input of a chunk; a buffer that is too small for the handle's mode is P2P_ENOMEM (-4, like the
    library's other workspace checks), never a fault."""
    import ctypes
    from patch2pix_amd import _lib
    _, _, mid_w, fine_w = weights
    n = 40
    mode = _lib.REGRESS_MODES[mid_w.mode]
    need = _lib.p2p_regress_workspace_bytes_mode(n, mode)
    assert need <= _lib.p2p_regress_workspace_bytes(n)
    if mid_w.mode == "f32":
        assert need == 0
        return
    assert _lib.p2p_regress_workspace_bytes_mode(n, _lib.REGRESS_MODES["fp16x2"]) < 1 << 20 < _lib.p2p_regress_workspace_bytes_mode(n, _lib.REGRESS_MODES["fp16x2w"])
    p1 = _gpu(synthetic.make_pyramid(1, 48, 64)[:4], dev)
    p2 = _gpu(synthetic.make_pyramid(2, 48, 64)[:4], dev)
    pa, ka = ops._pyramid(p1)
    pb, kb = ops._pyramid(p2)
    props = torch.randint(0, 48, (n, 4), device=dev)
    m1, q1 = torch.empty((n, 4), device=dev), torch.empty((n,), device=dev)
    m2, q2 = torch.empty((n, 4), device=dev), torch.empty((n,), device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(nbytes):
        ws = torch.empty(max(nbytes, 128), dtype=torch.uint8, device=dev)
        return _lib.p2p_regress(mid_w.handle, fine_w.handle, ctypes.byref(pa), ctypes.byref(pb), props.data_ptr(), 0, n,
                                m1.data_ptr(), q1.data_ptr(), None, m2.data_ptr(), q2.data_ptr(), None, ws.data_ptr(), nbytes, stream)
    assert call(need) == 0
    torch.cuda.synchronize()
    want = m2.clone()
    assert call(need - 256) == -4 and b"workspace" in _lib.p2p_last_error()
    assert call(_lib.p2p_regress_workspace_bytes(n)) == 0
    torch.cuda.synchronize()
    assert torch.equal(m2, want)


# ------------------------------------------------------------------------------------------ whole path
def _model(dev, contrast=None):
    from patch2pix_amd.utils.eval import model_helper
    return model_helper.load_model(synthetic.make_checkpoint(0, contrast=contrast), lprint=lambda *a: None)


def _near_integer_rows(mid, eps=2e-4):
    """Rows with a coordinate within eps of an integer without being one: trunc() (networks/utils.py:19) of a value
    that differs in the last bits may land on the other side.  EXACT integers are stable and not counted: they are the
    clamp bounds 0 / W / H, or base - 8 where relu() zeroed the raw output (patch2pix.py:141-145: the coordinate is
    base - 8 + 16 tanh(relu(o)) >= base - 8, so a differing last bit of o can only move it upwards, never across)."""
    frac = mid - mid.floor()
    near = ((frac > 0) & (frac < eps)) | (frac > 1 - eps)
    return near.any(dim=1)


@pytest.mark.parametrize("name", gu.PAIR_CASES)
def test_predict_fine_golden(name, dev):
    net = _model(dev)
    g = gu.load(name)
    p1, p2 = gu.pair_inputs(g)
    f1 = [t[None].to(dev) for t in p1]
    f2 = [t[None].to(dev) for t in p2]
    fine, fine_scores, mid, mid_scores, coarse = net.predict_fine_from_feats(f1, f2, return_all=True)
    assert np.array_equal(coarse[0].cpu().numpy(), g["coarse"])          # indices bit-exact
    _compare_matches(mid[0].cpu(), torch.from_numpy(g["mid"]))
    unstable = _near_integer_rows(torch.from_numpy(g["mid"]))
    ok = ~unstable
    _compare_matches(fine[0].cpu()[ok], torch.from_numpy(g["fine"])[ok])
    assert (fine_scores[0].cpu()[ok] - torch.from_numpy(g["fine_scores"])[ok]).abs().max() <= SCORE_TOL
    assert unstable.sum() <= 2


@pytest.mark.parametrize("name", ["estimate_matches_240x320", "estimate_matches_imsize256"])
def test_estimate_matches_golden(name, dev, tmp_path):
    """The drop-in entry point on image files.  The backbone here runs on the GPU (HIP convolutions, csrc/backbone.hip) while the
    golden run used the CPU, so a small fraction of coarse argmaxes may legitimately differ;
    rows are compared by coarse match."""
    from PIL import Image
    from patch2pix_amd.utils.eval import model_helper
    g = gu.load(name)
    im1, im2 = synthetic.make_image_pair(int(g["seed"]), int(g["H"]), int(g["W"]))
    assert abs(float(im1.astype(np.float64).sum() + im2.astype(np.float64).sum()) - float(g["input_checksum"])) < 1
    Image.fromarray(im1).save(tmp_path / "1.png")
    Image.fromarray(im2).save(tmp_path / "2.png")
    net = _model(dev)
    imsize = None if int(g["imsize"]) < 0 else int(g["imsize"])
    m, s, c = model_helper.estimate_matches(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2,
                                            io_thres=0.25, eval_type="fine", imsize=imsize)
    assert m.dtype == np.float64 and s.dtype == np.float32 and c.dtype == np.float64
    assert m.shape[1] == 4 and c.shape == m.shape and s.shape == (m.shape[0],)
    ref = {tuple(np.round(r, 6)): i for i, r in enumerate(g["fine_coarse"])}
    hits = [(i, ref[tuple(np.round(r, 6))]) for i, r in enumerate(c) if tuple(np.round(r, 6)) in ref]
    assert len(hits) >= 0.8 * len(g["fine_coarse"]), f"only {len(hits)} of {len(g['fine_coarse'])} coarse matches agree"
    gi = np.array([h[0] for h in hits]); ri = np.array([h[1] for h in hits])
    err = np.abs(m[gi] - g["fine_matches"][ri]).max(axis=1)
    # backbone numerics differ (GPU vs CPU summation order): the regressed coordinates agree to well below a pixel
    assert np.median(err) < 0.05 and (err < 0.5).mean() > 0.9
    mc, sc, cc = model_helper.estimate_matches(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2,
                                               ncn_thres=0.0, eval_type="coarse", imsize=imsize)
    assert mc.shape[1] == 4 and np.array_equal(mc, cc)
    # The strict form: the same entry point with the backbone evaluated on the CPU like in the golden run (identical
    # pyramids on both sides; everything after the backbone stays on the HIP path).  Now every row must be there, in
    # the reference's order, within 1e-3 px / 1e-5.
    import copy
    cpu_extract = copy.deepcopy(net.extract).to("cpu")
    gpu_pyramid = net.extract.pyramid
    net.extract.pyramid = lambda im: [f.to(dev) for f in cpu_extract.pyramid(im.cpu())]
    try:
        m, s, c = model_helper.estimate_matches(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2,
                                                io_thres=0.25, eval_type="fine", imsize=imsize)
        mc, sc, cc = model_helper.estimate_matches(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2,
                                                   ncn_thres=0.0, eval_type="coarse", imsize=imsize)
    finally:
        net.extract.pyramid = gpu_pyramid
    assert c.shape == g["fine_coarse"].shape and np.array_equal(c, g["fine_coarse"]), "coarse rows of the fine path differ"
    scale = max(1.0, max(int(g["H"]), int(g["W"])) / float(imsize) * 1.1) if imsize else 1.0     # original-pixel factor
    assert np.abs(m - g["fine_matches"]).max() <= COORD_TOL * scale and np.abs(s - g["fine_scores"]).max() <= SCORE_TOL
    # coarse scores = 1 / sum(exp(x - max)) over 1200 cells of the consensus output: same bar as the other coarse tests
    assert np.array_equal(mc, g["coarse_matches"]) and np.allclose(sc, g["coarse_scores"], rtol=1e-4, atol=0)


def test_batched_launch_equals_per_pair(dev, ops, weights):
    """p2p_regress_batch over pairs of *different* sizes == one launch per pair (bit-exact),
    including an empty item and more items than one launch holds."""
    _, _, mid_w, fine_w = weights
    sizes = [(48, 64), (96, 128), (64, 48), (48, 64), (80, 80), (48, 64), (96, 64), (64, 64), (48, 96), (56, 72),
             (48, 64), (64, 96), (72, 56), (48, 48), (88, 64), (64, 80), (48, 72), (56, 56)]   # 18 > 16 items per launch
    g = torch.Generator().manual_seed(3)
    pyr1, pyr2, props = [], [], []
    for i, (H, W) in enumerate(sizes):
        pyr1.append(_gpu(synthetic.make_pyramid(200 + i, H, W)[:4], dev))
        pyr2.append(_gpu(synthetic.make_pyramid(300 + i, H, W)[:4], dev))
        n = 0 if i == 3 else 5 + (3 * i) % 17
        props.append(torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                                  torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)],
                                 1).to(dev))
    outs = ops.regress_batch(mid_w, fine_w, pyr1, pyr2, props)
    for i in range(len(sizes)):
        single = ops.regress(mid_w, fine_w, pyr1[i], pyr2[i], props[i])
        for k in ("matches1", "probs1", "matches2", "probs2"):
            assert torch.equal(outs[i][k], single[k]), (i, k)


def test_ragged_launches_are_bit_identical_per_proposal(dev, ops, weights):
    """A proposal's result does not depend on what else is in the launch: counts that are no multiple of 8 (rows of a Winograd
    row block that no proposal owns are zeros), of 16 (a partly filled FC batch), a single proposal -- every prefix of one
    proposal list gives the first rows of the full launch, bit for bit."""
    _, _, mid_w, fine_w = weights
    H, W = 64, 96
    g = torch.Generator().manual_seed(11)
    pyr1, pyr2 = _gpu(synthetic.make_pyramid(410, H, W)[:4], dev), _gpu(synthetic.make_pyramid(411, H, W)[:4], dev)
    n = 41
    props = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                         torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1).to(dev)
    full = ops.regress(mid_w, fine_w, pyr1, pyr2, props)
    for k in (1, 7, 8, 9, 13, 17, 40):
        part = ops.regress(mid_w, fine_w, pyr1, pyr2, props[:k].contiguous())
        for key in ("matches1", "probs1", "matches2", "probs2"):
            assert torch.equal(part[key], full[key][:k]), (k, key)


def test_many_items_and_more_proposals_than_a_chunk(dev, ops, weights):
    """More items than one launch holds (18 > 16) AND more proposals than one chunk of the Winograd path (2 x 1500 + 14 x 40 >
    3328 in the first launch): the launches of one call share the scratch with different offsets; results == one call per item."""
    _, _, mid_w, fine_w = weights
    if mid_w.mode == "f32":
        pytest.skip("3640 proposals through the exact-f32 kernel: covered by the smaller batch tests")
    g = torch.Generator().manual_seed(12)
    sizes = [(48, 64)] * 18
    counts = [1500, 1500] + [40] * 16
    pyr1 = [_gpu(synthetic.make_pyramid(500 + i, H, W)[:4], dev) for i, (H, W) in enumerate(sizes)]
    pyr2 = [_gpu(synthetic.make_pyramid(600 + i, H, W)[:4], dev) for i, (H, W) in enumerate(sizes)]
    props = [torch.stack([torch.randint(0, W + 1, (c,), generator=g), torch.randint(0, H + 1, (c,), generator=g),
                          torch.randint(0, W + 1, (c,), generator=g), torch.randint(0, H + 1, (c,), generator=g)], 1).to(dev)
             for (H, W), c in zip(sizes, counts)]
    outs = ops.regress_batch(mid_w, fine_w, pyr1, pyr2, props)
    for i in (0, 1, 2, 15, 16, 17):
        single = ops.regress(mid_w, fine_w, pyr1[i], pyr2[i], props[i])
        for key in ("matches1", "probs1", "matches2", "probs2"):
            assert torch.equal(outs[i][key], single[key]), (i, key)


# ------------------------------------------------------------------------------------------ full sizes
def _adjudicate_delta_flips(fa, fb, got, ref_code, ksize=2):
    """Every pooled cell whose relocalisation argmax differs from the fp32 oracle's is re-evaluated in fp64: the
    two candidates' correlations must be closer than the fp32 error bound of the two dot products (tests/adjudicate.py:
    eps_256 * sum |a_i b_i| per candidate), otherwise the difference is an error.  Returns [(cell, got, ref, gap, bound)]."""
    from adjudicate import _eps, U
    k = ksize
    na = fa.double() / (fa.double().pow(2).sum(0, keepdim=True) + 1e-6).sqrt()          # modules.py:6 in fp64
    nb = fb.double() / (fb.double().pow(2).sum(0, keepdim=True) + 1e-6).sqrt()
    eps_c = _eps(fa.shape[0], 1.0) + 2 * (0.5 * _eps(fa.shape[0], 1.0) + 3 * U)
    out = []
    for a, b, c, d in np.argwhere(got != ref_code):
        def corr_of(code):
            di, dj, dk, dl = code // (k * k * k), (code // (k * k)) % k, (code // k) % k, code % k
            va, vb = na[:, k * a + di, k * b + dj], nb[:, k * c + dk, k * d + dl]
            return float((va * vb).sum()), eps_c * float((va * vb).abs().sum())
        (g, eg), (r, er) = corr_of(int(got[a, b, c, d])), corr_of(int(ref_code[a, b, c, d]))
        out.append(((int(a), int(b), int(c), int(d)), int(got[a, b, c, d]), int(ref_code[a, b, c, d]), abs(g - r), eg + er))
    return out


def test_coarse_full_size_vs_oracle(dev, ops, cweights):
    """BASELINE configuration (480x640, ksize 2): the whole coarse stage against the CPU oracle.  All 2400 coarse
    match rows must be EQUAL; the few relocalisation argmaxes (of 1.44 M) that differ from the fp32 oracle must be
    near-ties in an fp64 evaluation (printed), and none of them may be consumed by a match row."""
    sd, ncn, _, _ = cweights
    p1, p2 = synthetic.make_correlated_pyramids(77, 480, 640)
    o_ncn, _, _ = orc.split_params(sd)
    rc, rd = orc.coarse_forward(p1[4], p2[4], 2, o_ncn)
    corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), 2, ncn)
    got = delta.cpu().numpy().astype(np.int64)
    np.testing.assert_allclose(corr.cpu().numpy(), rc.numpy(), rtol=2e-4, atol=1e-7)
    ref_code = (((rd[0] * 2 + rd[1]) * 2 + rd[2]) * 2 + rd[3]).numpy()
    flips = _adjudicate_delta_flips(p1[4], p2[4], got, ref_code)
    print(f"\n{len(flips)} of {got.size} relocalisation argmaxes differ from the fp32 oracle:")
    for cell, g, r, gap, bound in flips:
        print(f"  cell {cell}: kernel {g}, oracle {r}, fp64 gap of the two candidates {gap:.3e} (fp32 error bound {bound:.1e})")
        assert gap <= 0.25 * bound, f"cell {cell}: argmax differs but the candidates are {gap:.3e} apart in fp64 (bound {bound:.1e}: not a near-tie)"
    assert len(flips) <= 8
    rm, rs = orc.cal_coarse_matches(rc, rd, 2, 8)
    m, s = ops.coarse_matches(corr, delta, 2, 8, True)
    assert torch.equal(m.cpu(), rm), f"{int((m.cpu() != rm).any(dim=1).sum())} of {rm.shape[0]} coarse match rows differ"
    assert torch.allclose(s.cpu(), rs, rtol=2e-4)


def test_full_size_reference_golden(dev, ops, weights):
    """The BASELINE configuration against the UNMODIFIED REFERENCE (tests/golden/full_480x640.npz, made by
    oracle/make_golden.py): all 2400 coarse rows and the mutual set equal; the ptmax=400 proposals the reference
    sampled go through both regressors within 1e-3 px / 1e-5."""
    from patch2pix_amd.networks.utils import filter_coarse
    sd, ncn, mid_w, fine_w = weights
    g = gu.load("full_480x640")
    p1, p2 = gu.pair_inputs(g)
    corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), 2, ncn)
    m, sc = ops.coarse_matches(corr, delta, 2, 8, True)
    assert np.array_equal(m.cpu().numpy(), g["all_matches"].astype(np.int64))
    np.testing.assert_allclose(sc.cpu().numpy(), g["all_scores"], rtol=2e-4)
    flat = corr.reshape(-1)[::int(g["corr_sample_stride"])].cpu().numpy()
    np.testing.assert_allclose(flat, g["corr_sample"], rtol=2e-4, atol=1e-7)
    hist = np.bincount(delta.cpu().numpy().reshape(-1), minlength=16)
    assert np.abs(hist - g["delta_hist"]).sum() <= 16, "relocalisation codes differ from the reference beyond near-ties"
    fm, _ = filter_coarse(m[None], sc[None], 0.0, True)
    assert np.array_equal(fm[0].cpu().numpy(), g["mutual_matches"].astype(np.int64))
    np.random.seed(0)                                   # the reference sampled with the global numpy RNG seeded 0
    cm, _ = filter_coarse(m[None], sc[None], 0.0, True, ptmax=int(g["ptmax"]))
    assert np.array_equal(cm[0].cpu().numpy(), g["proposals"].astype(np.int64))
    out = ops.regress(mid_w, fine_w, _gpu(p1[:4], dev), _gpu(p2[:4], dev), cm[0])
    _compare_matches(out["matches1"].cpu(), torch.from_numpy(g["mid"]))
    assert (out["probs1"].cpu() - torch.from_numpy(g["mid_scores"])).abs().max() <= SCORE_TOL
    ok = ~_near_integer_rows(torch.from_numpy(g["mid"]))
    _compare_matches(out["matches2"].cpu()[ok], torch.from_numpy(g["fine"])[ok])
    assert (out["probs2"].cpu()[ok] - torch.from_numpy(g["fine_scores"])[ok]).abs().max() <= SCORE_TOL
    assert (~ok).sum() <= 4


@pytest.mark.parametrize("mode", MODES)
def test_benched_batch_path_vs_oracle(mode, dev):
    """Exactly what bench.py times: B = 16 pairs of 480x640 through Patch2Pix.coarse_async / fine_from_ticket with
    ptmax = 400, every pair against the CPU oracle: all 2400 coarse rows equal, the sampled proposals equal, mid and
    fine coordinates within 1e-3 px, scores within 1e-5 (fine level: oracle fed with the kernel's own mid matches)."""
    net = _model(dev)
    for w in net._weights()[1:]:
        w.set_mode(mode)
    B, H, W, ptmax = 16, 480, 640, 400
    ckpt_sd = gu.state_dict(0)
    o_ncn, mid_p, fine_p = orc.split_params(ckpt_sd)
    pairs = [synthetic.make_correlated_pyramids(2000 + i, H, W) for i in range(B)]
    f1 = [torch.stack([p[0][j] for p in pairs]).to(dev) for j in range(5)]
    f2 = [torch.stack([p[1][j] for p in pairs]).to(dev) for j in range(5)]
    np.random.seed(99)
    with torch.no_grad():
        ticket = net.coarse_async(f1, f2, ksize=2)
        fine, fine_s, mid, mid_s, coarse = net.fine_from_ticket(ticket, 0.0, True, return_all=True, ptmax=ptmax)
    torch.cuda.synchronize()
    from adjudicate import ErrorModel, differing_rows_are_near_ties
    rng = np.random.RandomState(99)                     # the product draws from the global numpy RNG, pair after pair
    worst = dict(mid=0.0, fine=0.0, score=0.0)
    near_ties = 0
    with torch.no_grad():
        for b in range(B if mode == DEFAULT_MODE else 3):   # all 16 pairs in the default mode, 3 in the others (CPU time)
            rc, rd = orc.coarse_forward(pairs[b][0][4], pairs[b][1][4], 2, o_ncn)
            rm, rs = orc.cal_coarse_matches(rc, rd, 2, 8)
            got_rows = ticket["matches"][b].cpu()
            cm, _ = orc.filter_coarse(rm, rs, 0.0, True, ptmax=ptmax, rng=rng)       # keeps the RNG streams in step
            if torch.equal(got_rows, rm):
                assert torch.equal(coarse[b].cpu(), cm), f"pair {b}: sampled proposals differ"
            else:
                # a differing row must be undecidable in fp32: the two candidates closer in fp64 than the error bound of an
                # fp32 evaluation (tests/adjudicate.py); the proposals then differ legitimately and the fine stage is
                # compared on the kernel's own proposals
                em = ErrorModel(pairs[b][0][4], pairs[b][1][4], ckpt_sd, 2)
                em.check(rc, "oracle fp32 volume")
                nd, gap = differing_rows_are_near_ties(got_rows, rm, em)
                print(f"\npair {b}: {nd} of {rm.shape[0]} coarse rows differ from the fp32 oracle, fp64 gap {gap:.3f} of the fp32 error bound")
                near_ties += nd
                cm = coarse[b].cpu()
            ref_mid, ref_mp, _ = orc.fine_level(pairs[b][0][:4], pairs[b][1][:4], cm, mid_p)
            ref_fine, ref_fp, _ = orc.fine_level(pairs[b][0][:4], pairs[b][1][:4], mid[b].cpu(), fine_p)
            worst["mid"] = max(worst["mid"], (mid[b].cpu() - ref_mid).abs().max().item())
            worst["fine"] = max(worst["fine"], (fine[b].cpu() - ref_fine).abs().max().item())
            # the plain chain (oracle fine level on the ORACLE's mid matches): equal within the tolerance wherever the
            # truncated mid coordinates agree; a mid coordinate within rounding error of an integer moves its fine patch
            # by one pixel (networks/utils.py:19) -- counted, and rare
            chain, _, _ = orc.fine_level(pairs[b][0][:4], pairs[b][1][:4], ref_mid, fine_p)
            moved = (mid[b].cpu().long() != ref_mid.long()).any(dim=1)
            worst["moved"] = worst.get("moved", 0) + int(moved.sum())
            worst["n"] = worst.get("n", 0) + int(moved.numel())
            if bool((~moved).any()):
                worst["chain"] = max(worst.get("chain", 0.0), (fine[b].cpu() - chain)[~moved].abs().max().item())
            worst["score"] = max(worst["score"], (mid_s[b].cpu() - ref_mp).abs().max().item(),
                                 (fine_s[b].cpu() - ref_fp).abs().max().item())
    assert near_ties <= 4, f"{near_ties} near-tie rows in {B} pairs"
    print(f"\n{mode}: max |d mid| {worst['mid']:.2e} px, |d fine| {worst['fine']:.2e} px, |d score| {worst['score']:.2e}; plain chain "
          f"{worst.get('chain', 0.0):.2e} px on {worst['n'] - worst['moved']} matches, {worst['moved']} fine patches moved by trunc()")
    assert worst["mid"] <= COORD_TOL and worst["fine"] <= COORD_TOL and worst["score"] <= SCORE_TOL
    assert worst.get("chain", 0.0) <= COORD_TOL and worst["moved"] <= max(2, worst["n"] // 1000)


@pytest.mark.parametrize("hw", [(480, 640), (960, 1280)])
def test_coarse_symmetry_property(hw, dev, ops, cweights):
    """Size-independent property at the BASELINE sizes (config E = 960x1280: 1.5 GB full-resolution volume,
    3 GB hidden layer): the consensus stage is symmetric by construction, so swapping the two images must
    transpose the volume -- corr(A,B)[a,b,c,d] == corr(B,A)[c,d,a,b] -- and swap the two match directions."""
    _, ncn, _, _ = cweights
    H, W = hw
    p1, p2 = synthetic.make_correlated_pyramids(5, H, W)
    fa, fb = p1[4].to(dev), p2[4].to(dev)
    c_ab, d_ab = ops.coarse_forward(fa, fb, 2, ncn)
    c_ab, d_ab = c_ab.clone(), d_ab.clone()
    c_ba, d_ba = ops.coarse_forward(fb, fa, 2, ncn)
    ref = c_ab.permute(2, 3, 0, 1)
    assert torch.isfinite(c_ab).all()
    rel = ((c_ba - ref).abs() / (ref.abs() + 1e-6)).max().item()
    assert rel < 1e-3, rel
    # relocalisation codes: (di,dj,dk,dl) of (A,B) == (dk,dl,di,dj) of (B,A) except on near-ties
    s_ab = d_ab.permute(2, 3, 0, 1).long()
    swapped = ((s_ab & 3) << 2) | (s_ab >> 2)
    assert (swapped != d_ba.long()).float().mean().item() < 1e-4
    m_ab, s1 = ops.coarse_matches(c_ab, d_ab, 2, 8, True)
    assert (s1 > 0).all() and (s1 <= 1.0 + 1e-6).all()
    assert (m_ab >= 4).all() and (m_ab[:, 0] < W).all() and (m_ab[:, 1] < H).all()


def test_regress_is_deterministic_and_anchor_path(dev, ops, weights):
    """Same launch twice -> bit-identical; config E style proposals (ptmax 800 x panc 8 = 6400) run through
    shift_to_anchors semantics and agree with the oracle on a sample."""
    sd, _, mid_w, fine_w = weights
    H, W = 192, 256
    p1 = synthetic.make_pyramid(11, H, W)
    p2 = synthetic.make_pyramid(12, H, W)
    g = torch.Generator().manual_seed(4)
    base = torch.stack([torch.randint(8, W - 8, (800,), generator=g), torch.randint(8, H - 8, (800,), generator=g),
                        torch.randint(8, W - 8, (800,), generator=g), torch.randint(8, H - 8, (800,), generator=g)], 1)
    props = orc.shift_to_anchors(base, 8, 8)
    assert props.shape == (6400, 4)
    g1, g2 = _gpu(p1[:4], dev), _gpu(p2[:4], dev)
    a = ops.regress(mid_w, fine_w, g1, g2, props.to(dev))
    b = ops.regress(mid_w, fine_w, g1, g2, props.to(dev))
    for k in ("matches1", "matches2", "probs1", "probs2"):
        assert torch.equal(a[k], b[k])
    _, mid_p, _ = orc.split_params(sd)
    idx = torch.arange(0, 6400, 97)
    ref_mid, ref_p, _ = orc.fine_level(p1[:4], p2[:4], props[idx], mid_p)
    _compare_matches(a["matches1"].cpu()[idx], ref_mid)
    assert (a["probs1"].cpu()[idx] - ref_p).abs().max() <= SCORE_TOL


def test_nc_only_model_and_predict_coarse(dev, tmp_path):
    """load_model(method='nc') (bare NCNet state_dict, model_helper.py:53-57) + predict_coarse /
    estimate_matches(eval_type='coarse') against the oracle's coarse stage on CPU-computed pyramids."""
    from patch2pix_amd.utils.eval import model_helper
    sd = gu.state_dict(0)
    nc_sd = {k: v for k, v in sd.items() if k.startswith(("extract.", "ncn."))}
    net = model_helper.load_model({"state_dict": nc_sd}, method="nc", lprint=lambda *a: None)
    assert net.regress_mid is None and net.upsample == 8
    g = gu.load("coarse_128x160_k2")
    p1, p2 = gu.coarse_inputs(g)
    corr4d, delta4d = net.forward_coarse_match(p1[4][None].to(dev), p2[4][None].to(dev), ksize=2)
    assert corr4d.shape == (1, 1, 8, 10, 8, 10) and len(delta4d) == 4
    assert delta4d[0].dtype == torch.int64 and delta4d[0].shape == corr4d.shape
    m, s = net.cal_coarse_matches(corr4d, delta4d, ksize=2, upsample=net.upsample, center=True)
    assert np.array_equal(m[0].cpu().numpy(), g["all_matches"])
    # the reference-format delta4d (four int64 planes) is accepted as well as the packed form
    planes = tuple(t.clone() for t in delta4d)
    m2, _ = net.cal_coarse_matches(corr4d, planes, ksize=2, upsample=net.upsample, center=True)
    assert torch.equal(m, m2)
    from patch2pix_amd.networks.utils import filter_coarse
    fm, fs = filter_coarse(m, s, 0.0, True)
    assert np.array_equal(fm[0].cpu().numpy(), g["mutual_matches"])
    fu, _ = filter_coarse(m, s, 0.0, False)
    assert np.array_equal(fu[0].cpu().numpy(), g["unique_matches"])
    # sort=True orders by descending score
    ms, ss = net.cal_coarse_matches(corr4d, delta4d, ksize=2, upsample=net.upsample, sort=True)
    assert (ss[0][:-1] >= ss[0][1:]).all()


def test_batch_of_pairs_through_model(dev):
    """B = 3 pairs through predict_fine_from_feats == the three pairs one at a time (lists per batch item)."""
    net = _model(dev)
    pairs = [synthetic.make_correlated_pyramids(60 + i, 128, 160) for i in range(3)]
    f1 = [torch.stack([p[0][j] for p in pairs]).to(dev) for j in range(5)]
    f2 = [torch.stack([p[1][j] for p in pairs]).to(dev) for j in range(5)]
    fine, scores, coarse = net.predict_fine_from_feats(f1, f2)
    assert len(fine) == len(scores) == len(coarse) == 3
    for b in range(3):
        one = net.predict_fine_from_feats([t[b:b + 1] for t in f1], [t[b:b + 1] for t in f2])
        assert torch.equal(coarse[b], one[2][0])
        assert torch.equal(fine[b], one[0][0]) and torch.equal(scores[b], one[1][0])


def test_refine_matches_and_empty_input(dev):
    net = _model(dev)
    im1, im2 = synthetic.make_image_pair(9, 96, 128)
    t1 = torch.from_numpy(im1).permute(2, 0, 1).float().div(255)[None].to(dev)
    t2 = torch.from_numpy(im2).permute(2, 0, 1).float().div(255)[None].to(dev)
    r, s, c = net.refine_matches(t1, t2, np.zeros((0, 4)), 0.25)
    assert r.shape == (0, 4) and s.shape == (0,) and c.shape == (0, 4)
    coarse = np.array([[20, 20, 28, 36], [60, 44, 68, 60], [100, 80, 92, 72]], dtype=np.int64)
    with torch.no_grad():
        r, s, c = net.refine_matches(t1, t2, coarse, 0.0)
    assert r.shape == (3, 4) and s.shape == (3,) and np.array_equal(c, coarse)
    assert np.isfinite(r).all() and (np.abs(r - coarse) <= 16.0 + 1e-3).all()


def test_images_of_different_sizes(dev, ops, weights):
    """Ragged pair: image 1 is 64x96, image 2 is 96x64 (the volume is 4x6 x 6x4 cells, not square)."""
    sd, ncn, mid_w, fine_w = weights
    p1 = synthetic.make_pyramid(71, 64, 96)
    p2 = synthetic.make_pyramid(72, 96, 64)
    o_ncn, mid_p, _ = orc.split_params(sd)
    for ksize in (1, 2):
        rc, rd = orc.coarse_forward(p1[4], p2[4], ksize, o_ncn)
        corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), ksize, ncn)
        assert tuple(corr.shape) == tuple(rc.shape)
        flips = _check_coarse(corr.cpu().numpy(), None if delta is None else delta.cpu().numpy().astype(np.int64),
                              rc.numpy(), None if rd is None else [d.numpy() for d in rd], ksize)
        assert flips == 0
        rm, rs = orc.cal_coarse_matches(rc, rd, ksize, 8)
        m, s = ops.coarse_matches(corr, delta, ksize, 8, True)
        assert torch.equal(m.cpu(), rm) and torch.allclose(s.cpu(), rs, rtol=2e-4)
    g = torch.Generator().manual_seed(1)
    n = 21
    props = torch.stack([torch.randint(0, 97, (n,), generator=g), torch.randint(0, 65, (n,), generator=g),
                         torch.randint(0, 65, (n,), generator=g), torch.randint(0, 97, (n,), generator=g)], 1)
    ref_mid, ref_p, _ = orc.fine_level(p1[:4], p2[:4], props, mid_p)
    out = ops.regress(mid_w, None, _gpu(p1[:4], dev), _gpu(p2[:4], dev), props.to(dev))
    _compare_matches(out["matches1"].cpu(), ref_mid)
    assert (out["probs1"].cpu() - ref_p).abs().max() <= SCORE_TOL
    # clamping uses each image's own size: x1 <= 96, y1 <= 64, x2 <= 64, y2 <= 96
    hi = torch.tensor([96.0, 64.0, 64.0, 96.0])
    assert (out["matches1"].cpu() <= hi).all() and (out["matches1"].cpu() >= 0).all()


def test_stream_equals_per_pair_calls(dev, tmp_path):
    """estimate_matches_stream (threaded loading, batched backbone, shared fine launch) returns what
    estimate_matches returns pair by pair.  The batched backbone may differ from the un-batched one in the last
    bits (MIOpen algorithm choice), so rows are matched by their coarse match and compared with a tolerance."""
    from PIL import Image
    from patch2pix_amd.utils.eval import model_helper
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    net = _model(dev)
    pairs = []
    for i, (h, w) in enumerate([(240, 320), (240, 320), (240, 320), (192, 256), (192, 256), (240, 320)]):
        a, b = synthetic.make_image_pair(300 + i, h, w)
        pa, pb = tmp_path / f"{i}a.jpg", tmp_path / f"{i}b.jpg"
        Image.fromarray(a).save(pa, quality=95)
        Image.fromarray(b).save(pb, quality=95)
        pairs.append((str(pa), str(pb)))
    streamed = list(estimate_matches_stream(net, pairs, ksize=2, io_thres=0.25, batch=3, workers=3))
    assert len(streamed) == len(pairs)
    for (m, s, c), (pa, pb) in zip(streamed, pairs):
        rm, rs, rc = model_helper.estimate_matches(net, pa, pb, ksize=2, io_thres=0.25)
        assert m.dtype == np.float64 and s.dtype == np.float32 and c.dtype == np.float64
        ref = {tuple(np.round(r, 6)): i for i, r in enumerate(rc)}
        hits = [(i, ref[tuple(np.round(r, 6))]) for i, r in enumerate(c) if tuple(np.round(r, 6)) in ref]
        assert len(hits) >= 0.9 * max(len(rc), 1), (len(hits), len(rc))
        if hits:
            gi = np.array([h[0] for h in hits]); ri = np.array([h[1] for h in hits])
            assert np.median(np.abs(m[gi] - rm[ri]).max(axis=1)) < 0.02


# ------------------------------------------------------------------------------------------ round-2 additions
def test_shift_to_anchors_product_method(dev):
    """Patch2Pix.shift_to_anchors with panc = 8 (networks/patch2pix.py:377-402) is the product's own method, not the
    oracle's: rows, order and dtype against the oracle, then both regressors on the expanded set."""
    net = _model(dev)
    assert net.panc == 1
    g = torch.Generator().manual_seed(5)
    base = torch.stack([torch.randint(8, 120, (37,), generator=g), torch.randint(8, 88, (37,), generator=g),
                        torch.randint(8, 120, (37,), generator=g), torch.randint(8, 88, (37,), generator=g)], 1)
    same = net.shift_to_anchors([base.to(dev)])
    assert torch.equal(same[0].cpu(), base)                              # panc == 1: identity (:380-381)
    net.panc = 8
    out = net.shift_to_anchors([base.to(dev), base[:5].to(dev)])
    ref = orc.shift_to_anchors(base, net.pshift, 8)
    assert out[0].dtype == torch.int64 and torch.equal(out[0].cpu(), ref) and out[0].shape == (37 * 8, 4)
    assert torch.equal(out[1].cpu(), orc.shift_to_anchors(base[:5], net.pshift, 8))
    p1, p2 = synthetic.make_pyramid(11, 96, 128), synthetic.make_pyramid(12, 96, 128)
    f1, f2 = [t[None].to(dev) for t in p1], [t[None].to(dev) for t in p2]
    fine, fine_s, mid, mid_s = net._fine_chain(f1, f2, [out[0]])
    _, mid_p, _ = orc.split_params(gu.state_dict(0))
    ref_mid, ref_p, _ = orc.fine_level(p1[:4], p2[:4], ref, mid_p)
    _compare_matches(mid[0].cpu(), ref_mid)
    assert (mid_s[0].cpu() - ref_p).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("hw", [(96, 128), (101, 135), (75, 83)])
def test_refine_matches_vs_oracle(hw, dev):
    """Patch2Pix.refine_matches (networks/patch2pix.py:278-318) on images whose size is NOT a multiple of 8 (its
    caller, model_helper.refine_matches, loads images with load_im_tensor, which does not round sizes): the backbone's
    maps have ceil(H / 2^j) rows while the gather clamps to H // 2^j - 1 (networks/utils.py:22-23)."""
    net = _model(dev)
    H, W = hw
    im1, im2 = synthetic.make_image_pair(9, 128, 160)
    t1 = torch.from_numpy(im1[:H, :W].copy()).permute(2, 0, 1).float().div(255)[None].to(dev)
    t2 = torch.from_numpy(im2[:H, :W].copy()).permute(2, 0, 1).float().div(255)[None].to(dev)
    g = torch.Generator().manual_seed(H)
    n = 40
    coarse = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                          torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
    coarse[0] = torch.tensor([W, H, W, H])
    coarse[1] = torch.tensor([W - 1, H - 1, 0, 0])
    with torch.no_grad():
        r, s, c = net.refine_matches(t1, t2, coarse.numpy(), 0.0)
        pyr1 = [f[0].cpu() for f in net.extract.pyramid(t1)]
        pyr2 = [f[0].cpu() for f in net.extract.pyramid(t2)]
    assert pyr1[1].shape[-2:] == ((H + 1) // 2, (W + 1) // 2) and pyr1[3].shape[-2:] == ((H + 7) // 8, (W + 7) // 8)
    assert r.shape == (n, 4) and np.array_equal(c, coarse.numpy())
    _, mid_p, fine_p = orc.split_params(gu.state_dict(0))
    ref_mid, _, _ = orc.fine_level(pyr1[:4], pyr2[:4], coarse, mid_p)
    # the oracle's fine level is fed its own mid matches here; rows whose mid coordinate is within 2e-4 px of an
    # integer may legitimately move by one patch pixel (trunc, networks/utils.py:19) and are excluded
    ref_fine, ref_p, _ = orc.fine_level(pyr1[:4], pyr2[:4], ref_mid, fine_p)
    ok = ~_near_integer_rows(ref_mid)
    assert (~ok).sum() <= 2
    _compare_matches(torch.from_numpy(r)[ok], ref_fine[ok])
    assert (torch.from_numpy(s)[ok] - ref_p[ok]).abs().max() <= SCORE_TOL
    # io_thres keeps the confident rows, unless none passes (:312-317)
    thr = float(np.median(s))
    r2, s2, c2 = net.refine_matches(t1, t2, coarse, thr)
    assert (s2 > thr).all() and len(s2) == int((s > thr).sum()) and np.array_equal(c2, coarse.numpy()[s > thr])
    r3, s3, c3 = net.refine_matches(t1, t2, coarse, 2.0)
    assert len(s3) == n


def test_model_helper_refine_matches_on_files(dev, tmp_path):
    """utils/eval/model_helper.py:111-127 -- refine the matches of a caller-supplied coarse matcher on image files,
    through load_im_tensor (grey + RGB, plain rounding to imsize)."""
    from PIL import Image
    from patch2pix_amd.utils.eval import model_helper
    net = _model(dev)
    im1, im2 = synthetic.make_image_pair(3, 150, 210)
    Image.fromarray(im1).save(tmp_path / "1.png")
    Image.fromarray(im2).save(tmp_path / "2.png")
    seen = {}

    def matcher(g1, g2):
        seen["shapes"] = (tuple(g1.shape), tuple(g2.shape), g1.device.type, float(g1.max()))
        return torch.tensor([[20, 20, 28, 36], [60, 44, 68, 60], [99, 70, 92, 68]], dtype=torch.int64)

    r, s, c = model_helper.refine_matches(str(tmp_path / "1.png"), str(tmp_path / "2.png"), net, matcher, io_thres=0.0, imsize=100)
    assert seen["shapes"][0] == (1, 1, 71, 100) and seen["shapes"][2] == "cuda" and seen["shapes"][3] <= 1.0
    assert r.shape == (3, 4) and s.shape == (3,) and c.shape == (3, 4) and r.dtype == np.float64
    scale = np.array([210 / 100, 150 / 71, 210 / 100, 150 / 71])
    np.testing.assert_allclose(c, scale * np.array([[20, 20, 28, 36], [60, 44, 68, 60], [99, 70, 92, 68]]))
    assert (np.abs(r / scale - c / scale) <= 16.0 + 1e-3).all()
    only, none1, none2 = model_helper.refine_matches(str(tmp_path / "1.png"), str(tmp_path / "2.png"), net, matcher, coarse_only=True)
    assert none1 is None and none2 is None and only.shape == (3, 4)


REAL_PAIRS = [("real_pair_1", None), ("real_pair_2", 640), ("real_pair_3", 1024),
              ("real_pair_1_contrast", None), ("real_pair_2_contrast", 640), ("real_pair_3_contrast", 1024)]


@pytest.mark.parametrize("name,imsize", REAL_PAIRS)
def test_real_image_pairs(name, imsize, dev, capsys):
    """The reference's three example pairs (real photographs), two checkpoints: the plain random-init one (dense positive
    features, nearly flat volume: the near-tie stress case) and the `_contrast` one whose backbone yields sparse features
    (synthetic.contrast_shift) so that the argmaxes of a photograph are decidable wherever the image content allows it
    (pair_2 is a night shot: 41 % / 73 % black pixels, a third of its feature cells have an identical twin).
    (1) HIP path vs CPU oracle vs the unmodified reference's rows on IDENTICAL pyramids (this implementation's backbone
    run on the CPU; drift vs the reference's features is recorded): every row that is decidable in fp32 (fp64 margin above
    the error bound of both candidates, tests/adjudicate.py) must hold the fp64 winner, every differing row must be
    undecidable; the contrast fixture of pair_1 must have NO differing row at all, pair_3 (6144 rows) at most two.
    Regressed coordinates within 1e-3 px on identical proposals.
    (2) The drop-in entry estimate_matches (backbone on MIOpen) against the reference's recorded output."""
    import os
    from patch2pix_amd.utils.datasets.preprocess import load_im_flexible
    from patch2pix_amd.utils.eval import model_helper
    from adjudicate import ErrorModel, assert_decidable_rows, differing_rows_are_near_ties
    g = gu.load(name)
    contrast = torch.from_numpy(g["contrast_shift"]) if "contrast_shift" in g else None
    d = os.path.join(gu.GOLDEN, "images", str(g["pair"]))
    net = _model(dev, contrast)
    t1, s1 = load_im_flexible(os.path.join(d, "1.jpg"), 2, net.upsample, imsize=imsize)
    t2, s2 = load_im_flexible(os.path.join(d, "2.jpg"), 2, net.upsample, imsize=imsize)
    # (1) identical pyramids on both sides
    cpu_net = net.extract.to("cpu")
    try:
        with torch.no_grad():
            pyr1 = [f[0] for f in cpu_net.pyramid(t1[None])]
            pyr2 = [f[0] for f in cpu_net.pyramid(t2[None])]
    finally:
        net.extract.to(dev)
    drift = abs(gu.checksum([pyr1[4][None]]) - float(g["feat1_checksum"])) / float(g["feat1_checksum"])
    sd = gu.state_dict(0) if contrast is None else synthetic.make_state_dict(0, contrast=contrast)
    with torch.no_grad():
        f1, f2 = [f[None].to(dev) for f in pyr1], [f[None].to(dev) for f in pyr2]
        ticket = net.coarse_async(f1, f2, ksize=2)
        fine, fine_s, mid, mid_s, coarse = net.fine_from_ticket(ticket, 0.0, True, return_all=True)
        o_ncn, mid_p, fine_p = orc.split_params(sd)
        rc, rd = orc.coarse_forward(pyr1[4], pyr2[4], 2, o_ncn)
        rm, rs = orc.cal_coarse_matches(rc, rd, 2, 8)
    got_rows = ticket["matches"][0].cpu()
    lists = {"oracle": rm}
    if "all_rows" in g:
        lists["reference"] = torch.from_numpy(g["all_rows"].astype(np.int64))
    ndiff = {k: int((got_rows != v).any(dim=1).sum()) for k, v in lists.items()}
    expect_exact = name == "real_pair_1_contrast"
    report = f"{name}: identical pyramids (feature drift vs the reference's CPU run {drift:.1e}): coarse rows differing from " + \
        ", ".join(f"the {k} {n} of {rm.shape[0]}" for k, n in ndiff.items())
    if expect_exact:
        assert all(n == 0 for n in ndiff.values()), report
    if name == "real_pair_3_contrast":
        assert all(n <= 2 for n in ndiff.values()), report
    if any(ndiff.values()) or name in ("real_pair_1", "real_pair_1_contrast", "real_pair_2_contrast"):
        # the error model costs minutes of CPU at 1024 px: built where a row differs and for the two smaller contrast pairs
        em = ErrorModel(pyr1[4], pyr2[4], sd, 2)
        slack = em.check(rc, "oracle fp32 volume")
        with torch.no_grad():
            corr_gpu, _ = net.forward_coarse_match(f1[4], f2[4], ksize=2)
        slack_gpu = em.check(corr_gpu[0, 0].cpu(), "HIP volume")
        ndec, nrows = assert_decidable_rows(got_rows, em)
        worst = 0.0
        for k, v in lists.items():
            _, w = differing_rows_are_near_ties(got_rows, v, em)
            worst = max(worst, w)
        report += (f"; {ndec} of {nrows} rows decidable in fp32, all equal to the fp64 winner; every differing row undecidable "
                   f"(worst fp64 gap {worst:.3f} of the fp32 error bound; |error| of the oracle / HIP volume {slack:.3f} / "
                   f"{slack_gpu:.3f} of the bound)")
    # fine stage on the kernel's own proposals against the oracle (identical pyramids, identical proposals)
    with torch.no_grad():
        ref_mid, ref_mp, _ = orc.fine_level(pyr1[:4], pyr2[:4], coarse[0].cpu(), mid_p)
        ref_fine, ref_fp, _ = orc.fine_level(pyr1[:4], pyr2[:4], mid[0].cpu(), fine_p)
    _compare_matches(mid[0].cpu(), ref_mid)
    _compare_matches(fine[0].cpu(), ref_fine)
    assert (mid_s[0].cpu() - ref_mp).abs().max() <= SCORE_TOL and (fine_s[0].cpu() - ref_fp).abs().max() <= SCORE_TOL
    with capsys.disabled():
        print(f"\n{report}; {coarse[0].shape[0]} proposals: max |d mid| {(mid[0].cpu() - ref_mid).abs().max():.2e} px, "
              f"|d fine| {(fine[0].cpu() - ref_fine).abs().max():.2e} px")
    # (2) the entry point on the files, against the reference's recorded output
    m, s, c = model_helper.estimate_matches(net, os.path.join(d, "1.jpg"), os.path.join(d, "2.jpg"), ksize=2, io_thres=0.25,
                                            eval_type="fine", imsize=imsize)
    refc = {tuple(np.round(r, 4)): i for i, r in enumerate(g["fine_coarse"])}
    hits = [(i, refc[tuple(np.round(r, 4))]) for i, r in enumerate(c) if tuple(np.round(r, 4)) in refc]
    frac = len(hits) / max(len(g["fine_coarse"]), 1)
    if hits:
        gi, ri = np.array([h[0] for h in hits]), np.array([h[1] for h in hits])
        err = np.abs(m[gi] - g["fine_matches"][ri]).max(axis=1)
        mma3 = float((err < 3.0).mean() * frac)
        with capsys.disabled():
            print(f"{name}: entry point (pyramid produced on the GPU): {len(c)} matches, reference {len(g['fine_coarse'])}; "
                  f"{frac:.3f} of the reference's coarse matches reproduced; of those: median |d| {np.median(err):.2e} px, max {err.max():.2e} px, "
                  f"within 1e-3 px {float((err < 1e-3).mean()):.3f}, within 3 px {float((err < 3).mean()):.3f}; "
                  f"MMA@3px-style agreement (reference = ground truth) {mma3:.3f}")
        assert np.median(err) < (1e-3 if contrast is not None else 0.05)
        if contrast is not None:
            assert float((err < 1e-3).mean()) >= 0.99 and mma3 >= 0.97
    assert frac >= (0.97 if contrast is not None else 0.7)


def test_homography_warp_pairs(dev, tmp_path, capsys):
    """BASELINE configs[2] substitute: seeded homography warps of the reference's example photographs
    (tools/hpatches_substitute.py) through the streaming entry point, contrast checkpoint.  The HPatches metric -- MMA@3px
    against the known homography -- of the HIP path equals the CPU oracle's (the same pipeline on pyramids from the CPU
    backbone), and the oracle's matches are reproduced."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(gu.GOLDEN)), "tools"))
    import hpatches_substitute as hs
    from patch2pix_amd.utils.datasets.preprocess import load_im_flexible
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    g = gu.load("real_pair_1_contrast")
    contrast = torch.from_numpy(g["contrast_shift"])
    net = _model(dev, contrast)
    sd = synthetic.make_state_dict(0, contrast=contrast)
    o_ncn, mid_p, fine_p = orc.split_params(sd)
    warps = hs.make_warp_pairs(str(tmp_path))[:2]
    imsize = 320
    got = list(estimate_matches_stream(net, [(a, b) for a, b, _ in warps], imsize=imsize, batch=2, workers=2))
    cpu_net = net.extract.to("cpu")
    try:
        for (a, b, H), (m, sc, c) in zip(warps, got):
            t1, s1 = load_im_flexible(a, 2, net.upsample, imsize=imsize)
            t2, s2 = load_im_flexible(b, 2, net.upsample, imsize=imsize)
            with torch.no_grad():
                pyr1 = [f[0] for f in cpu_net.pyramid(t1[None])]
                pyr2 = [f[0] for f in cpu_net.pyramid(t2[None])]
                rc, rd = orc.coarse_forward(pyr1[4], pyr2[4], 2, o_ncn)
                rm, rs = orc.cal_coarse_matches(rc, rd, 2, 8)
                cm, _ = orc.filter_coarse(rm, rs, 0.0, True)
                mid, _, _ = orc.fine_level(pyr1[:4], pyr2[:4], cm, mid_p)
                fine, fp, _ = orc.fine_level(pyr1[:4], pyr2[:4], mid, fine_p)
            keep = torch.nonzero(fp > 0.25).flatten()
            if keep.numel() == 0:
                keep = torch.arange(fp.numel())
            scale = np.array([s1[0], s1[1], s2[0], s2[1]], dtype=np.float64)
            ref_m, ref_c = fine[keep].double().numpy() * scale, cm[keep].double().numpy() * scale
            refc = {tuple(np.round(r, 4)): i for i, r in enumerate(ref_c)}
            hits = [(i, refc[tuple(np.round(r, 4))]) for i, r in enumerate(c) if tuple(np.round(r, 4)) in refc]
            frac = len(hits) / max(len(ref_c), 1)
            mma_hip, mma_ref = hs.mma(m, H), hs.mma(ref_m, H)
            with capsys.disabled():
                print(f"\nwarp {os.path.basename(b)}: {len(c)} matches (oracle {len(ref_c)}), {frac:.3f} of the oracle's reproduced, "
                      f"MMA@3px HIP {mma_hip:.4f} / oracle {mma_ref:.4f}")
            assert frac >= 0.97 and abs(len(c) - len(ref_c)) <= max(2, 0.03 * len(ref_c))
            assert abs(mma_hip - mma_ref) <= 0.03
            if hits:
                gi, ri = np.array([h[0] for h in hits]), np.array([h[1] for h in hits])
                assert np.median(np.abs(m[gi] - ref_m[ri]).max(axis=1)) < 1e-3
    finally:
        net.extract.to(dev)


def test_config_E_vs_oracle(dev, ops, cweights):
    """BASELINE configs[4]: 960x1280, ptmax 800, panc 8 -> 6400 proposals per pair (training-time options, opt-in).
    Coarse stage against the CPU oracle (19200 x 19200 correlation, 23 M-cell volume: all 9600 rows equal) and the
    fine stage on all 6400 proposals in the default arithmetic."""
    from patch2pix_amd.networks.utils import filter_coarse
    sd, ncn, _, _ = cweights
    H, W = 960, 1280
    p1, p2 = synthetic.make_correlated_pyramids(31, H, W)
    o_ncn, mid_p, fine_p = orc.split_params(sd)
    with torch.no_grad():
        rc, rd = orc.coarse_forward(p1[4], p2[4], 2, o_ncn)
        rm, rs = orc.cal_coarse_matches(rc, rd, 2, 8)
    corr, delta = ops.coarse_forward(p1[4].to(dev), p2[4].to(dev), 2, ncn)
    m, s = ops.coarse_matches(corr, delta, 2, 8, True)
    np.testing.assert_allclose(corr.cpu().numpy(), rc.numpy(), rtol=3e-4, atol=1e-7)
    # 9600 argmaxes over a 23 M-cell volume: one or two can be fp32 near-ties.  They go through the same fp32 error model as
    # every other size, evaluated locally (oracle/error_model.py: LocalErrorModel -- the fp64 consensus output only on the
    # A rows / B columns the candidates of a differing row need): fp64 gap <= 0.25 x the bound of the two candidates, and the
    # kernel's own volume within 0.25 x the bound of the fp64 value at those cells
    from adjudicate import differing_rows_are_near_ties_local
    ndiff, worst = differing_rows_are_near_ties_local(m.cpu(), rm, p1[4], p2[4], sd, 2, volume_got=corr.cpu())
    assert ndiff <= 2, f"{ndiff} of {rm.shape[0]} coarse rows differ at 960x1280"
    if ndiff:
        print(f"\nconfig E: {ndiff} coarse row(s) differ from the fp32 oracle on near-ties (fp64 gap {worst:.3f} of the fp32 error bound)")
    np.random.seed(3)
    cm, _ = filter_coarse(m[None], s[None], 0.0, True, ptmax=800)
    ref_cm, _ = orc.filter_coarse(rm, rs, 0.0, True, ptmax=800, rng=np.random.RandomState(3))
    if ndiff == 0:
        assert torch.equal(cm[0].cpu(), ref_cm)
    props = orc.shift_to_anchors(ref_cm, 8, 8)
    assert props.shape == (6400, 4)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid_w, fine_w = ops.RegressorWeights(sub("regress_mid."), dev), ops.RegressorWeights(sub("regress_fine."), dev)
    out = ops.regress(mid_w, fine_w, _gpu(p1[:4], dev), _gpu(p2[:4], dev), props.to(dev))
    with torch.no_grad():
        ref_mid, ref_mp, _ = orc.fine_level(p1[:4], p2[:4], props, mid_p)
        ref_fine, ref_fp, _ = orc.fine_level(p1[:4], p2[:4], out["matches1"].cpu(), fine_p)
    _compare_matches(out["matches1"].cpu(), ref_mid)
    _compare_matches(out["matches2"].cpu(), ref_fine)
    assert (out["probs1"].cpu() - ref_mp).abs().max() <= SCORE_TOL and (out["probs2"].cpu() - ref_fp).abs().max() <= SCORE_TOL
    # the device path at this size (9600 coarse rows > the 8192 that fit LDS: workspace sort of csrc/filter.hip): device
    # filter_coarse against the ORACLE's filter on the oracle's rows, regressors on the device-side counts against the oracle
    net = _model(dev)
    with torch.no_grad():
        dfine, dscores, dcoarse, counts = net.predict_fine_device([t[None].to(dev) for t in p1], [t[None].to(dev) for t in p2], ksize=2)
        fine_l, scores_l, coarse_l = net.unpad(dfine, dscores, dcoarse, counts)
        ref_sel, _ = orc.filter_coarse(m.cpu(), s.cpu(), 0.0, True)      # the oracle's filter on the rows the device path filters
        assert torch.equal(coarse_l[0].cpu(), ref_sel), "device filter_coarse differs from the oracle at 9600 rows"
        o_mid, _, _ = orc.fine_level(p1[:4], p2[:4], ref_sel, mid_p)
        o_fine, o_fp, _ = orc.fine_level(p1[:4], p2[:4], o_mid, fine_p)
    ok = ~_near_integer_rows(o_mid)
    _compare_matches(fine_l[0].cpu()[ok], o_fine[ok])
    assert (scores_l[0].cpu()[ok] - o_fp[ok]).abs().max() <= SCORE_TOL and int((~ok).sum()) <= 4


def test_documented_import_route_on_gpu(dev, tmp_path):
    """INTEGRATION.md, literally: sys.path.append('<repo>/patch2pix_amd'); from utils.eval.model_helper import ... in a
    fresh interpreter, then load a checkpoint FILE and match an image pair."""
    import os
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.save(synthetic.make_checkpoint(0), tmp_path / "ckpt.pth")
    im1, im2 = synthetic.make_image_pair(51, 240, 320)
    Image.fromarray(im1).save(tmp_path / "1.png")
    Image.fromarray(im2).save(tmp_path / "2.png")
    code = f"""
import sys
sys.path.append({os.path.join(root, 'patch2pix_amd')!r})
from utils.eval.model_helper import load_model, estimate_matches
from networks.patch2pix import Patch2Pix
model = load_model({str(tmp_path / 'ckpt.pth')!r}, method='patch2pix')
assert isinstance(model, Patch2Pix)
m, s, c = estimate_matches(model, {str(tmp_path / '1.png')!r}, {str(tmp_path / '2.png')!r}, ksize=2, io_thres=0.25, eval_type='fine', imsize=1024)
print('ROUTE_OK', m.shape, m.dtype, s.dtype, c.shape)
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert res.returncode == 0 and "ROUTE_OK" in res.stdout, res.stderr[-2000:]
    assert "float64 float32" in res.stdout


def test_device_filter_against_reference_golden(dev, ops):
    """p2p_filter_coarse_batch on the 2400 coarse rows the UNMODIFIED REFERENCE produced at 480x640
    (tests/golden/full_480x640.npz): the mutual set it returns equals the reference's filter_coarse output
    (networks/utils.py:38-72), rows and order; scores are those of the first occurrences."""
    g = gu.load("full_480x640")
    rows = torch.from_numpy(g["all_matches"].astype(np.int64))
    scores = torch.from_numpy(g["all_scores"])
    out_m, out_s, counts = ops.filter_coarse_batch(rows[None].to(dev), scores[None].to(dev), 0.0, True)
    c = int(counts[0])
    assert np.array_equal(out_m[0, :c].cpu().numpy(), g["mutual_matches"].astype(np.int64))
    ref_m, ref_s = orc.filter_coarse(rows, scores, 0.0, True)
    assert torch.equal(out_s[0, :c].cpu(), ref_s) and torch.equal(out_m[0, :c].cpu(), ref_m)


def _coarse_like_rows(seed, cells_a, cells_b, mutual_frac=0.3, wmax=1280):
    """A match list with the structure cal_coarse_matches produces (patch2pix.py:340-375): one row per B cell, then one
    row per A cell; a fraction of the A->B rows repeats a B->A row (a mutual match)."""
    g = torch.Generator().manual_seed(seed)
    n1, n2 = cells_b, cells_a
    first = torch.randint(0, wmax // 8, (n1, 4), generator=g) * 8 + 4
    second = torch.randint(0, wmax // 8, (n2, 4), generator=g) * 8 + 4
    nm = int(mutual_frac * min(n1, n2))
    second[torch.randperm(n2, generator=g)[:nm]] = first[torch.randperm(n1, generator=g)[:nm]]
    return torch.cat((first, second)), torch.rand(n1 + n2, generator=g)


@pytest.mark.parametrize("cells", [1200, 4800, 7500])      # 2400 rows (480x640), 9600 (960x1280), 15000 (1600-pixel images)
def test_device_filter_against_oracle(cells, dev, ops):
    """p2p_filter_coarse_batch against the oracle's filter_coarse (networks/utils.py:38-72) on a batch of three lists:
    mutual on/off, a threshold some rows pass, a threshold no row passes (second keep-all fall-back, :65-67), a list
    without any repeated row under `mutual` (first keep-all fall-back, :48-50)."""
    lists = [_coarse_like_rows(7 * cells + b, cells, cells) for b in range(2)]
    lists.append(_coarse_like_rows(5, cells, cells, mutual_frac=0.0))
    rows = torch.stack([l[0] for l in lists]).to(dev)
    scores = torch.stack([l[1] for l in lists]).to(dev)
    for mutual, thres in ((True, 0.0), (False, 0.0), (True, 0.6), (False, 0.6), (True, 2.0), (False, 2.0)):
        out_m, out_s, counts = ops.filter_coarse_batch(rows, scores, thres, mutual)
        for b in range(3):
            ref_m, ref_s = orc.filter_coarse(lists[b][0], lists[b][1], thres, mutual)
            c = int(counts[b])
            assert c == ref_m.shape[0], (cells, mutual, thres, b, c, ref_m.shape[0])
            assert torch.equal(out_m[b, :c].cpu(), ref_m) and torch.equal(out_s[b, :c].cpu(), ref_s), (cells, mutual, thres, b)


@pytest.mark.parametrize("n", [2400, 9600])
def test_device_tail_against_reference_semantics(n, dev, ops):
    """p2p_match_tail_batch against the numpy tail of the reference's estimate_matches (utils/eval/model_helper.py:92-109):
    rows with fine score > io_thres in order, EVERY row if none passes (:97-105), refined and coarse coordinates scaled
    to original pixels in float64; a count of -1 passes through."""
    g = torch.Generator().manual_seed(n)
    B = 4
    fine = torch.rand(B, n, 4, generator=g) * 1200
    scores = torch.rand(B, n, generator=g)
    scores[2] *= 0.2                                   # item 2: nothing passes 0.25
    coarse = torch.randint(0, 1280, (B, n, 4), generator=g)
    counts = torch.tensor([n, n // 3, n - 1, -1], dtype=torch.int32)
    scale = torch.tensor([[1.0, 1.0, 1.0, 1.0], [1.6, 1.5, 2.0, 2.25], [1.0 / 3.0, 1.7, 1.1, 1.3], [1, 1, 1, 1]], dtype=torch.float64)
    for io_thres in (0.25, 2.0):
        om, osc, oc, on = ops.match_tail_batch(fine.to(dev), scores.to(dev), coarse.to(dev), counts.to(dev), scale, io_thres)
        assert int(on[3]) == -1
        for b in range(3):
            c = int(counts[b])
            f, s, co = fine[b, :c].numpy(), scores[b, :c].numpy(), coarse[b, :c].numpy()
            pos = np.where(s > io_thres)[0]
            if len(pos) > 0:
                f, s, co = f[pos], s[pos], co[pos]
            up = scale[b].numpy()[None]
            k = int(on[b])
            assert k == len(s)
            assert np.array_equal(om[b, :k].cpu().numpy(), up * f) and np.array_equal(osc[b, :k].cpu().numpy(), s)
            assert np.array_equal(oc[b, :k].cpu().numpy(), up * co) and om.dtype == torch.float64


def test_unpad_refuses_host_fallback_sentinel(dev):
    """counts == -1 (a coordinate outside the device filter's packed key) must not turn into a slice of garbage."""
    net = _model(dev)
    z = torch.zeros(1, 4, 4, device=dev)
    with pytest.raises(RuntimeError):
        net.unpad(z, z[..., 0], z.long(), torch.tensor([-1], dtype=torch.int32, device=dev))


def test_device_side_filter_path_equals_host_path(dev):
    """predict_fine_device (filter_coarse on the device, regressors reading the counts from device memory, no host round
    trip) returns exactly what predict_fine_from_feats returns through the host-side filter."""
    from patch2pix_amd.utils.eval import model_helper
    net = model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)
    pairs = [synthetic.make_correlated_pyramids(900 + i, 128, 160) for i in range(3)]
    f1 = [torch.stack([p[0][j] for p in pairs]).to(dev) for j in range(5)]
    f2 = [torch.stack([p[1][j] for p in pairs]).to(dev) for j in range(5)]
    for mutual, thres in ((True, 0.0), (False, 0.0), (True, 0.9)):
        fine, scores, coarse = net.predict_fine_from_feats(f1, f2, ksize=2, ncn_thres=thres, mutual=mutual)
        dfine, dscores, dcoarse = net.unpad(*net.predict_fine_device(f1, f2, ksize=2, ncn_thres=thres, mutual=mutual))
        for b in range(3):
            assert torch.equal(dcoarse[b], coarse[b]), (mutual, thres, b)
            assert torch.equal(dfine[b], fine[b]) and torch.equal(dscores[b], scores[b]), (mutual, thres, b)


def test_graphed_matcher_equals_eager_path(dev):
    """utils/eval/graphed.py: the device path captured as ONE hipGraph (pyramids in, matches out; and images in, with
    the backbone inside the graph) replays to exactly what the eager calls return, for two different inputs per graph."""
    from patch2pix_amd.utils.eval import model_helper
    from patch2pix_amd.utils.eval.graphed import GraphedMatcher
    net = model_helper.load_model(synthetic.make_checkpoint(0), lprint=lambda *a: None)
    H, W = 128, 160
    # (1) pyramids as the static inputs: bit-identical to the eager host-filter path
    g = GraphedMatcher(net, H, W, with_backbone=False)
    for seed in (910, 911):
        p1, p2 = synthetic.make_correlated_pyramids(seed, H, W)
        f1, f2 = [t[None].to(dev) for t in p1], [t[None].to(dev) for t in p2]
        fine, scores, coarse = net.predict_fine_from_feats(f1, f2, ksize=2)
        gfine, gscores, gcoarse = g(f1, f2)
        assert torch.equal(gcoarse[0], coarse[0]) and torch.equal(gfine[0], fine[0]) and torch.equal(gscores[0], scores[0])
    # (2) images as the static inputs, backbone inside the graph
    g2 = GraphedMatcher(net, H, W, with_backbone=True)
    for seed in (5, 6):
        a, b = synthetic.make_image_pair(seed, H, W)
        norm = lambda x: (torch.from_numpy(x).permute(2, 0, 1).float() / 255.0 - 0.45)[None].to(dev) / 0.225
        ia, ib = norm(a), norm(b)
        with torch.no_grad():
            fine, scores, coarse = net.predict_fine(ia, ib, ksize=2)
        gfine, gscores, gcoarse = g2(ia, ib)
        assert torch.equal(gcoarse[0], coarse[0])
        assert (gfine[0] - fine[0]).abs().max() <= COORD_TOL and (gscores[0] - scores[0]).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("io_thres", [0.25, 0.99])
def test_estimate_matches_device_equals_host_tail(io_thres, dev, tmp_path):
    """model_helper.estimate_matches_device (filter_coarse, regressors AND the io_thres / scaling tail on the device, one
    copy back) returns exactly the triple of estimate_matches; io_thres 0.99 exercises the keep-everything fall-back."""
    from PIL import Image
    from patch2pix_amd.utils.eval import model_helper
    im1, im2 = synthetic.make_image_pair(21, 300, 420)
    Image.fromarray(im1).save(tmp_path / "1.png")
    Image.fromarray(im2).save(tmp_path / "2.png")
    net = _model(dev)
    # MIOpen may pick a different (atomics-based) convolution algorithm from one call to the next, so two backbone runs
    # on the same image differ in the last bits: both entry points get the SAME pyramids through a memo
    memo, real_pyramid = {}, net.extract.pyramid

    def pyramid(im):
        key = (tuple(im.shape), float(im.double().sum()))
        if key not in memo:
            memo[key] = real_pyramid(im)
        return memo[key]
    net.extract.pyramid = pyramid
    try:
        a = model_helper.estimate_matches(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2, io_thres=io_thres,
                                          imsize=256)
        b = model_helper.estimate_matches_device(net, str(tmp_path / "1.png"), str(tmp_path / "2.png"), ksize=2,
                                                 io_thres=io_thres, imsize=256)
    finally:
        net.extract.pyramid = real_pyramid
    assert a[0].shape[0] > 0
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)


def test_stream_against_reference_golden(dev, tmp_path):
    """estimate_matches_stream (loader threads, batched backbone, shared launches) against the REFERENCE's recorded
    estimate_matches output, not against this implementation's per-pair calls: the golden pair three times in one batch,
    backbone on the CPU like in the golden run -> the reference's rows in order, 1e-3 px / 1e-5."""
    import copy
    from PIL import Image
    from patch2pix_amd.utils.eval.stream import estimate_matches_stream
    g = gu.load("estimate_matches_240x320")
    im1, im2 = synthetic.make_image_pair(int(g["seed"]), int(g["H"]), int(g["W"]))
    Image.fromarray(im1).save(tmp_path / "1.png")
    Image.fromarray(im2).save(tmp_path / "2.png")
    net = _model(dev)
    cpu_extract = copy.deepcopy(net.extract).to("cpu")
    gpu_pyramid = net.extract.pyramid
    net.extract.pyramid = lambda im: [f.to(dev) for f in cpu_extract.pyramid(im.cpu())]
    try:
        out = list(estimate_matches_stream(net, [(str(tmp_path / "1.png"), str(tmp_path / "2.png"))] * 3, ksize=2, io_thres=0.25,
                                           batch=3, workers=2))
    finally:
        net.extract.pyramid = gpu_pyramid
    assert len(out) == 3
    for m, s, c in out:
        assert m.dtype == np.float64 and s.dtype == np.float32 and c.dtype == np.float64
        assert np.array_equal(c, g["fine_coarse"])
        assert np.abs(m - g["fine_matches"]).max() <= COORD_TOL and np.abs(s - g["fine_scores"]).max() <= SCORE_TOL


def test_device_normalisation_is_bit_identical(dev):
    """utils/datasets/preprocess.py::normalise_pixels on the GPU == the host normalisation of load_im_flexible (the
    reference's /255, -mean, /std, preprocess.py:32-60) for every uint8 value in every channel."""
    from PIL import Image
    from patch2pix_amd.utils.datasets.preprocess import _normalised, normalise_pixels
    px = np.zeros((16, 16, 3), np.uint8)
    flat = px.reshape(-1, 3)
    flat[:, 0], flat[:, 1], flat[:, 2] = np.arange(256), np.arange(256)[::-1], (np.arange(256) * 7) % 256
    host = _normalised(Image.fromarray(px))
    got = normalise_pixels(torch.from_numpy(px)[None].to(dev))[0].cpu()
    assert torch.equal(host, got)


# ---- pyramid producer: the HIP convolutions of conv1 ... layer3 against torch ---------------------------
def _extract(dev, seed=0):
    from patch2pix_amd.networks import resnet
    net = resnet.ResNet34()
    net.change_stride("layer3")
    sd = synthetic.make_state_dict(seed)
    net.load_state_dict({k[len("extract."):]: v for k, v in sd.items() if k.startswith("extract.") and "layer4" not in k}, strict=False)
    return net.eval()


def _image_batch(seeds, h, w):
    ims = [synthetic.make_image_pair(s, h, w)[0] for s in seeds]
    x = torch.stack([torch.from_numpy(a).permute(2, 0, 1).float() / 255.0 for a in ims])
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (x - mean) / std


@pytest.mark.parametrize("hw", [(96, 128), (101, 135), (240, 320)])
def test_backbone_pyramid_vs_fp64(hw, dev, monkeypatch, capsys):
    """Every level of the pyramid against the same module evaluated in fp64 on the CPU: the HIP producer has to be as close to
    it as the fp32 library path (MIOpen) is -- both are fp32-accumulating evaluations of the same sums (reference
    networks/resnet.py:138-157)."""
    net = _extract(dev)
    x = _image_batch([3, 4], *hw)
    with torch.no_grad():
        ref = [t for t in net.double().pyramid(x.double())]
        net = net.float().to(dev)
        monkeypatch.setenv("P2P_BACKBONE", "hip")
        got = net.pyramid(x.to(dev))
        monkeypatch.setenv("P2P_BACKBONE", "miopen")
        lib = net.pyramid(x.to(dev))
    assert len(got) == 5 and torch.equal(got[0].cpu(), x)
    rows = []
    for lv in range(1, 5):
        assert got[lv].shape == ref[lv].shape and got[lv].is_contiguous() and got[lv].dtype == torch.float32
        scale = ref[lv].abs().amax(dim=(1, 2, 3), keepdim=True)
        e_hip = ((got[lv].cpu().double() - ref[lv]).abs() / scale).max().item()
        e_lib = ((lib[lv].cpu().double() - ref[lv]).abs() / scale).max().item()
        rows.append((e_hip, e_lib))
        assert e_hip <= max(3e-6, 2.5 * e_lib), f"level {lv}: HIP {e_hip:.2e} vs MIOpen {e_lib:.2e} of the largest value"
    with capsys.disabled():
        print(f"\npyramid {hw}: max |err| / max |ref| per level, HIP / MIOpen: " + "  ".join(f"{a:.1e} / {b:.1e}" for a, b in rows))


def test_backbone_batch_and_tile_independence(dev, monkeypatch):
    """An image's pyramid does not depend on its batch mates (operand scales are per image) nor on the tile the launch
    heuristic picks (every tile sums an output's K axis in the same order): bit-identical."""
    from patch2pix_amd import ops
    net = _extract(dev).to(dev)
    x = _image_batch([5, 6, 7], 120, 168).to(dev)
    x[1] *= 40.0                                                     # a batch mate on a very different scale
    monkeypatch.setenv("P2P_BACKBONE", "hip")
    with torch.no_grad():
        full = net.pyramid(x)
        for i in range(3):
            one = net.pyramid(x[i:i + 1].contiguous())
            for lv in range(1, 5):
                assert torch.equal(one[lv][0], full[lv][i]), f"image {i} level {lv}"
        for tile in ((2, 4, 2), (1, 4, 2), (2, 2, 2), (1, 2, 2)):
            monkeypatch.setattr(ops, "FORCED_CONV_TILE", tile)
            other = net.pyramid(x)
            for lv in range(1, 5):
                assert torch.equal(other[lv], full[lv]), f"tile {tile} level {lv}"


def test_backbone_full_size_vs_library(dev, monkeypatch):
    """480x640 (the benchmark's image size), two images: HIP producer against the MIOpen evaluation of the same module."""
    net = _extract(dev).to(dev)
    x = _image_batch([8, 9], 480, 640).to(dev)
    with torch.no_grad():
        monkeypatch.setenv("P2P_BACKBONE", "hip")
        got = net.pyramid(x)
        monkeypatch.setenv("P2P_BACKBONE", "miopen")
        lib = net.pyramid(x)
    for lv in range(1, 5):
        assert got[lv].shape == lib[lv].shape
        scale = lib[lv].abs().amax(dim=(1, 2, 3), keepdim=True)
        assert ((got[lv] - lib[lv]).abs() / scale).max().item() < 1e-5
