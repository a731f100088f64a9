"""The HIP kernels' own code, executed on the CPU by the test-suite's HIP stand-in (tests/hipemu: the unmodified
.hip sources compiled for x86, work-items as fibers, wave64 collectives and the two MFMA shapes emulated) and checked
against the golden vectors of the reference and against the oracle.

This is TEST INFRASTRUCTURE for the index arithmetic of the kernels when no GPU is at hand; the emulated library is
built and loaded only here.  The product has no CPU path (tests/test_cabi_exports.py::test_no_cpu_fallback), and the
parity claims rest on the -m gpu tests, which run the same sources on the MI355X.
Tolerances are those of tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import p2p_oracle as orc
from patch2pix_amd.utils import synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
import emu_lib  # noqa: E402

COORD_TOL, SCORE_TOL = 1e-3, 1e-5


@pytest.fixture(scope="module")
def emu():
    return emu_lib.load()


@pytest.fixture(scope="module")
def sd():
    return gu.state_dict(0)


@pytest.fixture(scope="module")
def ncn(emu, sd):
    return emu_lib.ncn_create(emu, sd)


@pytest.mark.parametrize("name", gu.COARSE_CASES)
def test_coarse_stage_against_reference_golden(name, emu, ncn):
    g = gu.load(name)
    p1, p2 = gu.coarse_inputs(g)
    ksize = int(g["ksize"])
    corr, delta = emu_lib.coarse_forward_batch(emu, ncn, p1[4][None], p2[4][None], ksize)
    np.testing.assert_allclose(corr[0].numpy(), g["corr4d"], rtol=2e-4, atol=1e-7)
    if ksize > 1:
        k, rd = ksize, g["delta4d"].astype(np.int64)
        assert np.array_equal(delta[0].numpy(), (((rd[0] * k + rd[1]) * k + rd[2]) * k + rd[3]).reshape(delta[0].shape))
    m, s = emu_lib.coarse_matches_batch(emu, corr, delta, ksize, 8)
    assert np.array_equal(m[0].numpy(), g["all_matches"])
    np.testing.assert_allclose(s[0].numpy(), g["all_scores"], rtol=2e-4)


@pytest.mark.parametrize("ksize", [2, 1])
def test_coarse_stage_is_tile_independent(ksize, emu, sd):
    """The consensus kernel's work-group tile (ta, tb, tc) -- picked from the volume and the batch size -- must not change a
    single bit of the coarse stage: every output cell sums the contributions of its 3 x 3 hidden strips in one fixed order
    (consensus.hip).  Forced tiles incl. marches that do not divide the first axis, against the automatic choice and
    against the oracle; last axis 11 and 22 (not multiples of 4)."""
    p1, p2 = synthetic.make_correlated_pyramids(321, 96 if ksize == 1 else 112, 176)
    o_ncn, _, _ = orc.split_params(sd)
    rc, _ = orc.coarse_forward(p1[4], p2[4], ksize, o_ncn)
    ncn = emu_lib.ncn_create(emu, sd)
    base, bdelta = emu_lib.coarse_forward_batch(emu, ncn, p1[4][None], p2[4][None], ksize)
    np.testing.assert_allclose(base[0].numpy(), rc.numpy(), rtol=2e-4, atol=1e-7)
    tiles = ((2, 3, 2), (4, 6, 6), (0, 2, 5), (3, 4, 3), (30, 6, 6), (1, 5, 8))
    for tile in (tiles if ksize == 2 else tiles[::2]):         # (the un-pooled volume is 16 x the cells: half the tiles there)
        emu_lib.check(emu, emu.p2p_ncn_set_tile(ncn, *tile), "p2p_ncn_set_tile")
        corr, delta = emu_lib.coarse_forward_batch(emu, ncn, p1[4][None], p2[4][None], ksize)
        assert torch.equal(corr, base), f"tile {tile} changes the volume (max |d| {float((corr - base).abs().max()):.3e})"
        assert bdelta is None or torch.equal(delta, bdelta)
    emu.p2p_ncn_destroy(ncn)


def test_coarse_batch_equals_single_pairs(emu, ncn):
    """One launch per kernel for B pairs == B single-pair calls, bit for bit, also when the workspace only holds two
    of the five pairs at a time."""
    pairs = [synthetic.make_correlated_pyramids(500 + i, 64, 96) for i in range(5)]
    fa = torch.stack([p[0][4] for p in pairs])
    fb = torch.stack([p[1][4] for p in pairs])
    singles = [emu_lib.coarse_forward_batch(emu, ncn, fa[i:i + 1], fb[i:i + 1], 2) for i in range(5)]
    for ws_pairs in (5, 2):
        corr, delta = emu_lib.coarse_forward_batch(emu, ncn, fa, fb, 2, ws_pairs=ws_pairs)
        m, s = emu_lib.coarse_matches_batch(emu, corr, delta, 2, 8)
        for i in range(5):
            assert torch.equal(corr[i], singles[i][0][0]) and torch.equal(delta[i], singles[i][1][0])
            m1, s1 = emu_lib.coarse_matches_batch(emu, singles[i][0], singles[i][1], 2, 8)
            assert torch.equal(m[i], m1[0]) and torch.equal(s[i], s1[0])


@pytest.mark.parametrize("mode", ["fp16x2w", "fp16x2", "f32"])
def test_regressors_against_reference_golden(mode, emu, sd):
    """The regressor kernels (fp16x2 with Winograd / direct conv2, exact fp32 MFMA) on the first proposals of the reference's
    forward_fine_match golden: integer proposals through the mid regressor, float proposals through the fine one."""
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    regs = {"int_mid": emu_lib.regressor_create(emu, sub("regress_mid."), mode),
            "float_fine": emu_lib.regressor_create(emu, sub("regress_fine."), mode)}
    g = gu.load("fine_48x64")
    p1, p2 = gu.fine_inputs(g)
    n = 4
    for tag, reg in regs.items():
        props = torch.from_numpy(g[tag + "_in"][:n])
        out = emu_lib.regress(emu, reg, None, p1[:4], p2[:4], props)
        assert (out["matches1"] - torch.from_numpy(g[tag + "_matches"][:n])).abs().max() <= COORD_TOL
        assert (out["probs1"] - torch.from_numpy(g[tag + "_probs"][:n])).abs().max() <= SCORE_TOL


def test_persistent_regressor_walks_many_proposals(sd, tmp_path):
    """The fp16x2 kernel's work-groups are persistent: each walks its share of the proposals and then runs the FC tail of all
    of them as batches of 16 rows on the f32 matrix path.  ONE emulated compute unit (a fresh process: the count is read
    once per process) and 17 proposals at one level: the proposal loop and two FC batches (16 + 1 rows) against the oracle
    (the mid -> fine hand-over through the scratch buffer is covered by the chain tests on three emulated units)."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tests/hipemu")
import emu_lib, golden_util as gu
from patch2pix_amd.utils import synthetic
emu = emu_lib.load(); sd = gu.state_dict(0)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
mid = emu_lib.regressor_create(emu, sub("regress_mid."), "fp16x2")
p1, p2 = synthetic.make_pyramid(7, 48, 64), synthetic.make_pyramid(8, 48, 64)
props = torch.randint(0, 48, (17, 4), generator=torch.Generator().manual_seed(3))
out = emu_lib.regress(emu, mid, None, p1[:4], p2[:4], props)
torch.save((props, out), sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = str(tmp_path / "out.pt")
    subprocess.check_call([sys.executable, "-c", code, f], env=dict(os.environ, HIPEMU_CUS="1"), cwd=root)
    props, out = torch.load(f)
    p1, p2 = synthetic.make_pyramid(7, 48, 64), synthetic.make_pyramid(8, 48, 64)
    _, mid_p, fine_p = orc.split_params(sd)
    ref_mid, ref_midp, ref_raw = orc.fine_level(p1[:4], p2[:4], props, mid_p)
    assert (out["raw1"] - ref_raw).abs().max() < 5e-5
    assert (out["matches1"] - ref_mid).abs().max() <= COORD_TOL and (out["probs1"] - ref_midp).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("mode", ["fp16x2w", "fp16x2"])
def test_regressor_chain_and_image_borders(mode, emu, sd):
    """Mid -> fine inside one launch (the fine patch is centred on the truncated mid match, its base is the
    un-truncated one), with proposals on the image corners where every level of the patch clamps."""
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid = emu_lib.regressor_create(emu, sub("regress_mid."), mode)
    fine = emu_lib.regressor_create(emu, sub("regress_fine."), mode)
    H, W = 48, 64
    p1, p2 = synthetic.make_pyramid(7, H, W), synthetic.make_pyramid(8, H, W)
    props = torch.tensor([[0, 0, W, H], [W, H, 0, 0], [31, 17, 5, 40]])
    _, mid_p, fine_p = orc.split_params(sd)
    ref_mid, ref_midp, ref_raw = orc.fine_level(p1[:4], p2[:4], props, mid_p)
    out = emu_lib.regress(emu, mid, fine, p1[:4], p2[:4], props)
    assert (out["raw1"] - ref_raw).abs().max() < 5e-5
    assert (out["matches1"] - ref_mid).abs().max() <= COORD_TOL
    assert (out["probs1"] - ref_midp).abs().max() <= SCORE_TOL
    # second level: the oracle is fed the kernel's own mid matches (a 1e-6 px wobble across an integer would move
    # the whole fine patch by one pixel, networks/utils.py:19)
    ref_fine, ref_finep, _ = orc.fine_level(p1[:4], p2[:4], out["matches1"], fine_p)
    assert (out["matches2"] - ref_fine).abs().max() <= COORD_TOL
    assert (out["probs2"] - ref_finep).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("mode", ["fp16x2w", "fp16x2"])
def test_regress_batch_items_of_different_sizes(mode, emu, sd):
    """p2p_regress_batch over items (pairs) of different image sizes, one of them empty == one call per item."""
    import ctypes
    from patch2pix_amd import _lib as real
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid = emu_lib.regressor_create(emu, sub("regress_mid."), mode)
    sizes, counts = [(16, 24), (24, 16), (8, 8), (16, 16)], [2, 1, 0, 1]
    g = torch.Generator().manual_seed(4)
    pyr1 = [synthetic.make_pyramid(200 + i, h, w)[:4] for i, (h, w) in enumerate(sizes)]
    pyr2 = [synthetic.make_pyramid(300 + i, h, w)[:4] for i, (h, w) in enumerate(sizes)]
    props = [torch.stack([torch.randint(0, w + 1, (n,), generator=g), torch.randint(0, h + 1, (n,), generator=g),
                          torch.randint(0, w + 1, (n,), generator=g), torch.randint(0, h + 1, (n,), generator=g)], 1)
             for (h, w), n in zip(sizes, counts)]
    n = sum(counts)
    allp = torch.cat(props).contiguous()
    arr_a, arr_b = (real.Pyramid * len(sizes))(), (real.Pyramid * len(sizes))()
    keep = []
    for i in range(len(sizes)):
        for arr, pyr in ((arr_a, pyr1[i]), (arr_b, pyr2[i])):
            lv = [t.contiguous() for t in pyr]
            keep.append(lv)
            for j in range(4):
                arr[i].level[j] = lv[j].data_ptr()
            arr[i].height, arr[i].width = lv[0].shape[-2:]
    m = torch.empty((n, 4)); p = torch.empty((n,))
    cnt = (ctypes.c_int * len(sizes))(*counts)
    ws, wsp, wsn = emu_lib.regress_scratch(emu, n)
    emu_lib.check(emu, emu.p2p_regress_batch(mid, None, len(sizes), arr_a, arr_b, cnt, allp.data_ptr(), 0, m.data_ptr(),
                                             p.data_ptr(), None, None, None, None, wsp, wsn, None), "p2p_regress_batch")
    start = 0
    for i, c in enumerate(counts):
        if c:
            single = emu_lib.regress(emu, mid, None, pyr1[i], pyr2[i], props[i])
            assert torch.equal(m[start:start + c], single["matches1"]) and torch.equal(p[start:start + c], single["probs1"])
        start += c


@pytest.mark.parametrize("ksize", [2, 1])
def test_plain_c_example_end_to_end(ksize, emu, ncn, sd, tmp_path):
    """examples/cabi_coarse.c itself (argument order, buffer sizes, file format), compiled against the stand-in and
    run on the CPU, against the ctypes calls into the same emulated library: bit-identical outputs.  On the GPU box
    tests/test_plain_c_host.py runs the gcc-built program against the real library."""
    import subprocess
    import build_emu
    import cabi_example_io as io
    exe = build_emu.build_example()
    pairs = [synthetic.make_correlated_pyramids(700 + i, 64, 96) for i in range(3)]
    fa = torch.stack([p[0][4] for p in pairs]).contiguous()
    fb = torch.stack([p[1][4] for p in pairs]).contiguous()
    io.write_input(tmp_path / "in.bin", sd, fa, fb, ksize)
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    corr, delta = emu_lib.coarse_forward_batch(emu, ncn, fa, fb, ksize)
    m, sc = emu_lib.coarse_matches_batch(emu, corr, delta, ksize, 8)
    c_corr, c_delta, c_m, c_s = io.read_output(tmp_path / "out.bin", corr.numel(), sc.numel(), ksize)
    assert np.array_equal(c_corr, corr.numpy().ravel())
    if ksize > 1:
        assert np.array_equal(c_delta, delta.numpy().ravel())
    assert np.array_equal(c_m, m.numpy().ravel()) and np.array_equal(c_s, sc.numpy().ravel())


@pytest.mark.parametrize("seed", range(8))
def test_coarse_stage_random_shapes(seed, emu, ncn, sd):
    """Feature maps of unrelated, odd sizes for the two images (down to 1x1), any channel count the library accepts,
    small batches: volume and relocalisation against the oracle, match extraction bit-exact on the kernel's volume."""
    rng = np.random.RandomState(100 + seed)
    ksize = int(rng.choice([1, 2]))
    hA, wA, hB, wB = [int(rng.randint(1, 8)) * ksize for _ in range(4)]
    C, B = int(rng.choice([32, 64, 128, 256])), int(rng.choice([1, 2, 3]))
    g = torch.Generator().manual_seed(seed)
    fa, fb = torch.randn(B, C, hA, wA, generator=g), torch.randn(B, C, hB, wB, generator=g)
    corr, delta = emu_lib.coarse_forward_batch(emu, ncn, fa, fb, ksize)
    m, s = emu_lib.coarse_matches_batch(emu, corr, delta, ksize, 8)
    o_ncn, _, _ = orc.split_params(sd)
    o64, _, _ = orc.split_params(sd, torch.float64)
    for b in range(B):
        rc, rd = orc.coarse_forward(fa[b], fb[b], ksize, o_ncn)
        # Unstructured features on tiny volumes can be ill-conditioned (the row/column maximum of the consensus output
        # is a near-cancelling sum and enters the final mutual matching cubed): the yardstick is how far the fp32
        # oracle itself is from an fp64 evaluation.
        r64, _ = orc.coarse_forward(fa[b].double(), fb[b].double(), ksize, o64)
        rel = lambda x: ((x.double() - r64).abs() / r64.abs().clamp_min(1e-30)).max().item()
        assert rel(corr[b]) <= max(3e-4, 4 * rel(rc)), (rel(corr[b]), rel(rc))
        kd = None
        if ksize > 1:
            k, d = ksize, delta[b].long()
            ref_s = ((rd[0] * k + rd[1]) * k + rd[2]) * k + rd[3]
            assert int((d != ref_s).sum()) <= 1, "relocalisation argmax differs beyond a near-tie"
            kd = [d // (k * k * k), (d // (k * k)) % k, (d // k) % k, d % k]
        rm, rs = orc.cal_coarse_matches(corr[b], kd, ksize, 8)
        assert torch.equal(m[b], rm)
        assert torch.allclose(s[b], rs, rtol=1e-4)


@pytest.mark.parametrize("mode", ["fp16x2", "f32"])
def test_regressor_on_image_sizes_that_are_not_multiples_of_8(mode, emu, sd):
    """refine_matches loads images without rounding their size (utils/datasets/preprocess.py:7-30): the backbone's maps
    then have ceil(H / 2^j) rows, while the gather clamps to H // 2^j - 1 (networks/utils.py:22-23) -- the last row /
    column of an odd-sized map is never read.  Oracle = the reference's indexing on the same maps."""
    H, W = 27, 37
    gen = torch.Generator().manual_seed(3)
    up = lambda d, j: (d + (1 << j) - 1) >> j

    def pyr():
        return [torch.randn(3, H, W, generator=gen)] + [torch.relu(torch.randn(c, up(H, j), up(W, j), generator=gen) + 0.3)
                                                        for c, j in ((64, 1), (64, 2), (128, 3))]
    p1, p2 = pyr(), pyr()
    props = torch.tensor([[W, H, 0, 0], [W - 1, H - 1, 3, 5], [18, 13, 30, 20], [0, H, W, 0]])
    _, mid_p, _ = orc.split_params(sd)
    ref_mid, ref_p, ref_raw = orc.fine_level(p1, p2, props, mid_p)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    out = emu_lib.regress(emu, emu_lib.regressor_create(emu, sub("regress_mid."), mode), None, p1, p2, props)
    assert (out["raw1"] - ref_raw).abs().max() < 5e-5
    assert (out["matches1"] - ref_mid).abs().max() <= COORD_TOL
    assert (out["probs1"] - ref_p).abs().max() <= SCORE_TOL


@pytest.mark.parametrize("seed", range(6))
def test_device_filter_coarse_against_oracle(seed, emu):
    """p2p_filter_coarse_batch (sort-unique, mutual, threshold, both keep-all fallbacks) == networks/utils.py:38-72."""
    g = torch.Generator().manual_seed(seed)
    n = int(torch.randint(1, 600, (1,), generator=g))
    distinct = int(torch.randint(1, max(2, n), (1,), generator=g))
    span, B = [8, 200, 32767][seed % 3], 1 + seed % 3
    pool = torch.randint(0, span + 1, (B, distinct, 4), generator=g)
    rows = torch.gather(pool, 1, torch.randint(0, distinct, (B, n), generator=g)[:, :, None].expand(-1, -1, 4))
    scores = torch.rand(B, n, generator=g)
    for mutual in (True, False):
        for thres in (0.0, 0.5, 2.0):
            got = emu_lib.filter_coarse_batch(emu, rows, scores, thres, mutual)
            for b in range(B):
                r, rs = orc.filter_coarse(rows[b], scores[b], thres, mutual)
                assert got[b] is not None and torch.equal(got[b][0], r) and torch.equal(got[b][1], rs)
    for bad in ([70000, 1, 2, 3], [-1, 1, 2, 3]):          # outside the packed key: the kernel asks for the host path
        assert emu_lib.filter_coarse_batch(emu, torch.tensor([[bad, [1, 2, 3, 4]]]), torch.rand(1, 2), 0.0, True) == [None]


@pytest.mark.parametrize("n,distinct", [(8192, 5000), (8193, 8000), (9600, 7000), (20000, 3), (40000, 30000)])
def test_device_filter_coarse_long_lists(n, distinct, emu):
    """Lists beyond the 8192 rows that fit LDS (a 960x1280 pair has 9600 rows) go through the workspace path: chunks
    sorted in LDS, chunk-spanning network steps on global memory.  Same contract, same oracle (networks/utils.py:38-72),
    incl. both keep-all fall-backs."""
    g = torch.Generator().manual_seed(n)
    B = 2
    pool = torch.randint(0, 1281, (B, distinct, 4), generator=g)
    rows = torch.gather(pool, 1, torch.randint(0, distinct, (B, n), generator=g)[:, :, None].expand(-1, -1, 4))
    scores = torch.rand(B, n, generator=g)
    for mutual, thres in ((True, 0.0), (False, 0.5), (True, 2.0)):
        got = emu_lib.filter_coarse_batch(emu, rows, scores, thres, mutual)
        for b in range(B):
            r, rs = orc.filter_coarse(rows[b], scores[b], thres, mutual)
            assert got[b] is not None and torch.equal(got[b][0], r) and torch.equal(got[b][1], rs)
    all_distinct = torch.arange(n)[None, :, None].expand(1, n, 4) % 30000                  # mutual selects nothing: keep all
    got = emu_lib.filter_coarse_batch(emu, all_distinct.contiguous(), scores[:1].contiguous(), 0.0, True)
    r, rs = orc.filter_coarse(all_distinct[0], scores[0], 0.0, True)
    assert torch.equal(got[0][0], r) and torch.equal(got[0][1], rs)


@pytest.mark.parametrize("mode,counts", [("fp16x2w", [2, -1, 1]), ("fp16x2", [2, 0, 1])])
def test_regress_with_device_counts(mode, counts, emu, sd):
    """p2p_regress_batch_dev: every item owns `stride` slots, the first counts[i] hold proposals; used slots equal the
    per-item call bit for bit, the others are not touched.  A count of -1 (the device filter's "take the host path") is an
    item without proposals and must not shift the items behind it."""
    import ctypes
    from patch2pix_amd import _lib as real
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    mid = emu_lib.regressor_create(emu, sub("regress_mid."), mode)
    fine = emu_lib.regressor_create(emu, sub("regress_fine."), mode)
    sizes, stride = [(16, 24), (24, 16), (8, 8)], 3
    g = torch.Generator().manual_seed(4)
    pyr1 = [synthetic.make_pyramid(200 + i, h, w)[:4] for i, (h, w) in enumerate(sizes)]
    pyr2 = [synthetic.make_pyramid(300 + i, h, w)[:4] for i, (h, w) in enumerate(sizes)]
    props = torch.zeros(len(sizes), stride, 4, dtype=torch.int64)
    for i, ((h, w), c) in enumerate(zip(sizes, counts)):
        c = max(c, 0)
        props[i, :c] = torch.stack([torch.randint(0, w + 1, (c,), generator=g), torch.randint(0, h + 1, (c,), generator=g),
                                    torch.randint(0, w + 1, (c,), generator=g), torch.randint(0, h + 1, (c,), generator=g)], 1)
    arr_a, arr_b, keep = (real.Pyramid * len(sizes))(), (real.Pyramid * len(sizes))(), []
    for i in range(len(sizes)):
        for arr, pyr in ((arr_a, pyr1[i]), (arr_b, pyr2[i])):
            lv = [t.contiguous() for t in pyr]
            keep.append(lv)
            for j in range(4):
                arr[i].level[j] = lv[j].data_ptr()
            arr[i].height, arr[i].width = lv[0].shape[-2:]
    n, mark = len(sizes) * stride, -777.0
    m1, p1, m2, p2 = torch.full((n, 4), mark), torch.full((n,), mark), torch.full((n, 4), mark), torch.full((n,), mark)
    cnt = torch.tensor(counts, dtype=torch.int32)
    ws, wsp, wsn = emu_lib.regress_scratch(emu, n)
    emu_lib.check(emu, emu.p2p_regress_batch_dev(mid, fine, len(sizes), arr_a, arr_b, cnt.data_ptr(), stride,
                                                 props.data_ptr(), 0, m1.data_ptr(), p1.data_ptr(), None, m2.data_ptr(),
                                                 p2.data_ptr(), None, wsp, wsn, None), "p2p_regress_batch_dev")
    for i, c in enumerate(counts):
        c = max(c, 0)
        used, rest = slice(i * stride, i * stride + c), slice(i * stride + c, (i + 1) * stride)
        if c:
            single = emu_lib.regress(emu, mid, fine, pyr1[i], pyr2[i], props[i, :c].contiguous())
            assert torch.equal(m1[used], single["matches1"]) and torch.equal(m2[used], single["matches2"])
            assert torch.equal(p1[used], single["probs1"]) and torch.equal(p2[used], single["probs2"])
        assert bool((m1[rest] == mark).all() and (m2[rest] == mark).all() and (p2[rest] == mark).all())


def test_match_tail_against_reference_semantics(emu):
    """p2p_match_tail_batch == the numpy tail of estimate_matches (utils/eval/model_helper.py:92-109): rows with score >
    io_thres in order, everything if none passes, float64 scaling; counts of -1 pass through."""
    import numpy as np
    g = torch.Generator().manual_seed(5)
    B, n = 4, 700
    fine = torch.rand(B, n, 4, generator=g) * 600
    scores = torch.rand(B, n, generator=g)
    scores[2] *= 0.2                                   # item 2: nothing passes 0.25 -> everything is kept
    coarse = torch.randint(0, 640, (B, n, 4), generator=g)
    counts = torch.tensor([700, 123, 300, -1], dtype=torch.int32)
    scale = torch.tensor([[1.0, 1.0, 1.0, 1.0], [1.6, 1.5, 2.0, 2.25], [1.0 / 3.0, 1.7, 1.1, 1.3], [1, 1, 1, 1]], dtype=torch.float64)
    got = emu_lib.match_tail_batch(emu, fine, scores, coarse, counts, scale, 0.25)
    assert got[3] is None
    for b in range(3):
        c = int(counts[b])
        f, s, co = fine[b, :c].numpy(), scores[b, :c].numpy(), coarse[b, :c].numpy()
        pos = np.where(s > 0.25)[0]
        if len(pos) > 0:
            f, s, co = f[pos], s[pos], co[pos]
        up = scale[b].numpy()[None]
        assert np.array_equal(got[b][0].numpy(), up * f) and np.array_equal(got[b][1].numpy(), s)
        assert np.array_equal(got[b][2].numpy(), up * co) and got[b][0].dtype == torch.float64
    assert len(got[2][1]) == 300


@pytest.mark.parametrize("dims", [(4, 6, 4, 6), (7, 9, 8, 13), (5, 3, 6, 70), (13, 14, 9, 11)])
def test_fused_consensus_against_oracle(dims, emu, sd):
    """p2p_neigh_consensus_batch (both consensus layers in one kernel on the fp16 matrix cores, csrc/consensus.hip) ==
    NeighConsensus.forward (ncn/model.py:145-155) of the oracle: volumes that are no multiples of the tile, a B row longer
    than one column tile (70 > 60), a batch of two."""
    g = torch.Generator().manual_seed(sum(dims))
    x = torch.rand(3, *dims, generator=g)
    x[1] = x[1] * 0.01                                  # a volume far below the scale the fp16 planes are laid out for
    x[2] = (x[2] - 0.5) * 300.0                         # ... and one far above it, with negative values (the kernel rescales by max |x|)
    ncn = emu_lib.ncn_create(emu, sd)
    y = emu_lib.neigh_consensus_batch(emu, ncn, x)
    o_ncn, _, _ = orc.split_params(sd)
    for b in range(3):
        ref = orc.neigh_consensus(x[b], o_ncn)
        assert (y[b] - ref).abs().max() <= 3e-6 * ref.abs().max(), (dims, b, float((y[b] - ref).abs().max()), float(ref.abs().max()))


# ---- pyramid producer (csrc/backbone.hip): every layer type of ResNet34 conv1 ... layer3 against torch in fp64 -----------
# The kernels are fp32-equivalent (operands to within 2^-24, fp32 accumulation); the emulated MFMA sums its K values one
# after the other in fp32, which is the worst summation order there is: errors of ~1e-6 of the largest output against
# 2-4e-7 for torch's blocked fp32 convolution.  On the MI355X the two are level (tests/test_gpu_parity.py).
BACKBONE_TOL = 4e-6


def _bn_params(co, gen):
    return [torch.rand(co, generator=gen) + 0.5, torch.randn(co, generator=gen) * 0.1, torch.randn(co, generator=gen) * 0.1,
            torch.rand(co, generator=gen) + 0.5]


def _bn_eval64(t, bn):
    g, b, m, v = [q.double().view(1, -1, 1, 1) for q in bn]
    return (t - m) / torch.sqrt(v + 1e-5) * g + b


@pytest.mark.parametrize("ci,co,ks,stride,n,h,w,res,relu,tile", [
    (64, 64, 3, 1, 2, 11, 19, True, True, None),           # layer1 block convolution (+ identity skip), two images of different scale
    (64, 128, 3, 2, 2, 13, 21, False, True, None),         # layer2[0].conv1: stride 2, odd extents
    (64, 128, 1, 2, 1, 13, 21, False, False, None),        # layer2[0].downsample: 1x1 stride 2, no ReLU
    (128, 128, 3, 1, 1, 9, 17, True, True, "2,2,2"),       # layer2, the large tile of a 128-channel layer
    (128, 256, 3, 1, 1, 9, 17, False, True, "2,4,2"),      # layer3[0].conv1 (stride patched to 1), the large tile
    (128, 256, 1, 1, 3, 7, 18, False, False, None),        # layer3[0].downsample
    (256, 256, 3, 1, 1, 5, 16, True, True, "1,4,2"),
    (256, 256, 3, 1, 1, 10, 16, True, True, "2,2,2"),
])
def test_backbone_convolution_against_torch(ci, co, ks, stride, n, h, w, res, relu, tile, emu):
    gen = torch.Generator().manual_seed(ci * 7 + co + ks + stride)
    wt = torch.randn(co, ci, ks, ks, generator=gen) * (2.0 / (ci * ks * ks)) ** 0.5
    bn = _bn_params(co, gen)
    x = torch.relu(torch.randn(n, ci, h, w, generator=gen)) * torch.tensor([1.0, 37.0, 0.003][:n]).view(n, 1, 1, 1)
    ref = _bn_eval64(torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, ks // 2), bn)
    skip = torch.randn(n, co, ref.shape[2], ref.shape[3], generator=gen) * ref.abs().amax(dim=(1, 2, 3), keepdim=True).float() if res else None
    if res:
        ref = ref + skip.double()
    if relu:
        ref = ref.relu()
    got, gmax = emu_lib.conv_bn(emu, wt, bn, stride, x, skip, relu, tile=tuple(int(v) for v in tile.split(",")) if tile else None)
    assert got.shape == ref.shape
    scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True)
    assert ((got.double() - ref).abs() / scale).max().item() < BACKBONE_TOL
    assert torch.equal(gmax, got.abs().amax(dim=(1, 2, 3)))          # what the next layer scales its operands by


@pytest.mark.parametrize("n,h,w", [(2, 37, 70), (1, 16, 130), (1, 64, 64)])
def test_backbone_stem_pool_transpose_against_torch(n, h, w, emu):
    gen = torch.Generator().manual_seed(h * w)
    wt = torch.randn(64, 3, 7, 7, generator=gen) * 0.1
    bn = _bn_params(64, gen)
    image = torch.randn(n, 3, h, w, generator=gen) * 1.3
    level1, pooled, back, pmax = emu_lib.stem_pool(emu, wt, bn, image)
    ref = _bn_eval64(torch.nn.functional.conv2d(image.double(), wt.double(), None, 2, 3), bn).relu()
    assert level1.shape == ref.shape
    assert ((level1.double() - ref).abs().max() / ref.abs().max()).item() < BACKBONE_TOL
    want = torch.nn.functional.max_pool2d(level1, 3, 2, 1)           # max-pool and transposition are exact
    assert torch.equal(pooled.permute(0, 3, 1, 2), want) and torch.equal(back, want)
    assert torch.equal(pmax, want.abs().amax(dim=(1, 2, 3)))


def test_backbone_basic_block_chain_against_torch(emu):
    """Two BasicBlocks the way networks/resnet.py composes the three kinds of call (reference resnet.py:26-60): a
    down-sampling block (conv1 stride 2 + ReLU, 1x1 stride-2 projection of the skip without ReLU, conv2 + skip + ReLU)
    followed by an identity block, each layer fed with the maximum its producer reported."""
    from patch2pix_amd.networks.resnet import _Basic
    torch.manual_seed(5)
    blocks = [_Basic(64, 128, 2).eval(), _Basic(128, 128, 1).eval()]
    for blk in blocks:
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    x = torch.relu(torch.randn(1, 64, 14, 18))
    with torch.no_grad():
        ref = x.double()
        for blk in blocks:
            ref = blk.double()(ref)
        for blk in blocks:
            blk.float()

    def bn(b):
        return [b.weight.data, b.bias.data, b.running_mean, b.running_var]
    cur = x
    for blk in blocks:
        skip = cur
        if blk.downsample is not None:
            skip, _ = emu_lib.conv_bn(emu, blk.downsample[0].weight.data, bn(blk.downsample[1]), blk.downsample[0].stride[0], cur, None, relu=False)
        y, ymax = emu_lib.conv_bn(emu, blk.conv1.weight.data, bn(blk.bn1), blk.conv1.stride[0], cur, None, relu=True)
        assert torch.equal(ymax, y.abs().amax(dim=(1, 2, 3)))
        cur, _ = emu_lib.conv_bn(emu, blk.conv2.weight.data, bn(blk.bn2), 1, y, skip, relu=True)
    assert cur.shape == ref.shape
    assert ((cur.double() - ref).abs().max() / ref.abs().max()).item() < 2 * BACKBONE_TOL


@pytest.mark.parametrize("case", ["one_pixel", "thin", "zeros", "signed", "huge"])
def test_backbone_convolution_edge_inputs(case, emu):
    """Extents smaller than a tile, an all-zero image (its recorded maximum is 0), signed inputs (the scale comes from
    max |x|) and magnitudes far outside fp16 (the power-of-two operand scale brings them back)."""
    gen = torch.Generator().manual_seed(11)
    ci, co = 64, 128
    wt = torch.randn(co, ci, 3, 3, generator=gen) * 0.05
    bn = _bn_params(co, gen)
    shape = {"one_pixel": (1, ci, 1, 1), "thin": (2, ci, 3, 37)}.get(case, (1, ci, 6, 9))
    x = torch.randn(shape, generator=gen)
    if case == "zeros":
        x.zero_()
    elif case == "huge":
        x = x.abs() * 3e20
    elif case != "signed":
        x = x.relu()
    for stride in (1, 2):
        ref = _bn_eval64(torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, 1), bn).relu()
        got, gmax = emu_lib.conv_bn(emu, wt, bn, stride, x, None, True)
        assert got.shape == ref.shape and torch.isfinite(got).all()
        scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
        assert ((got.double() - ref).abs() / scale).max().item() < BACKBONE_TOL
        assert torch.equal(gmax, got.abs().amax(dim=(1, 2, 3)))
