"""Host-side pieces of the product path that need no GPU: filter_coarse semantics against the
oracle (which is pinned to the reference), image sizing, checkpoint schema."""
import numpy as np
import torch

from oracle import p2p_oracle as orc
from patch2pix_amd.networks.utils import filter_coarse
from patch2pix_amd.utils import synthetic
from patch2pix_amd.utils.datasets.preprocess import cal_rescale_size, load_im_flexible


def test_filter_coarse_matches_oracle_including_fallbacks():
    g = torch.Generator().manual_seed(0)
    for trial in range(10):
        n = 150
        m = torch.randint(0, 5, (n, 4), generator=g) * 8 + 4
        s = torch.rand(n, generator=g)
        for mutual in (True, False):
            for thres in (0.0, 0.5, 2.0):          # 2.0: nothing passes -> keep-all fallback
                a, b = filter_coarse(m[None], s[None], thres, mutual)
                r, rs = orc.filter_coarse(m, s, thres, mutual)
                assert torch.equal(a[0], r) and torch.equal(b[0], rs)
        np.random.seed(5)
        a, _ = filter_coarse([m], [s], 0.0, True, ptmax=37)
        r, _ = orc.filter_coarse(m, s, 0.0, True, ptmax=37, rng=np.random.RandomState(5))
        assert torch.equal(a[0], r) and a[0].shape[0] == 37
    # no duplicates at all + mutual -> the reference keeps every row (utils.py:48-50)
    m = torch.arange(40).reshape(10, 4)
    a, b = filter_coarse([m], [torch.ones(10)], 0.0, True)
    assert torch.equal(a[0], m)
    # coordinates too large for the packed key fall back to the generic row-unique
    big = torch.tensor([[70000, 1, 2, 3], [70000, 1, 2, 3], [5, 1, 2, 3]])
    a, _ = filter_coarse([big], [torch.tensor([0.3, 0.2, 0.9])], 0.0, True)
    assert torch.equal(a[0], big[:1])


def test_rescale_size_rounds_down_to_multiple_of_16():
    assert cal_rescale_size(640, 640, 480, k_size=2, scale_factor=1 / 8) == (640, 480)
    assert cal_rescale_size(1024, 1600, 1200, k_size=2, scale_factor=1 / 8) == (1024, 768)
    assert cal_rescale_size(400, 400, 300, k_size=2, scale_factor=1 / 8) == (400, 288)


def test_load_im_flexible(tmp_path):
    from PIL import Image
    im1, _ = synthetic.make_image_pair(3, 150, 203)
    Image.fromarray(im1).save(tmp_path / "a.png")
    t, scale = load_im_flexible(str(tmp_path / "a.png"), 2, 8, imsize=None)
    assert t.shape == (3, 144, 192) and t.dtype == torch.float32
    assert scale == (203 / 192, 150 / 144)
    t2, _ = load_im_flexible(str(tmp_path / "a.png"), 2, 8, imsize=4096)      # never up-sample
    assert t2.shape == t.shape


def test_synthetic_checkpoint_has_reference_schema():
    ck = synthetic.make_checkpoint(1, backbone=False)
    assert {"backbone", "feat_idx", "change_stride", "regressor_config", "state_dict"} <= set(ck)
    sd = ck["state_dict"]
    assert tuple(sd["ncn.conv.0.weight"].shape) == (3, 16, 1, 3, 3, 3)
    assert tuple(sd["regress_fine.conv.0.weight"].shape) == (512, 518, 3, 3)
    assert ck["regressor_config"].psize == [16, 16]


def test_filter_coarse_property_based():
    """Random match lists (many duplicates, pixel coordinates of any magnitude the packed 64-bit key supports and
    beyond) against the oracle's restatement of networks/utils.py:38-72."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 80), st.integers(1, 6), st.sampled_from([8, 1000, 65535, 70000]), st.booleans(),
           st.sampled_from([0.0, 0.3, 0.9, 1.5]), st.integers(0, 2 ** 31 - 1))
    def check(n, distinct, span, mutual, thres, seed):
        g = torch.Generator().manual_seed(seed)
        pool = torch.randint(0, span + 1, (distinct, 4), generator=g)
        rows = pool[torch.randint(0, distinct, (n,), generator=g)]
        scores = torch.rand(n, generator=g)
        a, b = filter_coarse([rows], [scores], thres, mutual)
        r, rs = orc.filter_coarse(rows, scores, thres, mutual)
        assert torch.equal(a[0], r) and torch.equal(b[0], rs)
        # rows come back in lexicographic order unless the selection was empty and everything was kept in input order
        # (networks/utils.py:48-50: `mutual` with no row occurring twice; the score threshold may still thin that list)
        has_duplicate = len({tuple(x) for x in rows.tolist()}) < n
        if not mutual or has_duplicate:
            keys = [tuple(x) for x in a[0].tolist()]
            assert keys == sorted(keys)

    check()


def test_gpu_local_cpu_list_parsing_and_no_gpu_behaviour():
    """utils/host.py: sysfs cpulist syntax; without a GPU (here) nothing is pinned and nothing raises."""
    from patch2pix_amd.utils import host
    assert host._parse_cpulist("64-67,192,194-195\n") == {64, 65, 66, 67, 192, 194, 195}
    assert host._parse_cpulist("") == set()
    if not torch.cuda.is_available():
        assert host.gpu_local_cpus(0) == set() and host.pin_process_to_gpu(0) is None


def test_backbone_pack_cache_signature_follows_the_module():
    """networks/resnet.py: the packed device-side weights are re-made whenever the signature moves -- after
    load_state_dict (version counters), .data assignment (storage), a replaced Parameter (identity), change_stride."""
    import copy
    from patch2pix_amd.networks import resnet
    net = resnet.ResNet34()
    net.change_stride("layer3")
    s0 = net._hip_signature("cuda:0")
    assert net._hip_signature("cuda:0") == s0 and net._hip_signature("cuda:1") != s0
    net.load_state_dict(net.state_dict())
    s1 = net._hip_signature("cuda:0")
    assert s1 != s0
    net.layer2[1].bn2.running_var.add_(1.0)
    s2 = net._hip_signature("cuda:0")
    assert s2 != s1
    net.layer1[0].conv1.weight.data = net.layer1[0].conv1.weight.data.clone()
    s3 = net._hip_signature("cuda:0")
    assert s3 != s2
    net.layer3[5].conv2.weight = torch.nn.Parameter(net.layer3[5].conv2.weight.detach().clone())
    s4 = net._hip_signature("cuda:0")
    assert s4 != s3
    net.change_stride("layer2")
    assert net._hip_signature("cuda:0") != s4
    net._hip_trunk_cache, net._hip_sig = object(), s4          # the cache itself is not copied or pickled
    twin = copy.deepcopy(net)
    assert not hasattr(twin, "_hip_trunk_cache") and not hasattr(twin, "_hip_sig")


def test_homography_substitute_metric():
    """tools/hpatches_substitute.py (BASELINE configs[2] stand-in): the seeded homography maps the four corners as drawn, and
    MMA@3px counts a match iff its first point, mapped by H, lands within 3 px of the second."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hpatches_substitute as hs
    rng = np.random.RandomState(3)
    H = hs.random_homography(rng, 640, 480)
    assert abs(H[2, 2] - 1.0) < 1e-12
    pts = np.array([[10.0, 20.0], [600.0, 400.0], [320.0, 240.0], [5.0, 470.0]])
    q = np.concatenate([pts, np.ones((4, 1))], 1) @ H.T
    q = q[:, :2] / q[:, 2:3]
    good = np.concatenate([pts, q + [[1.0, -1.5], [0.0, 0.0], [2.0, 2.0], [0.5, 0.5]]], 1)
    assert hs.mma(good, H) == 1.0
    bad = good.copy()
    bad[0, 2] += 3.5                                   # 3.5 px beyond the 1 px offset of the first match
    bad[3, 3] -= 10.0
    assert hs.mma(bad, H) == 0.5
    assert hs.mma(np.zeros((0, 4)), H) == 0.0
    corners = np.array([[0, 0, 1], [640, 0, 1], [640, 480, 1], [0, 480, 1]], dtype=float) @ H.T
    corners = corners[:, :2] / corners[:, 2:3]
    assert np.abs(corners - [[0, 0], [640, 0], [640, 480], [0, 480]]).max() <= 0.12 * 640 + 1e-6
