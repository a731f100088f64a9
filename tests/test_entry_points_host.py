"""Host logic of the two entry points (utils/eval/model_helper.py::estimate_matches and
utils/eval/stream.py::estimate_matches_stream) with a stand-in network that runs on the CPU: loading and scaling,
grouping of equally sized pairs, ordering, io_thres filtering with its keep-all fallback, return dtypes
(reference utils/eval/model_helper.py:64-109).  The numerical path itself is GPU-only and tested in -m gpu."""
import numpy as np
import pytest
import torch
from PIL import Image

from patch2pix_amd.utils import synthetic
from patch2pix_amd.utils.eval import model_helper
from patch2pix_amd.utils.eval.stream import estimate_matches_stream


class _Extract:
    def pyramid(self, im):
        # "features" that remember which image they came from: per-image mean and the spatial size
        return [im.mean(dim=(1, 2, 3))]


class FakeNet:
    """Deterministic stand-in: the matches of a pair are a function of the two images' means, so that any mix-up of
    pairs, order or scale factors shows."""
    device = torch.device("cpu")
    upsample = 8

    def __init__(self):
        self.extract = _Extract()
        self.calls = []

    @staticmethod
    def _rows(m1, m2, n):
        base = torch.arange(n, dtype=torch.float32)[:, None]
        fine = torch.cat([base + m1, base * 2 + m1, base + m2, base * 3 + m2], dim=1)
        conf = torch.linspace(0.0, 1.0, n)                 # row i has confidence i / (n-1)
        coarse = (fine + 100).round().long()
        return fine, conf, coarse

    def _n(self, m1):
        return 3 + int(abs(float(m1)) * 1000) % 5

    # --- per-pair API used by estimate_matches
    def predict_fine(self, im1, im2, ksize=2, ncn_thres=0.0, mutual=True):
        self.calls.append(("fine", tuple(im1.shape), ksize, ncn_thres, mutual))
        m1, m2 = im1.mean(), im2.mean()
        f, c, co = self._rows(m1, m2, self._n(m1))
        return [f], [c], [co]

    def predict_coarse(self, im1, im2, ksize=2, ncn_thres=0.0, mutual=True):
        self.calls.append(("coarse", tuple(im1.shape), ksize, ncn_thres, mutual))
        f, c, co = self._rows(im1.mean(), im2.mean(), 4)
        return [co], [c]

    # --- batched API used by the streaming entry point
    def coarse_async(self, feats1, feats2, ksize=2):
        self.calls.append(("batch", feats1[0].shape[0], ksize))
        return {"m1": feats1[0], "m2": feats2[0]}

    def fine_from_ticket(self, ticket, ncn_thres=0.0, mutual=True):
        fine, conf, coarse = [], [], []
        for m1, m2 in zip(ticket["m1"], ticket["m2"]):
            f, c, co = self._rows(m1, m2, self._n(m1))
            fine.append(f); conf.append(c); coarse.append(co)
        return fine, conf, coarse


def _save_pairs(tmp_path, sizes):
    pairs = []
    for i, (h, w) in enumerate(sizes):
        a, b = synthetic.make_image_pair(50 + i, h, w)
        pa, pb = tmp_path / f"{i}_a.png", tmp_path / f"{i}_b.png"
        Image.fromarray(a).save(pa); Image.fromarray(b).save(pb)
        pairs.append((str(pa), str(pb)))
    return pairs


def test_estimate_matches_contract(tmp_path):
    (p1, p2), = _save_pairs(tmp_path, [(150, 203)])
    net = FakeNet()
    m, s, c = model_helper.estimate_matches(net, p1, p2, ksize=2, io_thres=0.5, eval_type="fine")
    assert m.dtype == np.float64 and c.dtype == np.float64 and s.dtype == np.float32
    assert m.shape[1] == 4 and c.shape == m.shape and s.shape == (m.shape[0],)
    assert (s > 0.5).all() and 0 < len(s)                       # only the confident rows
    assert net.calls[-1] == ("fine", (1, 3, 144, 192), 2, 0.0, True)      # 150x203 -> multiples of upsample*ksize
    # rows are scaled back to the ORIGINAL image: x by 203/192, y by 150/144, for both images
    t1, sc1 = model_helper.load_im_flexible(p1, 2, 8)
    t2, sc2 = model_helper.load_im_flexible(p2, 2, 8)
    f, conf, co = FakeNet._rows(t1.mean(), t2.mean(), net._n(t1.mean()))
    keep = conf.numpy() > 0.5
    scale = np.array([sc1 + sc2])
    np.testing.assert_allclose(m, scale * f.numpy()[keep], rtol=1e-6)
    np.testing.assert_allclose(c, scale * co.numpy()[keep].astype(np.float64), rtol=1e-12)
    # nothing clears the threshold -> every row is returned (reference :97-105)
    m_all, s_all, _ = model_helper.estimate_matches(net, p1, p2, io_thres=2.0)
    assert len(s_all) == len(conf)
    # coarse evaluation returns the coarse matches twice and passes ncn_thres / mutual through
    mc, sc, cc = model_helper.estimate_matches(net, p1, p2, ksize=1, ncn_thres=0.3, mutual=False, eval_type="coarse")
    assert net.calls[-1] == ("coarse", (1, 3, 144, 200), 1, 0.3, False)
    assert mc is cc or np.array_equal(mc, cc)
    assert mc.dtype == np.float64 and sc.dtype == np.float32
    with pytest.raises(ValueError):
        model_helper.estimate_matches(net, p1, p2, eval_type="bogus")


def test_imsize_limits_the_longer_side(tmp_path):
    (p1, p2), = _save_pairs(tmp_path, [(300, 400)])
    net = FakeNet()
    model_helper.estimate_matches(net, p1, p2, imsize=256)
    assert net.calls[-1][1] == (1, 3, 192, 256)
    model_helper.estimate_matches(net, p1, p2, imsize=4096)          # never up-sampled
    assert net.calls[-1][1] == (1, 3, 288, 400)


@pytest.mark.parametrize("batch,workers", [(1, 1), (3, 2), (8, 4)])
def test_stream_equals_per_pair_host_logic(tmp_path, batch, workers):
    """Pairs of mixed sizes: the stream groups consecutive equally sized pairs (never more than `batch`), keeps the
    input order and returns exactly what per-pair estimate_matches returns."""
    sizes = [(96, 128), (96, 128), (96, 128), (128, 96), (96, 128), (96, 128), (160, 160), (160, 160), (96, 128)]
    pairs = _save_pairs(tmp_path, sizes)
    ref_net, net = FakeNet(), FakeNet()
    expected = [model_helper.estimate_matches(ref_net, a, b, io_thres=0.4) for a, b in pairs]
    got = list(estimate_matches_stream(net, pairs, io_thres=0.4, batch=batch, workers=workers))
    assert len(got) == len(expected)
    for (m, s, c), (em, es, ec) in zip(got, expected):
        assert m.dtype == np.float64 and s.dtype == np.float32 and c.dtype == np.float64
        np.testing.assert_allclose(m, em, rtol=1e-6)
        np.testing.assert_allclose(s, es, rtol=1e-6)
        np.testing.assert_allclose(c, ec, rtol=1e-6)
    groups = [n for tag, n, _ in net.calls if tag == "batch"]
    assert sum(groups) == len(pairs) and max(groups) <= batch
    if batch >= 3:
        assert groups[:3] == [3, 1, 2]        # 3 equal pairs, the transposed one alone, then two more


def test_stream_handles_empty_input():
    assert list(estimate_matches_stream(FakeNet(), [])) == []


def test_bounded_loader_map_keeps_order_and_bound():
    """stream._bounded_map: results in input order, never more than `ahead` jobs submitted beyond what was consumed."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from patch2pix_amd.utils.eval import stream
    lock, state = threading.Lock(), {"submitted": 0, "consumed": 0, "worst": 0}

    def work(j):
        with lock:
            state["submitted"] += 1
            state["worst"] = max(state["worst"], state["submitted"] - state["consumed"])
        return j * j

    with ThreadPoolExecutor(max_workers=3) as pool:
        got = []
        for v in stream._bounded_map(pool, work, range(40), ahead=5):
            with lock:
                state["consumed"] += 1
            got.append(v)
    assert got == [j * j for j in range(40)]
    assert state["worst"] <= 5 + 1          # the look-ahead window (+ the job submitted as its result is handed out)
    with ThreadPoolExecutor(max_workers=2) as pool:
        assert list(stream._bounded_map(pool, work, [], ahead=4)) == []
