"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the UNMODIFIED reference
(imported through oracle/ref_shim.py) on seeded synthetic inputs, on CPU fp32.

Run in the build container (where /root/reference exists):   python -m oracle.make_golden
The fixtures hold only the reference's *outputs* plus the seed recipe and a checksum of the
inputs; inputs are regenerated from the seeds at test time (same torch build on the GPU box).
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_shim import build_reference_net, load_reference   # noqa: E402
from patch2pix_amd.utils import synthetic                          # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def checksum(tensors):
    return float(sum(t.double().abs().sum().item() for t in tensors))


def np_(t):
    return t.detach().cpu().numpy()


def proposals(seed, n, H, W, as_float):
    g = torch.Generator().manual_seed(seed)
    m = torch.stack([torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g),
                     torch.randint(0, W + 1, (n,), generator=g), torch.randint(0, H + 1, (n,), generator=g)], 1)
    m[0] = torch.tensor([0, 0, W, H])
    m[1] = torch.tensor([W, H, 0, 0])
    m[2] = torch.tensor([3, H - 2, W - 5, 6])
    if as_float:
        m = m.float() + torch.rand(n, 4, generator=g) * 0.99
        m[:, 0::2].clamp_(0, W)
        m[:, 1::2].clamp_(0, H)
    return m


def case_coarse(ref, net, sd, name, seed, H, W, ksize):
    p1, p2 = synthetic.make_correlated_pyramids(seed, H, W)
    with torch.no_grad():
        corr, delta = net.forward_coarse_match(p1[4][None], p2[4][None], ksize=ksize)
        m, s = net.cal_coarse_matches(corr, delta, ksize=ksize, upsample=8, center=True)
        fm, fs = ref.utils.filter_coarse(m, s, 0.0, True)
        fa, fsa = ref.utils.filter_coarse(m, s, 0.0, False)
    out = dict(seed=seed, H=H, W=W, ksize=ksize, sd_seed=SD_SEED,
               input_checksum=checksum([p1[4], p2[4]]),
               corr4d=np_(corr[0, 0]), all_matches=np_(m[0]), all_scores=np_(s[0]),
               mutual_matches=np_(fm[0]), mutual_scores=np_(fs[0]),
               unique_matches=np_(fa[0]), unique_scores=np_(fsa[0]))
    if delta is not None:
        out["delta4d"] = np.stack([np_(d[0, 0]) for d in delta]).astype(np.int8)
    np.savez_compressed(os.path.join(GOLDEN, name), **out)
    print(name, "mutual", fm[0].shape[0], "corr", tuple(corr.shape))


def case_fine(ref, net, sd, name, seed, H, W, n):
    p1 = synthetic.make_pyramid(seed, H, W)
    p2 = synthetic.make_pyramid(seed + 1, H, W)
    f1 = [t[None] for t in p1]
    f2 = [t[None] for t in p2]
    out = dict(seed=seed, H=H, W=W, n=n, sd_seed=SD_SEED, input_checksum=checksum(p1 + p2))
    for tag, as_float, reg in (("int_mid", False, net.regress_mid), ("float_fine", True, net.regress_fine)):
        m = proposals(seed + 100, n, H, W, as_float)
        with torch.no_grad():
            rm, rp = net.forward_fine_match(f1, f2, [m], psize=16, ptype="center", regressor=reg)
        out[tag + "_in"] = np_(m)
        out[tag + "_matches"] = np_(rm[0])
        out[tag + "_probs"] = np_(rp[0])
    np.savez_compressed(os.path.join(GOLDEN, name), **out)
    print(name, "ok")


def case_predict_fine(ref, net, sd, name, seed, H, W):
    p1, p2 = synthetic.make_correlated_pyramids(seed, H, W)
    f1 = [t[None] for t in p1]
    f2 = [t[None] for t in p2]
    with torch.no_grad():
        corr, delta = net.forward_coarse_match(p1[4][None], p2[4][None], ksize=2)
        cm, cs = net.cal_coarse_matches(corr, delta, ksize=2, upsample=8, center=True)
        cm, cs = ref.utils.filter_coarse(cm, cs, 0.0, True)
        mid, midp = net.forward_fine_match(f1, f2, cm, 16, "center", net.regress_mid)
        fine, finep = net.forward_fine_match(f1, f2, mid, 16, "center", net.regress_fine)
    np.savez_compressed(os.path.join(GOLDEN, name), seed=seed, H=H, W=W, sd_seed=SD_SEED,
                        input_checksum=checksum(p1 + p2), coarse=np_(cm[0]), coarse_scores=np_(cs[0]),
                        mid=np_(mid[0]), mid_scores=np_(midp[0]), fine=np_(fine[0]), fine_scores=np_(finep[0]))
    print(name, "matches", cm[0].shape[0])


def case_estimate_matches(ref, name, seed, H, W, imsize):
    """Full reference entry point utils/eval/model_helper.py:64-109 on a synthetic image pair."""
    from PIL import Image
    ckpt = synthetic.make_checkpoint(SD_SEED)
    im1, im2 = synthetic.make_image_pair(seed, H, W)
    with tempfile.TemporaryDirectory() as td:
        torch.save(ckpt, os.path.join(td, "ckpt.pth"))
        Image.fromarray(im1).save(os.path.join(td, "1.png"))
        Image.fromarray(im2).save(os.path.join(td, "2.png"))
        net = ref.model_helper.load_model(os.path.join(td, "ckpt.pth"), method="patch2pix", lprint=lambda *a: None)
        res = {}
        for tag, kw in (("fine", dict(eval_type="fine", io_thres=0.25)),
                        ("coarse", dict(eval_type="coarse", ncn_thres=0.0))):
            m, s, c = ref.model_helper.estimate_matches(net, os.path.join(td, "1.png"), os.path.join(td, "2.png"),
                                                        ksize=2, imsize=imsize, **kw)
            res[tag + "_matches"], res[tag + "_scores"], res[tag + "_coarse"] = m, s, c
    np.savez_compressed(os.path.join(GOLDEN, name), seed=seed, H=H, W=W, imsize=(-1 if imsize is None else imsize), sd_seed=SD_SEED,
                        input_checksum=float(im1.astype(np.float64).sum() + im2.astype(np.float64).sum()), **res)
    print(name, "fine", res["fine_matches"].shape, res["fine_matches"].dtype, res["fine_scores"].dtype,
          "coarse", res["coarse_matches"].shape)


def case_full_size(ref, net, name, seed, H, W, ptmax):
    """BASELINE configuration (480x640, ksize 2, ptmax 400) through the unmodified reference: all coarse rows, the
    ptmax-sampled proposals (training-time option of networks/utils.py:55-63, global numpy RNG seeded with 0) and both
    regressors on them.  The 1.44 M-cell volume itself is not stored: a strided sample of corr4d and the histogram
    of the relocalisation codes stand in for it (the rows pin every argmax that is consumed)."""
    p1, p2 = synthetic.make_correlated_pyramids(seed, H, W)
    f1 = [t[None] for t in p1]
    f2 = [t[None] for t in p2]
    with torch.no_grad():
        corr, delta = net.forward_coarse_match(p1[4][None], p2[4][None], ksize=2)
        m, s = net.cal_coarse_matches(corr, delta, ksize=2, upsample=8, center=True)
        fm, fs = ref.utils.filter_coarse(m, s, 0.0, True)
        np.random.seed(0)
        cm, cs = ref.utils.filter_coarse(m, s, 0.0, True, ptmax=ptmax)
        mid, midp = net.forward_fine_match(f1, f2, cm, 16, "center", net.regress_mid)
        fine, finep = net.forward_fine_match(f1, f2, mid, 16, "center", net.regress_fine)
    code = ((delta[0] * 2 + delta[1]) * 2 + delta[2]) * 2 + delta[3]
    flat = corr[0, 0].reshape(-1)
    np.savez_compressed(os.path.join(GOLDEN, name), seed=seed, H=H, W=W, ptmax=ptmax, sd_seed=SD_SEED,
                        input_checksum=checksum(p1 + p2), all_matches=np_(m[0]).astype(np.int16), all_scores=np_(s[0]),
                        mutual_matches=np_(fm[0]).astype(np.int16), corr_sample_stride=997,
                        corr_sample=np_(flat[::997]), delta_hist=np.bincount(np_(code).reshape(-1), minlength=16),
                        proposals=np_(cm[0]).astype(np.int16), proposal_scores=np_(cs[0]),
                        mid=np_(mid[0]), mid_scores=np_(midp[0]), fine=np_(fine[0]), fine_scores=np_(finep[0]))
    print(name, "rows", m.shape[1], "mutual", fm[0].shape[0], "proposals", cm[0].shape[0])


def case_real_pair(ref, name, pair_dir, imsize, contrast=None):
    """The reference's estimate_matches (utils/eval/model_helper.py:64-109) on a real image pair of its examples/
    directory (copied to tests/golden/images), synthetic checkpoint in the reference's own schema.  `contrast`: the
    checkpoint variant whose backbone yields sparse, discriminative features (synthetic.contrast_shift); the shift the
    fixture was made with is stored in it."""
    extra = {}
    if contrast is not None:
        extra["contrast_shift"] = np_(synthetic.contrast_shift(synthetic.make_state_dict(SD_SEED), contrast))
        ckpt = synthetic.make_checkpoint(SD_SEED, contrast=torch.from_numpy(extra["contrast_shift"]))
    else:
        ckpt = synthetic.make_checkpoint(SD_SEED)
    im1, im2 = os.path.join(GOLDEN, "images", pair_dir, "1.jpg"), os.path.join(GOLDEN, "images", pair_dir, "2.jpg")
    with tempfile.TemporaryDirectory() as td:
        torch.save(ckpt, os.path.join(td, "ckpt.pth"))
        net = ref.model_helper.load_model(os.path.join(td, "ckpt.pth"), method="patch2pix", lprint=lambda *a: None)
        res = {}
        for tag, kw in (("fine", dict(eval_type="fine", io_thres=0.25)), ("coarse", dict(eval_type="coarse", ncn_thres=0.0))):
            m, s, c = ref.model_helper.estimate_matches(net, im1, im2, ksize=2, imsize=imsize, **kw)
            res[tag + "_matches"], res[tag + "_scores"], res[tag + "_coarse"] = m, s, c
        # the layer-3 features the reference's backbone produced on this CPU: lets a test tell backbone drift
        # (another CPU / MIOpen) from a difference in the matching path
        t1, _ = ref.model_helper.load_im_flexible(im1, 2, net.upsample, imsize=imsize)
        t2, _ = ref.model_helper.load_im_flexible(im2, 2, net.upsample, imsize=imsize)
        with torch.no_grad():
            feat = net.extract(t1[None], early_feat=True)
            # every coarse row before filter_coarse (patch2pix.py:240-248 without the filter): pins each argmax
            corr4d, delta4d = net.forward(t1[None], t2[None], ksize=2)
            rows, row_scores = net.cal_coarse_matches(corr4d, delta4d, ksize=2, upsample=net.upsample, center=True)
        extra["all_rows"] = np_(rows[0]).astype(np.int16)
        extra["all_scores"] = np_(row_scores[0])
    np.savez_compressed(os.path.join(GOLDEN, name), pair=pair_dir, imsize=(-1 if imsize is None else imsize), sd_seed=SD_SEED,
                        feat1_checksum=checksum([feat]), feat1_shape=np.array(feat.shape), **extra, **res)
    print(name, "fine", res["fine_matches"].shape, "coarse", res["coarse_matches"].shape, "feat", tuple(feat.shape))


SD_SEED = 0


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    import contextlib
    import io
    ref = load_reference()
    sd = synthetic.make_state_dict(SD_SEED)
    net = build_reference_net(sd, synthetic.default_regressor_config())
    np.savez_compressed(os.path.join(GOLDEN, "weights_checksum"), sd_seed=SD_SEED,
                        checksum=checksum([v for v in sd.values() if v.is_floating_point()]))
    case_coarse(ref, net, sd, "coarse_64x96_k2", 31, 64, 96, 2)
    case_coarse(ref, net, sd, "coarse_96x64_k2", 32, 96, 64, 2)
    case_coarse(ref, net, sd, "coarse_48x64_k1", 33, 48, 64, 1)
    case_coarse(ref, net, sd, "coarse_128x160_k2", 34, 128, 160, 2)
    case_fine(ref, net, sd, "fine_48x64", 21, 48, 64, 24)
    case_fine(ref, net, sd, "fine_96x128", 23, 96, 128, 40)
    case_predict_fine(ref, net, sd, "predict_fine_128x160", 41, 128, 160)
    case_predict_fine(ref, net, sd, "predict_fine_192x256", 42, 192, 256)
    with contextlib.redirect_stdout(io.StringIO()):
        pass
    case_estimate_matches(ref, "estimate_matches_240x320", 51, 240, 320, None)
    case_estimate_matches(ref, "estimate_matches_imsize256", 52, 300, 400, 256)
    new_cases(ref, net)
    round3_cases(ref)


def new_cases(ref, net):
    """Round-2 fixtures (kept separate so that `python -m oracle.make_golden --new` leaves the older files alone)."""
    case_full_size(ref, net, "full_480x640", 77, 480, 640, 400)
    case_real_pair(ref, "real_pair_1", "pair_1", None)
    case_real_pair(ref, "real_pair_2", "pair_2", 640)
    case_real_pair(ref, "real_pair_3", "pair_3", 1024)


def round3_cases(ref):
    """Round-3 fixtures: the example photographs with the contrast checkpoint (`python -m oracle.make_golden --r3`)."""
    case_real_pair(ref, "real_pair_1_contrast", "pair_1", None, contrast=4.0)
    case_real_pair(ref, "real_pair_2_contrast", "pair_2", 640, contrast=4.0)
    case_real_pair(ref, "real_pair_3_contrast", "pair_3", 1024, contrast=4.0)


if __name__ == "__main__":
    if "--r3" in sys.argv:
        round3_cases(load_reference())
    elif "--new" in sys.argv:
        os.makedirs(GOLDEN, exist_ok=True)
        _sd = synthetic.make_state_dict(SD_SEED)
        new_cases(load_reference(), build_reference_net(_sd, synthetic.default_regressor_config()))
    else:
        main()
