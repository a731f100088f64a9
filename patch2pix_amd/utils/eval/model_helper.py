"""Entry points of the matching path with the reference's names and contracts.

Role of reference utils/eval/model_helper.py: `load_model(ckpt_path, method, lprint)`,
`estimate_matches(net, im1, im2, ksize, ncn_thres, mutual, io_thres, eval_type, imsize)`,
`refine_matches(im1_path, im2_path, net, coarse_matcher, io_thres, imsize, coarse_only)` and the two matcher
factories `init_patch2pix_matcher(args)` / `init_ncn_matcher(args)`.  Signatures, defaults and return layouts are the
reference's (what image-matching-toolbox binds to); the bodies are organised around the HIP library underneath.

Return contract of `estimate_matches` (reference :64-109):
    matches         float64 [M,4]  (x1, y1, x2, y2) in ORIGINAL image pixels
    scores          float32 [M]
    coarse_matches  float64 [M,4]  the coarse match each row was refined from (same as matches for eval_type='coarse')
"""
from argparse import Namespace
from functools import partial

import numpy as np
import torch

from ..common.setup_helper import load_weights
from ..datasets.preprocess import load_im_flexible, load_im_pixels, load_im_tensor, normalise_pixels
from ...networks.patch2pix import Patch2Pix
from ... import ops

_SILENT = lambda *a, **k: None


# ------------------------------------------------------------------------------------------ model construction
def _base_config(device):
    """Inference configuration of reference :32-39 (change_stride on, proposals in chunks of 1200)."""
    return Namespace(device=device, training=False, backbone="ResNet34", change_stride=True, regr_batch=1200,
                     feat_idx=None, regressor_config=None, weights_dict=None)


def _configure_patch2pix(config, ckpt):
    """A full checkpoint (utils/train/helper.py:10-20) carries its own architecture description."""
    for field in ("backbone", "feat_idx", "regressor_config"):
        setattr(config, field, ckpt[field])
    config.weights_dict = ckpt["state_dict"]
    config.regressor_config.panc = 1          # evaluation never expands anchors (reference :46)


def _configure_ncn(config, ckpt):
    """NCNet-only checkpoints are a bare state_dict or {'state_dict': ...} (reference :53-57)."""
    config.weights_dict = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt


def load_model(ckpt_path, method="patch2pix", lprint=print):
    """Build the network on the current HIP device and load a reference-format checkpoint.
    `ckpt_path` may also be an already loaded checkpoint dict."""
    if not torch.cuda.is_available():
        raise RuntimeError("patch2pix_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    device = torch.device("cuda", torch.cuda.current_device())
    in_memory = isinstance(ckpt_path, dict)
    ckpt = ckpt_path if in_memory else load_weights(ckpt_path, device)
    config = _base_config(device)
    lprint("\nLoad model method:{} ".format(method))
    if "patch2pix" in method:
        _configure_patch2pix(config, ckpt)
        origin = "<in-memory checkpoint>" if in_memory else ckpt_path
        lprint(f"Ckpt:{origin} epochs:{ckpt['last_epoch'] + 1}" if "last_epoch" in ckpt else f"Ckpt:{origin}")
    elif "nc" in method:
        _configure_ncn(config, ckpt)
        lprint("Load pretrained weights: {}".format("<in-memory checkpoint>" if in_memory else ckpt_path))
    else:
        lprint("Wrong method name.")
    return Patch2Pix(config).eval()


def init_patch2pix_matcher(args):
    net = load_model(args.ckpt, method="patch2pix")
    return partial(_call_fine, net, args)


def init_ncn_matcher(args):
    net = load_model(args.ckpt, method="nc")
    return partial(_call_coarse, net, args)


def _call_fine(net, args, imq, imr):
    return estimate_matches(net, imq, imr, ksize=args.ksize, io_thres=args.io_thres, eval_type="fine", imsize=args.imsize)


def _call_coarse(net, args, imq, imr):
    return estimate_matches(net, imq, imr, ksize=args.ksize, ncn_thres=args.ncn_thres, eval_type="coarse",
                            imsize=args.imsize)


# ------------------------------------------------------------------------------------------ matching
_decoders = None        # two threads: the second image of a pair is decoded while the first one is (PIL releases the GIL)


def _load_pair(net, im1, im2, ksize, imsize):
    """Both images as [1,3,H,W] tensors on the device + the (1,4) factors back to original pixels."""
    global _decoders
    if _decoders is None:
        from concurrent.futures import ThreadPoolExecutor
        _decoders = ThreadPoolExecutor(max_workers=2, thread_name_prefix="p2p-decode")
    # PIL decode + bicubic resize on the host like the reference (load_im_flexible); the uint8 pixels go to the device and
    # are normalised there (bit-identical, a quarter of the upload)
    second = _decoders.submit(load_im_pixels, im2, ksize, net.upsample, imsize=imsize)
    loaded = [load_im_pixels(im1, ksize, net.upsample, imsize=imsize), None]
    loaded[1] = second.result()
    tensors, factors = [], ()
    for pixels, scale_wh in loaded:
        tensors.append(normalise_pixels(pixels.unsqueeze(0).to(net.device)))
        factors += tuple(scale_wh)
    return tensors[0], tensors[1], np.array([factors])


def _host(t):
    return t.detach().cpu().numpy()


def estimate_matches(net, im1, im2, ksize=2, ncn_thres=0.0, mutual=True, io_thres=0.25, eval_type="fine",
                     imsize=None):
    """Match one image pair (batch size 1, like the reference)."""
    t1, t2, to_original = _load_pair(net, im1, im2, ksize, imsize)

    if eval_type == "coarse":
        with torch.no_grad():
            rows, row_scores = net.predict_coarse(t1, t2, ksize=ksize, ncn_thres=ncn_thres, mutual=mutual)
        pixels = to_original * _host(rows[0])
        return pixels, _host(row_scores[0]), pixels

    if eval_type != "fine":
        raise ValueError(f"eval_type must be 'coarse' or 'fine', got {eval_type!r}")
    with torch.no_grad():
        refined, confidence, proposals = net.predict_fine(t1, t2, ksize=ksize, ncn_thres=ncn_thres, mutual=mutual)
    refined, confidence, proposals = _host(refined[0]), _host(confidence[0]), _host(proposals[0])

    # keep the confident matches; when none clears the threshold every match is returned (reference :97-105)
    confident = np.flatnonzero(confidence > io_thres)
    if confident.size:
        refined, confidence, proposals = refined[confident], confidence[confident], proposals[confident]
    return to_original * refined, confidence, to_original * proposals


def estimate_matches_device(net, im1, im2, ksize=2, ncn_thres=0.0, mutual=True, io_thres=0.25, imsize=None):
    """estimate_matches(eval_type='fine') with NOTHING between the image tensors and the result on the host
    (non-reference entry point): coarse stage, filter_coarse, both regressors and the io_thres / scaling tail of
    model_helper.py:92-109 all run on the device; one device-to-host copy at the end.  Same return triple."""
    t1, t2, to_original = _load_pair(net, im1, im2, ksize, imsize)
    with torch.no_grad():
        fine, scores, coarse, counts = net.predict_fine_device(net.extract.pyramid(t1), net.extract.pyramid(t2), ksize=ksize,
                                                               ncn_thres=ncn_thres, mutual=mutual)
        m, s, c, n = ops.match_tail_batch(fine, scores, coarse, counts, to_original, io_thres)
    k = int(n[0])
    if k < 0:          # a coordinate outside the device filter's packed key: the reference path
        return estimate_matches(net, im1, im2, ksize, ncn_thres, mutual, io_thres, "fine", imsize)
    return _host(m[0, :k]), _host(s[0, :k]), _host(c[0, :k])


def refine_matches(im1_path, im2_path, net, coarse_matcher, io_thres=0.0, imsize=None, coarse_only=False):
    """Refine the matches of a third-party coarse matcher (reference :111-127): `coarse_matcher(grey1, grey2)` gets
    the two grey images [1,1,H,W] and returns [N,4] pixel matches in the loaded images' frame."""
    im1, grey1, sc1 = load_im_tensor(im1_path, net.device, imsize, with_gray=True)
    im2, grey2, sc2 = load_im_tensor(im2_path, net.device, imsize, with_gray=True)
    to_original = np.array([sc1 + sc2])
    coarse = coarse_matcher(grey1, grey2)
    if coarse_only:
        return to_original * coarse.cpu().data.numpy(), None, None
    with torch.no_grad():
        refined, scores, coarse = net.refine_matches(im1, im2, coarse, io_thres)
    return to_original * refined, scores, to_original * coarse
