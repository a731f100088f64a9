"""Drop-in for reference utils/eval/model_helper.py: `load_model`, `estimate_matches`,
`init_patch2pix_matcher`, `init_ncn_matcher` with identical signatures and return layouts
(float64 [M,4] matches in original-image pixels, float32 [M] scores, float64 [M,4] coarse)."""
from argparse import Namespace

import numpy as np
import torch

from ..common.setup_helper import load_weights
from ..datasets.preprocess import load_im_flexible
from ...networks.patch2pix import Patch2Pix


def init_patch2pix_matcher(args):
    net = load_model(args.ckpt, method="patch2pix")
    return lambda imq, imr: estimate_matches(net, imq, imr, ksize=args.ksize, io_thres=args.io_thres,
                                             eval_type="fine", imsize=args.imsize)


def init_ncn_matcher(args):
    net = load_model(args.ckpt, method="nc")
    return lambda imq, imr: estimate_matches(net, imq, imr, ksize=args.ksize, ncn_thres=args.ncn_thres,
                                             eval_type="coarse", imsize=args.imsize)


def load_model(ckpt_path, method="patch2pix", lprint=print):
    """model_helper.py:28-62.  The device is cuda:<current> -- this package has no CPU path."""
    if not torch.cuda.is_available():
        raise RuntimeError("patch2pix_amd needs an MI355X (torch.cuda.is_available() is False)")
    device = torch.device("cuda:{}".format(torch.cuda.current_device()))
    ckpt = ckpt_path if isinstance(ckpt_path, dict) else load_weights(ckpt_path, device)
    config = Namespace(training=False, device=device, regr_batch=1200, backbone="ResNet34", feat_idx=None,
                       weights_dict=None, regressor_config=None, change_stride=True)
    lprint("\nLoad model method:{} ".format(method))
    if "patch2pix" in method:
        config.backbone = ckpt["backbone"]
        config.feat_idx = ckpt["feat_idx"]
        config.weights_dict = ckpt["state_dict"]
        config.regressor_config = ckpt["regressor_config"]
        config.regressor_config.panc = 1          # evaluation always uses panc 1 (model_helper.py:46)
        if "last_epoch" in ckpt:
            lprint(f"Ckpt:{ckpt_path if not isinstance(ckpt_path, dict) else '<dict>'} epochs:{ckpt['last_epoch'] + 1}")
    elif "nc" in method:
        if isinstance(ckpt, dict) and "state_dict" in ckpt:
            ckpt = ckpt["state_dict"]
        config.weights_dict = ckpt
    else:
        lprint("Wrong method name.")
    net = Patch2Pix(config)
    net.eval()
    return net


def estimate_matches(net, im1, im2, ksize=2, ncn_thres=0.0, mutual=True, io_thres=0.25, eval_type="fine",
                     imsize=None):
    """model_helper.py:64-109 (batch size 1)."""
    im1, sc1 = load_im_flexible(im1, ksize, net.upsample, imsize=imsize)
    im2, sc2 = load_im_flexible(im2, ksize, net.upsample, imsize=imsize)
    upscale = np.array([sc1 + sc2])
    im1 = im1.unsqueeze(0).to(net.device)
    im2 = im2.unsqueeze(0).to(net.device)

    if eval_type == "coarse":
        with torch.no_grad():
            coarse_matches, scores = net.predict_coarse(im1, im2, ksize=ksize, ncn_thres=ncn_thres, mutual=mutual)
        matches = upscale * coarse_matches[0].cpu().data.numpy()
        return matches, scores[0].cpu().data.numpy(), matches

    if eval_type == "fine":
        with torch.no_grad():
            fine_matches, fine_scores, coarse_matches = net.predict_fine(im1, im2, ksize=ksize, ncn_thres=ncn_thres,
                                                                         mutual=mutual)
        coarse_matches = coarse_matches[0].cpu().data.numpy()
        fine_matches = fine_matches[0].cpu().data.numpy()
        fine_scores = fine_scores[0].cpu().data.numpy()

    pos_ids = np.where(fine_scores > io_thres)[0]
    if len(pos_ids) > 0:
        coarse_matches = coarse_matches[pos_ids]
        matches = fine_matches[pos_ids]
        scores = fine_scores[pos_ids]
    else:
        matches, scores = fine_matches, fine_scores
    return upscale * matches, scores, upscale * coarse_matches
