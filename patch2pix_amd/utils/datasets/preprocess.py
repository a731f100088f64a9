"""Image loading for matching -- role of reference utils/datasets/preprocess.py:32-60,83-91.

`load_im_tensor` (reference :7-30) is the loader of `refine_matches`: optional down-scaling to `imsize` with plain
rounding (no multiple-of-16 constraint), a grey copy for third-party coarse matchers, batch axis, on the device.
Host side (PIL) like the reference; the output feeds the backbone.  Target size = the original
size scaled so that max(w,h) == imsize (never up-sampled) and rounded *down* to a multiple of
upsample*k_size; bicubic resize; /255; ImageNet mean/std."""
import numpy as np
import torch
from PIL import Image

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def _normalised(img):
    arr = np.array(img, dtype=np.float32).transpose(2, 0, 1)
    arr /= 255.0
    return (torch.from_numpy(arr) - torch.from_numpy(_MEAN)) / torch.from_numpy(_STD)


_LUT = {}


def normalise_pixels(pixels):
    """uint8 [B,H,W,3] (any device) -> float32 [B,3,H,W], bit-identical to `_normalised`: the normalised value depends
    only on (channel, byte), so it is looked up in a [3,256] table computed ON THE HOST with the reference's three
    operations (/255, -mean, /std).  (The same arithmetic on the GPU is not bit-identical: its division by a constant is
    a multiplication by the reciprocal.)  Lets the entry points upload uint8 pixels -- a quarter of the bytes -- and
    skip the float conversion on the host."""
    dev = pixels.device
    if dev not in _LUT:
        v = np.arange(256, dtype=np.float32)[None, :].repeat(3, 0).reshape(3, 256, 1)
        v /= 255.0
        _LUT[dev] = ((torch.from_numpy(v) - torch.from_numpy(_MEAN)) / torch.from_numpy(_STD)).reshape(3, 256).to(dev)
    lut = _LUT[dev]
    idx = pixels.permute(0, 3, 1, 2).long()                                  # [B,3,H,W]
    return torch.gather(lut[None, :, :, None].expand(idx.shape[0], 3, 256, idx.shape[3]), 2, idx)


def load_im_pixels(im_path, k_size=2, upsample=16, imsize=None):
    """`load_im_flexible` up to (not including) the normalisation: -> (uint8 tensor [H,W,3], (wo/wt, ho/ht))."""
    img = Image.open(im_path).convert("RGB")
    wo, ho = img.width, img.height
    if not (imsize and imsize > 0) or imsize > max(wo, ho):
        imsize = max(wo, ho)
    wt, ht = cal_rescale_size(imsize, wo, ho, k_size=k_size, scale_factor=1.0 / upsample)
    img = img.resize((wt, ht), Image.BICUBIC)
    return torch.from_numpy(np.array(img, dtype=np.uint8)), (wo / wt, ho / ht)


def load_im_tensor(im_path, device, imsize=None, with_gray=True):
    """-> (im [1,3,H,W] normalised, gray [1,1,H,W] in [0,1], (wo/wt, ho/ht)) or (im, scale) without grey."""
    im = Image.open(im_path)
    wt, ht = wo, ho = im.width, im.height
    if imsize and max(wo, ho) > imsize and imsize > 0:
        s = imsize / max(wo, ho)
        ht, wt = int(round(ho * s)), int(round(wo * s))
        im = im.resize((wt, ht), Image.BICUBIC)
    scale = (wo / wt, ho / ht)
    gray = None
    if with_gray:
        g = np.array(im.convert("L"), dtype=np.float32) / 255.0
        gray = torch.from_numpy(g)[None, None].to(device)
    rgb = im if im.mode == "RGB" else im.convert("RGB")
    t = _normalised(rgb).unsqueeze(0).to(device)
    if with_gray:
        return t, gray, scale
    return t, scale


def cal_rescale_size(image_size, w, h, k_size=2, scale_factor=1 / 16, no_print=True):
    ratio = max(w, h) / image_size
    wt = int(np.floor(w / ratio * scale_factor / k_size) / scale_factor * k_size)
    ht = int(np.floor(h / ratio * scale_factor / k_size) / scale_factor * k_size)
    return wt, ht


def load_im_flexible(im_path, k_size=2, upsample=16, imsize=None, crop_square=False):
    img = Image.open(im_path).convert("RGB")
    wo, ho = img.width, img.height
    if not (imsize and imsize > 0) or imsize > max(wo, ho):
        imsize = max(wo, ho)
    wt, ht = cal_rescale_size(imsize, wo, ho, k_size=k_size, scale_factor=1.0 / upsample)
    img = img.resize((wt, ht), Image.BICUBIC)
    t = _normalised(img)
    if crop_square:
        t = t[:, :t.shape[2], :]
    return t, (wo / wt, ho / ht)
