"""Image loading for matching -- role of reference utils/datasets/preprocess.py:32-60,83-91.

Host side (PIL) like the reference; the output feeds the backbone.  Target size = the original
size scaled so that max(w,h) == imsize (never up-sampled) and rounded *down* to a multiple of
upsample*k_size; bicubic resize; /255; ImageNet mean/std."""
import numpy as np
import torch
from PIL import Image

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


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
    arr = np.array(img, dtype=np.float32).transpose(2, 0, 1)
    arr /= 255.0
    t = torch.from_numpy(arr)
    t = (t - torch.from_numpy(_MEAN)) / torch.from_numpy(_STD)
    if crop_square:
        t = t[:, :t.shape[2], :]
    return t, (wo / wt, ho / ht)
