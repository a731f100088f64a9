"""Seeded synthetic checkpoints and inputs (the pretrained weights are not redistributable
and there is no network: reference pretrained/download.sh:1-5).

`make_state_dict(seed)` produces a state_dict with exactly the key names / shapes / dtypes of
a reference checkpoint's `state_dict` (utils/train/helper.py:10-17; 276 entries minus the
never-used `extract.layer4.*`), so the same dict can be loaded by the reference model (as a
checkpoint) and by this package.  The distributions follow the reference's own initialiser
`xavier_init_func_` (networks/modules.py:154-166) with BatchNorm statistics randomised so
that the folded BN is not a no-op, and the NCNet filters biased towards
a positive centre tap so that the coarse stage yields a useful number of mutual matches.
"""
import math
from argparse import Namespace

import torch


def default_regressor_config():
    """Defaults of train_patch2pix.py:46-54 as stored in a checkpoint ('regressor_config')."""
    return Namespace(conv_dims=[512, 512], conv_kers=[3, 3], conv_strs=[2, 1], fc_dims=[512, 256],
                     feat_comb="pre", psize=[16, 16], pshift=8, panc=1, shared=False)


def _xavier(gen, *shape):
    rf = 1
    for s in shape[2:]:
        rf *= s
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(*shape, generator=gen) * 2 - 1) * a


def _bn(gen, sd, prefix, n, randomise=True):
    if randomise:
        sd[prefix + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=gen)
        sd[prefix + ".bias"] = 0.1 * torch.randn(n, generator=gen)
        sd[prefix + ".running_mean"] = 0.1 * torch.randn(n, generator=gen)
        sd[prefix + ".running_var"] = 0.5 + torch.rand(n, generator=gen)
    else:
        sd[prefix + ".weight"] = torch.ones(n)
        sd[prefix + ".bias"] = torch.zeros(n)
        sd[prefix + ".running_mean"] = torch.zeros(n)
        sd[prefix + ".running_var"] = torch.ones(n)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def _backbone(gen, sd):
    def conv(name, cout, cin, k):
        std = math.sqrt(2.0 / (cin * k * k))
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=gen) * std

    conv("extract.conv1", 64, 3, 7)
    _bn(gen, sd, "extract.bn1", 64, randomise=False)
    cin = 64
    for lname, planes, blocks in (("layer1", 64, 3), ("layer2", 128, 4), ("layer3", 256, 6)):
        for b in range(blocks):
            p = f"extract.{lname}.{b}"
            conv(p + ".conv1", planes, cin if b == 0 else planes, 3)
            _bn(gen, sd, p + ".bn1", planes, randomise=False)
            conv(p + ".conv2", planes, planes, 3)
            _bn(gen, sd, p + ".bn2", planes, randomise=False)
            sd[p + ".bn2.weight"] *= 0.5          # keep the residual stream from blowing up
            if b == 0 and cin != planes:
                conv(p + ".downsample.0", planes, cin, 1)
                _bn(gen, sd, p + ".downsample.1", planes, randomise=False)
        cin = planes


def _ncn(gen, sd, peaky):
    # natural layout [c_out, c_in, 3,3,3,3]; stored layout [3, c_out, c_in, 3,3,3] (conv4d.py:119-120)
    w1 = _xavier(gen, 16, 1, 3, 3, 3, 3)
    w2 = _xavier(gen, 1, 16, 3, 3, 3, 3)
    if peaky:
        w1 = w1 * 0.25
        w1[:, 0, 1, 1, 1, 1] = 0.5 + 0.5 * torch.rand(16, generator=gen)
        w2 = w2.abs() * 0.25
        w2[0, :, 1, 1, 1, 1] = 0.5 + 0.5 * torch.rand(16, generator=gen)
    sd["ncn.conv.0.weight"] = w1.permute(2, 0, 1, 3, 4, 5).contiguous()
    sd["ncn.conv.0.bias"] = 0.01 * torch.randn(16, generator=gen)
    sd["ncn.conv.2.weight"] = w2.permute(2, 0, 1, 3, 4, 5).contiguous()
    sd["ncn.conv.2.bias"] = 0.01 * torch.randn(1, generator=gen)


def _regressor(gen, sd, prefix, feat_dim=259):
    sd[prefix + ".conv.0.weight"] = _xavier(gen, 512, 2 * feat_dim, 3, 3)
    _bn(gen, sd, prefix + ".conv.1", 512)
    sd[prefix + ".conv.2.weight"] = _xavier(gen, 512, 512, 3, 3)
    _bn(gen, sd, prefix + ".conv.3", 512)
    sd[prefix + ".fc.0.weight"] = _xavier(gen, 512, 512)
    sd[prefix + ".fc.0.bias"] = 0.05 * torch.randn(512, generator=gen)
    _bn(gen, sd, prefix + ".fc.1", 512)
    sd[prefix + ".fc.3.weight"] = _xavier(gen, 256, 512)
    sd[prefix + ".fc.3.bias"] = 0.05 * torch.randn(256, generator=gen)
    _bn(gen, sd, prefix + ".fc.4", 256)
    sd[prefix + ".fc.6.weight"] = 8.0 * _xavier(gen, 5, 256)
    # Centre the five outputs in the sensitive range of 16*tanh(relu(.)) / sigmoid(.): with random
    # features the pooled activations barely vary between proposals, so an un-calibrated bias
    # saturates tanh and makes every proposal regress to the same corner.
    target = torch.tensor([0.45, 0.5, 0.55, 0.4, 0.2]) + 0.05 * torch.randn(5, generator=gen)
    sd[prefix + ".fc.6.bias"] = torch.zeros(5)
    sd[prefix + ".fc.6.bias"] = target - _pilot_outputs(gen, sd, prefix, feat_dim).mean(dim=0)


def _pilot_outputs(gen, sd, prefix, feat_dim, n=6):
    """Raw regressor outputs on a few random unit-norm patches (plain torch; calibration only)."""
    import torch.nn.functional as F

    def bn(x, name, shape):
        g = lambda k: sd[f"{prefix}.{name}.{k}"].view(shape)
        return (x - g("running_mean")) / torch.sqrt(g("running_var") + 1e-5) * g("weight") + g("bias")

    halves = []
    for _ in range(2):
        t = torch.relu(torch.randn(n, feat_dim, 16, 16, generator=gen) + 0.3)
        halves.append(t / (t.pow(2).sum(dim=1, keepdim=True) + 1e-6).sqrt())
    z = torch.cat(halves, dim=1)
    u = bn(F.conv2d(z, sd[prefix + ".conv.0.weight"], stride=2, padding=1), "conv.1", (1, -1, 1, 1))
    u = bn(F.conv2d(u, sd[prefix + ".conv.2.weight"], padding=1), "conv.3", (1, -1, 1, 1))
    v = torch.relu(u).amax(dim=(2, 3))
    v = torch.relu(bn(F.linear(v, sd[prefix + ".fc.0.weight"], sd[prefix + ".fc.0.bias"]), "fc.1", (1, -1)))
    v = torch.relu(bn(F.linear(v, sd[prefix + ".fc.3.weight"], sd[prefix + ".fc.3.bias"]), "fc.4", (1, -1)))
    return F.linear(v, sd[prefix + ".fc.6.weight"], sd[prefix + ".fc.6.bias"])


def contrast_shift(sd, kappa=4.0, images=2, height=240, width=320):
    """Per-channel shift [256] that makes the layer-3 features of the random-init backbone SPARSE: mean + kappa *
    standard deviation of the last block's pre-ReLU output over a few synthetic images, rounded to 1/64.  With the plain
    He-init backbone every feature vector of a photograph is dense and positive (mean cosine between two cells 0.95),
    the correlation volume is nearly flat and most argmaxes of the coarse stage are undecidable in fp32; subtracting
    the shift in the last BatchNorm (bias of `extract.layer3.5.bn2`) leaves ~20 % of the channels active per cell
    (mean cosine 0.2-0.4).  Calibration runs the backbone on the CPU, so fixtures store the shift they were made with
    (`make_state_dict(contrast=<tensor>)`) instead of re-deriving it on another machine."""
    import torch.nn.functional as F
    from ..networks import resnet

    net = resnet.ResNet34()
    net.change_stride("layer3")
    net.load_state_dict({k[len("extract."):]: v for k, v in sd.items() if k.startswith("extract.")}, strict=False)
    net.eval()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    blk, cols = net.layer3[5], []
    with torch.no_grad():
        for s in range(images):
            for im in make_image_pair(900 + s, height, width):
                t = (torch.from_numpy(im).permute(2, 0, 1).float() / 255 - mean) / std
                x = F.relu(net.bn1(net.conv1(t[None])))
                x = net.layer2(net.layer1(F.max_pool2d(x, 3, 2, 1)))
                for b in net.layer3[:5]:
                    x = b(x)
                z = blk.bn2(blk.conv2(F.relu(blk.bn1(blk.conv1(x))))) + x
                cols.append(z[0].reshape(z.shape[1], -1))
    z = torch.cat(cols, dim=1)
    return torch.round((z.mean(dim=1) + kappa * z.std(dim=1)) * 64) / 64


def make_state_dict(seed=0, peaky_ncn=True, backbone=True, contrast=None):
    """Reference-layout state_dict (fp32 CPU tensors) from one integer seed.  `contrast`: None (plain backbone), a
    float kappa (calibrate `contrast_shift` here) or the [256] shift tensor of a fixture."""
    gen = torch.Generator().manual_seed(int(seed))
    sd = {}
    if backbone:
        _backbone(gen, sd)
        if contrast is not None:
            shift = contrast_shift(sd, float(contrast)) if isinstance(contrast, (int, float)) else torch.as_tensor(contrast).float()
            sd["extract.layer3.5.bn2.bias"] = sd["extract.layer3.5.bn2.bias"] - shift
    _ncn(gen, sd, peaky_ncn)
    _regressor(gen, sd, "regress_mid")
    _regressor(gen, sd, "regress_fine")
    return sd


def make_checkpoint(seed=0, **kw):
    """A dict with the reference checkpoint schema (utils/train/helper.py:10-20)."""
    return {"last_epoch": 0, "best_vals": None, "backbone": "ResNet34", "feat_idx": [0, 1, 2, 3],
            "change_stride": True, "regressor_config": default_regressor_config(),
            "state_dict": make_state_dict(seed, **kw), "optim": None}


def make_pyramid(seed, height, width, device="cpu"):
    """Synthetic post-ReLU-like feature pyramid of one image, shapes as reference
    resnet.py:138-157 with change_stride: [3,H,W],[64,H/2,W/2],[64,H/4,W/4],[128,H/8,W/8],[256,H/8,W/8]."""
    gen = torch.Generator().manual_seed(int(seed))
    dims = ((3, 1, False), (64, 2, True), (64, 4, True), (128, 8, True), (256, 8, True))
    out = []
    for c, ds, relu in dims:
        t = torch.randn(c, height // ds, width // ds, generator=gen)
        if relu:
            t = torch.relu(t + 0.3)
        out.append(t.to(device))
    return out


def make_correlated_pyramids(seed, height, width, shift=(8, 16), noise=0.25, device="cpu"):
    """Two pyramids where image 2 is image 1 translated by `shift` pixels (multiples of 8)
    plus noise, so that the coarse stage finds many mutual matches."""
    p1 = make_pyramid(seed, height, width)
    gen = torch.Generator().manual_seed(int(seed) + 7919)
    p2 = []
    for t, ds in zip(p1, (1, 2, 4, 8, 8)):
        r = torch.roll(t, shifts=(shift[0] // ds, shift[1] // ds), dims=(1, 2))
        r = r + noise * torch.randn(r.shape, generator=gen)
        if ds > 1:
            r = torch.relu(r)
        p2.append(r.to(device))
    return [t.to(device) for t in p1], p2


def make_correlated_pyramids_device(pair_id, height, width, device, shift=(8, 16), noise=0.25):
    """The same construction as `make_correlated_pyramids`, generated ON the device from a generator seeded with the
    pair id (another random stream than the CPU version: used where no CPU counterpart of the inputs is needed -- the
    pair stream of bench.py --pairs N, whose 10 000 pairs cannot all be resident)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(pair_id))
    dims = ((3, 1, False), (64, 2, True), (64, 4, True), (128, 8, True), (256, 8, True))
    p1, p2 = [], []
    for c, ds, relu in dims:
        t = torch.randn(c, height // ds, width // ds, generator=gen, device=device)
        if relu:
            t = torch.relu(t + 0.3)
        r = torch.roll(t, shifts=(shift[0] // ds, shift[1] // ds), dims=(1, 2))
        r = r + noise * torch.randn(r.shape, generator=gen, device=device)
        if ds > 1:
            r = torch.relu(r)
        p1.append(t)
        p2.append(r)
    return p1, p2


def make_image_pair(seed, height=480, width=640, shift=(16, 24)):
    """Two uint8 RGB images [H,W,3]: a multi-scale random texture and a shifted, re-noised copy."""
    import numpy as np
    from PIL import Image

    rng = np.random.RandomState(int(seed))
    big_h, big_w = height + 2 * abs(shift[0]) + 8, width + 2 * abs(shift[1]) + 8
    canvas = np.zeros((big_h, big_w, 3), np.float32)
    for cells, amp in ((6, 1.0), (17, 0.7), (41, 0.5), (97, 0.35)):
        low = rng.rand(cells, int(cells * big_w / big_h) + 1, 3).astype(np.float32)
        img = Image.fromarray((low * 255).astype(np.uint8)).resize((big_w, big_h), Image.BICUBIC)
        canvas += amp * (np.asarray(img, np.float32) / 255.0 - 0.5)
    canvas = (canvas - canvas.min()) / (canvas.max() - canvas.min())
    oy, ox = abs(shift[0]) + 4, abs(shift[1]) + 4
    im1 = canvas[oy:oy + height, ox:ox + width]
    im2 = canvas[oy - shift[0]:oy - shift[0] + height, ox - shift[1]:ox - shift[1] + width]
    im2 = np.clip(im2 * 0.9 + 0.05 + 0.02 * rng.randn(*im2.shape).astype(np.float32), 0, 1)
    return (im1 * 255).astype(np.uint8), (im2 * 255).astype(np.uint8)
