"""Feature-pyramid producer for the matching hot path.

Role of reference networks/resnet.py:125-173 (ResNet34 truncated after layer3, with the
layer3 stride patch of `change_stride`), SURVEY.md section 8 row f1.  The module keeps the torch
parameters (checkpoints load unchanged); on the GPU in eval mode the 32 convolutions of
stem and of layer1..layer3 run as the library's fp32-equivalent fp16 MFMA kernels (csrc/backbone.hip;
NHWC activations between the layers, NCHW pyramid levels out), the max-pool as its own kernel.
`P2P_BACKBONE=miopen` selects the all-PyTorch path (the round-1/2 producer).  Parameter names follow the torchvision ResNet convention so that
reference checkpoints (`extract.*` keys, utils/train/helper.py:10-17) load unchanged;
`layer4.*` keys of a checkpoint are never used by the path (reference
networks/patch2pix.py:72-74 freezes them as "never used") and are skipped on load.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# (planes, blocks, stride of first block) for ResNet34 up to layer3
_STAGES = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2))


class _Basic(nn.Module):
    """Two 3x3 conv+BN with identity / projected skip."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + skip, inplace=True)


class ResNet34(nn.Module):
    """conv1/bn1 + layer1..layer3 of ResNet34; `pyramid()` returns the 5 maps the path reads."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for name, planes, blocks, stride in _STAGES:
            seq = [_Basic(cin, planes, stride)] + [_Basic(planes, planes, 1) for _ in range(blocks - 1)]
            setattr(self, name, nn.Sequential(*seq))
            cin = planes

    def change_stride(self, target="layer3"):
        """Make `target`'s first block stride-1 (reference resnet.py:169-173)."""
        blk = getattr(self, target)[0]
        for conv in (blk.conv1, blk.conv2, blk.downsample[0]):
            conv.stride = (1, 1)

    def __getstate__(self):
        """copy.deepcopy / pickling: the packed device-side weights are a cache, not part of the module."""
        state = self.__dict__.copy()
        state.pop("_hip_trunk_cache", None)
        state.pop("_hip_sig", None)
        return state

    def _hip_signature(self, device):
        """(device, identity, version counter and storage of every tensor the packed weights were made from): changes with
        load_state_dict / in-place edits (version), .to() / .data assignment (storage) and replaced Parameter objects
        (identity).  Walks the known structure through the modules' own dictionaries: the generic parameters() /
        buffers() iterators cost a millisecond per call, more than launching the 36 kernels."""
        sig = [str(device)]
        pairs = [(self.conv1, self.bn1)]
        for name, _, _, _ in _STAGES:
            for blk in self._modules[name]._modules.values():
                m = blk._modules
                pairs += [(m["conv1"], m["bn1"]), (m["conv2"], m["bn2"])]
                if m.get("downsample") is not None:
                    d = m["downsample"]._modules
                    pairs.append((d["0"], d["1"]))
        for conv, bn in pairs:
            for t in (conv._parameters["weight"], bn._parameters["weight"], bn._parameters["bias"],
                      bn._buffers["running_mean"], bn._buffers["running_var"]):
                sig.append((id(t), t._version, t.data_ptr()))
            sig.append(conv.stride)
        return tuple(sig)

    def _hip_trunk(self, device):
        """The packed convolutions of layer1..layer3 for `device`, re-packed when a parameter changed."""
        from .. import ops
        sig = self._hip_signature(device)
        if getattr(self, "_hip_sig", None) != sig:
            def pack(conv, bn):
                if abs(bn.eps - 1e-5) > 1e-12:      # the packed BatchNorm fold uses the reference's eps (resnet.py: nn.BatchNorm2d default)
                    raise NotImplementedError(f"BatchNorm eps {bn.eps}: the HIP pyramid producer folds eps = 1e-5 (set P2P_BACKBONE=miopen)")
                return ops.ConvBN(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.stride[0], device)
            trunk = []
            for name, _, _, _ in _STAGES:
                blocks = []
                for blk in getattr(self, name):
                    down = pack(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                    blocks.append((pack(blk.conv1, blk.bn1), pack(blk.conv2, blk.bn2), down))
                trunk.append(blocks)
            if abs(self.bn1.eps - 1e-5) > 1e-12:
                raise NotImplementedError(f"BatchNorm eps {self.bn1.eps}: the HIP pyramid producer folds eps = 1e-5 (set P2P_BACKBONE=miopen)")
            stem = ops.Stem(self.conv1.weight, self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var, device)
            self._hip_trunk_cache, self._hip_sig = (stem, trunk), sig
        return self._hip_trunk_cache

    def _pyramid_hip(self, x):
        # every launch below goes to the current stream of the CURRENT device: make the input's device current (a model on
        # cuda:1 driven from a process whose current device is cuda:0 would otherwise launch there with device-1 pointers)
        with torch.cuda.device(x.device):
            return self._pyramid_hip_on_current_device(x)

    def _pyramid_hip_on_current_device(self, x):
        from .. import ops
        stem, trunk = self._hip_trunk(x.device)
        feats = [x]
        x = stem.forward(x)
        feats.append(x)
        # float bits of max |activation| per image, one row per layer output: raised by the producing kernel (atomicMax)
        maxima = torch.zeros((1 + 2 * sum(len(b) for b in trunk), x.shape[0]), device=x.device, dtype=torch.int32)
        xmax, row = maxima[0], 1
        x = ops.maxpool_nhwc(x, xmax)                                          # NHWC from here on
        for blocks in trunk:
            for conv1, conv2, down in blocks:
                skip = x if down is None else down.forward(x, xmax, relu=False)
                y = conv1.forward(x, xmax, ymax=maxima[row])
                x = conv2.forward(y, maxima[row], residual=skip, ymax=maxima[row + 1])
                xmax, row = maxima[row + 1], row + 2
            feats.append(ops.nhwc_to_nchw(x))
        return feats

    def pyramid(self, x):
        """[image, relu(bn1(conv1)), layer1, layer2, layer3] -- reference forward_all (resnet.py:138-157)."""
        # the HIP producer: fp32 inference only -- a half / double image, or one autograd has to flow through, takes the torch path
        if (x.is_cuda and not self.training and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad)
                and os.environ.get("P2P_BACKBONE", "hip") != "miopen"):
            return self._pyramid_hip(x)
        feats = [x]
        x = F.relu(self.bn1(self.conv1(x)), inplace=True)
        feats.append(x)
        x = self.layer1(F.max_pool2d(x, 3, 2, 1))
        feats.append(x)
        x = self.layer2(x)
        feats.append(x)
        x = self.layer3(x)
        feats.append(x)
        return feats

    def forward_all(self, x, feat_list=None, early_feat=True):
        """Reference-compatible signature: appends the pyramid to `feat_list`."""
        feats = self.pyramid(x)
        if feat_list is not None:
            feat_list.extend(feats)
        return feats

    def forward(self, x, early_feat=True):
        return self.pyramid(x)[-1]
