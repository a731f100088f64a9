"""Feature-pyramid producer for the matching hot path (stays on PyTorch-ROCm / MIOpen).

Role of reference networks/resnet.py:125-173 (ResNet34 truncated after layer3, with the
layer3 stride patch of `change_stride`).  This is the *boundary* of the hot path
(SURVEY.md section 8 row a20): it is deliberately left to PyTorch, the HIP library starts
at its outputs.  Parameter names follow the torchvision ResNet convention so that
reference checkpoints (`extract.*` keys, utils/train/helper.py:10-17) load unchanged;
`layer4.*` keys of a checkpoint are never used by the path (reference
networks/patch2pix.py:72-74 freezes them as "never used") and are skipped on load.
"""
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

    def pyramid(self, x):
        """[image, relu(bn1(conv1)), layer1, layer2, layer3] -- reference forward_all (resnet.py:138-157)."""
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
