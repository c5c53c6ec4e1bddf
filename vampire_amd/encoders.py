"""Build-owned image / BEV encoders standing where the reference calls mmdet's `build_backbone` and
mmdet3d's `build_neck` (base_vampire2.py:167-168, bev_depth_head.py:131-134): ResNet-18/34/50 and the
SECOND FPN, with mmdet's constructor arguments and attribute names and torchvision's parameter
names (so `torchvision://resnet50` state dicts load).  mmdet / mmdet3d are not in this image; these
are implementations of the published architectures, not copies -- "parity unpinned" against the
registry versions."""
import torch
import torch.nn.functional as F
from torch import nn


# =============================================================================================
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        return F.relu(self.bn2(self.conv2(y)) + idt, inplace=True)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)       # style='pytorch': stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = F.relu(self.bn2(self.conv2(y)), inplace=True)
        return F.relu(self.bn3(self.conv3(y)) + idt, inplace=True)


class ResNet(nn.Module):
    """ResNet-18/34/50 trunk with mmdet's constructor arguments and attribute names (`conv1 norm1 relu
    maxpool res_layers out_indices deep_stem`, layers `layer1..`), torchvision parameter names."""
    ARCH = {18: (_BasicBlock, (2, 2, 2, 2)), 34: (_BasicBlock, (3, 4, 6, 3)), 50: (_Bottleneck, (3, 4, 6, 3))}

    def __init__(self, depth=50, in_channels=3, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), norm_eval=False, frozen_stages=-1,
                 init_cfg=None, type=None, **_):
        super().__init__()
        block, blocks = self.ARCH[depth]
        self.deep_stem = False
        self.out_indices = list(out_indices)
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(base_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.res_layers = []
        cin = base_channels
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            layers = []
            for j in range(blocks[i]):
                stride = strides[i] if j == 0 else 1
                down = None
                if stride != 1 or cin != planes * block.expansion:
                    down = nn.Sequential(nn.Conv2d(cin, planes * block.expansion, 1, stride, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                layers.append(block(cin, planes, stride, down))
                cin = planes * block.expansion
            name = f"layer{i + 1}"
            setattr(self, name, nn.Sequential(*layers))
            self.res_layers.append(name)

    @property
    def norm1(self):
        return self.bn1

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        if hasattr(self, "maxpool"):
            x = self.maxpool(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return outs


class SECONDFPN(nn.Module):
    """Every level resampled to one stride (transposed conv up, strided conv down), BN + ReLU, concat."""

    def __init__(self, in_channels, out_channels, upsample_strides, type=None, **_):
        super().__init__()
        blocks = []
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s >= 1:
                s = int(round(s))
                layer = nn.ConvTranspose2d(cin, cout, s, s, bias=False)
            else:
                k = int(round(1.0 / s))
                layer = nn.Conv2d(cin, cout, k, k, bias=False)
            blocks.append(nn.Sequential(layer, nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, xs):
        return [torch.cat([blk(x) for blk, x in zip(self.deblocks, xs)], dim=1)]


def build_backbone(conf):
    return ResNet(**conf)


def build_neck(conf):
    return SECONDFPN(**conf)


