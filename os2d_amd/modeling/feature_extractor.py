"""ResNet-C4 feature extractor kept in PyTorch-ROCm (MIOpen), as BASELINE.json's north_star prescribes: the backbone
is an input producer for the hot path, not part of it.

Mirrors reference os2d/modeling/feature_extractor.py:13-129 (``build_feature_extractor``, ``ResNetFeatureExtractor``,
``resnet50_c4`` / ``resnet101_c4``).  torchvision is not available in this image, so the ResNet (v1.5 bottleneck:
stride on the 3x3 convolution, as torchvision >= 0.5) is defined here with torchvision's module / parameter names
(``conv1, bn1, layer1.0.conv1, layer1.0.downsample.0, ...``) so reference checkpoints load key-for-key.
"""
from itertools import chain

import torch.nn as nn

from ..structures.feature_map import FeatureMapSize

GROUPNORM_NUMGROUPS = 32


def get_norm_layer(use_group_norm):
    if use_group_norm:
        return lambda width: nn.GroupNorm(GROUPNORM_NUMGROUPS, width)
    return nn.BatchNorm2d


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNetFeatureExtractor(nn.Module):
    """conv1-bn1-relu-maxpool-layer1..layer{level-1}; level 4 = C4: stride 16, 1024 channels
    (reference feature_extractor.py:23-65)."""

    def __init__(self, layers, level, feature_map_stride, feature_map_receptive_field, use_group_norm=False):
        super(ResNetFeatureExtractor, self).__init__()
        assert level in [1, 2, 3, 4, 5], "Feature level should be one of 1, 2, 3, 4, 5"
        norm_layer = get_norm_layer(use_group_norm)
        self._norm_layer = norm_layer
        self.feature_map_receptive_field = feature_map_receptive_field
        self.feature_map_stride = feature_map_stride
        self._feature_level = level
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        names = ["layer1", "layer2", "layer3", "layer4"][:level - 1]
        planes = [64, 128, 256, 512]
        strides = [1, 2, 2, 2]
        for i, name in enumerate(names):
            setattr(self, name, self._make_layer(planes[i], layers[i], strides[i]))
        self._block_names = names
        for m in self.modules():   # torchvision's default initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @property
    def resnet_blocks(self):
        return [getattr(self, n) for n in self._block_names]

    def _make_layer(self, planes, blocks, stride):
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                norm_layer(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for layer in self.resnet_blocks:
            x = layer(x)
        return x

    def freeze_bn(self):
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def freeze_blocks(self, num_blocks=0):
        """Freeze the first ``num_blocks`` blocks; block 0 = conv1+bn1 (reference feature_extractor.py:73-83)."""
        layer0 = [nn.ModuleList([self.conv1, self.bn1])]
        remaining = num_blocks
        for b in chain(layer0, chain.from_iterable(self.resnet_blocks)):
            if remaining > 0:
                for p in b.parameters():
                    p.requires_grad = False
                remaining -= 1

    def get_num_blocks_in_feature_extractor(self):
        return 1 + sum(len(b) for b in self.resnet_blocks)


def resnet50_c4(use_group_norm=False):
    """R-50-C4 (reference feature_extractor.py:108-117): stride 16, declared receptive field 16."""
    return ResNetFeatureExtractor([3, 4, 6, 3], 4, FeatureMapSize(h=16, w=16), FeatureMapSize(h=16, w=16), use_group_norm)


def resnet101_c4(use_group_norm=False):
    """R-101-C4 (reference feature_extractor.py:120-129)."""
    return ResNetFeatureExtractor([3, 4, 23, 3], 4, FeatureMapSize(h=16, w=16), FeatureMapSize(h=16, w=16), use_group_norm)


def build_feature_extractor(backbone_arch, use_group_norm=False):
    """reference feature_extractor.py:13-20."""
    arch = backbone_arch.lower()
    if arch == "resnet50":
        return resnet50_c4(use_group_norm=use_group_norm)
    if arch == "resnet101":
        return resnet101_c4(use_group_norm=use_group_norm)
    raise RuntimeError("Unknown backbone arch: {0}".format(backbone_arch))
