"""2D feature extractor of ``MVSNet`` without torchvision -- "next" row of SURVEY.md §8f (rank 3), the stage that feeds the
hot path: ``mv3d/subnetworks/mvsnet.py:55-105`` builds it from ``torchvision.models.mnasnet1_0(pretrained=True)`` and
``torchvision.ops.FeaturePyramidNetwork`` (torchvision 0.8.2, un-vendored and absent here).  This module restates the
two architectures in plain PyTorch with the same module tree, so the ``state_dict`` keys and shapes are the ones a
reference checkpoint carries under ``mvsnet.feat_extractor.*`` / ``mvsnet.feat_shrinker.*``:

  * ``FeatureExtractor``: the MnasNet-1.0 trunk (stem 3->32->16, then six stacks of inverted-residual blocks with
    channel widths 24, 40, 80, 96, 192, 320), regrouped by the reference into ``layer1..layer5`` =
    ``layers[0:8]``, ``[8:9]``, ``[9:10]``, ``[10:12]``, ``[12:14]`` -> 16 / 24 / 40 / 96 / 320 channels at 1/2 .. 1/32
    resolution (mvsnet.py:60-64, 66-73);
  * ``FeatureShrinker``: a feature pyramid over those five maps -- 1x1 lateral convolutions, nearest-neighbour top-down
    additions, 3x3 output convolutions, all with bias -- returning ``feat_dim`` channels per level, finest first
    (mvsnet.py:83-105).

Stock 2D convolutions executed by PyTorch-ROCm (MIOpen), as in the reference; nothing here is a hand-written kernel.
PARITY UNPINNED: torchvision is not installed, so neither the module tree nor the arithmetic can be compared with the real
package here; ``tests/test_backbone.py`` pins shapes, strides and key names as documented for torchvision 0.8.2.
Pretrained ImageNet weights are unavailable offline: ``synthetic.backbone_weights`` provides seeded ones.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_BN_MOMENTUM = 1 - 0.9997          # torchvision's MnasNet batch-norm momentum (irrelevant in eval mode)


class _InvertedResidual(nn.Module):
    """MnasNet block: 1x1 expand -> kxk depthwise (stride s) -> 1x1 project, BN after each, ReLU after the first two;
    identity shortcut when the shape is preserved.  The convolutions live under ``.layers`` like in torchvision."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, expansion):
        super().__init__()
        mid = in_ch * expansion
        self.apply_residual = in_ch == out_ch and stride == 1
        self.layers = nn.Sequential(
            nn.Conv2d(in_ch, mid, 1, bias=False), nn.BatchNorm2d(mid, momentum=_BN_MOMENTUM), nn.ReLU(inplace=True),
            nn.Conv2d(mid, mid, kernel_size, padding=kernel_size // 2, stride=stride, groups=mid, bias=False),
            nn.BatchNorm2d(mid, momentum=_BN_MOMENTUM), nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch, momentum=_BN_MOMENTUM))

    def forward(self, x):
        y = self.layers(x)
        return y + x if self.apply_residual else y


def _stack(in_ch, out_ch, kernel_size, stride, expansion, repeats):
    blocks = [_InvertedResidual(in_ch, out_ch, kernel_size, stride, expansion)]
    blocks += [_InvertedResidual(out_ch, out_ch, kernel_size, 1, expansion) for _ in range(repeats - 1)]
    return nn.Sequential(*blocks)


def mnasnet1_0_trunk():
    """The first 14 children of ``mnasnet1_0().layers`` (the classifier head is never used by the reference)."""
    return [
        nn.Conv2d(3, 32, 3, padding=1, stride=2, bias=False), nn.BatchNorm2d(32, momentum=_BN_MOMENTUM),
        nn.ReLU(inplace=True),
        nn.Conv2d(32, 32, 3, padding=1, stride=1, groups=32, bias=False), nn.BatchNorm2d(32, momentum=_BN_MOMENTUM),
        nn.ReLU(inplace=True),
        nn.Conv2d(32, 16, 1, padding=0, stride=1, bias=False), nn.BatchNorm2d(16, momentum=_BN_MOMENTUM),
        _stack(16, 24, 3, 2, 3, 3), _stack(24, 40, 5, 2, 3, 3), _stack(40, 80, 5, 2, 6, 3),
        _stack(80, 96, 3, 1, 6, 2), _stack(96, 192, 5, 2, 6, 4), _stack(192, 320, 3, 1, 6, 1)]


class FeatureExtractor(nn.Module):
    """``forward(image[B,3,H,W]) -> (layer1 .. layer5)`` with 16 / 24 / 40 / 96 / 320 channels at H/2 .. H/32
    (mvsnet.py:55-73)."""

    OUT_CHANNELS = (16, 24, 40, 96, 320)

    def __init__(self):
        super().__init__()
        trunk = mnasnet1_0_trunk()
        self.layer1 = nn.Sequential(*trunk[0:8])
        self.layer2 = nn.Sequential(*trunk[8:9])
        self.layer3 = nn.Sequential(*trunk[9:10])
        self.layer4 = nn.Sequential(*trunk[10:12])
        self.layer5 = nn.Sequential(*trunk[12:14])

    def forward(self, image):
        layer1 = self.layer1(image)
        layer2 = self.layer2(layer1)
        layer3 = self.layer3(layer2)
        layer4 = self.layer4(layer3)
        layer5 = self.layer5(layer4)
        return layer1, layer2, layer3, layer4, layer5


class FeaturePyramidNetwork(nn.Module):
    """Top-down pyramid with lateral connections (``inner_blocks`` 1x1, ``layer_blocks`` 3x3, both with bias), no
    extra blocks: coarsest level first, each finer level = lateral + nearest-upsampled coarser inner map."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels_list])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, maps):
        inner = self.inner_blocks[-1](maps[-1])
        out = [self.layer_blocks[-1](inner)]
        for i in range(len(maps) - 2, -1, -1):
            lateral = self.inner_blocks[i](maps[i])
            inner = lateral + F.interpolate(inner, size=lateral.shape[-2:], mode='nearest')
            out.insert(0, self.layer_blocks[i](inner))
        return out


class FeatureShrinker(nn.Module):
    """``forward(layer1 .. layer5) -> (features_half, features_quarter, features_eighth, features_sixteenth,
    features_thirtysecond)``, ``feat_dim`` channels each (mvsnet.py:83-105)."""

    def __init__(self, feat_dim):
        super().__init__()
        self.fpn = FeaturePyramidNetwork(list(FeatureExtractor.OUT_CHANNELS), feat_dim)

    def forward(self, layer1, layer2, layer3, layer4, layer5):
        return tuple(self.fpn([layer1, layer2, layer3, layer4, layer5]))


def build_backbone(feat_dim):
    """(feat_extractor, feat_shrinker) as ``MVSNet.__init__`` creates them (mvsnet.py:172-173), in eval mode."""
    return FeatureExtractor().eval(), FeatureShrinker(feat_dim).eval()
