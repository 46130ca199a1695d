"""Oracle of SURVEY.md 8f rank 3 -- the 2D feature extractor of MVSNet (mv3d/subnetworks/mvsnet.py:55-105) -- TEST
INFRASTRUCTURE (tests/, smoke and bench.py's checker legs only; never imported by the product package).

A functional CPU restatement from STATE DICTS, independent of the product's module containers (3dvnet_amd/backbone.py):

  * ``extractor``: ``torchvision.models.mnasnet1_0().layers[0:14]`` as the reference regroups it (mvsnet.py:60-64:
    layer1 = layers[0:8], layer2 = [8:9], layer3 = [9:10], layer4 = [10:12], layer5 = [12:14]; forward :66-73) --
    stem Conv(3->32, k3 s2 p1) BN ReLU, depthwise Conv(32, k3 s1 p1) BN ReLU, Conv(32->16, 1x1) BN, then six stacks of
    inverted-residual blocks (expand 1x1 BN ReLU -> depthwise kxk stride s BN ReLU -> project 1x1 BN, identity shortcut when
    in == out and stride 1) with (in, out, kernel, stride, expansion, repeats) = (16,24,3,2,3,3) (24,40,5,2,3,3) (40,80,5,2,6,3)
    (80,96,3,1,6,2) (96,192,5,2,6,4) (192,320,3,1,6,1), as torchvision 0.8.2's mnasnet.py documents them;
  * ``shrinker``: ``torchvision.ops.FeaturePyramidNetwork([16, 24, 40, 96, 320], feat_dim)`` (mvsnet.py:86-88, forward
    :89-105): P5 = layer5(inner5(C5)); inner_k = inner_k(C_k) + nearest(inner_{k+1}, size of C_k); P_k = layer_k(inner_k).

PARITY UNPINNED: torchvision is an un-vendored dependency absent from this image and from /root/reference, so this restatement
cannot be checked against the real package here (tests/test_backbone.py pins the documented key names, shapes and strides).
"""
import torch
import torch.nn.functional as F

STACKS = {'layer2': [(16, 24, 3, 2, 3, 3)], 'layer3': [(24, 40, 5, 2, 3, 3)],
          'layer4': [(40, 80, 5, 2, 6, 3), (80, 96, 3, 1, 6, 2)], 'layer5': [(96, 192, 5, 2, 6, 4), (192, 320, 3, 1, 6, 1)]}


def _bn(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'],
                        sd[prefix + '.bias'], False, 0.0, eps)


def _inverted_residual(x, sd, prefix, kernel, stride, shortcut):
    mid = sd[prefix + '.layers.0.weight'].shape[0]
    y = F.relu(_bn(F.conv2d(x, sd[prefix + '.layers.0.weight']), sd, prefix + '.layers.1'))
    y = F.relu(_bn(F.conv2d(y, sd[prefix + '.layers.3.weight'], stride=stride, padding=kernel // 2, groups=mid), sd,
                   prefix + '.layers.4'))
    y = _bn(F.conv2d(y, sd[prefix + '.layers.6.weight']), sd, prefix + '.layers.7')
    return y + x if shortcut else y


def extractor(sd, image):
    """mvsnet.py:66-73 -> (layer1 .. layer5): 16 / 24 / 40 / 96 / 320 channels at 1/2 .. 1/32 resolution."""
    x = F.relu(_bn(F.conv2d(image, sd['layer1.0.weight'], stride=2, padding=1), sd, 'layer1.1'))
    x = F.relu(_bn(F.conv2d(x, sd['layer1.3.weight'], padding=1, groups=32), sd, 'layer1.4'))
    x = _bn(F.conv2d(x, sd['layer1.6.weight']), sd, 'layer1.7')
    maps = [x]
    for name in ('layer2', 'layer3', 'layer4', 'layer5'):
        for si, (cin, cout, kernel, stride, _, repeats) in enumerate(STACKS[name]):
            for bi in range(repeats):
                x = _inverted_residual(x, sd, '%s.%d.%d' % (name, si, bi), kernel, stride if bi == 0 else 1,
                                       shortcut=bi > 0 or (cin == cout and stride == 1))
        maps.append(x)
    return tuple(maps)


def shrinker(sd, maps):
    """mvsnet.py:89-105 -> (features_half, quarter, eighth, sixteenth, thirtysecond)."""
    inner = F.conv2d(maps[4], sd['fpn.inner_blocks.4.weight'], sd['fpn.inner_blocks.4.bias'])
    out = [F.conv2d(inner, sd['fpn.layer_blocks.4.weight'], sd['fpn.layer_blocks.4.bias'], padding=1)]
    for i in (3, 2, 1, 0):
        lateral = F.conv2d(maps[i], sd['fpn.inner_blocks.%d.weight' % i], sd['fpn.inner_blocks.%d.bias' % i])
        inner = lateral + F.interpolate(inner, size=lateral.shape[-2:], mode='nearest')
        out.insert(0, F.conv2d(inner, sd['fpn.layer_blocks.%d.weight' % i], sd['fpn.layer_blocks.%d.bias' % i], padding=1))
    return tuple(out)


def backbone_features(sd_extractor, sd_shrinker, images):
    """``feat_shrinker(*feat_extractor(images))`` (mvsnet.py:177-178) on the CPU in fp32."""
    with torch.no_grad():
        f32 = lambda sd: {k: v.detach().float().cpu() for k, v in sd.items() if v.dtype.is_floating_point}
        return shrinker(f32(sd_shrinker), extractor(f32(sd_extractor), images.detach().float().cpu()))
