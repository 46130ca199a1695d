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

The ``nn.Module`` classes are the PARAMETER CONTAINERS (torchvision's key names) and, on the CPU, the restated arithmetic the
device path is tested against.  On a HIP device ``NativeBackbone`` runs both networks on the library's own kernels
on channels-last fp32 activations with eval-mode BatchNorm folded into weights / bias.  ``precision='split_bf16'`` (the default,
round 6): ONE kernel per inverted-residual block and for the stem's three layers (csrc/irb.hip: expand on matrix cores into an LDS
slice, depthwise taps out of LDS, projection accumulated in registers -- the expanded tensor never reaches HBM) and per fine pyramid
level (csrc/fpn.hip: lateral + top-down + 3x3, output in the reference layout): 1.44 ms per 71 images at 256 x 320, 26 launches.
``precision='fp32'``: the per-layer kernels of round 5 (csrc/backbone.hip: 1x1 / 3x3 convolutions as GEMMs on exact-fp32 matrix
instructions, depthwise and stem kernels, bias + ReLU + residual / top-down addition in the epilogues): 3.3 ms, 55 launches.  The
stock modules on MIOpen took 8.0 ms.
PARITY UNPINNED: torchvision is not installed, so neither the module tree nor the arithmetic can be compared with the real
package here; ``tests/test_backbone.py`` pins shapes, strides and key names as documented for torchvision 0.8.2 and the device
kernels against these modules on the CPU.  Pretrained ImageNet weights are unavailable offline: ``synthetic.backbone_weights``
provides seeded ones.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

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


# ----------------------------------------------------------------------------------------------------------------------------
# HIP forward (csrc/backbone.hip)
# ----------------------------------------------------------------------------------------------------------------------------
def _fold(conv, bn):
    """(weight * scale per output channel, bias) of Conv2d followed by eval-mode BatchNorm2d."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    return conv.weight.detach() * scale.view(-1, 1, 1, 1), bn.bias.detach() - bn.running_mean.detach() * scale


class _Gemm:
    """A packed 1x1 / 3x3 convolution (v3d_conv_pack): weight [Cout, Cin, kh, kw] + bias."""

    def __init__(self, weight, bias, device):
        lib = _lib.load()
        co, ci, kh, kw = weight.shape
        self.taps, self.cin, self.cout = kh * kw, ci, co
        w = weight.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).float().contiguous().cpu()
        b = bias.float().contiguous().cpu()
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):          # the library allocates the weight image on the CURRENT device
            _lib.check(lib.v3d_conv_pack(ctypes.cast(w.data_ptr(), _lib.c_float_p), ctypes.cast(b.data_ptr(), _lib.c_float_p),
                                         co, kh * kw * ci, ctypes.byref(self.handle)), 'v3d_conv_pack')

    def __del__(self):
        try:       # (at interpreter shutdown the binding module may already be gone: the process frees the image anyway)
            if getattr(self, 'handle', None) and self.handle.value and _lib is not None:
                _lib.load().v3d_conv_free(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    def __call__(self, x, relu=False, res=None, res_mode=0):
        n, H, W, _ = x.shape
        out = torch.empty((n, H, W, self.cout), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().v3d_conv_nhwc_f32(self.handle, x.data_ptr(), n, H, W, self.cin, self.taps, int(relu), res_mode,
                                                 _lib.ptr(res), out.data_ptr(), _lib.stream_ptr(x.device)), 'v3d_conv_nhwc_f32')
        return out


class _Depthwise:
    def __init__(self, weight, bias, stride, device):
        c, _, k, _ = weight.shape
        self.k, self.stride, self.c = k, stride, c
        self.w = weight.reshape(c, k * k).t().float().contiguous().to(device)          # [k*k][C]
        self.b = bias.float().contiguous().to(device)

    def __call__(self, x, relu=True):
        n, H, W, _ = x.shape
        s = self.stride
        out = torch.empty((n, (H + s - 1) // s, (W + s - 1) // s, self.c), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().v3d_depthwise_nhwc_f32(x.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), n, H, W, self.c, self.k, s,
                                                      int(relu), out.data_ptr(), _lib.stream_ptr(x.device)), 'v3d_depthwise_nhwc_f32')
        return out


class _Block:
    """An inverted-residual block as ONE kernel (csrc/irb.hip, v3d_irb_*): split-bf16 matrix operands, the expanded tensor never
    reaches HBM.  ``supported(H, W)`` says whether the library has a kernel instance for this block at that input size."""

    def __init__(self, blk, device):
        lib = _lib.load()
        L = blk.layers
        we, be = _fold(L[0], L[1])
        wd, bd = _fold(L[3], L[4])
        wp, bp = _fold(L[6], L[7])
        mid, cin = we.shape[0], we.shape[1]
        self.cin, self.mid, self.cout, self.k, self.stride = cin, mid, wp.shape[0], wd.shape[-1], L[3].stride[0]
        host = [t.float().contiguous().cpu() for t in (we.reshape(mid, cin), be, wd.reshape(mid, -1), bd, wp.reshape(self.cout, mid), bp)]
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.v3d_irb_pack(*[ctypes.cast(t.data_ptr(), _lib.c_float_p) for t in host], cin, mid, self.cout, self.k,
                                        self.stride, int(blk.apply_residual), ctypes.byref(self.handle)), 'v3d_irb_pack')

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.handle.value and _lib is not None:
                _lib.load().v3d_irb_free(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    def supported(self, H, W):
        return bool(_lib.load().v3d_irb_supported(self.handle, H, W))

    def __call__(self, x):
        n, H, W, _ = x.shape
        s = self.stride
        out = torch.empty((n, (H + s - 1) // s, (W + s - 1) // s, self.cout), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        nbytes = lib.v3d_irb_workspace_bytes(self.handle, n, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None      # (the caching allocator: no device call)
        _lib.check(lib.v3d_irb_nhwc_f32(self.handle, x.data_ptr(), n, H, W, out.data_ptr(), _lib.ptr(ws), nbytes, _lib.stream_ptr(x.device)),
                   'v3d_irb_nhwc_f32')
        return out


class _StemBlock:
    """``FeatureExtractor.layer1`` (mnasnet layers 0-7: stem convolution, depthwise, pointwise; image -> 16 channels at half
    resolution) as ONE kernel (csrc/irb.hip, v3d_stem_block_*), split-bf16 matrix operands."""

    def __init__(self, layer1, device):
        lib = _lib.load()
        ws, bs = _fold(layer1[0], layer1[1])
        wd, bd = _fold(layer1[3], layer1[4])
        wp, bp = _fold(layer1[6], layer1[7])
        host = [t.float().contiguous().cpu() for t in (ws.reshape(32, 27), bs, wd.reshape(32, 9), bd, wp.reshape(16, 32), bp)]
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.v3d_stem_block_pack(*[ctypes.cast(t.data_ptr(), _lib.c_float_p) for t in host], ctypes.byref(self.handle)),
                       'v3d_stem_block_pack')

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.handle.value and _lib is not None:
                _lib.load().v3d_irb_free(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    def __call__(self, img):
        n, _, H, W = img.shape
        out = torch.empty((n, H // 2, W // 2, 16), dtype=torch.float32, device=img.device)
        _lib.check(_lib.load().v3d_stem_block_f32(self.handle, img.data_ptr(), n, H, W, out.data_ptr(), _lib.stream_ptr(img.device)),
                   'v3d_stem_block_f32')
        return out


class _PyramidLevel:
    """One level of the feature pyramid as ONE kernel (csrc/fpn.hip, v3d_fpn_*): lateral 1x1 + top-down addition + 3x3 output
    convolution, the result in the reference layout; feat_dim 32 and at most 48 input channels (the three fine levels)."""

    @staticmethod
    def fits(lateral, output):
        return lateral.out_channels == 32 and output.out_channels == 32 and lateral.in_channels % 8 == 0 and lateral.in_channels <= 48

    def __init__(self, lateral, output, device):
        lib = _lib.load()
        self.cin = lateral.in_channels
        host = [t.detach().float().contiguous().cpu() for t in (lateral.weight.reshape(32, self.cin), lateral.bias, output.weight, output.bias)]
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.v3d_fpn_pack(*[ctypes.cast(t.data_ptr(), _lib.c_float_p) for t in host], self.cin, ctypes.byref(self.handle)),
                       'v3d_fpn_pack')

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.handle.value and _lib is not None:
                _lib.load().v3d_fpn_free(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    def __call__(self, x, coarse_inner, want_inner):
        """x [n, H, W, cin], coarse_inner [n, ceil(H/2), ceil(W/2), 32] or None -> (inner [n, H, W, 32] or None, out [n, 32, H, W])"""
        n, H, W, _ = x.shape
        inner = torch.empty((n, H, W, 32), dtype=torch.float32, device=x.device) if want_inner else None
        out = torch.empty((n, 32, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().v3d_fpn_level_f32(self.handle, x.data_ptr(), _lib.ptr(coarse_inner), n, H, W, _lib.ptr(inner), out.data_ptr(),
                                                 _lib.stream_ptr(x.device)), 'v3d_fpn_level_f32')
        return inner, out


class NativeBackbone:
    """``(feat_extractor, feat_shrinker)`` on the HIP kernels: ``forward(images [n, 3, H, W]) -> (half, quarter, eighth,
    sixteenth, thirtysecond)`` in the reference layout [n, feat_dim, h, w] -- what ``feat_shrinker(*feat_extractor(images))``
    returns (mvsnet.py:66-73, 89-105).  H and W must be multiples of 8 (round 6: the reference's default 240 x 320 gives 15 x 20
    and 8 x 10 maps at 1/16 and 1/32 -- odd sizes are handled by the strided kernels and by the FPN's nearest top-down
    addition) and ``feat_dim`` a multiple of 32; ``why_not()`` names the reason when a call does not qualify.

    ``precision`` (as everywhere in this package): 'split_bf16' runs every inverted-residual block the library has a fused kernel
    for as ONE launch (csrc/irb.hip); 'fp32' keeps the exact-fp32 three-launch blocks of round 5."""

    def __init__(self, feat_extractor, feat_shrinker, precision='split_bf16'):
        _lib.precision_code(precision)
        self.fe, self.fs, self.precision = feat_extractor, feat_shrinker, precision
        self._key, self._ops = None, None

    # (the packed kernels' weight images are a cache: a copy of the owning MVSNet packs again)
    def __deepcopy__(self, memo):
        import copy
        return NativeBackbone(copy.deepcopy(self.fe, memo), copy.deepcopy(self.fs, memo), self.precision)

    def __reduce__(self):
        return (NativeBackbone, (self.fe, self.fs, self.precision))

    def is_package_pair(self):
        """The two modules are this package's own containers (torchvision's module tree restated): the pair the kernels serve."""
        return isinstance(self.fe, FeatureExtractor) and isinstance(self.fs, FeatureShrinker)

    def why_not(self, images):
        """None when the call qualifies for the HIP kernels, else the reason (MVSNet.forward raises with it)."""
        from .mvsnet import module_device
        if not self.is_package_pair():
            return 'feat_extractor / feat_shrinker are not backbone.FeatureExtractor / FeatureShrinker'
        if not images.is_cuda:
            return 'images are not on a HIP device (there is no CPU path)'
        if images.dim() != 4 or images.shape[1] != 3:
            return 'images must be [n, 3, H, W], got %s' % (tuple(images.shape),)
        if images.shape[2] % 8 or images.shape[3] % 8:
            return 'image sides must be multiples of 8, got %d x %d' % (images.shape[2], images.shape[3])
        if self.fs.fpn.layer_blocks[0].out_channels % 32:
            return 'feat_dim must be a multiple of 32 (channel-last -> reference layout kernel), got %d' \
                % self.fs.fpn.layer_blocks[0].out_channels
        if self.fe.training or self.fs.training:
            return 'the modules are in training mode (BatchNorm is folded with running statistics)'
        if module_device(self.fe) != images.device or module_device(self.fs) != images.device:
            return 'modules on %s / %s, images on %s' % (module_device(self.fe), module_device(self.fs), images.device)
        return None

    def supports(self, images):
        return self.why_not(images) is None

    def _build(self, device):
        from .mvsnet import module_state_key
        key = (str(device), self.precision) + module_state_key(self.fe) + module_state_key(self.fs)
        if key == self._key:
            return self._ops
        l1 = self.fe.layer1
        w0, b0 = _fold(l1[0], l1[1])
        ops = dict(stem_w=w0.permute(1, 2, 3, 0).reshape(27, 32).float().contiguous().to(device), stem_b=b0.float().contiguous().to(device))
        w1, b1 = _fold(l1[3], l1[4])
        ops['stem_dw'] = _Depthwise(w1, b1, 1, device)
        ops['stem_pw'] = _Gemm(*_fold(l1[6], l1[7]), device)
        ops['stem_block'] = _StemBlock(l1, device) if self.precision == 'split_bf16' else None
        stages = []
        for layer in (self.fe.layer2, self.fe.layer3, self.fe.layer4, self.fe.layer5):
            blocks = []
            for stack in layer:
                for blk in stack:
                    L = blk.layers
                    wd, bd = _fold(L[3], L[4])
                    blocks.append((_Gemm(*_fold(L[0], L[1]), device), _Depthwise(wd, bd, L[3].stride[0], device), _Gemm(*_fold(L[6], L[7]), device),
                                   blk.apply_residual, _Block(blk, device) if self.precision == 'split_bf16' else None))
            stages.append(blocks)
        ops['stages'] = stages
        fpn = self.fs.fpn
        ops['inner'] = [_Gemm(m.weight.detach(), m.bias.detach(), device) for m in fpn.inner_blocks]
        ops['outer'] = [_Gemm(m.weight.detach(), m.bias.detach(), device) for m in fpn.layer_blocks]
        ops['level'] = [_PyramidLevel(a, b, device) if self.precision == 'split_bf16' and i < 3 and _PyramidLevel.fits(a, b) else None
                        for i, (a, b) in enumerate(zip(fpn.inner_blocks, fpn.layer_blocks))]
        self._key, self._ops = key, ops
        return ops

    def __call__(self, images):
        lib = _lib.load()
        dev = images.device
        ops = self._build(dev)
        img = images.contiguous().float()
        n, _, H, W = img.shape
        if ops['stem_block'] is not None:
            x = ops['stem_block'](img)
        else:
            x = torch.empty((n, H // 2, W // 2, 32), dtype=torch.float32, device=dev)
            _lib.check(lib.v3d_stem_f32(img.data_ptr(), ops['stem_w'].data_ptr(), ops['stem_b'].data_ptr(), n, H, W, x.data_ptr(),
                                        _lib.stream_ptr(dev)), 'v3d_stem_f32')
            x = ops['stem_pw'](ops['stem_dw'](x, relu=True), relu=False)
        maps = [x]                                                  # C1 .. C5, channels-last
        for blocks in ops['stages']:
            for expand, dw, project, residual, fused in blocks:
                if fused is not None and fused.supported(x.shape[1], x.shape[2]):
                    x = fused(x)
                    continue
                y = project(dw(expand(x, relu=True), relu=True), relu=False, res=x if residual else None, res_mode=1 if residual else 0)
                x = y
            maps.append(x)
        inner = ops['inner'][4](maps[4])
        outs = [None] * 5
        outs[4] = ops['outer'][4](inner)
        for i in (3, 2, 1, 0):
            if ops['level'][i] is not None:                         # one launch, the result already in the reference layout
                inner, outs[i] = ops['level'][i](maps[i], inner, want_inner=i > 0)
                continue
            inner = ops['inner'][i](maps[i], res=inner, res_mode=2)
            outs[i] = ops['outer'][i](inner)
        res = []
        for i, o in enumerate(outs):                                # -> the reference layout [n, C, h, w]
            if ops['level'][i] is not None:
                res.append(o)
                continue
            nn_, h, w, c = o.shape
            t = torch.empty((nn_, c, h, w), dtype=torch.float32, device=dev)
            _lib.check(lib.v3d_nhwc_to_nchw_f32(o.data_ptr(), t.data_ptr(), nn_, c, h * w, _lib.stream_ptr(dev)), 'v3d_nhwc_to_nchw_f32')
            res.append(t)
        return tuple(res)
