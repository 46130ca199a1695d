"""the MnasNet-1.0 + FPN feature extractor restated without torchvision
(mv3d/subnetworks/mvsnet.py:55-105).  torchvision is absent here, so parity with the real package is UNPINNED; these tests
pin what the reference relies on: the five slice points and their channel widths / strides, the FPN's outputs, and the
state_dict key names and shapes torchvision 0.8.2 documents for ``mnasnet1_0().layers[0:14]`` and
``ops.FeaturePyramidNetwork`` (so a reference checkpoint's ``mvsnet.feat_extractor.*`` / ``feat_shrinker.*`` entries load)."""
import pytest
import torch

from conftest import v3d


def test_extractor_slice_points_channels_and_strides():
    bb = v3d('backbone')
    fe = bb.FeatureExtractor().eval()
    with torch.no_grad():
        outs = fe(torch.zeros(1, 3, 64, 96))
    assert [o.shape[1] for o in outs] == [16, 24, 40, 96, 320]
    assert [tuple(o.shape[2:]) for o in outs] == [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    # MnasNet-1.0 stacks: (blocks, kernel, expansion) per slice (torchvision mnasnet.py: depths 24/40/80/96/192/320)
    assert [len(s) for s in (fe.layer2[0], fe.layer3[0], fe.layer4[0], fe.layer4[1], fe.layer5[0], fe.layer5[1])] == \
        [3, 3, 3, 2, 4, 1]
    assert fe.layer3[0][0].layers[3].kernel_size == (5, 5) and fe.layer3[0][0].layers[3].stride == (2, 2)
    assert fe.layer4[1][0].layers[3].stride == (1, 1) and fe.layer2[0][1].apply_residual
    n_params = sum(p.numel() for p in fe.parameters())
    assert n_params == 3102312 - (320 * 1280 + 2 * 1280)      # mnasnet1_0 features (3.10 M) minus the unused 1x1 head


def test_state_dict_keys_follow_torchvision_naming():
    bb = v3d('backbone')
    fe, fs = bb.build_backbone(32)
    k = fe.state_dict()
    assert k['layer1.0.weight'].shape == (32, 3, 3, 3) and k['layer1.3.weight'].shape == (32, 1, 3, 3)
    assert k['layer1.6.weight'].shape == (16, 32, 1, 1) and 'layer1.7.running_var' in k
    assert k['layer2.0.0.layers.0.weight'].shape == (48, 16, 1, 1)
    assert k['layer2.0.0.layers.3.weight'].shape == (48, 1, 3, 3)
    assert k['layer2.0.2.layers.6.weight'].shape == (24, 72, 1, 1)
    assert k['layer3.0.0.layers.3.weight'].shape == (72, 1, 5, 5)
    assert k['layer4.0.0.layers.0.weight'].shape == (240, 40, 1, 1)
    assert k['layer4.1.1.layers.6.weight'].shape == (96, 576, 1, 1)
    assert k['layer5.0.3.layers.3.weight'].shape == (1152, 1, 5, 5)
    assert k['layer5.1.0.layers.6.weight'].shape == (320, 1152, 1, 1) and 'layer5.1.0.layers.7.running_mean' in k
    f = fs.state_dict()
    assert sorted(f) == sorted(['fpn.%s.%d.%s' % (b, i, p) for b in ('inner_blocks', 'layer_blocks')
                                for i in range(5) for p in ('weight', 'bias')])
    assert f['fpn.inner_blocks.4.weight'].shape == (32, 320, 1, 1) and f['fpn.layer_blocks.0.weight'].shape == (32, 32, 3, 3)


def test_shrinker_is_a_top_down_pyramid():
    """Against a direct evaluation of the documented recursion: P5 = out5(lat5(C5)); Pk = outk(latk(Ck) + up(inner k+1))."""
    bb, syn = v3d('backbone'), v3d('synthetic')
    fe, fs = bb.build_backbone(16)
    sd_e, sd_s = syn.backbone_weights(16, seed=6)
    assert not fe.load_state_dict(sd_e, strict=False).unexpected_keys
    fs.load_state_dict(sd_s)
    img = syn.make_images(2, (64, 96), seed=1)
    with torch.no_grad():
        maps = fe(img)
        half, quarter, eighth, sixteenth, thirtysecond = fs(*maps)
        inner = None
        for i in (4, 3, 2, 1, 0):
            lat = torch.nn.functional.conv2d(maps[i], sd_s['fpn.inner_blocks.%d.weight' % i], sd_s['fpn.inner_blocks.%d.bias' % i])
            inner = lat if inner is None else lat + torch.nn.functional.interpolate(inner, size=lat.shape[-2:], mode='nearest')
            want = torch.nn.functional.conv2d(inner, sd_s['fpn.layer_blocks.%d.weight' % i], sd_s['fpn.layer_blocks.%d.bias' % i], padding=1)
            got = (half, quarter, eighth, sixteenth, thirtysecond)[i]
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    assert tuple(quarter.shape) == (2, 16, 16, 24) and tuple(half.shape) == (2, 16, 32, 48)
    assert float(quarter.std()) > 1e-3 and torch.isfinite(quarter).all()


def test_oracle_restatement_equals_the_containers_stock_forward():
    """oracle/backbone.py (functional, from state dicts) against the product's parameter containers run as stock PyTorch
    modules -- the explicit `native_backbone = False` path -- at the cfg2 size and at the reference's default 240 x 320
    (odd 15 x 20 and 8 x 10 pyramid levels): the same convolutions in the same order, equal to the last bit on one host."""
    from oracle import backbone as ob
    bb, syn = v3d('backbone'), v3d('synthetic')
    fe, fs = bb.build_backbone(32)
    sd_e, sd_s = syn.backbone_weights(32, seed=6)
    assert not fe.load_state_dict(sd_e, strict=False).unexpected_keys
    fs.load_state_dict(sd_s)
    for size in ((64, 96), (240, 320)):
        img = syn.make_images(2, size, seed=3)
        with torch.no_grad():
            want = fs(*fe(img))
        got = ob.backbone_features(fe.state_dict(), fs.state_dict(), img)
        assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want]
        assert all(torch.equal(g, w) for g, w in zip(got, want))
    assert tuple(got[3].shape[2:]) == (15, 20) and tuple(got[4].shape[2:]) == (8, 10)


def test_package_backbone_has_no_silent_stock_path():
    """MVSNet.forward with the package's own containers raises when the HIP backbone cannot take the call (here: CPU
    tensors) instead of running the stock modules; `native_backbone = False` is the explicit opt-in."""
    bb, syn, mvs = v3d('backbone'), v3d('synthetic'), v3d('mvsnet')
    Batch = v3d('batch').Batch
    fe, fs = bb.build_backbone(32)
    net = mvs.MVSNet(32, (64, 96), fe, fs).eval()
    edges, n_img = syn.make_edges(1, 1, 1)
    R, tv, K = syn.make_cameras(n_img, (64, 96), seed=5)
    b = Batch(syn.make_images(n_img, (64, 96), seed=2), R, tv, K, None, edges)
    with pytest.raises(v3d('_lib').V3DLibraryError, match='native_backbone = False'):
        net(b, 0.5, 0.05, 8, (16, 24))
    nb = bb.NativeBackbone(fe, fs)
    assert nb.is_package_pair() and 'HIP device' in nb.why_not(b.images)
    assert not bb.NativeBackbone(torch.nn.Identity(), fs).is_package_pair()


@pytest.mark.gpu
def test_mvsnet_forward_from_images(cuda):
    """MVSNet.forward(batch.images) end to end (mvsnet.py:176-229): HIP backbone -> HIP cost volume; the depth must
    equal the cost-volume path fed with the same quarter features, and match the oracle run on those features."""
    import numpy as np
    from oracle import costvolume as ocv
    syn, mvs, bb = v3d('synthetic'), v3d('mvsnet'), v3d('backbone')
    Batch = v3d('batch').Batch
    img_size, plane_size = (128, 160), (32, 40)
    edges, n_img = syn.make_edges(2, 1, 1)
    R, tv, K = syn.make_cameras(n_img, img_size, seed=5)
    fe, fs = bb.build_backbone(32)
    sd_e, sd_s = syn.backbone_weights(32, seed=6)
    fe.load_state_dict(sd_e, strict=False)
    fs.load_state_dict(sd_s)
    sd = syn.costregnet_weights(seed=0, sharpen=1.0)
    net = mvs.MVSNet(32, img_size, fe, fs).eval()
    net.cnn_3d.load_state_dict(sd, strict=False)
    net = net.to(cuda)
    b = Batch(syn.make_images(n_img, img_size, seed=2), R, tv, K, None, edges).to(cuda)
    with torch.no_grad():
        depth, f_half, f_quarter, f_eighth = net(b, 0.5, 0.05, 32, plane_size)
        assert tuple(f_half.shape) == (n_img, 32, 64, 80) and tuple(f_quarter.shape) == (n_img, 32, 32, 40)
        assert tuple(f_eighth.shape) == (n_img, 32, 16, 20)
        depth2, var, reg = net.cost_volume_depth(f_quarter, b, 0.5, 0.05, 32, plane_size, return_intermediates=True)
        depth_o, var_o, reg_o = ocv.mvsnet_depth(f_quarter.cpu(), R, tv, K, edges, sd, 0.5, 0.05, 32, img_size, plane_size)
    assert torch.equal(depth, depth2)
    # random-init backbone features are not O(1) like the U[0,1) features of the other tests: tolerances relative to the
    # volumes' ranges (the unsharpened soft-argmin keeps the depth well conditioned)
    np.testing.assert_allclose(var.cpu().numpy(), var_o.numpy(), rtol=0, atol=1e-4 * float(var_o.abs().max()))
    np.testing.assert_allclose(reg.cpu().numpy(), reg_o.numpy(), rtol=0, atol=2e-4 * float(reg_o.abs().max()))
    # random-init features saturate the depth softmax (a near-argmax over noise): an isolated pixel may flip between two
    # planes on a 1e-5 difference of the regularised volume, so the depth is checked on all but a handful of pixels
    rel = (depth.cpu() - depth_o).abs() / depth_o
    assert float((rel < 1e-4).float().mean()) > 0.995


@pytest.mark.gpu
def test_backbone_device_arithmetic_matches_cpu(cuda):
    """The only pin available for row 8f-3 without torchvision: the backbone's arithmetic on the device (MIOpen / rocBLAS
    through stock PyTorch modules) against the same modules on the CPU, seeded weights and images, at the cfg2 image size:
    all five FPN outputs within 1e-5 of their range (fp32 convolutions, summation order differs)."""
    bb, syn = v3d('backbone'), v3d('synthetic')
    fe, fs = bb.build_backbone(32)
    sd_e, sd_s = syn.backbone_weights(32, seed=6)
    assert not fe.load_state_dict(sd_e, strict=False).unexpected_keys
    fs.load_state_dict(sd_s)
    fe, fs = fe.eval(), fs.eval()
    img = syn.make_images(3, (256, 320), seed=4)
    with torch.no_grad():
        cpu_maps = fe(img)
        cpu_out = fs(*cpu_maps)
        fe_d, fs_d = fe.to(cuda), fs.to(cuda)
        dev_maps = fe_d(img.to(cuda))
        dev_out = fs_d(*dev_maps)
    assert [tuple(o.shape) for o in dev_out] == [tuple(o.shape) for o in cpu_out]
    assert tuple(dev_out[1].shape) == (3, 32, 64, 80)                 # the quarter-resolution map the cost volume consumes
    for name, a, b in [('C%d' % (i + 1), x, y) for i, (x, y) in enumerate(zip(dev_maps, cpu_maps))] + \
                      [('P%d' % (i + 1), x, y) for i, (x, y) in enumerate(zip(dev_out, cpu_out))]:
        scale = float(b.abs().max())
        assert scale > 1e-3 and torch.isfinite(a).all(), name
        err = float((a.cpu() - b).abs().max()) / scale
        assert err < 1e-5, '%s: device vs CPU %.2e of range' % (name, err)


@pytest.mark.gpu
def test_native_backbone_matches_the_oracle(cuda):
    """csrc/backbone.hip through ``NativeBackbone`` (stem, depthwise and conv-as-GEMM kernels on channels-last activations,
    BatchNorm folded, residuals in the epilogues) against oracle/backbone.py on the CPU: all five pyramid outputs within 2e-5
    of their range at the cfg2 image size (exact-fp32 products, other summation orders), for an odd batch size, for a second
    image size and for the reference's default 240 x 320 (MVSNet(img_size=(240, 320)), mvsnet.py:167: odd 15 x 20 / 8 x 10
    levels); ``MVSNet.forward`` takes this path, and `native_backbone = False` (the explicit stock path) gives the same maps."""
    from oracle import backbone as ob
    bb, syn, mvs = v3d('backbone'), v3d('synthetic'), v3d('mvsnet')
    fe, fs = bb.build_backbone(32)
    sd_e, sd_s = syn.backbone_weights(32, seed=6)
    assert not fe.load_state_dict(sd_e, strict=False).unexpected_keys
    fs.load_state_dict(sd_s)
    fe, fs = fe.eval().to(cuda), fs.eval().to(cuda)
    # 'split_bf16' (the default, as everywhere in the package): fused inverted-residual blocks (csrc/irb.hip) on split-bf16 matrix
    # operands -- an operand pair (hi, lo) carries 16 mantissa bits, 2^-17 relative per operand, and the finest map sits behind 17
    # blocks and four top-down additions: 8e-5 of the range (measured: at most 4.0e-5 at these sizes, P1 at 240 x 320; 5.3e-5 over 30
    # random sizes, scripts/fuzz_backbone.py); 'fp32': the exact-fp32
    # three-launch blocks, 2e-5 (summation orders only).  The depth the cost volume makes of these features is held to the path's
    # 1e-4 in test_mvsnet_forward_from_images.
    for precision, bound in (('split_bf16', 8e-5), ('fp32', 2e-5)):
        nb = bb.NativeBackbone(fe, fs, precision=precision)
        for n, size, seed in ((3, (256, 320), 4), (2, (96, 160), 5), (2, (240, 320), 6), (1, (248, 328), 7)):
            img = syn.make_images(n, size, seed=seed)
            want = ob.backbone_features(fe.state_dict(), fs.state_dict(), img)
            with torch.no_grad():
                assert nb.why_not(img.to(cuda)) is None
                got = nb(img.to(cuda))
                got2 = nb(img.to(cuda))
            assert [tuple(o.shape) for o in got] == [tuple(o.shape) for o in want]
            for i, (a, b) in enumerate(zip(got, want)):
                scale = float(b.abs().max())
                assert scale > 1e-3 and torch.isfinite(a).all()
                err = float((a.cpu() - b).abs().max()) / scale
                print('backbone %s P%d at %s: %.2e of range' % (precision, i + 1, size, err))
                assert err < bound, 'P%d at %s (%s): native vs oracle %.2e of range' % (i + 1, size, precision, err)
                assert torch.equal(a, got2[i])                      # deterministic
    # sides that are not multiples of 8: a reason, not a silent second path
    assert 'multiples of 8' in nb.why_not(torch.zeros(1, 3, 244, 320, device=cuda))
    # the explicit stock path (MIOpen / rocBLAS) agrees with the kernels
    with torch.no_grad():
        stock = fs(*fe(img.to(cuda)))
    for a, b in zip(got, stock):
        assert float((a - b).abs().max()) < 2e-5 * float(b.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('taps,cin,cout,H,W,res_mode', [(1, 16, 48, 24, 40, 0), (1, 72, 24, 17, 23, 1), (9, 32, 32, 24, 40, 0),
                                                        (9, 40, 32, 16, 20, 2), (1, 24, 32, 64, 80, 2), (9, 8, 32, 9, 7, 0), (1, 96, 32, 15, 20, 2),
                                                        (9, 32, 32, 31, 41, 2)])
def test_conv_entry_point_against_the_plain_convolution(taps, cin, cout, H, W, res_mode, cuda):
    """v3d_conv_nhwc_f32 through the C ABI against torch's conv2d on the CPU: channel counts that are not multiples of 32, the image
    borders of the 3x3 taps, a ragged last row block, ReLU and both residual modes (same-resolution and nearest-upsampled from the
    ceil(H / 2) x ceil(W / 2) map, even and odd sides)."""
    import ctypes
    libm = v3d('_lib')
    lib = libm.load()
    g = torch.Generator().manual_seed(taps * 1000 + cin)
    n = 3
    w = (torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) * 0.1).contiguous()
    bias = torch.randn(cout, generator=g)
    wk = w.permute(0, 2, 3, 1).reshape(cout, taps * cin).contiguous()           # [co, tap * cin + c] (include/v3d.h: v3d_conv_pack)
    x = torch.randn(n, H, W, cin, generator=g).to(cuda)
    res = None
    if res_mode == 1:
        res = torch.randn(n, H, W, cout, generator=g).to(cuda)
    elif res_mode == 2:
        res = torch.randn(n, (H + 1) // 2, (W + 1) // 2, cout, generator=g).to(cuda)      # a stride-2 map: ceil(H / 2) x ceil(W / 2)
    handle = ctypes.c_void_p()
    libm.check(lib.v3d_conv_pack(wk.numpy().ctypes.data_as(libm.c_float_p), bias.numpy().ctypes.data_as(libm.c_float_p), cout, taps * cin,
                                 ctypes.byref(handle)), 'v3d_conv_pack')
    try:
        out = torch.empty(n, H, W, cout, device=cuda)
        libm.check(lib.v3d_conv_nhwc_f32(handle, x.data_ptr(), n, H, W, cin, taps, 1, res_mode,
                                         res.data_ptr() if res is not None else None, out.data_ptr(), libm.stream_ptr(cuda)),
                   'v3d_conv_nhwc_f32')
        torch.cuda.synchronize()
    finally:
        lib.v3d_conv_free(handle)
    assert torch.isfinite(out).all()
    # fp32 matrix instructions are an fmaf chain in another order than the CPU's: 1e-5 of the range
    ref = torch.nn.functional.conv2d(x.cpu().permute(0, 3, 1, 2), w, bias, padding=1 if taps == 9 else 0).relu()
    if res_mode == 1:
        ref = ref + res.cpu().permute(0, 3, 1, 2)
    elif res_mode == 2:
        ref = ref + torch.nn.functional.interpolate(res.cpu().permute(0, 3, 1, 2), size=(H, W), mode='nearest')
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max() < 1e-5 * ref.abs().max()



BLOCKS = [(16, 24, 3, 2, 3, 128, 160), (24, 24, 3, 1, 3, 64, 80), (24, 40, 5, 2, 3, 64, 80), (40, 40, 5, 1, 3, 32, 40), (40, 80, 5, 2, 6, 32, 40),
          (80, 80, 5, 1, 6, 16, 20), (80, 96, 3, 1, 6, 16, 20), (96, 96, 3, 1, 6, 16, 20),
          # 1/32 resolution: few tiles, the expanded channels of a tile shared out over slice groups + the reducing launch
          (96, 192, 5, 2, 6, 16, 20), (192, 192, 5, 1, 6, 8, 10), (192, 320, 3, 1, 6, 8, 10), (96, 192, 5, 2, 6, 15, 20),
          # the reference's default 240 x 320 (odd 15 x 20 level, ragged tiles) and sizes whose tiles hang over two borders
          (16, 24, 3, 2, 3, 120, 160), (24, 24, 3, 1, 3, 60, 80), (40, 40, 5, 1, 3, 30, 40), (80, 80, 5, 1, 6, 15, 20), (40, 80, 5, 2, 6, 30, 40),
          (24, 24, 3, 1, 3, 13, 19), (24, 40, 5, 2, 3, 21, 11), (40, 40, 5, 1, 3, 9, 9), (16, 24, 3, 2, 3, 24, 40)]


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,k,stride,expansion,H,W', BLOCKS)
def test_fused_inverted_residual_block(cin, cout, k, stride, expansion, H, W, cuda):
    """csrc/irb.hip through the C ABI (v3d_irb_pack / v3d_irb_supported / v3d_irb_nhwc_f32) against the block's module on the CPU
    (torchvision's _InvertedResidual restated, backbone.py): every block shape of the trunk the library has a fused kernel for, at the
    cfg2 sizes, at the 240 x 320 sizes (odd maps, ragged tiles) and at small odd sizes; random BatchNorm statistics; 2e-5 of the
    range (split-bf16 matrix operands, fp32 depthwise taps); repeated launches bit-identical; the three-launch path agrees."""
    bb = v3d('backbone')
    torch.manual_seed(cin * 100 + H)
    blk = bb._InvertedResidual(cin, cout, k, stride, expansion).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    x = torch.randn(3, H, W, cin)
    with torch.no_grad():
        ref = blk(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    fused = bb._Block(blk, cuda)
    assert fused.supported(H, W), 'no fused kernel for this block'
    y = fused(x.to(cuda))
    y2 = fused(x.to(cuda))
    torch.cuda.synchronize()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, 'fused block vs module: %.2e of range' % err
    assert torch.equal(y, y2)
    L = blk.layers
    wd, bd = bb._fold(L[3], L[4])
    three = bb._Gemm(*bb._fold(L[6], L[7]), cuda)(bb._Depthwise(wd, bd, stride, cuda)(bb._Gemm(*bb._fold(L[0], L[1]), cuda)(x.to(cuda), relu=True), relu=True),
                                                   relu=False, res=x.to(cuda) if blk.apply_residual else None, res_mode=1 if blk.apply_residual else 0)
    assert float((y - three).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.gpu
def test_fused_block_reports_what_it_cannot_take(cuda):
    """A block shape without a kernel instance: `supported` says so and the entry point refuses -- the caller (NativeBackbone) runs
    the three-launch path instead, nothing is silently approximated."""
    bb, libm = v3d('backbone'), v3d('_lib')
    blk = bb._InvertedResidual(64, 64, 3, 1, 2).eval()             # not a MnasNet-1.0 block
    fused = bb._Block(blk, cuda)
    assert not fused.supported(32, 40)
    with pytest.raises(libm.V3DLibraryError):
        fused(torch.zeros(1, 32, 40, 64, device=cuda))


@pytest.mark.gpu
@pytest.mark.parametrize('cin,H,W,coarse,want_inner', [(16, 128, 160, True, False), (24, 64, 80, True, True), (40, 32, 40, True, True),
                                                       (24, 60, 80, True, True), (16, 23, 37, True, True), (40, 8, 16, False, True),
                                                       (8, 5, 3, True, False), (48, 15, 20, True, True)])
def test_pyramid_level_kernel(cin, H, W, coarse, want_inner, cuda):
    """csrc/fpn.hip through the C ABI (v3d_fpn_pack / v3d_fpn_level_f32) against torch on the CPU: lateral 1x1 + bias + nearest-
    upsampled coarser inner map (ceil(H/2) x ceil(W/2), even and odd sides) -> 3x3 / pad 1 + bias, the result in the reference
    layout; ragged tiles, images smaller than a tile, the level without a coarser one, the inner map only where asked for; 2e-5 of the
    range (split-bf16 operands); repeated launches bit-identical."""
    bb = v3d('backbone')
    g = torch.Generator().manual_seed(cin * 7 + H)
    lat = torch.nn.Conv2d(cin, 32, 1)
    outc = torch.nn.Conv2d(32, 32, 3, padding=1)
    with torch.no_grad():
        for m in (lat, outc):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.2)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g))
    n = 3
    x = torch.randn(n, H, W, cin, generator=g)
    ci = torch.randn(n, (H + 1) // 2, (W + 1) // 2, 32, generator=g) if coarse else None
    with torch.no_grad():
        inner_ref = lat(x.permute(0, 3, 1, 2))
        if coarse:
            inner_ref = inner_ref + torch.nn.functional.interpolate(ci.permute(0, 3, 1, 2), size=(H, W), mode='nearest')
        out_ref = outc(inner_ref)
    assert bb._PyramidLevel.fits(lat, outc)
    lvl = bb._PyramidLevel(lat, outc, cuda)
    inner, out = lvl(x.to(cuda), ci.to(cuda) if coarse else None, want_inner)
    inner2, out2 = lvl(x.to(cuda), ci.to(cuda) if coarse else None, want_inner)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (n, 32, H, W) and torch.isfinite(out).all()
    assert float((out.cpu() - out_ref).abs().max() / out_ref.abs().max()) < 2e-5
    assert torch.equal(out, out2)
    if want_inner:
        assert float((inner.cpu().permute(0, 3, 1, 2) - inner_ref).abs().max() / inner_ref.abs().max()) < 2e-5
        assert torch.equal(inner, inner2)
    else:
        assert inner is None


@pytest.mark.gpu
@pytest.mark.parametrize('H,W', [(256, 320), (240, 320), (32, 48), (18, 14), (8, 8)])
def test_stem_block_kernel(H, W, cuda):
    """The trunk's first three layers as one kernel (v3d_stem_block_*: the stride-2 3x3 convolution's 27 taps gathered as the K
    dimension of the first matrix product, depthwise, pointwise) against ``FeatureExtractor.layer1`` on the CPU: image borders, ragged
    tiles, an image smaller than a tile; 2e-5 of the range; repeated launches bit-identical."""
    bb, syn = v3d('backbone'), v3d('synthetic')
    fe, _ = bb.build_backbone(32)
    sd_e, _ = syn.backbone_weights(32, seed=6)
    fe.load_state_dict(sd_e, strict=False)
    fe = fe.eval()
    img = syn.make_images(2, (H, W), seed=H)
    with torch.no_grad():
        ref = fe.layer1(img).permute(0, 2, 3, 1).contiguous()
    stem = bb._StemBlock(fe.layer1, cuda)
    y = stem(img.to(cuda))
    y2 = stem(img.to(cuda))
    torch.cuda.synchronize()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    assert float((y.cpu() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert torch.equal(y, y2)
