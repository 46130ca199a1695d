"""GPU parity net added in round 2: a bounded random-shape fuzz of path A, the HIP sparse convolutions
against a dense conv3d recipe with ONE-HOT kernels (a wrong offset <-> weight-index convention cannot hide behind random
weights), stage 3 of the scene driver on device, re-packing of cached weights, and the range checks of the fixed-size
device tables."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import v3d
from oracle import costvolume as ocv
from oracle import scene as osc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', range(4))
def test_costvolume_fuzz_random_shapes(case, cuda):
    """Four seeded random configurations (plane count, grid, feature / image sizes, camera count, ragged unsorted edge
    lists with 1..11 sources, random weights and depth ranges) against the pinned oracle (host-independent evaluation
    orders, oracle/pinned.py): variance 5e-7 abs, regularised volume 5e-5 of max, depth 1e-4 relative (north_star) for
    BOTH operand precisions, the exact-fp32 chain within 2e-5."""
    syn, mvs = v3d('synthetic'), v3d('mvsnet')
    Batch = v3d('batch').Batch
    rng = np.random.default_rng(1000 + case)
    D = int(rng.choice([8, 16, 24, 32]))
    h, w = int(rng.choice([8, 16, 24, 40])), int(rng.choice([8, 16, 24, 56]))
    Hf, Wf = int(rng.integers(12, 40)), int(rng.integers(12, 50))
    H, W = 4 * Hf, 4 * Wf
    n_img = int(rng.integers(3, 14))
    R, tv, K = syn.make_cameras(n_img, (H, W), seed=int(rng.integers(1 << 30)), yaw_step_deg=float(rng.uniform(2, 12)))
    feat = syn.make_features(n_img, 32, Hf, Wf, seed=int(rng.integers(1 << 30)))
    refs, srcs = [], []
    for r in rng.choice(n_img, size=int(rng.integers(1, min(n_img, 4) + 1)), replace=False):
        ns = int(rng.integers(1, 12))
        refs += [int(r)] * ns
        srcs += [int(x) for x in rng.integers(0, n_img, ns)]
    perm = rng.permutation(len(refs))
    edges = torch.tensor([refs, srcs])[:, perm]
    sd = syn.costregnet_weights(seed=int(rng.integers(1 << 30)), sharpen=200.0)
    d0, dd = float(rng.uniform(0.3, 1.0)), float(rng.uniform(0.02, 0.2))
    with torch.no_grad():
        depth_o, var_o, reg_o = ocv.mvsnet_depth(feat, R, tv, K, edges, sd, d0, dd, D, (H, W), (h, w), pinned=True)
        net = mvs.MVSNet(32, (H, W)).eval()
        net.cnn_3d.load_state_dict(sd, strict=False)
        net = net.to(cuda)
        b = Batch(None, R, tv, K, None, edges).to(cuda)
        depth, var, reg = net.cost_volume_depth(feat.to(cuda), b, d0, dd, D, (h, w), return_intermediates=True)
        depth2 = net.cost_volume_depth(feat.to(cuda), b, d0, dd, D, (h, w))
        depth32 = net.cost_volume_depth(feat.to(cuda), b, d0, dd, D, (h, w), precision='fp32')
    torch.cuda.synchronize()
    assert torch.equal(depth, depth2)                      # split-variance hand-off == fp32-variance hand-off
    np.testing.assert_allclose(var.cpu().numpy(), var_o.numpy(), rtol=0, atol=5e-7)
    np.testing.assert_allclose(reg.cpu().numpy(), reg_o.numpy(), rtol=0, atol=5e-5 * float(reg_o.abs().max()))
    np.testing.assert_allclose(depth.cpu().numpy(), depth_o.numpy(), rtol=1e-4, atol=0)
    np.testing.assert_allclose(depth32.cpu().numpy(), depth_o.numpy(), rtol=2e-5, atol=0)


# ---- B6 through the HIP path with one-hot kernels ---------------------------------------------------------------------

def _random_sparse(n=400, c=16, extent=12, seed=0, ts=1):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randint(0, extent, (n, 3), generator=g) * ts
    b = torch.randint(0, 2, (n, 1), generator=g)
    coords = torch.unique(torch.cat((b, xyz), 1), dim=0)
    feats = torch.randn((coords.shape[0], c), generator=g)
    return coords, feats


def _dense(coords, feats, ts, extent):
    vol = torch.zeros((2, feats.shape[1], extent, extent, extent))
    ix = coords[:, 1:] // ts
    vol[coords[:, 0], :, ix[:, 0], ix[:, 1], ix[:, 2]] = feats
    return vol


def _dense_weight(kernel, transpose=False):
    # Wt[co, ci, ox+1, oy+1, oz+1] = kernel[k(o), ci, co], k = (ox+1) + 3 (oy+1) + 9 (oz+1)
    w = kernel.view(3, 3, 3, kernel.shape[1], kernel.shape[2]).permute(4, 3, 2, 1, 0)     # [co, ci, ox, oy, oz]
    return w.permute(1, 0, 2, 3, 4).contiguous() if transpose else w.contiguous()


def _hip_sparse_conv(sm, kernel, in_coords, in_feats, in_ts, out_coords, step, cuda):
    """out[p] = sum_k in[row of (out_coords[p] + step * o_k)] @ kernel[k] through SparseLevel.neighbors + the gather-GEMM
    (no norm, no ReLU): exactly what SparseUNet._conv feeds the C ABI with."""
    _, ci, co = kernel.shape
    lv = sm.SparseLevel(in_coords.int().to(cuda), in_ts)
    oc = out_coords.int().to(cuda).contiguous()
    nbr = lv.neighbors(oc, step)
    pack = sm.PackedGemm(kernel, ci * co, 1, co, 27, co, ci)
    x = in_feats.to(cuda).contiguous()
    n_out = oc.shape[0]
    idxs = [nbr.data_ptr() + 4 * k * n_out for k in range(27)]
    out = {pr: pack(n_out, [x] * 27, idxs=idxs, precision=pr).cpu() for pr in ('split_bf16', 'fp32')}
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('mode', ['conv_s1', 'conv_s2', 'conv_transpose_s2'])
def test_hip_sparse_conv_one_hot_kernels_vs_dense(mode, cuda):
    """For every kernel offset k: a kernel that is zero except kernel[k] = a non-symmetric [Ci, Co] matrix.  The HIP path
    (hash neighbour table + 27-segment gather-GEMM) must equal the dense conv3d / conv_transpose3d recipe of SURVEY
    Appendix A sampled at the active output coordinates -- per offset, so a permuted offset order, a flipped axis or a
    transposed [Ci, Co] slab fails for the offsets it affects."""
    sm = v3d('scenemodeling')
    ci, co = 16, 16
    g = torch.Generator().manual_seed(7)
    wmat = torch.randn((ci, co), generator=g)                    # non-symmetric
    if mode == 'conv_transpose_s2':
        fine, _ = _random_sparse(seed=3, c=ci)
        coarse = osc.strided_coords(fine, 1)
        feats = torch.randn((coarse.shape[0], ci), generator=g)
        in_coords, in_ts, out_coords, step = coarse, 2, fine, -1
    else:
        in_coords, feats = _random_sparse(seed=5, c=ci)
        in_ts = 1
        out_coords = in_coords if mode == 'conv_s1' else osc.strided_coords(in_coords, 1)
        step = 1
    for k in range(27):
        kernel = torch.zeros((27, ci, co))
        kernel[k] = wmat
        got = _hip_sparse_conv(sm, kernel, in_coords, feats, in_ts, out_coords, step, cuda)
        if mode == 'conv_transpose_s2':
            dense = F.conv_transpose3d(_dense(in_coords, feats, 2, 6), _dense_weight(kernel, transpose=True), stride=2,
                                       padding=1, output_padding=1)
            ref = dense[out_coords[:, 0], :, out_coords[:, 1], out_coords[:, 2], out_coords[:, 3]]
        else:
            s = 1 if mode == 'conv_s1' else 2
            dense = F.conv3d(_dense(in_coords, feats, 1, 12), _dense_weight(kernel), stride=s, padding=1)
            ix = out_coords[:, 1:] // s
            ref = dense[out_coords[:, 0], :, ix[:, 0], ix[:, 1], ix[:, 2]]
        assert float(ref.abs().max()) > 0.1, 'offset %d reaches no voxel: the test would be vacuous' % k
        np.testing.assert_allclose(got['fp32'].numpy(), ref.numpy(), rtol=1e-5, atol=1e-5, err_msg='offset %d' % k)
        np.testing.assert_allclose(got['split_bf16'].numpy(), ref.numpy(), rtol=0, atol=2e-4, err_msg='offset %d' % k)


# ---- stage 3 of the scene driver on device (eval-3dvnet.py:101-125) ---------------------------------------------------

def test_process_scene_with_upsampling_matches_oracle_chain(cuda):
    """process_scene(upsample=True) on the HIP path: stages 1-2 as in test_driver, then the three nearest + PropagationNet
    steps guided by the quarter / half features kept from stage 1 and by the images (features_half must come
    from stage 1, not from a batch attribute the Batch class does not have).  Checked against the oracle-backed driver
    followed by the oracle's propagation_net chain."""
    from oracle_net import OracleNet
    import test_driver as td
    syn, lm, drv = v3d('synthetic'), v3d('lightningmodel'), v3d('eval_3dvnet')
    cr, pn, un, dec = td.weights()
    net = lm.PL3DVNet(None, td.CFG, 0.16, feat_dim=32, img_size=td.IMG).eval()
    net.mvsnet.cnn_3d.load_state_dict(cr, strict=False)
    net.pointnet.load_state_dict(pn)
    net.sparse_conv.load_state_dict(un)
    net.decoder.load_state_dict(dec, strict=False)
    sds = [syn.propagation_weights(33, 32, 5), syn.propagation_weights(33, 32, 6), syn.propagation_weights(4, 32, 7)]
    for m, sd in zip((net.refine_quarter, net.refine_half, net.refine_full), sds):
        m.load_state_dict(sd, strict=False)
    net = net.to(cuda)
    scene = td.make_scene()
    n_img = scene.rotmats.shape[0]
    scene.features_half = syn.make_features(n_img, 32, 2 * td.FEAT[0], 2 * td.FEAT[1], seed=42)
    scene.images = syn.make_images(n_img, td.IMG, seed=43)
    out = drv.process_scene(scene, net, 1, cuda, td.CFG, td.OFFSETS, 2, 3, upsample=True)
    assert tuple(out.shape) == (5,) + td.IMG
    # oracle: refined plane-grid depths, then the stage-3 chain of the reference
    ref = td.run_oracle()
    k, n = 1, 5
    guides = [scene.features_quarter[k:k + n], scene.features_half[k:k + n], scene.images[k:k + n]]
    with torch.no_grad():
        for sd, gd in zip(sds, guides):
            ref = F.interpolate(ref.unsqueeze(1), gd.shape[-2:], mode='nearest')
            ref = osc.propagation_net(gd, ref, sd)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=0)
    # without half-resolution features stage 3 must fail loudly, not with an AttributeError deep inside
    scene2 = td.make_scene()
    scene2.images = scene.images
    with pytest.raises(ValueError):
        drv.process_scene(scene2, net, 1, cuda, td.CFG, td.OFFSETS, 2, 3, upsample=True)


def test_hip_propagation_net_matches_reference_golden_and_oracle_chain(cuda):
    """SURVEY 8f rank 2 on the HIP path (v3d_propagation_f32: four 3x3 conv layers on split-bf16 matrix cores + the fused
    softmax / 3x3 propagation kernel): (1) the reference golden N_propagation (33 guide+depth channels); (2) the stage-3
    chain of eval-3dvnet.py:101-125 -- 1/4, 1/2 and full resolution, 33 / 33 / 4 input channels, image sizes that are not
    multiples of the 4 x 14 tiles, chunked -- against the oracle chain; (3) chunking does not change bits."""
    import test_oracle_scene as tos
    syn, up = v3d('synthetic'), v3d('upsampling')
    g = tos.load_golden('N_propagation')
    sd = tos._sd(syn.propagation_weights, g, 'weights_seed', 'weights_checksum', in_dim=33, h_dim=32)
    net = up.PropagationNet(33, 32).eval()
    net.load_state_dict(sd, strict=False)
    net = net.to(cuda)
    with torch.no_grad():
        out = net(torch.from_numpy(g['features']).to(cuda), torch.from_numpy(g['depth']).to(cuda))
    np.testing.assert_allclose(out.cpu().numpy(), g['out'], rtol=2e-5, atol=0)
    gen = torch.Generator().manual_seed(3)
    depth = 1 + torch.rand((5, 7, 9), generator=gen)
    guides = [torch.rand((5, 32, 15, 19), generator=gen), torch.rand((5, 32, 30, 38), generator=gen),
              torch.rand((5, 3, 60, 76), generator=gen)]
    sds = [syn.propagation_weights(33, 32, 5), syn.propagation_weights(33, 32, 6), syn.propagation_weights(4, 32, 7)]
    nets = []
    for sdi, cin in zip(sds, (33, 33, 4)):
        n = up.PropagationNet(cin, 32).eval()
        n.load_state_dict(sdi, strict=False)
        nets.append(n.to(cuda))
    with torch.no_grad():
        out = up.upsample_depth(depth.to(cuda), [(n, gd.to(cuda)) for n, gd in zip(nets, guides)], chunk=2)
        whole = up.upsample_depth(depth.to(cuda), [(n, gd.to(cuda)) for n, gd in zip(nets, guides)], chunk=100)
        ref = depth
        for sdi, gd in zip(sds, guides):
            ref = F.interpolate(ref.unsqueeze(1), gd.shape[-2:], mode='nearest')
            ref = osc.propagation_net(gd, ref, sdi)
    assert torch.equal(out, whole)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=0)
    assert out.shape == (5, 60, 76)
    # round 6 (csrc/propz.hip): a stage is ONE row-marching kernel with the nearest resize in its addressing.  (a) the folded
    # resize equals torch's resize followed by forward(); (b) views are independent: any subset of the batch gives the same
    # bits; (c) the per-layer kernels of round 4 (developer option prop_fused = 0: other summation orders) agree to 2e-5
    libm = v3d('_lib')
    with torch.no_grad():
        d0 = depth.to(cuda)
        for n, gd in zip(nets, guides):
            gd = gd.to(cuda)
            folded = n.forward_resized(gd, d0)
            resized = F.interpolate(d0.unsqueeze(1), gd.shape[-2:], mode='nearest')
            assert torch.equal(folded, n(gd, resized))
            assert torch.equal(n.forward_resized(gd[1:4], d0[1:4]), folded[1:4])
            old = libm.set_option('prop_fused', 0)
            try:
                per_layer = n(gd, resized)
            finally:
                libm.set_option('prop_fused', old)
            np.testing.assert_allclose(folded.cpu().numpy(), per_layer.cpu().numpy(), rtol=2e-5, atol=0)
            d0 = folded
        assert torch.equal(d0, out)
        # widths / heights that are not multiples of the 40-column strips, a single row, a single column strip of 3 columns
        for (hh, ww) in ((1, 40), (3, 3), (17, 41), (9, 83)):
            gd = torch.rand((2, 3, hh, ww), generator=gen)
            dd = 1 + torch.rand((2, 1, hh, ww), generator=gen)
            got = nets[2](gd.to(cuda), dd.to(cuda))
            np.testing.assert_allclose(got.cpu().numpy(), osc.propagation_net(gd, dd, sds[2]).numpy(), rtol=2e-5, atol=0)
    assert float((ref - F.interpolate(depth.unsqueeze(1), (60, 76), mode='nearest')[:, 0]).abs().max()) > 1e-3    # it did something
    # feat_dim = 16 (the reference's signature default): 17 guide + depth channels
    sd17 = syn.propagation_weights(17, 32, 8)
    n17 = up.PropagationNet(17, 32).eval()
    n17.load_state_dict(sd17, strict=False)
    g17, d17 = torch.rand((3, 16, 30, 38), generator=gen), 1 + torch.rand((3, 1, 30, 38), generator=gen)
    with torch.no_grad():
        o17 = n17.to(cuda)(g17.to(cuda), d17.to(cuda))
        r17 = osc.propagation_net(g17, d17, sd17)
    np.testing.assert_allclose(o17.cpu().numpy(), r17.numpy(), rtol=2e-5, atol=0)
    # the reference's arithmetic type (round 6): the same kernel on exact-fp32 matrix instructions -- golden, the three-resolution
    # chain with the folded resize, and the 17-channel net, an order of magnitude inside the split-bf16 bound
    net32 = up.PropagationNet(33, 32, precision='fp32').eval()
    net32.load_state_dict(sd, strict=False)
    with torch.no_grad():
        o32 = net32.to(cuda)(torch.from_numpy(g['features']).to(cuda), torch.from_numpy(g['depth']).to(cuda))
        np.testing.assert_allclose(o32.cpu().numpy(), g['out'], rtol=3e-6, atol=0)
        d32 = depth.to(cuda)
        for sdi, cin, gd in zip(sds, (33, 33, 4), guides):
            n32 = up.PropagationNet(cin, 32, precision='fp32').eval()
            n32.load_state_dict(sdi, strict=False)
            d32 = n32.to(cuda).forward_resized(gd.to(cuda), d32)
        np.testing.assert_allclose(d32.cpu().numpy(), ref.numpy(), rtol=3e-6, atol=0)
        n17f = up.PropagationNet(17, 32, precision='fp32').eval()
        n17f.load_state_dict(sd17, strict=False)
        np.testing.assert_allclose(n17f.to(cuda)(g17.to(cuda), d17.to(cuda)).cpu().numpy(), r17.numpy(), rtol=3e-6, atol=0)


# ---- cached packed weights -----------------------------------------------------------------------------------------------

def test_packed_weights_follow_replaced_parameters(cuda):
    """`p.data = ...` and load_state_dict(assign=True) do not bump tensor._version; the packed MFMA weight
    image must still be rebuilt."""
    syn, mvs = v3d('synthetic'), v3d('mvsnet')
    sd_a, sd_b = syn.costregnet_weights(seed=1), syn.costregnet_weights(seed=2)
    net = mvs.CostRegNet(32, 8).eval()
    net.load_state_dict(sd_a, strict=False)
    net = net.to(cuda)
    x = torch.randn((1, 32, 8, 8, 16), generator=torch.Generator().manual_seed(0)).to(cuda)
    ya = net(x).clone()
    for name, p in net.named_parameters():
        p.data = sd_b[name].to(cuda)                       # new storage, version counter unchanged
    for name, buf in net.named_buffers():
        if name in sd_b:
            buf.data = sd_b[name].to(cuda)
    yb = net(x).clone()
    ref = mvs.CostRegNet(32, 8).eval()
    ref.load_state_dict(sd_b, strict=False)
    yb_ref = ref.to(cuda)(x)
    assert torch.equal(yb, yb_ref) and not torch.equal(ya, yb)
    net.load_state_dict({k: v.to(cuda) for k, v in sd_a.items()}, strict=False, assign=True)
    assert torch.equal(net(x), ya)


# ---- range checks of the fixed-size device tables -----------------------------------------------------------

def test_voxelize_rejects_out_of_range_batch_and_grid(cuda):
    utils, libm = v3d('utils'), v3d('_lib')
    g = torch.Generator().manual_seed(0)
    pts = torch.rand((500, 3), generator=g).to(cuda)
    ok = utils.voxelize(pts, torch.zeros(500, dtype=torch.long, device=cuda), 0.1)
    assert ok[0].shape[0] > 0
    bad_batch = torch.zeros(500, dtype=torch.long, device=cuda)
    bad_batch[7] = 1024                                      # the per-batch table has 1024 rows
    with pytest.raises(libm.V3DLibraryError, match='batch id'):
        utils.voxelize(pts, bad_batch, 0.1)
    with pytest.raises(libm.V3DLibraryError, match='cells per axis'):
        utils.voxelize(pts, torch.zeros(500, dtype=torch.long, device=cuda), 1e-6)
    nan_pts = pts.clone()
    nan_pts[3, 1] = float('nan')
    with pytest.raises(libm.V3DLibraryError):
        utils.voxelize(nan_pts, torch.zeros(500, dtype=torch.long, device=cuda), 0.1)


def test_hash_build_reports_coordinates_outside_the_key_range(cuda):
    sm, libm = v3d('scenemodeling'), v3d('_lib')
    lib = libm.load()
    coords = torch.tensor([[0, 1, 2, 3], [0, 65519, 0, 0], [1, -8, 5, 5]], dtype=torch.int32, device=cuda)
    lv = sm.SparseLevel(coords, 1)
    assert lib.v3d_hash_status(lv.table.data_ptr(), lv.n, libm.stream_ptr(cuda)) == 0
    bad = torch.tensor([[0, 1, 2, 3], [0, 70000, 0, 0]], dtype=torch.int32, device=cuda)
    lv2 = sm.SparseLevel(bad, 1)
    assert lib.v3d_hash_status(lv2.table.data_ptr(), lv2.n, libm.stream_ptr(cuda)) == -1      # V3D_ERR_BAD_SHAPE
    assert b'packed-key range' in lib.v3d_last_error()


# ---- fused hypothesis decoder ---------------------------------------------------------------------------

def test_fused_decoder_matches_unfused_chain_and_golden(cuda):
    """v3d_decoder_fused_f32 (interpolation -> 3 conv1d layers -> head -> softmax -> expectation in one kernel) against the
    5-launch chain (v3d_sparse_interp_f32 + 3 x v3d_gemm_gather_f32 + v3d_decoder_head_f32) on the C_forloop scene, with and
    without the per-point variance feature, for a point count that is not a multiple of the 8-point tile; and against the
    reference's own dense formulation (forward_forloop golden)."""
    from helpers import load_golden, t
    import test_scene_gpu as tsg
    syn, sm, rf = v3d('synthetic'), v3d('scenemodeling'), v3d('refinement')
    g, u = tsg._unet_inputs(cuda)
    net = sm.SparseUNet().eval()
    net.load_state_dict(syn.sparse_unet_weights(seed=int(g['unet_seed'])))
    xs = net.to(cuda)(u['F'], u['pts'], u['idx'], u['batch'], float(g['edge_len']))
    pts = t(g['pts_hyp']).to(cuda)[:667]                      # 667 = 83 tiles + 3 points
    pts_batch = t(g['pts_batch']).to(cuda)[:667]
    vals = torch.linspace(-0.15, 0.15, 7).to(cuda)
    for in_dim, seed in ((320, int(g['dec_seed'])), (352, 3)):
        dec = rf.HypothesisDecoder(in_dim, 128, 3, 1).eval()
        dec.load_state_dict(syn.decoder_weights(in_dim=in_dim, h_dim=128, seed=seed, sharpen=float(g['sharpen'])),
                            strict=False)
        dec = dec.to(cuda)
        assert dec.fused                                        # the default since round 3
        pf = None if in_dim == 320 else torch.rand((667, 7, 32), generator=torch.Generator().manual_seed(1)).to(cuda) * 0.1
        assert dec.can_fuse(xs, pts, pf)
        p_f, e_f = dec.decode_fused(xs, pts, pf, pts_batch, vals)
        p_f2, _ = dec.decode_fused(xs, pts, pf, pts_batch, vals)
        assert torch.equal(p_f, p_f2)                           # deterministic (see the repeated-launch test below)
        d0 = torch.rand(667, device=cuda)                        # the driver's `depth += offset` inside the kernel
        d1 = d0.clone()
        _, e_f3 = dec.decode_fused(xs, pts, pf, pts_batch, vals, depth_inout=d1)
        assert torch.equal(e_f3, e_f) and torch.equal(d1, d0 + e_f)
        p_u, e_u = dec.decode(dec.features(xs, pts, pf, pts_batch), vals)
        torch.cuda.synchronize()
        assert float(p_u.max()) > 0.5                      # peaked softmax: the comparison is not vacuous
        np.testing.assert_allclose(p_f.cpu().numpy(), p_u.cpu().numpy(), rtol=0, atol=1e-4)
        np.testing.assert_allclose(e_f.cpu().numpy(), e_u.cpu().numpy(), rtol=0, atol=2e-5)
        if in_dim == 320:
            np.testing.assert_allclose(p_f.cpu().numpy(), g['preds'][:667], rtol=0, atol=2e-4)
        assert torch.equal(dec(xs, pts, pf, pts_batch), p_f)   # forward() takes the fused kernel by default
        dec.fused = False
        assert not dec.can_fuse(xs, pts, pf)
        assert torch.equal(dec(xs, pts, pf, pts_batch), p_u)


@pytest.mark.parametrize('n_hyp', [1, 2, 5, 8])
def test_fused_decoder_other_hypothesis_counts_and_two_scenes(n_hyp, cuda):
    """The round-5 kernel owns 8 column slots per query point: hypothesis counts below 8 leave padding slots (their tap masks cut
    them off), 8 leaves none (the masks alone separate neighbouring points); 1 has no neighbours at all.  Fused path against the
    5-launch chain on the C_forloop scene (two batch elements), points far outside the scene included (all corners absent:
    bias-only features), a point count that is not a multiple of the 32-point tile, and the in-place depth update."""
    import test_scene_gpu as tsg
    syn, sm, rf = v3d('synthetic'), v3d('scenemodeling'), v3d('refinement')
    g, u = tsg._unet_inputs(cuda)
    net = sm.SparseUNet().eval()
    net.load_state_dict(syn.sparse_unet_weights(seed=int(g['unet_seed'])))
    xs = net.to(cuda)(u['F'], u['pts'], u['idx'], u['batch'], float(g['edge_len']))
    from helpers import t
    base = t(g['pts_hyp']).to(cuda)[:333, 3]                      # centre hypothesis of 333 points
    pts_batch = t(g['pts_batch']).to(cuda)[:333]
    gen = torch.Generator().manual_seed(40 + n_hyp)
    step = torch.randn((333, 1, 3), generator=gen).to(cuda) * 0.03
    pts = base[:, None, :] + step * (torch.arange(n_hyp, device=cuda).float() - (n_hyp - 1) / 2)[None, :, None]
    pts[7] += 50.0                                                 # far outside the scene
    pf = (torch.rand((333, n_hyp, 32), generator=gen) * 0.1).to(cuda)
    vals = torch.linspace(-0.1, 0.1, n_hyp).to(cuda) if n_hyp > 1 else torch.tensor([0.05], device=cuda)
    dec = rf.HypothesisDecoder(352, 128, 3, 1).eval()
    dec.load_state_dict(syn.decoder_weights(in_dim=352, h_dim=128, seed=5, sharpen=20.0), strict=False)
    dec = dec.to(cuda)
    assert dec.can_fuse(xs, pts, pf)
    d0 = torch.rand(333, device=cuda)
    d1 = d0.clone()
    p_f, e_f = dec.decode_fused(xs, pts, pf, pts_batch, vals, depth_inout=d1)
    p_u, e_u = dec.decode(dec.features(xs, pts, pf, pts_batch), vals)
    torch.cuda.synchronize()
    assert tuple(p_f.shape) == (333, n_hyp) and torch.isfinite(p_f).all()
    np.testing.assert_allclose(p_f.sum(dim=1).cpu().numpy(), 1.0, rtol=0, atol=1e-5)
    np.testing.assert_allclose(p_f.cpu().numpy(), p_u.cpu().numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(e_f.cpu().numpy(), e_u.cpu().numpy(), rtol=0, atol=2e-5)
    assert torch.equal(d1, d0 + e_f)
    assert set(pts_batch.unique().tolist()) == {0, 1} or int(pts_batch.max()) == 0


def test_fused_decoder_is_deterministic_under_load(cuda):
    """Round 2 shipped the fused decoder behind a 96 KB LDS request because results changed from launch to launch with two
    waves per SIMD; round 3 traced that to vectorised (ds_read_b128) reads of a self-written LDS corner table under co-resident
    matrix instructions.  The round-5 kernel (column-owner waves, 8 waves per workgroup = two per SIMD, wave-private corner
    table read with 8-byte reads, LDS-DMA weight ring) must stay clean: a 6-view cfg3-shaped scene (18 816 query points = 588
    tiles, more than two per CU), 60 launches while a GEMM runs on a second stream: every launch bit-identical to the first, and
    within 2e-5 of the unfused chain (same products, another summation order: the chain sums k in 32-wide chunks on 16x16x32
    matrix instructions, the fused kernel in 16-wide steps on 32x32x16 ones, and its softmax sums are shuffle trees)."""
    syn, lm = v3d('synthetic'), v3d('lightningmodel')
    cfg = syn.CONFIGS['cfg3']
    n_ref, k = 6, 2
    edges, n_img = syn.make_edges(n_ref, k, k)
    rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5)
    feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(cuda)
    depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56))
    depth = (depth + 0.02 * torch.randn(depth.shape, generator=torch.Generator().manual_seed(1))).to(cuda)
    rot, tv, K, edges = rot.to(cuda), tv.to(cuda), K.to(cuda), edges.to(cuda)
    db = torch.zeros(n_ref, dtype=torch.long, device=cuda)
    net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
    net.pointnet.load_state_dict(syn.pointnet_weights())
    net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
    net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False)
    net = net.to(cuda)
    with torch.no_grad():
        xs = net.model_scene(depth, db, feat, rot, tv, K, edges)
        pts_hyp, pts_feat = lm.backproject_variance(depth, feat, rot, tv, K, edges, cfg['img_size'], offset=0.05, n=3)
        pb = db.unsqueeze(1).expand(n_ref, 3136).reshape(-1)
        vals = torch.linspace(-0.15, 0.15, 7).to(cuda)
        assert net.decoder.can_fuse(xs, pts_hyp, pts_feat)
        p_u, e_u = net.decoder.decode(net.decoder.features(xs, pts_hyp, pts_feat, pb), vals)
        side = torch.cuda.Stream()
        junk = torch.randn(2048, 2048, device=cuda)
        first = None
        for i in range(60):
            with torch.cuda.stream(side):
                junk = (junk @ junk).tanh_()
            p_f, e_f = net.decoder.decode_fused(xs, pts_hyp, pts_feat, pb, vals)
            torch.cuda.synchronize()
            if first is None:
                first = (p_f.clone(), e_f.clone())
            assert torch.equal(p_f, first[0]) and torch.equal(e_f, first[1]), 'launch %d differs from launch 0' % i
        assert float((first[0] - p_u).abs().max()) < 2e-5 and float((first[1] - e_u).abs().max()) < 5e-6


@pytest.mark.parametrize('case', range(7))
def test_device_edge_csr_equals_torch_unique_and_stable_sort(case, cuda):
    """v3d_edges_csr (the device-side replacement of mvsnet.py:179's torch.unique + the scatter grouping) against the
    torch construction, bit for bit: unsorted ragged edge lists, duplicate edges, self edges, references that are nobody's
    source, 1 .. 300 references, up to 5 000 edges (more than one 1024-thread pass)."""
    mvs = v3d('mvsnet')
    rng = np.random.default_rng(77 + case)
    n_img = [3, 17, 71, 400, 1200, 64, 5000][case]            # the last case is beyond the kernel's LDS-resident tables
    n_ref = [1, 5, 64, 300, 7, 64, 40][case]
    refs = rng.choice(n_img, size=n_ref, replace=False)
    r_list, s_list = [], []
    for r in refs:
        ns = int(rng.integers(1, [4, 12, 9, 20, 700, 12, 30][case]))
        r_list += [int(r)] * ns
        s_list += [int(x) for x in rng.integers(0, n_img, ns)]
    perm = rng.permutation(len(r_list))
    edges = torch.tensor([r_list, s_list], dtype=torch.int64)[:, perm].to(cuda)
    dev = mvs.edges_to_csr(edges, n_ref=n_ref, n_img=n_img).check()
    ref = mvs.edges_to_csr(edges)
    for a, b, name in zip(dev, ref, ('ref_idx', 'ref_img', 'edge_ofs', 'edge_src')):
        assert a.dtype == b.dtype and torch.equal(a, b), name


def test_device_edge_csr_reports_a_wrong_reference_count(cuda):
    """A wrong n_ref (or an index outside [0, n_img)) must not pass silently: the tables come back empty -- all offsets 0 and
    every reference slot naming image 0, also the slots beyond the references actually found, which the builder never wrote
    (the output tensors are poisoned first) -- so the warp kernel stays inside its buffers, and .check() raises.  The module
    keeps the tables of its last forward: MVSNet.check_edges() raises too."""
    mvs = v3d('mvsnet')
    edges = torch.tensor([[0, 0, 2, 2, 5], [1, 2, 0, 3, 4]], dtype=torch.int64, device=cuda)
    ok = mvs.edges_to_csr(edges, n_ref=3, n_img=6).check()
    assert ok[1].tolist() == [0, 2, 5] and ok[2].tolist() == [0, 2, 4, 5] and ok[3].tolist() == [1, 2, 0, 3, 4]
    real_empty = torch.empty

    def poisoned_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.dtype == torch.int32:
            t.fill_(0x7fffffff)
        return t
    for n_ref, n_img in ((2, 6), (4, 6), (3, 5), (6, 6)):
        torch.empty = poisoned_empty
        try:
            bad = mvs.edges_to_csr(edges, n_ref=n_ref, n_img=n_img)
        finally:
            torch.empty = real_empty
        assert bad[2].shape[0] == n_ref + 1 and int(bad[2].abs().sum()) == 0
        assert bad[1].shape[0] == n_ref and int(bad[1].abs().sum()) == 0
        with pytest.raises(RuntimeError):
            bad.check()
    # the product path: a wrong reference count yields a finite, all-zero variance volume and check_edges() raises
    syn = v3d('synthetic')
    Batch = v3d('batch').Batch
    inp = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=5)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval().to(cuda)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(cuda)
    with torch.no_grad():
        _, var, _ = net.cost_volume_depth(inp['feat'].to(cuda), b, d0, dd, D, inp['plane_size'], return_intermediates=True,
                                          n_ref=4)
    assert var.shape[0] == 4 and float(var.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        net.check_edges()
    with torch.no_grad():
        net.cost_volume_depth(inp['feat'].to(cuda), b, d0, dd, D, inp['plane_size'], n_ref=3)
    net.check_edges()


def test_cost_volume_depth_with_reference_count_hint_is_bit_identical(cuda):
    """MVSNet.cost_volume_depth(..., n_ref=) (device-built edge tables, no host synchronisation) returns the same bits as
    the torch.unique path."""
    syn, mvs = v3d('synthetic'), v3d('mvsnet')
    Batch = v3d('batch').Batch
    inp = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=5)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net = net.to(cuda)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(cuda)
    feat = inp['feat'].to(cuda)
    with torch.no_grad():
        a = net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        c = net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=3)
    assert torch.equal(a, c)


def test_cost_volume_graph_replay_equals_eager_and_follows_updates(cuda):
    """mvsnet.CostVolumeGraph (the step captured into a HIP graph): replay == eager bit for bit; after update() with other
    features / cameras of the same shapes the replay equals the eager result on those."""
    syn, mvs = v3d('synthetic'), v3d('mvsnet')
    Batch = v3d('batch').Batch
    inp = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=5)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net = net.to(cuda)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(cuda)
    feat = inp['feat'].to(cuda).clone()
    with torch.no_grad():
        eager = net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=3).clone()
    g = mvs.CostVolumeGraph(net, feat, b, d0, dd, D, inp['plane_size'], n_ref=3)
    assert torch.equal(g.replay(), eager)
    assert torch.equal(g.replay(), eager)
    inp2 = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=6)
    g.update(features_quarter=inp2['feat'].to(cuda), rotmats=inp2['rotmats'].to(cuda), tvecs=inp2['tvecs'].to(cuda))
    b2 = Batch(None, inp2['rotmats'], inp2['tvecs'], inp['K'], None, inp['edges']).to(cuda)
    with torch.no_grad():
        eager2 = net.cost_volume_depth(inp2['feat'].to(cuda), b2, d0, dd, D, inp['plane_size'], n_ref=3)
    assert not torch.equal(eager2, eager)
    assert torch.equal(g.replay(), eager2)


def test_cost_volume_graph_survives_workspace_growth_and_refuses_stale_weights(cuda):
    """The captured launches carry raw addresses.  (i) A later, LARGER eager call replaces the module's grow-only scratch
    buffers: the graph keeps its own alive and still replays the captured step bit for bit.  (ii) Changing a regulariser
    parameter releases the packed weight image the graph points to: replay() raises instead of running on freed memory, and a
    freshly captured graph works.  (iii) update(ref_src_edges=) with an edge list that no longer holds n_ref references
    raises."""
    syn, mvs = v3d('synthetic'), v3d('mvsnet')
    Batch = v3d('batch').Batch
    inp = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=5)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net = net.to(cuda)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(cuda)
    feat = inp['feat'].to(cuda).clone()
    g = mvs.CostVolumeGraph(net, feat, b, d0, dd, D, inp['plane_size'], n_ref=3)
    want = g.replay().clone()
    old_ptrs = {k: v.data_ptr() for k, v in list(net._ws._bufs.items()) + list(net.cnn_3d._ws._bufs.items())}
    big = syn.make_costvolume_inputs('cfg1', n_ref=7, seed=9)          # more views: every workspace grows
    bb = Batch(None, big['rotmats'], big['tvecs'], big['K'], None, big['edges']).to(cuda)
    with torch.no_grad():
        net.cost_volume_depth(big['feat'].to(cuda), bb, d0, dd, D, big['plane_size'], n_ref=7)
        junk = [torch.full((1 << 22,), float('nan'), device=cuda) for _ in range(8)]   # recycle whatever was freed
    new_ptrs = {k: v.data_ptr() for k, v in list(net._ws._bufs.items()) + list(net.cnn_3d._ws._bufs.items())}
    assert any(old_ptrs[k] != new_ptrs[k] for k in old_ptrs), 'the larger call was expected to replace a workspace'
    assert torch.equal(g.replay(), want)
    del junk
    assert not g.stale()
    with torch.no_grad():
        net.cnn_3d.prob.bias.add_(0.25)
    assert g.stale()
    with pytest.raises(RuntimeError):
        g.replay()
    g2 = mvs.CostVolumeGraph(net, feat, b, d0, dd, D, inp['plane_size'], n_ref=3)
    with torch.no_grad():
        assert torch.equal(g2.replay(), net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=3))
    with pytest.raises(RuntimeError):
        e = b.ref_src_edges.clone()
        e[0] = e[0, 0]                                                   # one reference image instead of three
        g2.update(ref_src_edges=e)


@pytest.mark.parametrize('n,n_seg,N', [(1000, 37, 128), (20000, 3000, 128), (513, 600, 64), (4097, 5, 256)])
def test_segment_csr_and_max_equal_scatter_amax(n, n_seg, N, cuda):
    """v3d_segment_csr + v3d_segment_max_f32 (PointNet's max-pool without atomics) against torch's scatter amax: the row list
    is the stable sort of the rows by segment id, the offsets delimit every segment (empty ones included), the pooled rows
    are bit-identical (max is order independent) and empty segments stay at -inf."""
    libm = v3d('_lib')
    lib = libm.load()
    gen = torch.Generator().manual_seed(n)
    seg = torch.randint(0, n_seg, (n,), generator=gen).int()
    if n_seg > 3:
        seg[seg == 2] = 1                                         # segment 2 is empty
    x = torch.randn((n, N + 4), generator=gen)                    # row stride > N
    segd, xd = seg.to(cuda), x.to(cuda)
    perm = torch.empty(n, dtype=torch.int32, device=cuda)
    offs = torch.empty(n_seg + 1, dtype=torch.int32, device=cuda)
    ws = torch.empty(lib.v3d_segment_csr_workspace_bytes(n), dtype=torch.uint8, device=cuda)
    st = libm.stream_ptr(cuda)
    libm.check(lib.v3d_segment_csr(segd.data_ptr(), n, n_seg, perm.data_ptr(), offs.data_ptr(), ws.data_ptr(), ws.numel(), st),
               'v3d_segment_csr')
    order = torch.sort(seg.long(), stable=True).indices
    assert torch.equal(perm.cpu().long(), order)
    counts = torch.bincount(seg.long(), minlength=n_seg)
    assert torch.equal(offs.cpu().long(), torch.cat((torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0))))
    out = torch.full((n_seg, N), 7.0, device=cuda)
    libm.check(lib.v3d_segment_max_f32(xd.data_ptr(), xd.stride(0), perm.data_ptr(), offs.data_ptr(), n_seg, N,
                                       out.data_ptr(), N, st), 'v3d_segment_max_f32')
    ref = torch.full((n_seg, N), float('-inf')).scatter_reduce_(0, seg.long().view(-1, 1).expand(-1, N), x[:, :N], 'amax')
    assert torch.equal(out.cpu(), ref)
    if n_seg > 3:
        assert bool(torch.isinf(out[2]).all())


@pytest.mark.gpu
@pytest.mark.parametrize('M', [7, 100, 9000])
def test_sparse_conv_without_any_neighbour_is_the_epilogue_of_zero(M, cuda):
    """A tile (here: every tile) whose 27 offsets are all absent runs no step at all on any of the three small-M kernels: the output
    is the epilogue of a zero accumulator -- GroupNorm of the bias (none: zeros -> the GroupNorm bias), + residual, ReLU."""
    sm, libm = v3d('scenemodeling'), v3d('_lib')
    C = N = 128
    g = torch.Generator().manual_seed(M)
    gn_w, gn_b = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    pk = sm.PackedGemm(torch.randn(27, C, N, generator=g) * 0.05, C * N, 1, N, 27, N, C, gn_w=gn_w, gn_b=gn_b)
    x = torch.randn(M, C, generator=g).to(cuda)
    nbr = torch.full((27, M), -1, dtype=torch.int32, device=cuda)
    res = torch.randn(M, N, generator=g).to(cuda)
    want = torch.relu(gn_b.to(cuda)[None, :] + res)
    for rounds, pipe in ((1, 1), (1, 0), (0, 0)):
        old, old_p = libm.set_option('gemm_rounds', rounds), libm.set_option('gemm_pipe', pipe)
        try:
            out = pk(M, [x] * 27, idxs=[nbr[k] for k in range(27)], use_gn=True, residual=res, relu_out=True)
        finally:
            libm.set_option('gemm_rounds', old), libm.set_option('gemm_pipe', old_p)
        torch.cuda.synchronize()
        assert torch.equal(out, want), (rounds, pipe)


@pytest.mark.gpu
def test_sparse_conv_entry_point_equals_the_gather_gemm_call(cuda):
    """v3d_sparse_conv_f32 (the U-Net's one-call convolution: source, neighbour table, stride) is v3d_gemm_gather_f32 with the 27
    segment arrays built in C: same bits, with GroupNorm, residual and ReLU; a short neighbour stride is refused."""
    sm, libm = v3d('scenemodeling'), v3d('_lib')
    lib = libm.load()
    M, C, N = 5000, 64, 128
    g = torch.Generator().manual_seed(11)
    w = torch.randn(27, C, N, generator=g) * 0.05
    pk = sm.PackedGemm(w, C * N, 1, N, 27, N, C, gn_w=torch.rand(N, generator=g) + 0.5, gn_b=torch.randn(N, generator=g) * 0.1)
    x = torch.randn(M + 7, C, generator=g).to(cuda)                 # more source rows than outputs (a strided convolution)
    nbr = torch.randint(0, M + 7, (27, M), generator=g)
    nbr[torch.rand(27, M, generator=g) < 0.5] = -1
    nbr = nbr.to(torch.int32).to(cuda).contiguous()
    res = torch.randn(M, N, generator=g).to(cuda)
    ref = pk(M, [x] * 27, idxs=[nbr[k] for k in range(27)], use_gn=True, residual=res, relu_out=True)
    out = torch.empty_like(ref)
    rc = lib.v3d_sparse_conv_f32(pk.handle, M, x.data_ptr(), C, nbr.data_ptr(), M, 16, 1e-5, res.data_ptr(), N, 1, out.data_ptr(), N,
                                 libm.precision_code('split_bf16'), libm.stream_ptr(cuda))
    libm.check(rc, 'v3d_sparse_conv_f32')
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    rc = lib.v3d_sparse_conv_f32(pk.handle, M, x.data_ptr(), C, nbr.data_ptr(), M - 1, 16, 1e-5, None, 0, 1, out.data_ptr(), N,
                                 libm.precision_code('split_bf16'), libm.stream_ptr(cuda))
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize('M,C,N', [(2816, 128, 128), (13500, 64, 64), (777, 32, 128), (33, 64, 32), (9001, 128, 128),
                                   (40000, 64, 64), (8200, 96, 64)])
def test_gather_gemm_rounds_kernel_bit_identical_to_one_step_kernel(M, C, N, cuda):
    """The sparse convolution's two small-M kernels -- the loader / matrix pipeline (gemm_gather_pipe_kernel, the default; 32-,
    64- and 128-row tiles by M and N) and the rounds of four (offset, K chunk) steps (gemm_gather_rounds_kernel) -- keep the
    step order and the MFMA order per accumulator of gemm_gather_kernel: same bits, with absent neighbours, whole absent
    offsets (skipped segments), a ragged last tile, GroupNorm + residual + ReLU in the epilogue."""
    sm = v3d('scenemodeling')
    g = torch.Generator().manual_seed(M)
    w = torch.randn(27, C, N, generator=g) * 0.05
    gn_w, gn_b = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    pk = sm.PackedGemm(w, C * N, 1, N, 27, N, C, gn_w=gn_w, gn_b=gn_b)
    x = torch.randn(M, C, generator=g).to(cuda)
    nbr = (torch.arange(M)[None, :] + torch.randint(-50, 51, (27, M), generator=g)).clamp_(0, M - 1)
    nbr[torch.rand(27, M, generator=g) < 0.6] = -1
    nbr[5] = -1                                    # an offset no row of any tile has
    nbr[13] = torch.arange(M)
    nbr = nbr.to(torch.int32).to(cuda).contiguous()
    res = torch.randn(M, N, generator=g).to(cuda)
    outs = {}
    for tag, rounds, pipe in (('pipe', 1, 1), ('rounds', 1, 0), ('one_step', 0, 0)):
        old = v3d('_lib').set_option('gemm_rounds', rounds)          # developer options (include/v3d.h: v3d_set_option)
        old_p = v3d('_lib').set_option('gemm_pipe', pipe)
        try:
            outs[tag] = pk(M, [x] * 27, idxs=[nbr[k] for k in range(27)], use_gn=True, residual=res, relu_out=True)
        finally:
            v3d('_lib').set_option('gemm_rounds', old)
            v3d('_lib').set_option('gemm_pipe', old_p)
    torch.cuda.synchronize()
    assert torch.isfinite(outs['rounds']).all()
    assert torch.equal(outs['rounds'], outs['one_step'])
    assert torch.equal(outs['pipe'], outs['one_step'])
