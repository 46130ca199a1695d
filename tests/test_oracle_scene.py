"""CPU: the oracle restatement of rows B1-B6, C1-C3, H1, H4 against golden vectors captured from the
reference's own Python, plus the dense-convolution cross-check of the MinkowskiEngine semantics
(row B6 is parity-unpinned against real ME, see oracle/__init__.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import v3d
from helpers import load_golden, t, weights_checksum
from oracle import scene as osc


def _sd(maker, g, seed_key, cs_key, **kw):
    sd = maker(seed=int(g[seed_key]), **kw)
    cs = weights_checksum(sd)
    assert abs(cs - float(g[cs_key])) <= 1e-9 * abs(cs)
    return sd


def test_pointcloud_B2():
    g = load_golden('B_pointcloud')
    pts, feat, batch = osc.feature_rich_pointcloud(
        t(g['depth']), t(g['depth_batch']), t(g['feat']), t(g['rotmats']), t(g['tvecs']), t(g['K']),
        t(g['edges']), tuple(int(v) for v in g['img_size']))
    np.testing.assert_allclose(pts.numpy(), g['pts'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(feat.numpy(), g['pts_feat'], rtol=0, atol=1e-6)
    assert np.array_equal(batch.numpy(), g['pts_batch'])


def test_pinned_backprojection_reproduces_the_reference_points_bit_for_bit():
    """oracle/pinned.py::backproject_points (the evaluation orders of the reference run behind the goldens, spelled out with
    elementwise ops: host-independent) against the reference's own points and hypothesis points (B_pointcloud / C_pointflow
    goldens): EQUAL.  One ulp of a point coordinate next to a voxel face changes the voxel set of the scene model, so the
    refinement leg is compared with this version of the oracle (oracle/net.py, pinned=True)."""
    b, g = load_golden('B_pointcloud'), load_golden('C_pointflow')
    args = (t(b['depth']), t(b['depth_batch']), t(b['feat']), t(b['rotmats']), t(b['tvecs']), t(b['K']), t(b['edges']))
    size = tuple(int(v) for v in b['img_size'])
    pts = osc.feature_rich_pointcloud(*args, size, pinned=True)[0]
    assert np.array_equal(pts.numpy(), b['pts'])
    hyp = osc.pointflow_hypotheses(*args, float(g['offset']), int(g['n']), size, pinned=True)[0]
    assert np.array_equal(hyp.numpy(), g['pts_hyp'])


def test_voxelize_B3():
    g = load_golden('B_voxelize')
    a_pts, a_idx, a_batch, a_edges = osc.voxelize(t(g['pts']), t(g['pts_batch']), float(g['edge_len']))
    assert np.array_equal(a_idx.numpy(), g['anchor_idx3d'])           # integer work: exact
    assert np.array_equal(a_batch.numpy(), g['anchor_batch'])
    assert np.array_equal(a_edges.numpy(), g['anchor_pts_edges'])
    np.testing.assert_allclose(a_pts.numpy(), g['anchor_pts'], rtol=0, atol=1e-6)
    assert g['anchor_batch'].max() == 1 and a_idx.min() == 0


def test_pointnet_B4():
    g = load_golden('B_pointnet')
    sd = _sd(v3d('synthetic').pointnet_weights, g, 'weights_seed', 'weights_checksum')
    out = osc.pointnet(t(g['x_in']), t(g['idx']), int(g['n_idx']), sd)
    np.testing.assert_allclose(out.numpy(), g['out'], rtol=1e-5, atol=1e-5)


def test_pointflow_C1_C3():
    g = load_golden('C_pointflow')
    b = load_golden('B_pointcloud')
    pts_hyp, pts_feat, pts_batch = osc.pointflow_hypotheses(
        t(b['depth']), t(b['depth_batch']), t(b['feat']), t(b['rotmats']), t(b['tvecs']), t(b['K']),
        t(b['edges']), float(g['offset']), int(g['n']), tuple(int(v) for v in b['img_size']))
    np.testing.assert_allclose(pts_hyp.numpy(), g['pts_hyp'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts_feat.numpy(), g['pts_feat'], rtol=0, atol=1e-6)
    assert np.array_equal(pts_batch.numpy(), g['pts_batch'])
    preds = torch.softmax(pts_feat.sum(-1) * 20.0, dim=1)
    off = osc.offset_expectation(preds, float(g['offset']), int(g['n']), b['depth'].shape)
    np.testing.assert_allclose(off.numpy(), g['offset_pred'], rtol=1e-5, atol=1e-7)


def test_decoder_net_C2b():
    g = load_golden('C_decoder_net')
    sd = _sd(v3d('synthetic').decoder_weights, g, 'weights_seed', 'weights_checksum', in_dim=352,
             h_dim=128, sharpen=float(g['sharpen']))
    preds = osc.decoder_net(t(g['features']), sd)
    np.testing.assert_allclose(preds.numpy(), g['preds'], rtol=1e-4, atol=1e-6)
    assert g['preds'].max() > 0.5          # the softmax is peaked: the check is not vacuous


def test_sparse_interpolation_C2a_against_reference_forloop():
    """The restated sparse trilinear interpolation + decoder equals the reference's own dense
    formulation HypothesisDecoder.forward_forloop (refinement.py:46-97) on the same sparse levels."""
    g = load_golden('C_forloop')
    syn = v3d('synthetic')
    sd_un = _sd(syn.sparse_unet_weights, g, 'unet_seed', 'unet_checksum')
    sd_dec = _sd(syn.decoder_weights, g, 'dec_seed', 'dec_checksum', in_dim=320, h_dim=128,
                 sharpen=float(g['sharpen']))
    xs = osc.sparse_unet(t(g['x_pointnet']), t(g['anchor_pts']), t(g['anchor_idx3d']),
                         t(g['anchor_batch']), float(g['edge_len']), sd_un)
    feats = osc.decoder_features(xs, t(g['pts_hyp']), None, t(g['pts_batch']))
    assert feats.shape[-1] == 320
    preds = osc.decoder_net(feats, sd_dec)
    np.testing.assert_allclose(preds.numpy(), g['preds'], rtol=0, atol=2e-5)
    assert [x['stride'] for x in xs] == [4, 2, 1]
    assert [x['feats'].shape[1] for x in xs] == [128, 128, 64]


def test_misc_H1_H4():
    g = load_golden('H_misc')
    assert np.array_equal(osc.slice_edges(t(g['edges']), 3, 5, 0).numpy(), g['sliced'])
    np.testing.assert_allclose(float(osc.abs_rel(t(g['depth_pred']), t(g['depth_gt']))),
                               float(g['abs_rel']), rtol=1e-6)


# ---- B6: dense cross-check of the restated MinkowskiEngine semantics ---------

def _random_sparse(n=300, c=8, extent=12, seed=0, ts=1):
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
    w = kernel.view(3, 3, 3, kernel.shape[1], kernel.shape[2])            # [oz, oy, ox, ci, co]
    w = w.permute(4, 3, 2, 1, 0)                                          # [co, ci, ox, oy, oz]
    return w.permute(1, 0, 2, 3, 4).contiguous() if transpose else w.contiguous()


@pytest.mark.parametrize('stride', [1, 2])
def test_sparse_conv_equals_dense_conv(stride):
    coords, feats = _random_sparse()
    g = torch.Generator().manual_seed(1)
    kernel = torch.randn((27, 8, 5), generator=g)
    oc, of, ts = osc.sparse_conv(coords, feats, 1, kernel, stride)
    dense = F.conv3d(_dense(coords, feats, 1, 12), _dense_weight(kernel), stride=stride, padding=1)
    ix = oc[:, 1:] // ts
    ref = dense[oc[:, 0], :, ix[:, 0], ix[:, 1], ix[:, 2]]
    np.testing.assert_allclose(of.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert ts == stride
    if stride == 2:
        assert (oc[:, 1:] % 2 == 0).all() and oc.shape[0] < coords.shape[0]


def test_sparse_conv_transpose_equals_dense():
    fine, _ = _random_sparse(seed=3)
    coarse = osc.strided_coords(fine, 1)
    g = torch.Generator().manual_seed(2)
    cf = torch.randn((coarse.shape[0], 8), generator=g)
    kernel = torch.randn((27, 8, 5), generator=g)
    out, ts = osc.sparse_conv_transpose(coarse, cf, 2, kernel, fine)
    dense = F.conv_transpose3d(_dense(coarse, cf, 2, 6), _dense_weight(kernel, transpose=True), stride=2,
                               padding=1, output_padding=1)
    ref = dense[fine[:, 0], :, fine[:, 1], fine[:, 2], fine[:, 3]]
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert ts == 1


def test_sparse_interpolate_equals_dense_grid_sample():
    coords, feats = _random_sparse(c=6, seed=5, ts=2)
    g = torch.Generator().manual_seed(6)
    q = torch.rand((200, 3), generator=g) * 20.0
    qb = torch.cat((torch.zeros(200, 1), q), 1)
    out = osc.sparse_interpolate(coords, feats, 2, qb)
    # dense: lattice index = coord / ts, grid_sample on batch 0 with zero padding, align_corners
    vol = _dense(coords, feats, 2, 12)[0:1]                                 # [1, C, X, Y, Z]
    grid = (q / 2.0) / 11.0 * 2 - 1
    samp = F.grid_sample(vol, grid[None, None, None][..., [2, 1, 0]], mode='bilinear',
                         padding_mode='zeros', align_corners=True)
    np.testing.assert_allclose(out.numpy(), samp[0, :, 0, 0].T.numpy(), rtol=1e-4, atol=1e-5)


def test_propagation_net_next_row():
    """the oracle against the reference golden; the product module carries the reference's state_dict keys
    and has no CPU path (its arithmetic is the HIP library's: tests/test_parity_net_gpu.py checks it against this golden)."""
    g = load_golden('N_propagation')
    syn = v3d('synthetic')
    sd = _sd(syn.propagation_weights, g, 'weights_seed', 'weights_checksum', in_dim=33, h_dim=32)
    out = osc.propagation_net(t(g['features']), t(g['depth']), sd)
    np.testing.assert_allclose(out.numpy(), g['out'], rtol=1e-5, atol=1e-6)
    net = v3d('upsampling').PropagationNet(33, 32).eval()
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all('num_batches' in k for k in r.missing_keys)
    with pytest.raises(v3d('_lib').V3DLibraryError):
        net(t(g['features']), t(g['depth']))


def test_upsample_chain_stage3_oracle():
    """eval-3dvnet.py:101-125 restated: nearest resize + propagation_net at 1/4, 1/2 and full resolution keep shapes and stay
    inside the depth range of their 3x3 neighbourhoods (a convex combination)."""
    syn = v3d('synthetic')
    g = torch.Generator().manual_seed(3)
    depth = 1 + torch.rand((3, 7, 7), generator=g)
    guides = [torch.rand((3, 32, 8, 10), generator=g), torch.rand((3, 32, 16, 20), generator=g),
              torch.rand((3, 3, 32, 40), generator=g)]
    sds = [syn.propagation_weights(33, 32, 5), syn.propagation_weights(33, 32, 6), syn.propagation_weights(4, 32, 7)]
    ref = depth
    for sd, gd in zip(sds, guides):
        ref = F.interpolate(ref.unsqueeze(1), gd.shape[-2:], mode='nearest')
        ref = osc.propagation_net(gd, ref, sd)
    assert ref.shape == (3, 32, 40)
    assert float(ref.min()) >= float(depth.min()) - 1e-6 and float(ref.max()) <= float(depth.max()) + 1e-6


def test_results_format_and_metrics_next_row(tmp_path):
    """2D metrics against the reference's calc_2d_depth_metrics golden; preds.npz keys and
    the intrinsics rescale of mv3d/eval/main.py:74-101."""
    g = load_golden('H_misc')
    res = v3d('results')
    mets = res.depth_metrics_2d(t(g['depth_pred']), t(g['depth_gt']))
    for k, v in mets.items():
        np.testing.assert_allclose(float(v), float(g['met_' + k]), rtol=1e-5, atol=1e-8, err_msg=k)
    Batch = v3d('batch').Batch
    K = torch.tensor([[[100., 0., 50.], [0., 120., 40.], [0., 0., 1.]]]).repeat(5, 1, 1)
    b = Batch(torch.zeros(5, 3, 80, 100), torch.eye(3).repeat(5, 1, 1), torch.zeros(5, 3), K, None, None)
    rec = res.write_preds(str(tmp_path / 'preds.npz'), '/data/scene0707_00', np.ones((3, 40, 50), np.float32), b,
                          [1, 2, 3], np.arange(100, 105))
    z = np.load(str(tmp_path / 'preds.npz'))
    assert sorted(z.files) == ['K', 'depth_preds', 'img_idx', 'rotmats', 'scene', 'tvecs']
    assert str(z['scene']) == 'scene0707_00' and z['img_idx'].tolist() == [101, 102, 103]
    np.testing.assert_allclose(z['K'][0], [[50., 0., 25.], [0., 60., 20.], [0., 0., 1.]])
    assert float(K[1, 0, 0]) == 100.0          # the batch's own intrinsics are left untouched
