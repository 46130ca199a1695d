"""CPU: the oracle restatement reproduces the golden vectors captured from the reference's own
Python (tests/golden/make_golden.py).  Rows A1-A6."""
import numpy as np
import pytest
import torch

from conftest import v3d
from helpers import golden_costreg_weights, load_golden, t
from oracle import costvolume as ocv


def _run_tiny(g):
    sd = golden_costreg_weights(g)
    d0, dd, D = g['depth_cfg']
    return ocv.mvsnet_depth(t(g['feat']), t(g['rotmats']), t(g['tvecs']), t(g['K']), t(g['edges']),
                            sd, float(d0), float(dd), int(D), tuple(int(v) for v in g['img_size']),
                            tuple(int(v) for v in g['plane_size']))


@pytest.mark.parametrize('name', ['A_tiny_flat', 'A_tiny_sharp', 'A_tiny_rotated'])
def test_oracle_matches_reference_tiny(name):
    g = load_golden(name)
    with torch.no_grad():
        depth, var, reg = _run_tiny(g)
    # same torch CPU ops as the reference => essentially bit-exact
    np.testing.assert_allclose(var.numpy(), g['var'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(reg.numpy(), g['reg'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(depth.numpy(), g['depth'], rtol=1e-5, atol=0)


def test_rotated_fixture_exercises_mirroring_and_padding():
    """The 'rotated' fixture must really contain points behind a source camera (|z| mirroring,
    mvsnet.py:201) and samples falling outside the source image (zero padding)."""
    g = load_golden('A_tiny_rotated')
    d0, dd, D = g['depth_cfg']
    img_size = tuple(int(v) for v in g['img_size'])
    R, tv, K, e = t(g['rotmats']), t(g['tvecs']), t(g['K']), t(g['edges'])
    pts = ocv.plane_sweep_points(float(d0), float(dd), int(D), R, tv, K, img_size,
                                 tuple(int(v) for v in g['plane_size']))
    P = torch.bmm(K, torch.cat((R, tv[..., None]), 2))
    q = torch.bmm(P[e[1]], torch.cat((pts[e[0]], torch.ones(e.shape[1], 1, pts.shape[2])), 1))
    assert (q[:, 2] < 0).any()
    grid = ocv.project_to_grid(pts[e[0]], R, tv, K, e[1], img_size)
    assert (grid.abs() > 1).any()


@pytest.mark.parametrize('name,cfg', [('A_cfg1', 'cfg1')])
def test_oracle_matches_reference_cfg(name, cfg):
    g = load_golden(name)
    syn = v3d('synthetic')
    inp = syn.make_costvolume_inputs(cfg, n_ref=int(g['n_ref']))
    assert abs(float(inp['feat'].double().sum()) - float(g['feat_checksum'])) < 1e-6
    sd = golden_costreg_weights(g)
    d0, dd, D = inp['depth']
    with torch.no_grad():
        depth, var, reg = ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                           inp['edges'], sd, d0, dd, D, inp['img_size'],
                                           inp['plane_size'])
    np.testing.assert_allclose(var[:, ::4, ::3, ::5, ::7].numpy(), g['var_sub'], rtol=0, atol=1e-6)
    assert abs(float(var.double().sum()) - float(g['var_sum'])) < 1e-3
    np.testing.assert_allclose(reg[:, ::3, ::5, ::7].numpy(), g['reg_sub'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(depth.numpy(), g['depth'], rtol=1e-5, atol=0)


def test_oracle_matches_reference_cfg5():
    """BASELINE config 5 at full size: the oracle against the reference-generated golden (about 15 GB of host
    memory and a minute of CPU: the largest case of the CPU suite)."""
    g = load_golden('A_cfg5')
    syn = v3d('synthetic')
    inp = syn.make_costvolume_inputs('cfg5', n_ref=1)
    assert abs(float(inp['feat'].double().sum()) - float(g['feat_checksum'])) < 1e-6
    sd = golden_costreg_weights(g)
    d0, dd, D = inp['depth']
    with torch.no_grad():
        depth, var, reg = ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                           inp['edges'], sd, d0, dd, D, inp['img_size'],
                                           inp['plane_size'])
    np.testing.assert_allclose(var[:, ::4, ::7, ::11, ::13].numpy(), g['var_sub'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(reg[:, ::7, ::11, ::13].numpy(), g['reg_sub'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(depth[:, ::3, ::3].numpy(), g['depth_sub'], rtol=1e-5, atol=0)


@pytest.mark.parametrize('name', ['A_tiny_flat', 'A_tiny_rotated'])
def test_pinned_oracle_reproduces_golden_variance_bitwise_tiny(name):
    """oracle/pinned.py (host-independent evaluation orders) against the reference-generated goldens: the variance
    volume BIT FOR BIT -- including the rotated fixture with points behind a camera and out-of-image samples."""
    from oracle import pinned
    g = load_golden(name)
    d0, dd, D = g['depth_cfg']
    var = pinned.warp_variance(t(g['feat']), t(g['rotmats']), t(g['tvecs']), t(g['K']), t(g['edges']), float(d0),
                               float(dd), int(D), tuple(int(v) for v in g['img_size']),
                               tuple(int(v) for v in g['plane_size']))
    assert np.array_equal(var.numpy(), g['var'])
    fused = pinned.warp_variance(t(g['feat']), t(g['rotmats']), t(g['tvecs']), t(g['K']), t(g['edges']), float(d0),
                                 float(dd), int(D), tuple(int(v) for v in g['img_size']),
                                 tuple(int(v) for v in g['plane_size']), fused_square=True)
    np.testing.assert_allclose(fused.numpy(), g['var'], rtol=0, atol=3e-7)        # what the HIP kernels compute


@pytest.mark.parametrize('name,cfg,sub', [
    ('A_cfg1', 'cfg1', (slice(None), slice(None, None, 4), slice(None, None, 3), slice(None, None, 5), slice(None, None, 7))),
    ('A_cfg2', 'cfg2', (slice(None), slice(None, None, 4), slice(None, None, 5), slice(None, None, 7), slice(None, None, 7)))])
def test_pinned_oracle_reproduces_golden_variance_bitwise_cfg(name, cfg, sub):
    from oracle import pinned
    g = load_golden(name)
    inp = v3d('synthetic').make_costvolume_inputs(cfg, n_ref=1)
    d0, dd, D = inp['depth']
    var = pinned.warp_variance(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'], inp['edges'], d0, dd, D,
                               inp['img_size'], inp['plane_size'])
    assert np.array_equal(var.numpy()[sub], g['var_sub'])
    assert abs(float(var.double().sum()) - float(g['var_sum'])) < 1e-6


def test_pinned_oracle_reproduces_golden_variance_bitwise_cfg5():
    """The same bit-for-bit check at BASELINE config 5 (480x640, 192 planes, 11 edges, IEEE division by 11): the pinned
    oracle evaluated on exactly the voxels the reference-generated golden sub-samples (planes ::7, rows ::11, columns ::13,
    channels ::4) -- voxels are independent, so no full 472 MB volume is needed."""
    from oracle import pinned
    g = load_golden('A_cfg5')
    inp = v3d('synthetic').make_costvolume_inputs('cfg5', n_ref=1)
    assert abs(float(inp['feat'].double().sum()) - float(g['feat_checksum'])) < 1e-6
    d0, dd, D = inp['depth']
    h, w = inp['plane_size']
    dz, yy, xx = torch.meshgrid(torch.arange(0, D, 7), torch.arange(0, h, 11), torch.arange(0, w, 13), indexing='ij')
    index = ((dz * h + yy) * w + xx).reshape(-1)
    var = pinned.warp_variance(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'], inp['edges'], d0, dd, D,
                               inp['img_size'], inp['plane_size'], voxel_index=index)
    sub = var[:, ::4].reshape((1, 8) + tuple(dz.shape)).numpy()
    assert sub.shape == g['var_sub'].shape
    assert np.array_equal(sub, g['var_sub'])
