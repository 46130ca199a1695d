"""Deterministic synthetic inputs and weights for the cost-volume / refinement path.

There is no dataset or checkpoint access, so benchmarks, smoke tests and parity tests run on
synthetic scenes that follow the reference's conventions exactly:

  * poses are world->camera ``rotmats = R_c2w^T``, ``tvecs = -rotmats @ c``
    (mv3d/dsets/dataset.py:214-216);
  * ``ref_src_edges[0]`` = reference image index, ``[1]`` = source image index, grouped per
    reference, the reference itself being one of its own sources (dataset.py:133-137);
  * intrinsics are ScanNet's 640x480 depth camera rescaled without crop
    (dataset.py:64-73);
  * weights are plain ``state_dict``-style dicts carrying the reference's key names and shapes
    (mvsnet.py:133-163, scenemodeling.py:116-237, refinement.py:16-25) with BatchNorm /
    GroupNorm statistics and affines randomised so that folding them is actually exercised.

Everything here is CPU torch/numpy; callers move tensors to the device.
"""
import math

import numpy as np
import torch

ROOM = (6.03, 5.01, 2.97)          # deliberately not multiples of any voxel size
SCANNET_K = (577.87, 577.87, 319.5, 239.5, 640, 480)


def intrinsics(img_size):
    """3x3 K for an (H, W) image, rescaled from ScanNet 640x480 like PreprocessImage does."""
    fx, fy, cx, cy, w0, h0 = SCANNET_K
    sy, sx = img_size[0] / h0, img_size[1] / w0
    return torch.tensor([[fx * sx, 0.0, cx * sx], [0.0, fy * sy, cy * sy], [0.0, 0.0, 1.0]],
                        dtype=torch.float32)


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * k + (1 - math.cos(ang)) * (k @ k)


def make_cameras(n_img, img_size, seed=0, yaw_step_deg=None, radius=0.8, height=1.5):
    """Cameras on a circle inside the box room, looking outward; consecutive views are
    neighbours.  Returns (rotmats [N,3,3], tvecs [N,3], K [N,3,3]) float32, world->camera."""
    rng = np.random.RandomState(seed)
    if yaw_step_deg is None:
        yaw_step_deg = min(360.0 / n_img, 6.0)
    cx, cy = ROOM[0] / 2, ROOM[1] / 2
    rotmats, tvecs = [], []
    for i in range(n_img):
        th = math.radians(yaw_step_deg * i)
        f = np.array([math.cos(th), math.sin(th), 0.0])
        d = np.array([0.0, 0.0, -1.0])
        r = np.cross(d, f)
        R_c2w = np.stack([r, d, f], axis=1)
        jit = rng.normal(0, math.radians(0.5), 3)
        for a, ang in zip(np.eye(3), jit):
            R_c2w = _rot(a, ang) @ R_c2w
        c = np.array([cx + radius * math.cos(th), cy + radius * math.sin(th), height])
        c = c + rng.normal(0, 0.01, 3)
        R = R_c2w.T
        rotmats.append(R)
        tvecs.append(-R @ c)
    rotmats = torch.from_numpy(np.stack(rotmats)).float()
    tvecs = torch.from_numpy(np.stack(tvecs)).float()
    K = intrinsics(img_size).unsqueeze(0).repeat(n_img, 1, 1)
    return rotmats, tvecs, K


def make_edges(n_ref, n_before, n_after):
    """Sliding-window edge list: reference i+n_before <-> sources i .. i+n_before+n_after.
    Returns (edges [2, n_ref*(n_before+n_after+1)] int64, n_img)."""
    per = n_before + n_after + 1
    e = torch.empty((2, n_ref * per), dtype=torch.long)
    for i in range(n_ref):
        e[0, i * per:(i + 1) * per] = i + n_before
        e[1, i * per:(i + 1) * per] = torch.arange(i, i + per)
    return e, n_ref + n_before + n_after


def make_features(n_img, feat_dim, hf, wf, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((n_img, feat_dim, hf, wf), generator=g, dtype=torch.float32)


def ray_box_depth(rotmats, tvecs, K, img_size, plane_size):
    """Analytic z-depth of the box-room walls seen from each camera on the (h, w) sampling grid
    used by the reference (x = linspace(0, W-1, w), y = linspace(0, H-1, h)).  [N, h, w]."""
    n = rotmats.shape[0]
    xs = torch.linspace(0, img_size[1] - 1, plane_size[1])
    ys = torch.linspace(0, img_size[0] - 1, plane_size[0])
    yy, xx = torch.meshgrid(ys, xs, indexing='ij')
    pix = torch.stack((xx, yy, torch.ones_like(xx)), 0).reshape(3, -1).double()
    out = torch.empty((n,) + tuple(plane_size), dtype=torch.float32)
    lo = torch.zeros(3, dtype=torch.float64)
    hi = torch.tensor(ROOM, dtype=torch.float64)
    for i in range(n):
        R = rotmats[i].double()
        c = -(R.T @ tvecs[i].double())
        dirs = R.T @ (torch.inverse(K[i].double()) @ pix)        # world ray per unit z-depth
        t1 = (lo[:, None] - c[:, None]) / dirs
        t2 = (hi[:, None] - c[:, None]) / dirs
        t_exit = torch.maximum(t1, t2).min(dim=0)[0]
        out[i] = t_exit.reshape(plane_size).float()
    return out


# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------

def _uniform(g, shape, bound):
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def _conv_weight(g, shape, fan_in):
    return _uniform(g, shape, 1.0 / math.sqrt(fan_in))


def _norm_stats(g, sd, prefix, c, running=True):
    sd[prefix + '.weight'] = torch.rand(c, generator=g) + 0.5
    sd[prefix + '.bias'] = torch.randn(c, generator=g) * 0.1
    if running:
        sd[prefix + '.running_mean'] = torch.randn(c, generator=g) * 0.1
        sd[prefix + '.running_var'] = torch.rand(c, generator=g) + 0.5


def costregnet_weights(in_channels=32, base=8, seed=0, sharpen=1.0):
    """state_dict for CostRegNet(in_channels, base) (mvsnet.py:133-163).  ``sharpen`` multiplies
    ``prob.weight`` so that the depth softmax is peaked like a trained net's."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    chans = [(in_channels, base), (base, 2 * base), (2 * base, 2 * base), (2 * base, 4 * base),
             (4 * base, 4 * base), (4 * base, 8 * base), (8 * base, 8 * base)]
    for i, (ci, co) in enumerate(chans):
        sd['conv%d.conv.weight' % i] = _conv_weight(g, (co, ci, 3, 3, 3), ci * 27)
        _norm_stats(g, sd, 'conv%d.bn' % i, co)
    for i, (ci, co) in zip((7, 8, 9), [(8 * base, 4 * base), (4 * base, 2 * base), (2 * base, base)]):
        sd['conv%d.deconv.weight' % i] = _conv_weight(g, (ci, co, 3, 3, 3), co * 27)
        _norm_stats(g, sd, 'conv%d.bn' % i, co)
    sd['prob.weight'] = _conv_weight(g, (1, base, 3, 3, 3), base * 27) * sharpen
    sd['prob.bias'] = _uniform(g, (1,), 1.0 / math.sqrt(base * 27))
    return sd


def pointnet_weights(hidden=128, out_dim=64, in_dim=35, seed=1):
    """state_dict for PointNet(hidden, out_dim, in_dim) (scenemodeling.py:116-125)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (ci, co) in (('fc_pos', (in_dim, hidden)), ('fc1', (hidden, hidden)),
                           ('fc2', (2 * hidden, hidden)), ('fc3', (2 * hidden, hidden)),
                           ('fc4', (2 * hidden, hidden)), ('fc_out', (hidden, out_dim))):
        b = 1.0 / math.sqrt(ci)
        sd[name + '.weight'] = _uniform(g, (co, ci), b)
        sd[name + '.bias'] = _uniform(g, (co,), b)
    return sd


def sparse_unet_weights(dims=(64, 128, 128), n_groups=(4, 8, 8), n_res=(1, 2, 3), seed=2):
    """state_dict for SparseUNet (scenemodeling.py:147-189) with MinkowskiEngine's parameter
    naming: conv ``.kernel`` [27, Ci, Co] (1x1: [Ci, Co]), GroupNorm ``.gn.weight/.gn.bias``.
    No conv biases (ME default bias=False; residual blocks has_bias = norm is None, :19)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def res_block(prefix, c):
        for j in (1, 2):
            sd['%s.conv%d.kernel' % (prefix, j)] = _conv_weight(g, (27, c, c), c * 27)
            _norm_stats(g, sd, '%s.n%d.gn' % (prefix, j), c, running=False)

    for i, n in enumerate(n_res):
        for l in range(n):
            res_block('res_down.%d.%d' % (i, l), dims[i])
    for i in range(1, len(dims)):
        sd['down.%d.0.kernel' % (i - 1)] = _conv_weight(g, (27, dims[i - 1], dims[i]), dims[i - 1] * 27)
        _norm_stats(g, sd, 'down.%d.1.gn' % (i - 1), dims[i], running=False)
    rd, rn = dims[::-1], n_res[::-1]
    for i, n in enumerate(rn[1:]):
        for l in range(n):
            res_block('res_up.%d.%d' % (i, l), rd[i + 1])
    for i in range(1, len(rd)):
        sd['up.%d.0.kernel' % (i - 1)] = _conv_weight(g, (27, rd[i - 1], rd[i]), rd[i - 1] * 27)
        _norm_stats(g, sd, 'up.%d.1.gn' % (i - 1), rd[i], running=False)
        sd['feat_adj.%d.0.kernel' % (i - 1)] = _conv_weight(g, (2 * rd[i], rd[i]), 2 * rd[i])
        _norm_stats(g, sd, 'feat_adj.%d.1.gn' % (i - 1), rd[i], running=False)
    return sd


def decoder_weights(in_dim=352, h_dim=128, ksize=3, seed=3, sharpen=1.0):
    """state_dict for HypothesisDecoder.net (refinement.py:16-25): three Conv1d(no bias)+BN1d+ReLU
    and a final Conv1d(h_dim, 1) with bias.  ``sharpen`` multiplies ``net.3.weight``."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, ci in enumerate((in_dim, h_dim, h_dim)):
        sd['net.%d.0.weight' % i] = _conv_weight(g, (h_dim, ci, ksize), ci * ksize)
        _norm_stats(g, sd, 'net.%d.1' % i, h_dim)
    sd['net.3.weight'] = _conv_weight(g, (1, h_dim, ksize), h_dim * ksize) * sharpen
    sd['net.3.bias'] = _uniform(g, (1,), 1.0 / math.sqrt(h_dim * ksize))
    return sd


def propagation_weights(in_dim=33, h_dim=32, seed=5):
    """state_dict for PropagationNet(in_dim, h_dim) (mv3d/subnetworks/upsampling.py:14-21)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    widths = [in_dim, h_dim, h_dim, h_dim, 9]
    for i in range(4):
        sd['conv%d.0.weight' % (i + 1)] = _conv_weight(g, (widths[i + 1], widths[i], 3, 3), widths[i] * 9)
        _norm_stats(g, sd, 'conv%d.1' % (i + 1), widths[i + 1])
    return sd


def backbone_weights(feat_dim=32, seed=6):
    """(state_dict of FeatureExtractor, state_dict of FeatureShrinker(feat_dim)) with torchvision's key names
    (``layerK.<i>[.<block>.layers.<j>].weight``, ``fpn.inner_blocks.<i>.*``): seeded fan-in-scaled convolutions,
    randomised BatchNorm statistics / affines (pretrained ImageNet weights are not available offline)."""
    from .backbone import FeatureExtractor, FeatureShrinker
    g = torch.Generator().manual_seed(seed)
    out = []
    for mod in (FeatureExtractor(), FeatureShrinker(feat_dim)):
        sd = {}
        for k, v in mod.state_dict().items():
            if k.endswith('num_batches_tracked'):
                continue
            if k.endswith('running_mean'):
                sd[k] = torch.randn(v.shape, generator=g) * 0.1
            elif k.endswith('running_var'):
                sd[k] = torch.rand(v.shape, generator=g) + 0.5
            elif v.dim() == 4:                                     # conv weight: ReLU-preserving scale
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                sd[k] = _uniform(g, v.shape, math.sqrt(6.0 / fan_in))
            elif k.endswith('bias') and '.fpn.' not in '.' + k:   # BatchNorm bias
                sd[k] = torch.randn(v.shape, generator=g) * 0.1
            elif k.endswith('bias'):                               # FPN conv bias
                sd[k] = _uniform(g, v.shape, 0.05)
            else:                                                  # BatchNorm weight
                sd[k] = torch.rand(v.shape, generator=g) + 0.5
        out.append(sd)
    return tuple(out)


def make_images(n_img, img_size, seed):
    """Smooth seeded RGB images [n_img, 3, H, W] in ImageNet-normalised range (the reference normalises with the
    ImageNet mean / std, mv3d/utils.py:9-14)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randn((n_img, 3, img_size[0] // 8, img_size[1] // 8), generator=g)
    return torch.nn.functional.interpolate(low, size=tuple(img_size), mode='bilinear', align_corners=False) \
        + 0.1 * torch.randn((n_img, 3) + tuple(img_size), generator=g)


# ----------------------------------------------------------------------------------------------
# benchmark configurations (BASELINE.json configs)
# ----------------------------------------------------------------------------------------------

CONFIGS = {
    # name: img_size, feat (Hf,Wf), planes (d0, dd, D), plane grid, src before/after, voxel size
    'cfg1': dict(img_size=(128, 160), feat_size=(32, 40), depth=(0.5, 0.05, 32),
                 plane_size=(32, 40), window=(1, 1), edge_len=0.08),
    'cfg2': dict(img_size=(256, 320), feat_size=(64, 80), depth=(0.5, 0.05, 96),
                 plane_size=(56, 56), window=(4, 3), edge_len=0.08),
    'cfg3': dict(img_size=(256, 320), feat_size=(64, 80), depth=(0.5, 0.05, 96),
                 plane_size=(56, 56), window=(4, 3), edge_len=0.04),
    'cfg5': dict(img_size=(480, 640), feat_size=(120, 160), depth=(0.5, 0.025, 192),
                 plane_size=(120, 160), window=(5, 5), edge_len=0.02),
}


def make_costvolume_inputs(cfg_name, n_ref, feat_dim=32, seed=None):
    """Synthetic inputs for rows A1-A6: n_ref reference views of one sliding-window scene.
    Returns dict(feat, rotmats, tvecs, K, edges, n_img) + the config entries."""
    cfg = dict(CONFIGS[cfg_name])
    nb, na = cfg['window']
    edges, n_img = make_edges(n_ref, nb, na)
    if seed is None:
        seed = 1234 + int(cfg_name[-1])
    rotmats, tvecs, K = make_cameras(n_img, cfg['img_size'], seed=seed)
    feat = make_features(n_img, feat_dim, cfg['feat_size'][0], cfg['feat_size'][1], seed)
    cfg.update(feat=feat, rotmats=rotmats, tvecs=tvecs, K=K, edges=edges, n_img=n_img,
               n_ref=n_ref)
    return cfg
