"""Oracle for SURVEY.md §8a rows A1-A7: plane-sweep warp, cross-view variance, dense 3D-conv
regulariser, soft-argmin depth.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: mv3d/subnetworks/mvsnet.py:133-229, mv3d/utils.py:86-108.
"""
import numpy as np
import torch
import torch.nn.functional as F


def scatter_mean(x, index, dim_size):
    """torch_scatter.scatter(x, index, dim=0, reduce='mean', dim_size=...) restated:
    sum / clamp(count, 1) (call sites mvsnet.py:214-215, lightningmodel.py:167-168,227-228)."""
    out = torch.zeros((dim_size,) + tuple(x.shape[1:]), dtype=x.dtype)
    out.index_add_(0, index, x)
    cnt = torch.zeros(dim_size, dtype=x.dtype)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=x.dtype))
    cnt = cnt.clamp_(min=1).view((-1,) + (1,) * (x.dim() - 1))
    return out / cnt


def plane_sweep_points(depth_start, depth_interval, n_planes, R, t, K, img_size, plane_size):
    """Row A1 -- world-space plane-sweep points for every image (mv3d/utils.py:86-108).

    Pixel grid and depths are float32 linspaces (:92-94); [x*z, y*z, z] is formed in float64
    because np.ones promotes (:98-99) and is then cast to float32 (:100); X = R^T (K^-1 p - t)
    (:103-106).  Returns [N_img, 3, D*h*w], point index = d*(h*w) + y*w + x."""
    n_batch = R.shape[0]
    depth_end = depth_start + (n_planes - 1) * depth_interval
    xs = np.linspace(0, img_size[1] - 1, plane_size[1], dtype=np.float32)
    ys = np.linspace(0, img_size[0] - 1, plane_size[0], dtype=np.float32)
    z = np.linspace(depth_start, depth_end, n_planes, dtype=np.float32)
    xx, yy = np.meshgrid(xs, ys)
    p = np.stack((xx, yy, np.ones_like(xx))).astype(np.float64)          # [3,h,w]
    p = p[:, None] * z.astype(np.float64)[None, :, None, None]            # [3,D,h,w] f64
    p = torch.from_numpy(p).float().view(3, -1).unsqueeze(0).repeat(n_batch, 1, 1)
    K_inv = torch.inverse(K)
    R_T = R.transpose(2, 1)
    pts_cam = torch.bmm(K_inv, p)
    return torch.bmm(R_T, pts_cam - t[..., None])


def project_to_grid(pts_ref, rotmats, tvecs, K, src_idx, img_size):
    """Row A2 -- project world points (one set per edge) into the edge's source image and
    normalise with the IMAGE size (mvsnet.py:192-206; lightningmodel.py:147-163,213-223).

    pts_ref: [E, 3, n_pts] world points already gathered per edge.  Returns grid [E, n_pts, 1, 2]."""
    n_e, _, n_pts = pts_ref.shape
    w = torch.ones((n_e, 1, n_pts), dtype=torch.float32)
    pts_H = torch.cat((pts_ref, w), dim=1)
    P = torch.cat((rotmats, tvecs[..., None]), dim=2)
    P = torch.bmm(K, P)
    q = torch.bmm(P[src_idx], pts_H)
    zb = torch.abs(q[:, 2]) + 1e-8                                         # mvsnet.py:200-201
    uv = q[:, :2] / zb[:, None]
    grid = uv.transpose(2, 1).reshape(n_e, n_pts, 1, 2).clone()
    grid[..., 0] = (grid[..., 0] / float(img_size[1] - 1)) * 2 - 1.0      # mvsnet.py:205
    grid[..., 1] = (grid[..., 1] / float(img_size[0] - 1)) * 2 - 1.0      # mvsnet.py:206
    return grid


def warp_variance(feat, rotmats, tvecs, K, edges, depth_start, depth_interval, n_planes,
                  img_size, plane_size):
    """Rows A1-A4 -- variance cost volume [n_ref, C, D, h, w] (mvsnet.py:176-216).

    feat: [N_img, C, Hf, Wf] quarter-resolution features; edges: [2, E] int64, row 0 = ref image
    index, row 1 = source image index (the ref itself is one of its sources, dataset.py:133-137)."""
    ref_idx, gather_idx = torch.unique(edges[0], return_inverse=True)      # mvsnet.py:179
    n_ref = len(ref_idx)
    pts = plane_sweep_points(depth_start, depth_interval, n_planes, rotmats, tvecs, K,
                             img_size, plane_size)                         # mvsnet.py:188
    grid = project_to_grid(pts[edges[0]], rotmats, tvecs, K, edges[1], img_size)
    x_vox = F.grid_sample(feat[edges[1]], grid, mode='bilinear', align_corners=True)  # :209
    x_vox = x_vox.squeeze(3).view(-1, feat.shape[1], n_planes, *plane_size)
    x_avg = scatter_mean(x_vox, gather_idx, n_ref)                         # :214
    x_avg_sq = scatter_mean(x_vox ** 2, gather_idx, n_ref)                 # :215
    return x_avg_sq - x_avg ** 2                                           # :216


# ----------------------------------------------------------------------------------------------
# Row A5: CostRegNet (mvsnet.py:133-163), functional form over a reference-named state_dict.
# ----------------------------------------------------------------------------------------------

def _bn(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], training=False, eps=eps)


def conv_bn_relu3d(x, sd, name, stride=1):
    """ConvBnRelu3d (mvsnet.py:18-25): Conv3d(k3, pad 1, no bias) -> BatchNorm3d(eval) -> ReLU."""
    y = F.conv3d(x, sd[name + '.conv.weight'], None, stride=stride, padding=1)
    return F.relu(_bn(y, sd, name + '.bn'))


def deconv_bn_relu3d(x, sd, name):
    """DeconvBnRelu3d (mvsnet.py:28-36): ConvTranspose3d(k3, s2, p1, output_padding 1, no bias)
    -> BatchNorm3d(eval) -> ReLU."""
    y = F.conv_transpose3d(x, sd[name + '.deconv.weight'], None, stride=2, padding=1,
                           output_padding=1)
    return F.relu(_bn(y, sd, name + '.bn'))


def costregnet(x, sd):
    """CostRegNet.forward (mvsnet.py:154-163).  x: [B, 32, D, h, w] -> [B, 1, D, h, w].
    Skip-adds happen after the ReLU of the deconv (:159-161); `prob` has a bias, no BN/ReLU (:152)."""
    conv0 = conv_bn_relu3d(x, sd, 'conv0')
    conv2 = conv_bn_relu3d(conv_bn_relu3d(conv0, sd, 'conv1', 2), sd, 'conv2')
    conv4 = conv_bn_relu3d(conv_bn_relu3d(conv2, sd, 'conv3', 2), sd, 'conv4')
    y = conv_bn_relu3d(conv_bn_relu3d(conv4, sd, 'conv5', 2), sd, 'conv6')
    y = conv4 + deconv_bn_relu3d(y, sd, 'conv7')
    y = conv2 + deconv_bn_relu3d(y, sd, 'conv8')
    y = conv0 + deconv_bn_relu3d(y, sd, 'conv9')
    return F.conv3d(y, sd['prob.weight'], sd['prob.bias'], stride=1, padding=1)


def soft_argmin_depth(x_reg, depth_start, depth_interval, n_planes):
    """Row A6 -- softmax(-x) over D, expectation over float32 linspace depths (mvsnet.py:219-227).
    x_reg: [n_ref, D, h, w] -> (depth [n_ref, h, w], prob [n_ref, D, h, w])."""
    depth_end = depth_start + depth_interval * (n_planes - 1)
    prob = F.softmax(-x_reg, dim=1)
    vals = torch.linspace(depth_start, depth_end, n_planes).view(1, n_planes, 1, 1)
    return torch.sum(vals * prob, dim=1), prob


def mvsnet_depth(feat, rotmats, tvecs, K, edges, sd, depth_start, depth_interval, n_planes,
                 img_size, plane_size, pinned=False):
    """Rows A1-A6 end to end from quarter features (mvsnet.py:186-227).
    Returns (depth [n_ref,h,w], var [n_ref,C,D,h,w], x_reg [n_ref,D,h,w]).
    pinned=True: the variance volume from oracle/pinned.py -- the evaluation orders of the reference run that produced
    the goldens, independent of this host's BLAS (torch.bmm's last bits are not)."""
    if pinned:
        from . import pinned as pin
        var = pin.warp_variance(feat, rotmats, tvecs, K, edges, depth_start, depth_interval, n_planes, img_size,
                                plane_size)
    else:
        var = warp_variance(feat, rotmats, tvecs, K, edges, depth_start, depth_interval, n_planes,
                            img_size, plane_size)
    x_reg = costregnet(var, sd).squeeze(1)
    depth, _ = soft_argmin_depth(x_reg, depth_start, depth_interval, n_planes)
    return depth, var, x_reg
