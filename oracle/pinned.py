"""Host-independent restatement of rows A1-A4 with PINNED fp32 evaluation orders.  TEST INFRASTRUCTURE ONLY.

Why it exists: ``oracle/costvolume.py`` restates the reference with the same torch ops, so its last bits depend on the
host's BLAS -- torch.bmm on the Intel build container (MKL with FMA: every large batched product is an FMA chain in k
order) and on the AMD EPYC host of the GPU box (MKL without FMA: rounded products, sequential additions) differ in ~35 % of
the sample coordinates by one ulp.  Neither is "the" reference arithmetic (the authors ran cuBLAS on an RTX 3090); the
golden vectors under tests/golden/ were captured from the reference's own code in the build container.  This module
spells the build container's orders out with elementwise torch ops (never fused; FMA emulated through float64, exact for
fp32 operands), so that

  * tests can check it against the goldens on ANY host (tests/test_oracle_golden.py: variance within ~1e-7), and
  * the HIP kernels, which implement exactly these orders (csrc/v3d_common.h: dot3_chain, world_point, sample_position),
    can be compared with it BIT FOR BIT on the GPU box (tests/test_costvolume_gpu.py).

Orders (scripts/coord_order_probe.py measures them against torch):
  K^-1 p, R^T c, P [X;1]   FMA chain in k order, first term a plain product        (mv3d/utils.py:104-106, mvsnet.py:199)
  P = K [R|t]               rounded products, sequential additions                   (mvsnet.py:196-197; bmm's small path)
  u / (W-1), x / cnt        true divisions                                           (mvsnet.py:205-206, torch_scatter mean)
  bilinear taps             ((nw v + ne v) + sw v) + se v as an FMA chain            (F.grid_sample, vectorised CPU kernel)
  sum over edges            sequential in edge order; squares rounded before summing (mvsnet.py:214-216)
"""
import numpy as np
import torch


def _t(x):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32)


def fma(a, b, c):
    """fl32(a*b + c) for fp32 tensors: the product is exact in float64; the double rounding (float64, then float32) can
    differ from a true FMA only when the float64 sum lands exactly on a float32 tie -- not observed on these inputs."""
    return (_t(a).double() * _t(b).double() + _t(c).double()).float()


def mul(a, b):
    return _t(a) * _t(b)


def dot3_chain(a0, b0, a1, b1, a2, b2):
    return fma(a2, b2, fma(a1, b1, mul(a0, b0)))


def camera_blocks(K, R, t):
    """K^-1 (float64 inverse rounded to float32 == torch.inverse for pinhole intrinsics) and P = K [R|t] per image."""
    K, R, t = _t(K), _t(R), _t(t)
    Kinv = torch.linalg.inv(K.double()).float()
    Rt = torch.cat((R, t[:, :, None]), dim=2)                             # [N,3,4]
    P = torch.zeros_like(Rt)
    for r in range(3):
        for j in range(4):
            P[:, r, j] = (K[:, r, 0] * Rt[:, 0, j] + K[:, r, 1] * Rt[:, 1, j]) + K[:, r, 2] * Rt[:, 2, j]
    return Kinv, P


def world_points(K, R, t, ref, depth_start, depth_interval, n_planes, img_size, plane_size):
    """Row A1 for image `ref`: [3, D*h*w] float32, point index d*(h*w) + y*w + x (utils.py:86-108)."""
    Kinv, _ = camera_blocks(K, R, t)
    H, W = img_size
    h, w = plane_size
    xs = torch.from_numpy(np.linspace(0, W - 1, w, dtype=np.float32))
    ys = torch.from_numpy(np.linspace(0, H - 1, h, dtype=np.float32))
    zs = torch.from_numpy(np.linspace(depth_start, depth_start + (n_planes - 1) * depth_interval, n_planes,
                                      dtype=np.float32))
    shape = (n_planes, h, w)
    Z = zs[:, None, None].expand(shape).reshape(-1)
    p0 = xs[None, None, :].expand(shape).reshape(-1) * Z                  # float64 product rounded once == fp32 product
    p1 = ys[None, :, None].expand(shape).reshape(-1) * Z
    p2 = Z.clone()
    Ki, Rr, tr = Kinv[ref], _t(R)[ref], _t(t)[ref]
    c = [dot3_chain(Ki[i, 0], p0, Ki[i, 1], p1, Ki[i, 2], p2) - tr[i] for i in range(3)]
    return torch.stack([dot3_chain(Rr[0, j], c[0], Rr[1, j], c[1], Rr[2, j], c[2]) for j in range(3)])


def backproject_points(K, R, t, depth, img_size):
    """Rows B1-B2 / C1 (lightningmodel.py:138-144, 201-205) with the pinned orders of row A1: X = R^T (K^-1 (p * depth) - t) for
    the pixel grid of build_img_pts, cameras [n,3,3] / [n,3] of the reference views themselves, depth [n, h, w] -> [n, 3, h*w].
    The same chains as ``world_points`` (csrc/v3d_common.h: world_point): torch.bmm's last bits depend on the host BLAS, and ONE
    ulp of a back-projected coordinate next to a voxel boundary moves a point into another cell -- which the sparse U-Net's global
    receptive field turns into centimetres of depth for thousands of pixels (scripts/parity_scene.py)."""
    K, R, t, depth = _t(K), _t(R), _t(t), _t(depth)
    Kinv = torch.linalg.inv(K.double()).float()
    H, W = img_size
    n, h, w = depth.shape
    xs = torch.from_numpy(np.linspace(0, W - 1, w, dtype=np.float32))
    ys = torch.from_numpy(np.linspace(0, H - 1, h, dtype=np.float32))
    d = depth.reshape(n, -1)
    p0 = xs[None, :].expand(h, w).reshape(1, -1) * d
    p1 = ys[:, None].expand(h, w).reshape(1, -1) * d
    p2 = d
    c = [dot3_chain(Kinv[:, i, 0:1], p0, Kinv[:, i, 1:2], p1, Kinv[:, i, 2:3], p2) - t[:, i:i + 1] for i in range(3)]
    return torch.stack([dot3_chain(R[:, 0, j:j + 1], c[0], R[:, 1, j:j + 1], c[1], R[:, 2, j:j + 1], c[2]) for j in range(3)], dim=1)


def sample_positions(X, P_src, img_size, feat_size):
    """Row A2 + grid_sample's un-normalisation for one edge: world points X [3,N] -> (ix, iy) float32 [N]."""
    H, W = img_size
    Hf, Wf = feat_size
    X, P_src = _t(X), _t(P_src)
    q = [dot3_chain(P_src[i, 0], X[0], P_src[i, 1], X[1], P_src[i, 2], X[2]) + P_src[i, 3] for i in range(3)]
    zb = q[2].abs() + 1e-8
    u, v = q[0] / zb, q[1] / zb
    gx = (u / float(W - 1)) * 2 - 1.0
    gy = (v / float(H - 1)) * 2 - 1.0
    return ((gx + 1) * 0.5) * float(Wf - 1), ((gy + 1) * 0.5) * float(Hf - 1)


def bilinear_sample(fmap, ix, iy):
    """F.grid_sample(bilinear, zeros, align_corners=True) for one image: fmap [C,Hf,Wf], positions [N] -> [C,N]."""
    fmap = _t(fmap)
    C, Hf, Wf = fmap.shape
    x0, y0 = torch.floor(ix), torch.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wx0, wx1, wy0, wy1 = x1 - ix, ix - x0, y1 - iy, iy - y0
    flat = fmap.reshape(C, Hf * Wf)

    def val(yy, xx):
        ok = (xx >= 0) & (xx <= Wf - 1) & (yy >= 0) & (yy <= Hf - 1)
        xi = torch.nan_to_num(xx).clamp(0, Wf - 1).long()
        yi = torch.nan_to_num(yy).clamp(0, Hf - 1).long()
        return torch.where(ok[None], flat[:, yi * Wf + xi], torch.zeros((), dtype=torch.float32))
    ws = [wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1]                     # nw, ne, sw, se
    acc = val(y0, x0) * ws[0][None]
    for k, (yy, xx) in zip((1, 2, 3), ((y0, x1), (y1, x0), (y1, x1))):
        acc = fma(val(yy, xx), ws[k][None], acc)
    return acc


def warp_variance(feat, R, t, K, edges, depth_start, depth_interval, n_planes, img_size, plane_size,
                  fused_square=False, voxel_index=None):
    """Rows A1-A4 (mvsnet.py:176-216): [n_ref, C, D, h, w] float32 tensor.  fused_square=False squares with a rounding
    before the sum (the reference: x_vox ** 2, then the scatter mean); True = fma(x, x, acc), what the HIP kernels do.
    ``voxel_index`` (1-D long tensor of flat voxel indices d*h*w + y*w + x): evaluate only those voxels and return
    [n_ref, C, len(voxel_index)] -- every voxel is independent, so this is a sub-sample of the full result, bit for bit
    (used to check full-size volumes such as cfg5 without materialising them)."""
    feat, R, t, K = _t(feat), _t(R), _t(t), _t(K)
    edges = torch.as_tensor(np.asarray(edges) if not torch.is_tensor(edges) else edges)
    refs = torch.unique(edges[0]).tolist()
    _, P = camera_blocks(K, R, t)
    C, Hf, Wf = feat.shape[1:]
    h, w = plane_size
    n_out = n_planes * h * w if voxel_index is None else int(voxel_index.numel())
    out = torch.zeros((len(refs), C, n_out), dtype=torch.float32)
    for r, ref in enumerate(refs):
        X = world_points(K, R, t, ref, depth_start, depth_interval, n_planes, img_size, plane_size)
        if voxel_index is not None:
            X = X[:, voxel_index]
        s = torch.zeros((C, X.shape[1]), dtype=torch.float32)
        q = torch.zeros_like(s)
        srcs = edges[1][edges[0] == ref].tolist()
        for src in srcs:                                                   # edge order (index_add_ is sequential)
            ix, iy = sample_positions(X, P[src], img_size, (Hf, Wf))
            x = bilinear_sample(feat[src], ix, iy)
            s = s + x
            q = fma(x, x, q) if fused_square else q + x * x
        cnt = float(max(len(srcs), 1))
        avg, avg_sq = s / cnt, q / cnt
        out[r] = avg_sq - avg * avg
    return out if voxel_index is not None else out.reshape(len(refs), C, n_planes, h, w)
