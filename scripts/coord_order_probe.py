#!/usr/bin/env python3
"""Developer probe (CPU only): which fp32 evaluation order reproduces torch's CPU arithmetic -- i.e. the oracle's and the
reference's -- for the plane-sweep sample coordinates and the bilinear taps?  Candidate orders are emulated in numpy (FMA via
float64) and compared BITWISE with the torch ops.  Findings (torch 2.10 CPU, MKL), which psv_variance.hip / backproject.hip follow:
  * the large batched products K^-1 p, R^T c and P [X;1] (MKL sgemm):      FMA chain in k order, first term a plain product
  * the small batched product P = K [R|t] (bmm's naive kernel, < 400 MACs): rounded products, sequential additions, NO FMA
  * tensor / python_float:                                                 true division
  * F.grid_sample bilinear (vectorised kernel):                            ((nw v + ne v) + sw v) + se v as an FMA chain
  * torch.inverse(K) == fp64 inverse rounded to fp32 for the pinhole intrinsics used by the tests and benches
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic')
from oracle import costvolume as ocv  # noqa: E402  (developer probe, not the product path)

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def chain(M, v, order, use_fma):
    acc = None
    for k in order:
        a, b = M[:, k:k + 1] * np.ones((1, v.shape[1]), np.float32), v[k:k + 1]
        if acc is None:
            acc = (a * b).astype(np.float32)
        elif use_fma:
            acc = fma(a, b, acc)
        else:
            acc = (acc + (a * b).astype(np.float32)).astype(np.float32)
    return acc


def main():
    inp = syn.make_costvolume_inputs('cfg2', n_ref=1)
    R, t, K = inp['rotmats'], inp['tvecs'], inp['K']
    d0, dd, D = inp['depth']
    H, W = inp['img_size']
    h, w = inp['plane_size']
    ref, src = 4, 1
    pts = ocv.plane_sweep_points(d0, dd, D, R, t, K, (H, W), (h, w))
    Kinv = torch.inverse(K)[ref].numpy()
    print('torch.inverse(K) == fp64 inverse rounded:', np.array_equal(Kinv, np.linalg.inv(K[ref].double().numpy()).astype(np.float32)))
    xs, ys = np.linspace(0, W - 1, w, dtype=np.float32), np.linspace(0, H - 1, h, dtype=np.float32)
    z = np.linspace(d0, d0 + (D - 1) * dd, D, dtype=np.float32)
    xx, yy = np.meshgrid(xs, ys)
    p = (np.stack((xx, yy, np.ones_like(xx))).astype(np.float64)[:, None] * z.astype(np.float64)[None, :, None, None])
    p = p.astype(np.float32).reshape(3, -1)
    cam_t = torch.bmm(torch.inverse(K)[ref:ref + 1], torch.from_numpy(p)[None])[0].numpy()
    for uf in (True, False):
        print('K^-1 p   order 012 fma=%-5s bit-equal fraction %.6f' % (uf, np.mean(chain(Kinv, p, (0, 1, 2), uf) == cam_t)))
    cm = (cam_t - t[ref].numpy()[:, None]).astype(np.float32)
    X = pts[ref].numpy()
    for uf in (True, False):
        print('R^T c    order 012 fma=%-5s bit-equal fraction %.6f' % (uf, np.mean(chain(R[ref].numpy().T.copy(), cm, (0, 1, 2), uf) == X)))
    P_t = torch.bmm(K, torch.cat((R, t[..., None]), 2))[src].numpy()
    Rt = np.concatenate((R[src].numpy(), t[src].numpy()[:, None]), 1)
    for uf in (True, False):
        print('P=K[R|t] order 012 fma=%-5s bit-equal %s' % (uf, np.array_equal(chain(K[src].numpy(), Rt, (0, 1, 2), uf), P_t)))
    XH = np.concatenate((X, np.ones((1, X.shape[1]), np.float32)), 0)
    q = torch.bmm(torch.from_numpy(P_t)[None], torch.from_numpy(XH)[None])[0].numpy()
    for uf in (True, False):
        print('q=P[X;1] order 0123 fma=%-5s bit-equal fraction %.6f' % (uf, np.mean(chain(P_t, XH, (0, 1, 2, 3), uf) == q)))
    zb = (np.abs(q[2]) + f32(1e-8)).astype(np.float32)
    u = (torch.from_numpy(q[0]) / torch.from_numpy(zb)).numpy()
    g_t = ((torch.from_numpy(u.copy()) / float(W - 1)) * 2 - 1.0).numpy()
    print('u / (W-1): true division %.6f, multiplication by the reciprocal %.6f'
          % (np.mean(((u / f32(W - 1)).astype(np.float32) * f32(2) - f32(1)).astype(np.float32) == g_t),
             np.mean(((u * (f32(1) / f32(W - 1))).astype(np.float32) * f32(2) - f32(1)).astype(np.float32) == g_t)))
    v = (torch.from_numpy(q[1]) / torch.from_numpy(zb))
    gy_t = ((v / float(H - 1)) * 2 - 1.0)
    grid = torch.stack((torch.from_numpy(g_t), gy_t), -1).view(1, -1, 1, 2)
    feat = inp['feat'][src:src + 1]
    Hf, Wf = feat.shape[2:]
    out_t = F.grid_sample(feat, grid, mode='bilinear', align_corners=True)[0, :, :, 0].numpy()
    gx, gy = g_t, gy_t.numpy()
    ix = (((gx + f32(1)) / f32(2)).astype(np.float32) * f32(Wf - 1)).astype(np.float32)
    iy = (((gy + f32(1)) / f32(2)).astype(np.float32) * f32(Hf - 1)).astype(np.float32)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wx0, wx1, wy0, wy1 = [a.astype(np.float32) for a in ((x1 - ix), (ix - x0), (y1 - iy), (iy - y0))]
    fm = feat[0].numpy()

    def val(yy_, xx_):
        ok = (xx_ >= 0) & (xx_ <= Wf - 1) & (yy_ >= 0) & (yy_ <= Hf - 1)
        return np.where(ok[None], fm[:, np.clip(yy_, 0, Hf - 1).astype(int), np.clip(xx_, 0, Wf - 1).astype(int)], f32(0)).astype(np.float32)
    ws = [(wy0 * wx0).astype(np.float32), (wy0 * wx1).astype(np.float32), (wy1 * wx0).astype(np.float32), (wy1 * wx1).astype(np.float32)]
    vs = [val(y0, x0), val(y0, x1), val(y1, x0), val(y1, x1)]
    for uf in (True, False):
        acc = (vs[0] * ws[0][None]).astype(np.float32)
        for k in (1, 2, 3):
            acc = fma(vs[k], ws[k][None] * np.ones_like(vs[k]), acc) if uf else (acc + (vs[k] * ws[k][None]).astype(np.float32)).astype(np.float32)
        print('grid_sample taps nw,ne,sw,se fma=%-5s bit-equal fraction %.6f' % (uf, np.mean(acc == out_t)))


if __name__ == '__main__':
    main()
