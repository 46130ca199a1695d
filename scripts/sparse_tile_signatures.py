#!/usr/bin/env python3
"""How many of the 27 kernel offsets does a 64-row tile of a sparse convolution touch, by row order?  (CPU only.)

The sparse-convolution kernel (csrc/gemm_gather.hip: gemm_gather_pipe_kernel) multiplies an offset for a whole tile when ANY
of the tile's rows has that neighbour: with the rows in key order (torch.unique of (batch, x, y, z)) almost every offset is
live in almost every tile -- 2.4x the algorithmic matrix work.  This script builds the three coordinate maps
of the cfg3 bench scene with the oracle's voxelisation and reports, per level and per candidate row order, the mean number of
live offsets per tile against the mean number of present neighbours per row.

    python scripts/sparse_tile_signatures.py [--refs 64] [--tile 64]
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def signatures(coords, step):
    """27-bit neighbour-presence word per row: bit k = row has a neighbour at offset k * step (k as oracle.scene.kernel_offsets)."""
    c = coords.astype(np.int64)
    lo = c.min(0) - step
    c0 = c - lo
    ext = c0.max(0) + step + 1
    key = (c0[:, 0] * ext[1] + c0[:, 1]) * ext[2] + c0[:, 2]
    srt = np.sort(key)
    sig = np.zeros(len(c), dtype=np.int64)
    k = 0
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                q = ((c0[:, 0] + ox * step) * ext[1] + (c0[:, 1] + oy * step)) * ext[2] + (c0[:, 2] + oz * step)
                pos = np.searchsorted(srt, q)
                pos[pos >= len(srt)] = len(srt) - 1
                sig |= (srt[pos] == q).astype(np.int64) << k
                k += 1
    return sig


def popcount(x):
    return np.array([bin(int(v)).count('1') for v in x])


def morton(c):
    c = c.astype(np.int64)
    out = np.zeros(len(c), dtype=np.int64)
    for b in range(16):
        for a in range(3):
            out |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return out


def tile_cost(sig, order, tile):
    s = sig[order]
    n = (len(s) + tile - 1) // tile
    tot = 0
    hist = np.zeros(28, dtype=np.int64)
    for t in range(n):
        u = np.bitwise_or.reduce(s[t * tile:(t + 1) * tile])
        pc = bin(int(u)).count('1')
        tot += pc
        hist[pc] += 1
    return tot / n, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=64)
    ap.add_argument('--tile', type=int, default=64)
    args = ap.parse_args()
    syn = importlib.import_module('3dvnet_amd.synthetic')
    from oracle import scene as osc
    from oracle import pinned
    cfg = syn.CONFIGS['cfg3']
    nb, na = 4, 3
    edges, n_img = syn.make_edges(args.refs, nb, na)
    rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=1237, yaw_step_deg=360.0 / max(n_img, 60))
    gt = syn.ray_box_depth(rot[nb:nb + args.refs], tv[nb:nb + args.refs], K[nb:nb + args.refs], cfg['img_size'], (56, 56))
    gt = gt + 0.02 * torch.randn(gt.shape, generator=torch.Generator().manual_seed(7))
    pts = pinned.backproject_points(K[nb:nb + args.refs], rot[nb:nb + args.refs], tv[nb:nb + args.refs], gt, cfg['img_size']).permute(0, 2, 1).reshape(-1, 3)
    _, idx3d, _, _ = osc.voxelize(pts, torch.zeros(pts.shape[0], dtype=torch.long), cfg['edge_len'])
    c1 = np.unique(idx3d.numpy().astype(np.int64), axis=0)
    levels = [(1, c1)]
    for s in (2, 4):
        levels.append((s, np.unique(levels[-1][1] // s * s, axis=0)))
    for stride, c in levels:
        sig = signatures(c, stride)
        pc = popcount(sig)
        n = len(c)
        orders = {
            'key order (x, y, z): the shipped order': np.lexsort((c[:, 2], c[:, 1], c[:, 0])),
            'signature as an integer': np.argsort(sig, kind='stable'),
            'popcount, then signature': np.lexsort((sig, pc)),
            'Morton code': np.argsort(morton(c // stride), kind='stable'),
        }
        # signature bits reordered by how evenly they split the rows (most balanced bit most significant)
        freq = np.array([((sig >> k) & 1).mean() for k in range(27)])
        bitorder = np.argsort(np.abs(freq - 0.5))
        key = np.zeros(n, dtype=np.int64)
        for rank, k in enumerate(bitorder):
            key |= ((sig >> k) & 1) << (26 - rank)
        orders['signature, balanced bits first'] = np.argsort(key, kind='stable')
        print('level stride %d: %d rows, %.2f present neighbours per row (of 27), %d distinct signatures' %
              (stride, n, pc.mean(), len(np.unique(sig))))
        for name, o in orders.items():
            mean, hist = tile_cost(sig, o, args.tile)
            print('   %-42s live offsets per %d-row tile: %.2f  (%.2fx the per-row mean)   histogram %s' %
                  (name, args.tile, mean, mean / pc.mean(), ' '.join('%d:%d' % (i, h) for i, h in enumerate(hist) if h)))


if __name__ == '__main__':
    main()
