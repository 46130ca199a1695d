#!/usr/bin/env python3
"""Developer check: the fused cost-volume path vs the CPU oracle on random shapes (volume extents any multiples of 8,
random image / feature sizes, ragged random edge lists).  Not part of the test suite (the oracle needs ~1 s per case).
Round 1: 16 cases, worst variance error 3.6e-5, regularised volume 2.2e-5 of max, depth 1.0e-4 (1.03e-4 for the exact-fp32
chain on the same cases), split-variance path bit-identical to the fp32-variance path in every case.
    python scripts/fuzz_costvolume.py [--cases 12] [--seed 0]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
Batch = importlib.import_module('3dvnet_amd.batch').Batch
from oracle import costvolume as ocv   # noqa: E402  (developer tool: the oracle is the checker here)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=12)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    dev = torch.device('cuda:0')
    worst = dict(var=0.0, reg=0.0, depth=0.0)
    for case in range(args.cases):
        D = int(rng.choice([8, 16, 24, 32, 48]))
        h, w = int(rng.integers(1, 9)) * 8, int(rng.integers(1, 10)) * 8
        Hf, Wf = int(rng.integers(10, 60)), int(rng.integers(10, 70))
        H, W = 4 * Hf + int(rng.integers(0, 4)), 4 * Wf + int(rng.integers(0, 4))
        n_img = int(rng.integers(3, 9))
        R, tv, K = syn.make_cameras(n_img, (H, W), seed=int(rng.integers(1 << 30)))
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
            depth_o, var_o, reg_o = ocv.mvsnet_depth(feat, R, tv, K, edges, sd, d0, dd, D, (H, W), (h, w))
            net = mvs.MVSNet(32, (H, W)).eval()
            net.cnn_3d.load_state_dict(sd, strict=False)
            net = net.to(dev)
            b = Batch(None, R, tv, K, None, edges).to(dev)
            depth, var, reg = net.cost_volume_depth(feat.to(dev), b, d0, dd, D, (h, w), return_intermediates=True)
            depth2 = depth if os.environ.get('V3D_COSTREG_GENERIC') else net.cost_volume_depth(feat.to(dev), b, d0, dd, D, (h, w))
        torch.cuda.synchronize()
        ev = float((var.cpu() - var_o).abs().max())
        er = float((reg.cpu() - reg_o).abs().max() / reg_o.abs().max())
        ed = float(((depth.cpu() - depth_o).abs() / depth_o.abs()).max())
        same = bool(torch.equal(depth, depth2))
        worst = dict(var=max(worst['var'], ev), reg=max(worst['reg'], er), depth=max(worst['depth'], ed))
        print('case %2d D=%d h=%d w=%d feat=%dx%d img=%dx%d n_img=%d edges=%d: var %.2e reg %.2e depth %.2e split==fp32var %s'
              % (case, D, h, w, Hf, Wf, H, W, n_img, edges.shape[1], ev, er, ed, same))
        # depth gate 2e-4 here: with soft-argmin weights sharpened x200 on random coarse feature grids the fp32 coordinate
        # noise of the variance volume alone (~3e-5) moves the depth by ~1e-4 -- the exact-fp32 chain (V3D_COSTREG_GENERIC=1)
        # shows the same figures; on the BASELINE shapes both sit at 5e-5 (tests, bench)
        assert ev <= 5e-5 and er <= 2e-4 and ed <= 2e-4 and same
    print('worst', worst)


if __name__ == '__main__':
    main()
