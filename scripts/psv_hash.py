#!/usr/bin/env python3
"""Developer check: SHA-256 of the variance volume for a few seeded inputs -- run once per kernel variant
(V3D_PSV_GATHER=1 / default) and compare: the variants must agree bit for bit."""
import hashlib
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
dev = torch.device('cuda:0')
for cfg, n_ref, seed in (('cfg2', 4, 1), ('cfg1', 3, 2), ('cfg2', 2, 3)):
    inp = syn.make_costvolume_inputs(cfg, n_ref=n_ref, seed=seed)
    d0, dd, D = inp['depth']
    var = mvs.plane_sweep_variance(inp['feat'].to(dev), inp['rotmats'], inp['tvecs'], inp['K'], inp['edges'].to(dev),
                                   d0, dd, D, inp['img_size'], inp['plane_size'])
    torch.cuda.synchronize()
    print(cfg, n_ref, hashlib.sha256(var.cpu().numpy().tobytes()).hexdigest()[:16], float(var.double().sum()))
# 7 edges per reference (not a power of two): the mean takes the division path
R, tv, K = syn.make_cameras(9, (64, 80), seed=5)
feat = syn.make_features(9, 32, 16, 20, seed=5)
edges = torch.tensor([[4] * 7, [0, 1, 2, 3, 5, 6, 7]])
var = mvs.plane_sweep_variance(feat.to(dev), R, tv, K, edges.to(dev), 0.5, 0.2, 8, (64, 80), (16, 16))
print('7-edge', hashlib.sha256(var.cpu().numpy().tobytes()).hexdigest()[:16], float(var.double().sum()))
