#!/usr/bin/env python3
"""Developer check: SHA-256 of the variance volume for a few seeded inputs -- run once per kernel variant (default = window
kernel, `--option psv_kernel=1` = round-2 reuse kernel, `--option psv_kernel=2` = plain gather) and compare: the variants must agree bit for
bit.  Besides the bench geometries: 7 edges (division path of the mean), and camera pairs for which the window kernel's
16 x 4-cell window cannot hold a wave's footprints (source camera zoomed 3x / rolled by 90 degrees / far off to the side), on a
plane grid that is not a multiple of the 8-pixel tiles and a plane count that is not a multiple of 8 -- fp32 and split output."""
import hashlib
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
dev = torch.device('cuda:0')
for a in sys.argv[1:]:
    if a.startswith('--option='):                      # developer options of the library (include/v3d.h: v3d_set_option)
        name, val = a[len('--option='):].split('=')
        importlib.import_module('3dvnet_amd._lib').set_option(name, int(val))
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

# windows that cannot hold the footprints: every sample (or most) takes the out-of-window path of the window kernel
import math
R, tv, K = syn.make_cameras(6, (64, 80), seed=9)
K = K.clone(); K[1, 0, 0] *= 3.0; K[1, 1, 1] *= 3.0                       # source 1: zoomed 3x (pixels 4 cells apart)
c, s_ = math.cos(math.pi / 2), math.sin(math.pi / 2)
roll = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
R = R.clone(); tv = tv.clone()
R[2] = roll @ R[2]; tv[2] = roll @ tv[2]                                  # source 2: rolled by 90 degrees (rows become columns)
tv[3] = tv[3] + torch.tensor([0.8, 0.0, 0.0])                             # source 3: far off to the side (long epipolar slides)
feat = syn.make_features(6, 32, 16, 20, seed=9)
edges = torch.tensor([[0] * 5 + [4] * 3, [0, 1, 2, 3, 5, 4, 1, 2]])
for split in (False, True):
    var = mvs.plane_sweep_variance(feat.to(dev), R, tv, K, edges.to(dev), 0.4, 0.11, 13, (64, 80), (15, 19), split=split)
    data = var.data if split else var
    print('exotic split=%d' % split, hashlib.sha256(data.cpu().numpy().tobytes()).hexdigest()[:16], float(data.double().sum()) if not split else 0.0)
