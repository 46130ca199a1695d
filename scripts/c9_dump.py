#!/usr/bin/env python3
"""Developer check: the regularised volume and depth of two seeded cost-volume inputs (a cfg1 batch; a volume with partial
28-wide x tiles and a ragged edge list) written to an .npz -- run once per conv9+prob kernel (default = tile kernel,
`--option c9_kernel=1` = the depth-march experiment, csrc/conv9z.hip, -DV3D_EXPERIMENTS builds only; `--option c12_march=0` = the conv1 /
conv2 tile kernels) and compare.  Exit code 3: the library refuses the option (experiment not in this build)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'):      # a variant build (scripts/build_variant.py)
    libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
Batch = importlib.import_module('3dvnet_amd.batch').Batch
dev = torch.device('cuda:0')
opts = {}
for a in sys.argv[2:]:
    if a.startswith('--option='):                      # developer options of the library (include/v3d.h: v3d_set_option)
        name, val = a[len('--option='):].split('=')
        opts[name] = int(val)
        if libm.load().v3d_set_option(name.encode(), int(val)) == -5:       # V3D_ERR_UNSUPPORTED
            sys.exit(3)
out = {}
sd = syn.costregnet_weights(seed=3, sharpen=200.0)
inp = syn.make_costvolume_inputs('cfg1', n_ref=3, seed=21)
net = mvs.MVSNet(32, inp['img_size']).eval()
net.cnn_3d.load_state_dict(sd, strict=False)
net = net.to(dev)
b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
d0, dd, D = inp['depth']
with torch.no_grad():
    depth, _, reg = net.cost_volume_depth(inp['feat'].to(dev), b, d0, dd, D, inp['plane_size'], return_intermediates=True)
out['depth_a'], out['reg_a'] = depth.cpu().numpy(), reg.cpu().numpy()
img_size, feat_size, plane_size, D = (96, 160), (24, 40), (24, 40), 24
R, tv, K = syn.make_cameras(12, img_size, seed=11)
feat = syn.make_features(12, 32, *feat_size, seed=11)
edges = torch.tensor([[4] + [7] * 3 + [2] * 9, [4] + [6, 7, 8] + list(range(0, 9))])
net2 = mvs.MVSNet(32, img_size).eval()
net2.cnn_3d.load_state_dict(sd, strict=False)
net2 = net2.to(dev)
b2 = Batch(None, R, tv, K, None, edges).to(dev)
with torch.no_grad():
    depth, _, reg = net2.cost_volume_depth(feat.to(dev), b2, 0.5, 0.1, D, plane_size, return_intermediates=True)
out['depth_b'], out['reg_b'] = depth.cpu().numpy(), reg.cpu().numpy()
out['kernel'] = 'conv9z_kernel' if opts.get('c9_kernel') == 1 else 'conv9_prob_kernel'
np.savez(sys.argv[1], **out)
