#!/usr/bin/env python3
"""Developer helper: depth + regularised volume of a seeded cfg1 batch, saved to the .npz given as argv[1];
argv[2] = 'split_bf16' (default) or 'fp32' selects the regulariser's MFMA operand precision."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
Batch = importlib.import_module('3dvnet_amd.batch').Batch
dev = torch.device('cuda:0')
inp = syn.make_costvolume_inputs('cfg1', n_ref=2, seed=21)
net = mvs.MVSNet(32, inp['img_size']).eval()
net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=3, sharpen=200.0), strict=False)
net = net.to(dev)
b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
d0, dd, D = inp['depth']
with torch.no_grad():
    depth, var, reg = net.cost_volume_depth(inp['feat'].to(dev), b, d0, dd, D, inp['plane_size'], return_intermediates=True,
                                            precision=sys.argv[2] if len(sys.argv) > 2 else 'split_bf16')
torch.cuda.synchronize()
np.savez(sys.argv[1], depth=depth.cpu().numpy(), reg=reg.cpu().numpy())
print('saved', sys.argv[1], float(depth.min()), float(depth.max()))
