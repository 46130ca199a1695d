#!/usr/bin/env python3
"""Experiment: one stream with 2R views per step vs two streams with R views each (independent half-batches)."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
Batch = importlib.import_module('3dvnet_amd.batch').Batch
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = syn.costregnet_weights(seed=0, sharpen=200.0)


def make(n_ref, seed):
    inp = syn.make_costvolume_inputs('cfg2', n_ref=n_ref, seed=seed)
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(sd, strict=False)
    net = net.to(dev)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)
    d0, dd, D = inp['depth']
    return lambda: net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])


one = make(2 * R, 1)
a, b2 = make(R, 2), make(R, 3)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    for _ in range(3):
        one(); a(); b2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        with torch.cuda.stream(s1):
            a()
        with torch.cuda.stream(s2):
            b2()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print('one stream x %d views: %.0f maps/s; two streams x %d views: %.0f maps/s' %
      (2 * R, 2 * R * 20 / (t1 - t0), R, 2 * R * 20 / (t2 - t1)))
