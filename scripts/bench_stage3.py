#!/usr/bin/env python3
"""Developer timing: stage 3 alone (three PropagationNets at 64x80, 128x160, 256x320, 64 views), per-kernel HIP-event times.
    [V3D_LIB_OVERRIDE=...] python scripts/bench_stage3.py [--views 64]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--views', type=int, default=64)
ap.add_argument('--tag', default='')
args = ap.parse_args()
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'):
    libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
syn = importlib.import_module('3dvnet_amd.synthetic')
up = importlib.import_module('3dvnet_amd.upsampling')
dev = torch.device('cuda:0')
n = args.views
guides = [syn.make_features(n, 32, 64, 80, seed=1).to(dev), syn.make_features(n, 32, 128, 160, seed=2).to(dev),
          syn.make_images(n, (256, 320), seed=3).to(dev)]
nets = []
for cin, seed in ((33, 5), (33, 6), (4, 7)):
    m = up.PropagationNet(cin, 32).eval()
    m.load_state_dict(syn.propagation_weights(cin, 32, seed), strict=False)
    nets.append(m.to(dev))
depth = 1 + torch.rand((n, 56, 56), device=dev)
with torch.no_grad():
    for _ in range(2):
        up.upsample_depth(depth.clone(), list(zip(nets, guides)))
    torch.cuda.synchronize()
    libm.timing_enable(True)
    for _ in range(5):
        up.upsample_depth(depth.clone(), list(zip(nets, guides)))
    torch.cuda.synchronize()
st = libm.timing_collect()
tot = sum(ms for ms, _ in st.values()) / 5
print('%s stage 3, %d views: %.3f ms | ' % (args.tag, n, tot) + ' '.join('%s=%.3f' % (k.replace('propagation_', ''), ms / 5) for k, (ms, c) in st.items()))
