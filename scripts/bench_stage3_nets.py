#!/usr/bin/env python3
"""Developer timing: each PropagationNet of stage 3 by itself (64x80, 128x160, 256x320; 64 views), HIP-event time per launch.
    python scripts/bench_stage3_nets.py [--views 64]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--views', type=int, default=64)
ap.add_argument('--tag', default='')
ap.add_argument('--precision', default='split_bf16')
args = ap.parse_args()
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'):
    libm.LIB_PATH = os.path.abspath(os.environ['V3D_LIB_OVERRIDE'])
syn = importlib.import_module('3dvnet_amd.synthetic')
up = importlib.import_module('3dvnet_amd.upsampling')
dev = torch.device('cuda:0')
n = args.views
guides = [syn.make_features(n, 32, 64, 80, seed=1).to(dev), syn.make_features(n, 32, 128, 160, seed=2).to(dev),
          syn.make_images(n, (256, 320), seed=3).to(dev)]
out = []
for (cin, seed), gd, lo in zip(((33, 5), (33, 6), (4, 7)), guides, ((56, 56), (64, 80), (128, 160))):
    m = up.PropagationNet(cin, 32, precision=args.precision).eval()
    m.load_state_dict(syn.propagation_weights(cin, 32, seed), strict=False)
    m = m.to(dev)
    d = 1 + torch.rand((n,) + lo, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m.forward_resized(gd, d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.forward_resized(gd, d)
        e1.record()
        torch.cuda.synchronize()
    out.append('%dx%d cin %d: %.3f ms' % (gd.shape[2], gd.shape[3], cin, e0.elapsed_time(e1) / 10))
print(args.tag, ' | '.join(out))
