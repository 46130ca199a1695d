#!/usr/bin/env python3
"""Per-kernel timing of the cost-volume path at BASELINE config-2 shapes (developer tool).

    python scripts/bench_layers.py [--refs 32] [--iters 10] [--only psv,conv0,...]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=32)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--tag', default='')
    ap.add_argument('--precision', default='split_bf16', choices=['split_bf16', 'fp32'])
    args = ap.parse_args()
    libm = importlib.import_module('3dvnet_amd._lib')
    if os.environ.get('V3D_LIB_OVERRIDE'):      # a variant build (scripts/build_variant.py)
        libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    dev = torch.device('cuda:0')
    inp = syn.make_costvolume_inputs('cfg2', n_ref=args.refs)
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net.cnn_3d.precision = args.precision
    net = net.to(dev)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)
    d0, dd, D = inp['depth']
    with torch.no_grad():
        for _ in range(2):
            net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        torch.cuda.synchronize()
        libm.timing_enable(True)
        for _ in range(args.iters):
            net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        torch.cuda.synchronize()
    st = libm.timing_collect()
    tot = sum(ms for ms, _ in st.values()) / args.iters
    print('%s total %.3f ms/step  %.0f maps/s | ' % (args.tag, tot, args.refs / tot * 1e3) +
          ' '.join('%s=%.3f' % (k.replace('costreg_', ''), ms / c) for k, (ms, c) in st.items()))


if __name__ == '__main__':
    main()
