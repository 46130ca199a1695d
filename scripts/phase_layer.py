#!/usr/bin/env python3
"""Developer tool (library built with -DV3D_PHASE_TIMING): per-phase cycles of ONE regulariser layer at cfg2 shapes.
    python scripts/phase_layer.py --layer 2 [--refs 32]"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {0: (32, 1), 1: (8, 1), 2: (16, 2), 3: (16, 2), 4: (32, 4), 5: (32, 4), 6: (64, 8), 7: (64, 8), 8: (32, 4), 9: (16, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', type=int, required=True)
    ap.add_argument('--refs', type=int, default=32)
    ap.add_argument('--blocks', type=int, default=0)
    ap.add_argument('--precision', default='split_bf16')
    args = ap.parse_args()
    libm = importlib.import_module('3dvnet_amd._lib')
    if os.environ.get('V3D_LIB_OVERRIDE'):
        libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    lib = libm.load()
    fn = lib.v3d_debug_phase_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    dev = torch.device('cuda:0')
    net = mvs.CostRegNet(32, 8).eval()
    net.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net = net.to(dev)
    ci, div = SHAPES[args.layer]
    x = torch.rand(args.refs, ci, 96 // div, 56 // div, 56 // div, device=dev)
    skip = torch.rand(args.refs, {7: 32, 8: 16, 9: 8}[args.layer], 2 * x.shape[2], 2 * x.shape[3], 2 * x.shape[4], device=dev) if args.layer >= 7 else None
    buf = (ctypes.c_ulonglong * 8)()
    with torch.no_grad():
        for _ in range(2):
            net.run_layer(args.layer, x, skip, precision=args.precision)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            net.run_layer(args.layer, x, skip, precision=args.precision)
        t1.record(); torch.cuda.synchronize()
    nb = args.blocks or 65536
    fn(buf, nb)
    tot = sum(buf)
    print('layer %d: %.3f ms; phase shares:' % (args.layer, t0.elapsed_time(t1) / 5),
          ' '.join('%d:%.1f%%' % (i, 100.0 * v / max(tot, 1)) for i, v in enumerate(buf)))


if __name__ == '__main__':
    main()
