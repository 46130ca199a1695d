#!/usr/bin/env python3
"""Developer tool: per-phase cycle counts of an instrumented kernel.

Builds a private copy of the library with -DV3D_PHASE_TIMING (wave 0 of every workgroup adds the cycles between
PHASE_MARK(i) points to a device array), runs the cfg2 cost-volume path once and prints the average cycles per
workgroup and phase.     python scripts/phase_timing.py [--refs 32]
"""
import argparse
import ctypes
import importlib
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=32)
    ap.add_argument('--blocks', type=int, default=0, help='workgroups of the kernel whose counters are read')
    args = ap.parse_args()
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    libm = importlib.import_module('3dvnet_amd._lib')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    lib = libm.load()
    fn = lib.v3d_debug_phase_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    dev = torch.device('cuda:0')
    inp = syn.make_costvolume_inputs('cfg2', n_ref=args.refs)
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net = net.to(dev)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)
    d0, dd, D = inp['depth']
    buf = (ctypes.c_ulonglong * 8)()
    n_blocks = args.blocks or args.refs * 24 * 7 * 2
    with torch.no_grad():
        net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        fn(buf, n_blocks)
    fn2 = lib.v3d_debug_psv_phase_read
    fn2.restype = ctypes.c_int
    fn2.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf2 = (ctypes.c_ulonglong * 8)()
    nb2 = min(args.refs * 24 * 392, 65536)
    fn2(buf2, nb2)
    print('psv cycles per workgroup and phase:', ['%.0f' % (v / nb2) for v in buf2], 'total %.0f' % (sum(buf2) / nb2))
    print('cycles per workgroup and phase:', ['%.0f' % (v / n_blocks) for v in buf], 'total %.0f' % (sum(buf) / n_blocks))


if __name__ == '__main__':
    main()
