#!/usr/bin/env python3
"""Developer tool (variant library built with -DV3D_PHASE_TIMING, scripts/build_variant.py conv0z.hip ...): cycles of wave 0
per phase of the depth-march conv0 kernel at cfg2 shapes, summed over the workgroups.
    V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_czph.so python scripts/phase_conv0z.py [--refs 64]"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ['M: wait at B', "M: MFMA phase (B', stores inside)", "H: wait at B'", 'M: task end barriers', 'H: task prologue', 'H: wait vmcnt', 'H: wait at B', 'H: issue DMA + epilogue']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=64)
    ap.add_argument('--kernel', default='conv0z', choices=('conv0z', 'conv9z', 'conv12z'))
    args = ap.parse_args()
    libm = importlib.import_module('3dvnet_amd._lib')
    if os.environ.get('V3D_LIB_OVERRIDE'):
        libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    lib = libm.load()
    fn = getattr(lib, 'v3d_debug_%s_phase_read' % args.kernel)
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
    with torch.no_grad():
        for _ in range(3):
            net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'])
        torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    fn(buf, 1024)
    tot = sum(buf)
    print(args.kernel + ' phases (cycles of wave 0, %d workgroups): total %.3e' % (256, tot))
    names12 = ['1: wait at B (+ loop tail)', '1: odd plane MFMAs', '1: park + even plane', "1: wait at B'", "2: B .. arrival at B' (MFMAs)", "2: wait at B' (red stores, B wait in the next)", "H: B .. arrival at B' (stores, DMA issue, take)", "H: wait at B', vmcnt, B"]
    names = NAMES if args.kernel == 'conv0z' else names12 if args.kernel == 'conv12z' else ['P: wait vmcnt', 'P: wait at B', 'P: issue DMA', 'P: MFMA', 'P: emit u9', 'C: prob conv + stores', 'C: wait at B', 'C: task tail']
    for n, v in zip(names, buf):
        print('  %-34s %6.1f %%  %.3e' % (n, 100.0 * v / max(tot, 1), v))


if __name__ == '__main__':
    main()
