#!/usr/bin/env python3
"""Developer tool: phase cycles of conv9_prob_kernel (the last instrumented kernel of the regulariser) from a library whose
costreg.hip was built with -DV3D_PHASE_TIMING (scripts/build_variant.py; V3D_LIB_OVERRIDE selects it).
    V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_x.so python scripts/phase_c9.py --blocks 512"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=64)
    ap.add_argument('--blocks', type=int, default=512)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    libm = importlib.import_module('3dvnet_amd._lib')
    if os.environ.get('V3D_LIB_OVERRIDE'):
        libm.LIB_PATH = os.path.abspath(os.environ['V3D_LIB_OVERRIDE'])
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
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
    with torch.no_grad():
        for _ in range(2):
            net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=args.refs)
        torch.cuda.synchronize()
        libm.timing_enable(True)
        for _ in range(3):
            net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=args.refs)
        torch.cuda.synchronize()
    st = libm.timing_collect()
    fn(buf, args.blocks)
    tot = sum(buf)
    print('%s conv9_prob %.3f ms; cycles per workgroup %.0f; phases:' % (args.tag, st['costreg_conv9_prob'][0] / st['costreg_conv9_prob'][1], tot / args.blocks),
          ' '.join('%d:%.0f(%.1f%%)' % (i, v / args.blocks, 100.0 * v / max(tot, 1)) for i, v in enumerate(buf)))


if __name__ == '__main__':
    main()
