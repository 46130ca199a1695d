#!/usr/bin/env python3
"""Developer tool (library built with -DV3D_PHASE_TIMING, e.g. scripts/build_variant.py costreg.hip phase -DV3D_PHASE_TIMING):
average cycles per workgroup and phase of ONE split-bf16 layer (convg / deconvg kernels) at cfg2 shapes.
    V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_phase.so python scripts/phase_convg.py --layer 2 [--refs 64]"""
import argparse, ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {1: (8, 1), 2: (16, 2), 3: (16, 2), 4: (32, 4), 5: (32, 4), 6: (64, 8), 7: (64, 8), 8: (32, 4)}
ap = argparse.ArgumentParser()
ap.add_argument('--layer', type=int, required=True)
ap.add_argument('--refs', type=int, default=64)
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
skip = torch.rand(args.refs, {7: 32, 8: 16}[args.layer], 2 * x.shape[2], 2 * x.shape[3], 2 * x.shape[4], device=dev) if args.layer >= 7 else None
with torch.no_grad():
    for _ in range(3):
        net.run_layer(args.layer, x, skip, split=True)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
nb = 4096
fn(buf, nb)
print('layer %d: cycles per workgroup (first %d):' % (args.layer, nb), ' '.join('%d:%.0f' % (i, v / nb) for i, v in enumerate(buf)), 'sum %.0f' % (sum(buf) / nb))
