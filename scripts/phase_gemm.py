#!/usr/bin/env python3
"""Developer tool (library built with -DV3D_PHASE_TIMING): phase shares of the gather-GEMM on a decoder-like conv1d
layer: M = 7 * points rows, K = 3 x cin, N = cout.     python scripts/phase_gemm.py [--cin 352] [--cout 128]"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, default=352)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--points', type=int, default=50176)
    args = ap.parse_args()
    sm = importlib.import_module('3dvnet_amd.scenemodeling')
    libm = importlib.import_module('3dvnet_amd._lib')
    lib = libm.load()
    fn = lib.v3d_debug_gemm_phase_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    dev = torch.device('cuda:0')
    w = torch.randn(args.cout, args.cin, 3) * 0.05
    pk = sm.PackedGemm(w, 1, 3 * args.cin, 3, 3, args.cout, args.cin)
    M = 7 * args.points
    x = torch.randn(M, args.cin, device=dev)
    out = torch.empty(M, args.cout, device=dev)
    def run():
        pk(M, [x, x, x], group_len=7, relu_out=True, out=out)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        run()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 5
    buf = (ctypes.c_ulonglong * 8)()
    nb = min((M + 127) // 128, 65536)
    fn(buf, nb)
    tot = sum(buf)
    print('M=%d K=3x%d N=%d: %.3f ms = %.0f TFLOP/s; cycles/workgroup %.0f; phase shares:' %
          (M, args.cin, args.cout, ms, 2.0 * M * 3 * args.cin * args.cout / ms / 1e9, tot / nb),
          ' '.join('%d:%.1f%%' % (i, 100.0 * v / max(tot, 1)) for i, v in enumerate(buf)))


if __name__ == '__main__':
    main()
