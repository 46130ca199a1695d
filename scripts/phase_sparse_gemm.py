#!/usr/bin/env python3
"""Developer tool: one sparse-convolution-shaped gather-GEMM (27 segments over one source, row maps = a synthetic neighbour
table: rows within +-span of the output row, a fraction absent), timed; with a library built with -DV3D_PHASE_TIMING also the
phase shares of the rounds kernel (0 row table, 1 barrier, 2 commit incl. the wait for the gathers, 3 barrier, 4 gather issue,
5 MFMA + fragment loads, 6 epilogue).    python scripts/phase_sparse_gemm.py --rows 13500 --c 128"""
import argparse, ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=13500); ap.add_argument('--c', type=int, default=128)
ap.add_argument('--span', type=int, default=2000); ap.add_argument('--absent', type=float, default=0.5)
args = ap.parse_args()
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
sm = importlib.import_module('3dvnet_amd.scenemodeling'); lib = libm.load()
dev = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
M, C = args.rows, args.c
w = torch.randn(27, C, C, generator=g) * 0.02
pk = sm.PackedGemm(w, C * C, 1, C, 27, C, C)
x = torch.randn(M, C, generator=g).to(dev)
nbr = (torch.arange(M)[None, :] + torch.randint(-args.span, args.span + 1, (27, M), generator=g)).clamp_(0, M - 1)
nbr[torch.rand(27, M, generator=g) < args.absent] = -1
nbr[13] = torch.arange(M)
nbr = nbr.to(torch.int32).to(dev).contiguous()
idxs = [nbr[k] for k in range(27)]
out = torch.empty(M, C, device=dev)
run = lambda: pk(M, [x] * 27, idxs=idxs, relu_out=True, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10): run()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 10
msg = 'M=%d C=%d: %.1f us = %.1f TFLOP/s' % (M, C, ms * 1e3, 2.0 * M * 27 * (1 - args.absent) * C * C / ms / 1e9)
if hasattr(lib, 'v3d_debug_gemm_phase_read'):
    fn = lib.v3d_debug_gemm_phase_read; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)(); rows = dict(i.split('=') for i in filter(None, os.environ.get('V3D_OPTIONS', '').split(','))).get('gemm_round_rows', '32'); rows = int(rows) or 32; nb = (M + rows - 1) // rows; fn(buf, nb); tot = sum(buf)
    msg += '; cycles/workgroup %.0f: ' % (tot / nb) + ' '.join('%d:%.0f' % (i, v / nb) for i, v in enumerate(buf))
print(msg)
