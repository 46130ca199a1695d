#!/usr/bin/env python3
"""Developer (library built with -DV3D_PHASE_TIMING): a PointNet-shaped dense layer (identity row map, optionally a second, pooled
segment) on the one-step gather-GEMM: time and phase cycles per workgroup (0 prologue, 1 barrier, 2 commit incl. the wait for the
loads, 3 barrier, 4 issue, 5 MFMA, 6 epilogue).    python scripts/micro/phase_dense_gemm.py --rows 200704 --k 128 --seg 2"""
import argparse, ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=200704); ap.add_argument('--k', type=int, default=128); ap.add_argument('--n', type=int, default=128)
ap.add_argument('--seg', type=int, default=1); ap.add_argument('--vox', type=int, default=59975)
args = ap.parse_args()
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
sm = importlib.import_module('3dvnet_amd.scenemodeling'); lib = libm.load()
dev = torch.device('cuda:0'); g = torch.Generator().manual_seed(0)
M, K, N, S = args.rows, args.k, args.n, args.seg
w = torch.randn(N, S * K, generator=g) * 0.05
pk = sm.PackedGemm(w, K, S * K, 1, S, N, K, bias=torch.randn(N, generator=g))
x = torch.randn(M, K, generator=g).to(dev)
pool = torch.randn(args.vox, K, generator=g).to(dev)
idx = (torch.arange(M) * args.vox // M).to(torch.int32).to(dev)
out = torch.empty(M, N, device=dev)
run = (lambda: pk(M, [x], relu_in=True, out=out)) if S == 1 else (lambda: pk(M, [x, pool], idxs=[None, idx], relu_in=True, out=out))
for _ in range(3): run()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10): run()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 10
msg = 'M=%d K=%d x %d N=%d: %.1f us, %.2f TB/s of rows in + out' % (M, K, S, N, ms * 1e3, (M * K * 4 + M * N * 4) / ms / 1e9)
if hasattr(lib, 'v3d_debug_gemm_phase_read'):
    fn = lib.v3d_debug_gemm_phase_read; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)(); nb = (M + 127) // 128; fn(buf, nb); tot = sum(buf)
    msg += '; cycles/workgroup %.0f: ' % (tot / nb) + ' '.join('%d:%.0f' % (i, v / nb) for i, v in enumerate(buf))
print(msg)
