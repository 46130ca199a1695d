#!/usr/bin/env python3
"""Developer check + timing of the fused inverted-residual kernel (csrc/irb.hip) block by block: against the module on the CPU and
against the three-launch path on the device, 71 images."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bb = importlib.import_module('3dvnet_amd.backbone'); libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'):
    libm.LIB_PATH = os.path.abspath(os.environ['V3D_LIB_OVERRIDE'])
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = int(os.environ.get('N', 71))
cfgs = [(16, 24, 3, 2, 3, 128, 160), (24, 24, 3, 1, 3, 64, 80), (24, 40, 5, 2, 3, 64, 80), (40, 40, 5, 1, 3, 32, 40), (40, 80, 5, 2, 6, 32, 40),
        (80, 80, 5, 1, 6, 16, 20), (80, 96, 3, 1, 6, 16, 20), (96, 96, 3, 1, 6, 16, 20), (96, 192, 5, 2, 6, 16, 20), (192, 192, 5, 1, 6, 8, 10),
        (192, 320, 3, 1, 6, 8, 10),
        (16, 24, 3, 2, 3, 120, 160), (24, 24, 3, 1, 3, 60, 80), (40, 40, 5, 1, 3, 30, 40), (80, 80, 5, 1, 6, 15, 20), (96, 192, 5, 2, 6, 15, 20)]
def timeit(f, it=10):
    """HIP-event time of the library's kernels inside f (ms per call)"""
    for _ in range(3): f()
    torch.cuda.synchronize(); libm.timing_collect(); libm.timing_enable(True)
    for _ in range(it): f()
    torch.cuda.synchronize(); st = libm.timing_collect(); libm.timing_enable(False)
    return sum(v[0] for v in st.values()) / it
for cin, cout, k, s, e, H, W in cfgs:
    blk = bb._InvertedResidual(cin, cout, k, s, e).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(N, H, W, cin)
    with torch.no_grad():
        ref = blk(x[:3].permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    fused = bb._Block(blk, dev)
    ok = fused.supported(H, W)
    L = blk.layers
    wd, bd = bb._fold(L[3], L[4])
    ops = (bb._Gemm(*bb._fold(L[0], L[1]), dev), bb._Depthwise(wd, bd, s, dev), bb._Gemm(*bb._fold(L[6], L[7]), dev))
    xd = x.to(dev)
    def three():
        return ops[2](ops[1](ops[0](xd, relu=True), relu=True), relu=False, res=xd if blk.apply_residual else None, res_mode=1 if blk.apply_residual else 0)
    y3 = three()
    line = '%3d->%3d k%d s%d e%d %3dx%3d  3-launch %.3f ms (err %.1e)' % (cin, cout, k, s, e, H, W, timeit(three), float((y3[:3].cpu() - ref).abs().max() / ref.abs().max()))
    if ok:
        y = fused(xd); y2 = fused(xd)
        err = float((y[:3].cpu() - ref).abs().max() / ref.abs().max())
        line += '  fused %.3f ms  err %.2e of range%s' % (timeit(lambda: fused(xd)), err, '' if torch.equal(y, y2) else '  NOT DETERMINISTIC')
        line += '  vs 3-launch %.2e' % float((y - y3).abs().max() / ref.abs().max())
    else:
        line += '  (no fused kernel)'
    print(line, flush=True)
    if ok and os.environ.get('V3D_LIB_OVERRIDE'):
        import ctypes
        libm.load().v3d_debug_irb_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = (ctypes.c_ulonglong * 8)()
        libm.load().v3d_debug_irb_phase(buf, 1); fused(xd); libm.load().v3d_debug_irb_phase(buf, 1)
        wg = max(1, buf[7])
        print('      cycles per workgroup (wave 0): prologue %d | X %d | barrier1+park %d | W %d | barrier2 %d | P %d | epilogue %d   (%d workgroups)'
              % tuple([buf[k] // wg for k in range(7)] + [wg]), flush=True)
