#!/usr/bin/env python3
"""Developer timing: the native backbone on 71 images of 256 x 320 (and 240 x 320), wall time per batch and HIP-event time per kernel
family, both precisions.   [V3D_LIB_OVERRIDE=...] python scripts/bench_backbone.py"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bb = importlib.import_module('3dvnet_amd.backbone'); syn = importlib.import_module('3dvnet_amd.synthetic'); libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'):
    libm.LIB_PATH = os.path.abspath(os.environ['V3D_LIB_OVERRIDE'])
dev = torch.device('cuda:0')
fe, fs = bb.build_backbone(32)
sd_e, sd_s = syn.backbone_weights(32, seed=6); fe.load_state_dict(sd_e, strict=False); fs.load_state_dict(sd_s)
fe, fs = fe.eval().to(dev), fs.eval().to(dev)
for size in ((256, 320), (240, 320)):
    imgs = syn.make_images(71, size, seed=8).to(dev)
    for precision in ('split_bf16', 'fp32'):
        nat = bb.NativeBackbone(fe, fs, precision=precision)
        with torch.no_grad():
            for _ in range(3): nat(imgs)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): nat(imgs)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
            libm.timing_collect(); libm.timing_enable(True)
            for _ in range(5): nat(imgs)
            torch.cuda.synchronize(); st = libm.timing_collect(); libm.timing_enable(False)
        print('%s %-10s  %.3f ms per batch (wall), kernels %.3f ms, %d launches' % (size, precision, ms, sum(v[0] for v in st.values()) / 5, sum(v[1] for v in st.values()) // 5))
        for k, v in sorted(st.items(), key=lambda kv: -kv[1][0]):
            print('      %-28s %7.3f ms  %3d launches' % (k, v[0] / 5, v[1] // 5))
