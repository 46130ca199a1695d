#!/usr/bin/env python3
"""Developer timing: every launch of the native backbone (csrc/backbone.hip) on 71 images of 256 x 320, HIP-event time per layer
with shapes, algorithmic bytes (in + out activations) and the bandwidth / FLOP rate they imply."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bb = importlib.import_module('3dvnet_amd.backbone'); syn = importlib.import_module('3dvnet_amd.synthetic'); libm = importlib.import_module('3dvnet_amd._lib')
dev = torch.device('cuda:0')
fe, fs = bb.build_backbone(32)
sd_e, sd_s = syn.backbone_weights(32, seed=6); fe.load_state_dict(sd_e, strict=False); fs.load_state_dict(sd_s)
fe, fs = fe.eval().to(dev), fs.eval().to(dev)
imgs = syn.make_images(71, (256, 320), seed=8).to(dev)
nat = bb.NativeBackbone(fe, fs)
rows = []
def wrap(cls, kind):
    orig = cls.__call__
    def call(self, x, *a, **k):
        torch.cuda.synchronize(); libm.timing_collect(); libm.timing_enable(True)
        for _ in range(5): out = orig(self, x, *a, **k)
        torch.cuda.synchronize(); st = libm.timing_collect(); libm.timing_enable(False)
        ms = sum(v[0] for v in st.values()) / 5
        byt = (x.numel() + out.numel()) * 4
        if kind == 'gemm':
            fl = 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * self.cout * self.cin * self.taps
            desc = 'conv %dx%d %4d -> %4d' % (3 if self.taps == 9 else 1, 3 if self.taps == 9 else 1, self.cin, self.cout)
        else:
            fl = 2.0 * out.numel() * self.k * self.k
            desc = 'dw k%d s%d %4d' % (self.k, self.stride, self.c)
        rows.append((desc, tuple(x.shape[1:3]), ms, byt / ms / 1e6, fl / ms / 1e9))
        return out
    cls.__call__ = call
wrap(bb._Gemm, 'gemm'); wrap(bb._Depthwise, 'dw')
with torch.no_grad():
    nat(imgs); rows.clear(); nat(imgs)
tot = sum(r[2] for r in rows)
for d, hw, ms, gbs, tf in rows:
    print('%-24s in %3dx%3d  %7.3f ms  %7.0f GB/s  %6.1f TFLOP/s' % (d, hw[0], hw[1], ms, gbs, tf))
print('sum of layers %.3f ms (stem and layout kernels not included)' % tot)
