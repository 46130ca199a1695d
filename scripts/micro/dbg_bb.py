import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
bb = importlib.import_module('3dvnet_amd.backbone'); syn = importlib.import_module('3dvnet_amd.synthetic')
dev = torch.device('cuda:0')
fe, fs = bb.build_backbone(32)
sd_e, sd_s = syn.backbone_weights(32, seed=6); fe.load_state_dict(sd_e, strict=False); fs.load_state_dict(sd_s)
fe, fs = fe.eval().to(dev), fs.eval().to(dev)
n = int(sys.argv[1])
imgs = syn.make_images(n, (256, 320), seed=8).to(dev)
nat = bb.NativeBackbone(fe, fs)
def wrap(cls, name):
    orig = cls.__call__
    def call(self, x, *a, **k):
        out = orig(self, x, *a, **k); torch.cuda.synchronize()
        print(name, tuple(x.shape), '->', tuple(out.shape), 'ok', flush=True)
        return out
    cls.__call__ = call
wrap(bb._Gemm, 'gemm'); wrap(bb._Depthwise, 'dw')
with torch.no_grad():
    out = nat(imgs)
torch.cuda.synchronize(); print('done', n)
