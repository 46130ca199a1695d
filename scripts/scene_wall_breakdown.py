#!/usr/bin/env python3
"""Developer timing: wall time of the cfg3 scene pipeline per stage (synchronised around each call) against the sum of the
library's kernel times in that stage -- the difference is host work, torch glue kernels and synchronisations."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel')
drv = importlib.import_module('3dvnet_amd.eval_3dvnet'); Batch = importlib.import_module('3dvnet_amd.batch').Batch
libm = importlib.import_module('3dvnet_amd._lib')
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; win = (4, 3); refs = 64
edges, n_img = syn.make_edges(refs, *win)
rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=1237, yaw_step_deg=360.0 / n_img)
b = Batch(None, rot, tv, K, None, edges); b.features_quarter = syn.make_features(n_img, 32, *cfg['feat_size'], seed=1237)
net = lm.PL3DVNet(None, drv.DEPTH_CONFIG, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.mvsnet.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False); net = net.to(dev)
gt = syn.ray_box_depth(rot[4:4 + refs], tv[4:4 + refs], K[4:4 + refs], cfg['img_size'], (56, 56))
gt = (gt + 0.02 * torch.randn(gt.shape, generator=torch.Generator().manual_seed(7))).to(dev)
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); libm.timing_collect(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        ks = sum(ms for ms, c in libm.timing_collect().values())
        w, kk, n = acc.get(label, (0, 0, 0)); acc[label] = (w + dt, kk + ks, n + 1)
        return r
    setattr(obj, name, g)
def run():
    return drv.process_scene(b, net, win, dev, init_depth_override=gt, gather_depth=False)
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); print('un-instrumented: %.2f ms per scene' % ((time.perf_counter() - t0) / 3 * 1e3))
libm.timing_enable(True)
wrap(net, 'make_initial_depth_predictions', 'stage 1 (cost volume)'); wrap(net, 'model_scene', 'model_scene'); wrap(net, 'run_pointflow', 'run_pointflow')
wrap(net.pointnet, 'forward', '  pointnet'); wrap(net.sparse_conv, 'forward', '  sparse unet')
ut = importlib.import_module('3dvnet_amd.utils'); 
for _ in range(3): run()
for k_, (w, kk, n) in acc.items(): print('%-26s calls/scene %3d  wall %.2f ms  kernels %.2f ms  gap %.2f ms' % (k_, n // 3, w / 3, kk / 3, (w - kk) / 3))
