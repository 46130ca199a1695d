#!/usr/bin/env python3
"""Developer: every gather-GEMM and segmented maximum of one PointNet.forward on the cfg3 scene with its shape and HIP-event time."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel'); ut = importlib.import_module('3dvnet_amd.utils')
sm = importlib.import_module('3dvnet_amd.scenemodeling'); libm = importlib.import_module('3dvnet_amd._lib')
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; n_ref, k = 64, 2
edges, n_img = syn.make_edges(n_ref, k, k); rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5, yaw_step_deg=360.0 / n_img)
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(dev)
depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56))
depth = (depth + 0.02 * torch.randn(depth.shape, generator=torch.Generator().manual_seed(1))).to(dev)
rot, tv, K, edges = rot.to(dev), tv.to(dev), K.to(dev), edges.to(dev); db = torch.zeros(n_ref, dtype=torch.long, device=dev)
net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(syn.pointnet_weights()); net = net.to(dev)
log = []
orig = sm.PackedGemm.__call__
def timed(self, M, srcs, idxs=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(self, M, srcs, idxs=idxs, **kw); e1.record()
    log.append((M, self.K, self.N, self.n_seg, e0, e1))
    return y
with torch.no_grad():
    pts, pf, pb = net.construct_feature_rich_pointcloud(depth, db, feat, rot, tv, K, edges)
    a_pts, a_idx, a_batch, a_e = ut.voxelize(pts, pb, 0.04)
    x = torch.cat((pts[a_e[1]] - a_pts[a_e[0]], pf[a_e[1]]), dim=1)
    for _ in range(3): net.pointnet(x, a_e[0], a_pts.shape[0])
    sm.PackedGemm.__call__ = timed
    libm.timing_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net.pointnet(x, a_e[0], a_pts.shape[0]); e1.record()
    torch.cuda.synchronize(); st = libm.timing_collect(); libm.timing_enable(False)
print('points', x.shape, 'voxels', a_pts.shape[0], 'forward %.3f ms' % e0.elapsed_time(e1))
for M, Kc, N, ns, a, b in log:
    print('M=%6d K=%3d N=%3d seg=%2d  %.1f us' % (M, Kc, N, ns, a.elapsed_time(b) * 1e3))
print({k_: (round(ms, 3), c) for k_, (ms, c) in st.items()})
