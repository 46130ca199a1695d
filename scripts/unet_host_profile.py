#!/usr/bin/env python3
"""Developer: host-side profile (cProfile, cumulative) of SparseUNet.forward on the cfg3 scene."""
import cProfile, pstats, io, importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel'); ut = importlib.import_module('3dvnet_amd.utils')
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; n_ref, k = 64, 2
edges, n_img = syn.make_edges(n_ref, k, k); rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5, yaw_step_deg=360.0 / n_img)
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(dev)
depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56))
depth = (depth + 0.02 * torch.randn(depth.shape, generator=torch.Generator().manual_seed(1))).to(dev)
rot, tv, K, edges = rot.to(dev), tv.to(dev), K.to(dev), edges.to(dev); db = torch.zeros(n_ref, dtype=torch.long, device=dev)
net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights()); net = net.to(dev)
with torch.no_grad():
    pts, pf, pb = net.construct_feature_rich_pointcloud(depth, db, feat, rot, tv, K, edges)
    a_pts, a_idx, a_batch, a_e = ut.voxelize(pts, pb, 0.04)
    x = torch.cat((pts[a_e[1]] - a_pts[a_e[0]], pf[a_e[1]]), dim=1)
    x = net.pointnet(x, a_e[0], a_pts.shape[0])
    for _ in range(3): net.sparse_conv(x, a_pts, a_idx, a_batch, 0.04)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net.sparse_conv(x, a_pts, a_idx, a_batch, 0.04)
    torch.cuda.synchronize(); print('voxels', a_pts.shape[0], 'forward %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): net.sparse_conv(x, a_pts, a_idx, a_batch, 0.04)
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:4500])
