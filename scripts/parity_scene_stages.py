#!/usr/bin/env python3
"""Developer check: stage-by-stage comparison of the HIP scene path with the oracle on an n-view cfg3-shaped scene (surface-like
depths): point cloud, voxelisation, PointNet, sparse U-Net levels, first point-flow offsets.   python scripts/parity_scene_stages.py [n] [seed]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel')
drv = importlib.import_module('3dvnet_amd.eval_3dvnet'); utils = importlib.import_module('3dvnet_amd.utils')
from oracle import scene as osc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 77
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; nb, na = 4, 3
edges, n_img = syn.make_edges(n, nb, na)
rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=seed, yaw_step_deg=360.0 / max(n_img, 60))
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=seed)
gt = syn.ray_box_depth(rot[nb:nb + n], tv[nb:nb + n], K[nb:nb + n], cfg['img_size'], drv.DEPTH_CONFIG['size'])
gt = gt + 0.02 * torch.randn(gt.shape, generator=torch.Generator().manual_seed(7))
sds = dict(pn=syn.pointnet_weights(), un=syn.sparse_unet_weights(), dec=syn.decoder_weights(sharpen=50.0))
net = lm.PL3DVNet(None, drv.DEPTH_CONFIG, cfg['edge_len'], feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(sds['pn']); net.sparse_conv.load_state_dict(sds['un']); net.decoder.load_state_dict(sds['dec'], strict=False); net = net.to(dev)
db = torch.zeros(n, dtype=torch.long)
if '--after-iter1' in sys.argv:      # the depths the oracle holds after its first outer iteration (where cell flips have been seen)
    from oracle.net import OracleNet
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    bb = Batch(None, rot, tv, K, None, edges); bb.features_quarter = feat
    onet = OracleNet(syn.costregnet_weights(seed=0, sharpen=200.0), sds['pn'], sds['un'], sds['dec'], cfg['img_size'], cfg['edge_len'], pinned=True)
    with torch.no_grad():
        gt = drv.process_scene(bb, onet, (nb, na), torch.device('cpu'), init_depth_override=gt, offsets_list=[drv.OFFSETS_LIST[0]])
torch.set_num_threads(min(32, os.cpu_count() or 1))
def mx(a, b): return float((a - b).abs().max())
with torch.no_grad():
    g = lambda t: t.to(dev)
    pts, pf, pb = net.construct_feature_rich_pointcloud(g(gt), g(db), g(feat), g(rot), g(tv), g(K), g(edges))
    pts_o, pf_o, pb_o = osc.feature_rich_pointcloud(gt, db, feat, rot, tv, K, edges, cfg['img_size'], pinned=True)
    print('points bit-identical to the pinned oracle:', torch.equal(pts.cpu(), pts_o))
    print('points: max |d| %.3e   features: %.3e' % (mx(pts.cpu(), pts_o), mx(pf.cpu(), pf_o)))
    a_pts, a_idx, a_b, e = utils.voxelize(pts, pb, cfg['edge_len'])
    a_pts_o, a_idx_o, a_b_o, e_o = osc.voxelize(pts_o, pb_o, cfg['edge_len'])
    same_n = a_pts.shape[0] == a_pts_o.shape[0]
    print('voxels: %d vs %d; idx equal %s; point->voxel equal %s (%d points differ)' % (a_pts.shape[0], a_pts_o.shape[0],
          same_n and torch.equal(a_idx.cpu(), a_idx_o), same_n and torch.equal(e.cpu(), e_o), int((e[0].cpu() != e_o[0]).sum()) if same_n else -1))
    # which cell does each point fall into?  (independent of the row order of the anchors)
    cell = lambda p_, a_, e_: torch.round((p_[e_[1]] - a_[e_[0]]) / cfg['edge_len'] * 0 + a_[e_[0]] / cfg['edge_len'] * 1e0).long()
    ca, cb = a_idx.cpu().long()[e[0].cpu()], a_idx_o.long()[e_o[0]]
    ndiff = int((ca != cb).any(dim=1).sum())
    mins = (float((pts.cpu().min(0).values - pts_o.min(0).values).abs().max()), float((pts.cpu().max(0).values - pts_o.max(0).values).abs().max()))
    print('points whose integer cell differs: %d; bbox min / max differ by %.3e / %.3e' % (ndiff, mins[0], mins[1]))
    if ndiff:
        i = int((ca != cb).any(dim=1).nonzero()[0])
        j = int(e_o[1][i])
        print('  e.g. point %d: HIP %s cell %s | oracle %s cell %s | (p - min)/edge oracle %s' % (j, pts.cpu()[e[1].cpu()[i]].tolist(), ca[i].tolist(), pts_o[j].tolist(), cb[i].tolist(),
              ((pts_o[j] - pts_o.min(0).values) / cfg['edge_len']).tolist()))
    # the HIP path on the ORACLE's point cloud: isolates the voxelisation / networks from the back-projection's last bits
    xs = net.model_scene(g(gt), g(db), g(feat), g(rot), g(tv), g(K), g(edges))
    x_o = torch.cat((pts_o[e_o[1]] - a_pts_o[e_o[0]], pf_o[e_o[1]]), dim=1)
    pn_o = osc.pointnet(x_o, e_o[0], a_pts_o.shape[0], sds['pn'])
    xs_o = osc.sparse_unet(pn_o, a_pts_o, a_idx_o, a_b_o, cfg['edge_len'], sds['un'])
    for lv, (a, b) in enumerate(zip(xs, xs_o)):
        fa, fb = a['feats'].cpu(), b['feats']
        if fa.shape != fb.shape:
            print('level %d: shapes %s vs %s' % (lv, tuple(fa.shape), tuple(fb.shape))); continue
        # rows may be ordered differently: match by integer coordinates
        ka = (a['idx'].cpu().long() * torch.tensor([1, 1 << 20, 1 << 40])).sum(1); kb = (b['idx'].long() * torch.tensor([1, 1 << 20, 1 << 40])).sum(1)
        oa, ob = ka.argsort(), kb.argsort()
        print('level %d (%d rows): coords equal %s, features max |d| %.3e (max |f| %.3e)' % (lv, fa.shape[0], torch.equal(ka[oa], kb[ob]), mx(fa[oa], fb[ob]), float(fb.abs().max())))
    off = net.run_pointflow(xs, g(gt), g(db), g(feat), g(rot), g(tv), g(K), g(edges), 0.05, 3).cpu()
    off_o = osc.run_pointflow(xs_o, gt, db, feat, rot, tv, K, edges, 0.05, 3, sds['dec'], cfg['img_size'], pinned=True)
    d = (off - off_o).abs()
    print('offsets: max |d| %.3e m, #pixels > 1e-5 m: %d of %d; max |offset| %.3f' % (float(d.max()), int((d > 1e-5).sum()), d.numel(), float(off_o.abs().max())))
