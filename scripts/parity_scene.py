#!/usr/bin/env python3
"""Developer check: the refinement leg of the cfg3 pipeline on an n-view scene, HIP driver against the oracle-backed driver, one
outer iteration (scene model + 3 sweeps) at a time, with the number of points that fall into DIFFERENT voxel cells in the two runs
at the start of every iteration.  One such point (a 1-ulp difference of a back-projected coordinate next to a cell boundary, or
5e-6 m of accumulated offset difference) adds or moves a voxel, and the sparse U-Net's global receptive field turns that into
centimetres of depth difference for thousands of pixels: the algorithm's discretisation is chaotic at that scale, so end-to-end
figures are only meaningful for runs without a flip.      python scripts/parity_scene.py [n_views] [seed ...]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel')
drv = importlib.import_module('3dvnet_amd.eval_3dvnet'); Batch = importlib.import_module('3dvnet_amd.batch').Batch
from oracle.net import OracleNet
from oracle import scene as osc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seeds = [int(a) for a in sys.argv[2:]] or [77]
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; nb, na = 4, 3
sds = dict(cr=syn.costregnet_weights(seed=0, sharpen=200.0), pn=syn.pointnet_weights(), un=syn.sparse_unet_weights(), dec=syn.decoder_weights(sharpen=50.0))
net = lm.PL3DVNet(None, drv.DEPTH_CONFIG, cfg['edge_len'], feat_dim=32, img_size=cfg['img_size']).eval()
net.mvsnet.cnn_3d.load_state_dict(sds['cr'], strict=False); net.pointnet.load_state_dict(sds['pn']); net.sparse_conv.load_state_dict(sds['un'])
net.decoder.load_state_dict(sds['dec'], strict=False); net = net.to(dev)
torch.set_num_threads(min(32, os.cpu_count() or 1))
onet = OracleNet(sds['cr'], sds['pn'], sds['un'], sds['dec'], cfg['img_size'], cfg['edge_len'], pinned=True)
def cells(p):
    p = p.double().cpu()
    return torch.floor((p - p.min(0).values) / cfg['edge_len']).long()
for seed in seeds:
    edges, n_img = syn.make_edges(n, nb, na)
    rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=seed, yaw_step_deg=360.0 / max(n_img, 60))
    bb = Batch(None, rot, tv, K, None, edges); bb.features_quarter = syn.make_features(n_img, 32, *cfg['feat_size'], seed=seed)
    gt = syn.ray_box_depth(rot[nb:nb + n], tv[nb:nb + n], K[nb:nb + n], cfg['img_size'], drv.DEPTH_CONFIG['size'])
    gt = gt + 0.02 * torch.randn(gt.shape, generator=torch.Generator().manual_seed(7))
    db = torch.zeros(n, dtype=torch.long)
    g = lambda t: t.to(dev)
    with torch.no_grad():
        sh, sc = gt.clone(), gt.clone()
        for it, offs in enumerate(drv.OFFSETS_LIST):
            ph = net.construct_feature_rich_pointcloud(g(sh), g(db), g(bb.features_quarter), g(rot), g(tv), g(K), g(edges))[0]
            pc = osc.feature_rich_pointcloud(sc, db, bb.features_quarter, rot, tv, K, edges, cfg['img_size'], pinned=True)[0]
            flips = int((cells(ph) != cells(pc)).any(dim=1).sum())
            sh = drv.process_scene(bb, net, (nb, na), dev, init_depth_override=sh.to(dev), offsets_list=[offs]).cpu()
            t0 = time.time()
            sc = drv.process_scene(bb, onet, (nb, na), torch.device('cpu'), init_depth_override=sc, offsets_list=[offs])
            rel = (sh - sc).abs() / sc
            print('seed %d outer iteration %d: %d of %d points in different cells; max rel depth error after it %.3e (#pixels > 1e-4: %d; oracle %.0f s)'
                  % (seed, it, flips, ph.shape[0], float(rel.max()), int((rel > 1e-4).sum()), time.time() - t0), flush=True)
