#!/usr/bin/env python3
"""Developer: hash-probe statistics of decoder_corner_kernel on the cfg3 scene (library built with -DV3D_CORNER_STATS:
python scripts/build_variant.py decoder.hip cstats -DV3D_CORNER_STATS; V3D_LIB_OVERRIDE=3dvnet_amd/build/ablate/lib_cstats.so)."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel'); libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; n_ref, k = 64, 2
edges, n_img = syn.make_edges(n_ref, k, k); rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5, yaw_step_deg=360.0 / n_img)
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(dev)
depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56)).to(dev)
rot, tv, K, edges = rot.to(dev), tv.to(dev), K.to(dev), edges.to(dev); db = torch.zeros(n_ref, dtype=torch.long, device=dev)
net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False); net = net.to(dev)
fn = libm.load().v3d_debug_corner_stats; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 12)()
with torch.no_grad():
    xs = net.model_scene(depth, db, feat, rot, tv, K, edges)
    torch.cuda.synchronize(); fn(buf, 1)
    net.run_pointflow(xs, depth, db, feat, rot, tv, K, edges, 0.05, 1)
    torch.cuda.synchronize(); fn(buf, 1)
for l in range(3):
    n, pr, mx, hit = buf[l * 4:l * 4 + 4]
    print('level %d: rows %d, %d lookups, %.2f probes each, longest chain %d, %.1f%% present' % (l, xs[l]['feats'].shape[0] if isinstance(xs, (list, tuple)) else -1, n, pr / max(n, 1), mx, 100.0 * hit / max(n, 1)))
