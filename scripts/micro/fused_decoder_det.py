import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
syn, lm = importlib.import_module('3dvnet_amd.synthetic'), importlib.import_module('3dvnet_amd.lightningmodel')
cuda = torch.device('cuda:0')
cfg = syn.CONFIGS['cfg3']
n_ref, k = 6, 2
edges, n_img = syn.make_edges(n_ref, k, k)
rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5)
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(cuda)
depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56))
depth = (depth + 0.02 * torch.randn(depth.shape, generator=torch.Generator().manual_seed(1))).to(cuda)
rot, tv, K, edges = rot.to(cuda), tv.to(cuda), K.to(cuda), edges.to(cuda)
db = torch.zeros(n_ref, dtype=torch.long, device=cuda)
net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False)
net = net.to(cuda)
with torch.no_grad():
    xs = net.model_scene(depth, db, feat, rot, tv, K, edges)
    pts_hyp, pts_feat = lm.backproject_variance(depth, feat, rot, tv, K, edges, cfg['img_size'], offset=0.05, n=3)
    pb = db.unsqueeze(1).expand(n_ref, 3136).reshape(-1)
    vals = torch.linspace(-0.15, 0.15, 7).to(cuda)
    p_u, e_u = net.decoder.decode(net.decoder.features(xs, pts_hyp, pts_feat, pb), vals)
    first = None; nbad = 0
    for i in range(20):
        p_f, e_f = net.decoder.decode_fused(xs, pts_hyp, pts_feat, pb, vals)
        torch.cuda.synchronize()
        if first is None: first = p_f.clone()
        d = (p_f != first).any(dim=1).nonzero().flatten()
        du = ((p_f - p_u).abs().max(dim=1).values > 1e-4).nonzero().flatten()
        if i < 3 or len(d): print(i, 'differs from first at', len(d), 'points', d[:8].tolist(), '| wrong vs unfused:', len(du), du[:12].tolist(), 'tiles', sorted(set((du // 32).tolist()))[:10])
