import cProfile, pstats, importlib, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel')
drv = importlib.import_module('3dvnet_amd.eval_3dvnet'); Batch = importlib.import_module('3dvnet_amd.batch').Batch
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; k = 2
edges, n_img = syn.make_edges(64, k, k)
rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=1237, yaw_step_deg=360.0 / n_img)
b = Batch(None, rot, tv, K, None, edges); b.features_quarter = syn.make_features(n_img, 32, *cfg['feat_size'], seed=1237)
net = lm.PL3DVNet(None, drv.DEPTH_CONFIG, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.mvsnet.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False); net = net.to(dev)
drv.process_scene(b, net, k, dev); torch.cuda.synchronize()
# make every call synchronous so host time attributes to the right python frame
os.environ['AMD_SERIALIZE_KERNEL'] = '3'
pr = cProfile.Profile(); pr.enable()
drv.process_scene(b, net, k, dev); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats('tottime').print_callers('item'); st.print_callers("'to' of"); print(s.getvalue()[:6000])
