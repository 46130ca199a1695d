#!/usr/bin/env python3
"""Developer timing: one point-flow sweep of the 64-view cfg3 scene (4 chunks of 16 views, 7 hypotheses), unfused chain against
the fused decoder, per-kernel HIP-event times per sweep (V3D_FUSED_LDS_KB selects the fused kernel's LDS request)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
syn = importlib.import_module('3dvnet_amd.synthetic'); lm = importlib.import_module('3dvnet_amd.lightningmodel'); libm = importlib.import_module('3dvnet_amd._lib')
if os.environ.get('V3D_LIB_OVERRIDE'): libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']      # an ablated build (fused_decoder_ablate.sh)
dev = torch.device('cuda:0'); cfg = syn.CONFIGS['cfg3']; n_ref, k = 64, 2
edges, n_img = syn.make_edges(n_ref, k, k); rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=5, yaw_step_deg=360.0 / n_img)
feat = syn.make_features(n_img, 32, *cfg['feat_size'], seed=5).to(dev)
depth = syn.ray_box_depth(rot[k:k + n_ref], tv[k:k + n_ref], K[k:k + n_ref], cfg['img_size'], (56, 56)).to(dev)
rot, tv, K, edges = rot.to(dev), tv.to(dev), K.to(dev), edges.to(dev); db = torch.zeros(n_ref, dtype=torch.long, device=dev)
net = lm.PL3DVNet(None, {'size': (56, 56)}, 0.04, feat_dim=32, img_size=cfg['img_size']).eval()
net.pointnet.load_state_dict(syn.pointnet_weights()); net.sparse_conv.load_state_dict(syn.sparse_unet_weights())
net.decoder.load_state_dict(syn.decoder_weights(sharpen=50.0), strict=False)
net = net.to(dev)
with torch.no_grad():
    xs = net.model_scene(depth, db, feat, rot, tv, K, edges)
    outs = {}
    for fused in ((True,) if os.environ.get('V3D_TIME_FUSED_ONLY') else (False, True)):
        net.decoder.fused = fused
        CH = int(os.environ.get('V3D_TIME_CHUNK', '16'))      # reference views per run_pointflow call (eval-3dvnet.py: 16)
        def sweep():
            o = []
            for b0 in range(0, 64, CH):
                e = edges[:, (edges[0] >= b0 + k) & (edges[0] < b0 + CH + k)] - b0
                o.append(net.run_pointflow(xs, depth[b0:b0 + CH], db[b0:b0 + CH], feat[b0:b0 + CH + 2 * k], rot[b0:b0 + CH + 2 * k], tv[b0:b0 + CH + 2 * k], K[b0:b0 + CH + 2 * k], e, 0.05, 3))
            return torch.cat(o)
        for _ in range(2): outs[fused] = sweep()
        torch.cuda.synchronize(); libm.timing_enable(True)
        for _ in range(3): sweep()
        torch.cuda.synchronize(); st = libm.timing_collect(); libm.timing_enable(False)
        tot = sum(ms for ms, c in st.values()) / 3
        print('LDS %s KB fused=%s: %.3f ms per 64-view sweep:' % (os.environ.get('V3D_FUSED_LDS_KB', 'default'), fused, tot),
              {k_: round(ms / 3, 3) for k_, (ms, c) in st.items() if ms / 3 > 0.05})
    if os.environ.get('V3D_FUSED_PHASES'):      # library built with -DV3D_PHASE_TIMING (fused_decoder_ablate.sh build "-DV3D_PHASE_TIMING")
        import ctypes
        fn = libm.load().v3d_debug_fused_phase_read; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        buf = (ctypes.c_ulonglong * 8)(); nb = 256; fn(buf, nb); tot = sum(buf)
        print('phases (cycles per workgroup of the last launch, wave 0): total %.0f:' % (tot / nb), ' '.join('%d:%.0f (%.1f%%)' % (i, v / nb, 100.0 * v / tot) for i, v in enumerate(buf)))
    if False in outs: print('max |offset fused - chain| = %.2e m' % float((outs[True] - outs[False]).abs().max()))
