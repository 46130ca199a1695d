#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own Python
(/root/reference, imported in place with the stubs of _ref_import.py) on small seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
Outputs tests/golden/*.npz (committed).  The reference's source never leaves /root/reference;
fixtures hold data only (inputs, expected outputs, seeds, checksums).
"""
import importlib
import math
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import  # noqa: E402

syn = importlib.import_module('3dvnet_amd.synthetic')
ref = _ref_import.reference()


def checksum(sd):
    return float(sum(v.double().abs().sum().item() * (i + 1) for i, (k, v) in enumerate(sorted(sd.items()))))


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def run_reference_mvsnet(feat, rotmats, tvecs, K, edges, sd, depth, img_size, plane_size):
    """MVSNet.forward called unbound with a duck-typed self (mvsnet.py:176-229); captures the
    variance volume (row A4) and the regularised volume (row A5)."""
    net = ref.mvsnet.CostRegNet(feat.shape[1], 8).eval()
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    cap = {}

    def cnn(x):
        cap['var'] = x.clone()
        y = net(x)
        cap['reg'] = y.clone()
        return y

    fake = NS(feat_dim=feat.shape[1], img_size=img_size, cnn_3d=cnn,
              feat_extractor=lambda im: (im,),
              feat_shrinker=lambda *a: (None, feat, None, None, None))
    batch = NS(images=torch.zeros(feat.shape[0], 3, *img_size), rotmats=rotmats, tvecs=tvecs, K=K,
               ref_src_edges=edges)
    with torch.no_grad():
        d, _, _, _ = ref.mvsnet.MVSNet.forward(fake, batch, depth[0], depth[1], depth[2], plane_size)
    return d, cap['var'], cap['reg'].squeeze(1)


def golden_costvolume():
    # --- tiny: full tensors -------------------------------------------------------------------
    img_size, feat_size, plane_size, depth = (64, 80), (16, 20), (8, 8), (0.5, 0.25, 8)
    edges, n_img = syn.make_edges(2, 1, 1)
    rot, tv, K = syn.make_cameras(n_img, img_size, seed=11)
    feat = syn.make_features(n_img, 32, *feat_size, seed=11)
    for tag, sharpen in (('flat', 1.0), ('sharp', 200.0)):
        sd = syn.costregnet_weights(seed=0, sharpen=sharpen)
        d, var, reg = run_reference_mvsnet(feat, rot, tv, K, edges, sd, depth, img_size, plane_size)
        save('A_tiny_' + tag, feat=feat, rotmats=rot, tvecs=tv, K=K, edges=edges,
             img_size=img_size, plane_size=plane_size, depth_cfg=depth, weights_seed=0,
             sharpen=sharpen, weights_checksum=checksum(sd), var=var, reg=reg, depth=d)

    # --- behind-camera / strongly rotated sources: |z| mirroring and zero padding ---------------
    edges = torch.tensor([[1, 1, 1, 1], [0, 1, 2, 3]], dtype=torch.long)
    rot, tv, K = syn.make_cameras(4, img_size, seed=5, yaw_step_deg=70.0)
    feat = syn.make_features(4, 32, *feat_size, seed=5)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d, var, reg = run_reference_mvsnet(feat, rot, tv, K, edges, sd, depth, img_size, plane_size)
    save('A_tiny_rotated', feat=feat, rotmats=rot, tvecs=tv, K=K, edges=edges, img_size=img_size,
         plane_size=plane_size, depth_cfg=depth, weights_seed=0, sharpen=200.0,
         weights_checksum=checksum(sd), var=var, reg=reg, depth=d)

    # --- BASELINE config 1 shape (the reference's CPU-runnable case): sub-sampled var ------------
    inp = syn.make_costvolume_inputs('cfg1', n_ref=1)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d, var, reg = run_reference_mvsnet(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                       inp['edges'], sd, inp['depth'], inp['img_size'],
                                       inp['plane_size'])
    save('A_cfg1', n_ref=1, weights_seed=0, sharpen=200.0, weights_checksum=checksum(sd),
         feat_checksum=float(inp['feat'].double().sum()),
         var_sub=var[:, ::4, ::3, ::5, ::7], var_sum=float(var.double().sum()),
         reg_sub=reg[:, ::3, ::5, ::7], depth=d)

    # --- BASELINE config 2 shape (ScanNet 256x320, 96 planes, 1 ref + 7 src) --------------------
    inp = syn.make_costvolume_inputs('cfg2', n_ref=1)
    d, var, reg = run_reference_mvsnet(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                       inp['edges'], sd, inp['depth'], inp['img_size'],
                                       inp['plane_size'])
    save('A_cfg2', n_ref=1, weights_seed=0, sharpen=200.0, weights_checksum=checksum(sd),
         feat_checksum=float(inp['feat'].double().sum()),
         var_sub=var[:, ::4, ::5, ::7, ::7], var_sum=float(var.double().sum()),
         reg_sub=reg[:, ::5, ::7, ::7], depth=d)


if __name__ == '__main__':
    which = sys.argv[1:] or ['A']
    if 'A' in which:
        golden_costvolume()
