#!/usr/bin/env python3
"""Developer tool: measured errors of the HIP path against the reference-generated goldens (tests/golden/A_*.npz) and the
oracle, for both operand precisions -- the numbers quoted in DESIGN.md §2.

    python scripts/parity_report.py [cfg1 cfg2 cfg5]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
syn = importlib.import_module('3dvnet_amd.synthetic')
mvs = importlib.import_module('3dvnet_amd.mvsnet')
Batch = importlib.import_module('3dvnet_amd.batch').Batch
from helpers import golden_costreg_weights, load_golden  # noqa: E402

SUB = {'cfg1': ((slice(None), slice(None, None, 4), slice(None, None, 3), slice(None, None, 5), slice(None, None, 7)),
                (slice(None), slice(None, None, 3), slice(None, None, 5), slice(None, None, 7)), None),
       'cfg2': ((slice(None), slice(None, None, 4), slice(None, None, 5), slice(None, None, 7), slice(None, None, 7)),
                (slice(None), slice(None, None, 5), slice(None, None, 7), slice(None, None, 7)), None),
       'cfg5': ((slice(None), slice(None, None, 4), slice(None, None, 7), slice(None, None, 11), slice(None, None, 13)),
                (slice(None), slice(None, None, 7), slice(None, None, 11), slice(None, None, 13)),
                (slice(None), slice(None, None, 3), slice(None, None, 3)))}


def main():
    dev = torch.device('cuda:0')
    for cfg in (sys.argv[1:] or ['cfg1', 'cfg2']):
        g = load_golden('A_' + cfg)
        inp = syn.make_costvolume_inputs(cfg, n_ref=1)
        sd = golden_costreg_weights(g)
        net = mvs.MVSNet(32, inp['img_size']).eval()
        net.cnn_3d.load_state_dict(sd, strict=False)
        net = net.to(dev)
        b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
        d0, dd, D = inp['depth']
        vs, rs, ds = SUB[cfg]
        for pr in ('split_bf16', 'fp32'):
            with torch.no_grad():
                depth, var, reg = net.cost_volume_depth(inp['feat'].to(dev), b, d0, dd, D, inp['plane_size'],
                                                        return_intermediates=True, precision=pr)
            var, reg, depth = var.cpu().numpy(), reg.cpu().numpy(), depth.cpu().numpy()
            gd = g['depth'] if ds is None else g['depth_sub']
            dd_ = depth if ds is None else depth[ds]
            print('%s %-10s var max abs err %.3e (bit-equal %.4f)  reg max err / max %.3e  depth max rel err %.3e'
                  % (cfg, pr, np.abs(var[vs] - g['var_sub']).max(), np.mean(var[vs] == g['var_sub']),
                     np.abs(reg[rs] - g['reg_sub']).max() / np.abs(g['reg_sub']).max(),
                     (np.abs(dd_ - gd) / gd).max()))


if __name__ == '__main__':
    main()
