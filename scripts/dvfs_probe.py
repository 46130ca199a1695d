#!/usr/bin/env python3
"""Developer probe: does the time of conv0 depend on what ran just before it (clock / power management) rather than on its
own work?  Runs warp -> [gap] -> regulariser with (a) no gap, (b) a low-power spin of ~1 / ~3 ms on one thread
(torch.cuda._sleep), (c) the warp kernel run twice; prints the library's per-kernel HIP-event times."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module('3dvnet_amd.synthetic'); mvs = importlib.import_module('3dvnet_amd.mvsnet')
libm = importlib.import_module('3dvnet_amd._lib'); Batch = importlib.import_module('3dvnet_amd.batch').Batch
dev = torch.device('cuda:0')
inp = syn.make_costvolume_inputs('cfg2', n_ref=64)
net = mvs.MVSNet(32, inp['img_size']).eval()
net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
net = net.to(dev)
b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
feat = inp['feat'].to(dev); d0, dd, D = inp['depth']
vals = net.depth_values(d0, dd, D, dev)
csr = mvs.edges_to_csr(b.ref_src_edges)

def step(gap_cycles=0, twice=False):
    var = mvs.plane_sweep_variance(feat, b.rotmats, b.tvecs, b.K, b.ref_src_edges, d0, dd, D, net.img_size, inp['plane_size'],
                                   workspace=net._ws, csr=csr, split=True)
    if twice:
        var = mvs.plane_sweep_variance(feat, b.rotmats, b.tvecs, b.K, b.ref_src_edges, d0, dd, D, net.img_size,
                                       inp['plane_size'], workspace=net._ws, csr=csr, split=True)
    if gap_cycles:
        torch.cuda._sleep(int(gap_cycles))
    return net.cnn_3d.regularize_depth(var, vals)

with torch.no_grad():
    for name, kw in (('no gap', {}), ('gap ~1 ms', dict(gap_cycles=2_000_000)), ('gap ~3 ms', dict(gap_cycles=6_000_000)),
                     ('warp twice', dict(twice=True)), ('no gap', {})):
        for _ in range(3):
            step(**kw)
        torch.cuda.synchronize()
        libm.timing_enable(True)
        for _ in range(10):
            step(**kw)
        torch.cuda.synchronize()
        st = libm.timing_collect(); libm.timing_enable(False)
        print('%-12s' % name, ' '.join('%s=%.3f' % (k.replace('costreg_', ''), ms / c) for k, (ms, c) in st.items()
                                       if k in ('psv_variance', 'costreg_conv0', 'costreg_conv1', 'costreg_conv2', 'costreg_conv9_prob')))
