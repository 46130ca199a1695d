#!/usr/bin/env python3
"""Developer tool: which Python lines of the scene path launch stock torch kernels (copies, elementwise, index, cat)?
One cfg3 scene under torch.profiler with stacks; device-side aten ops grouped by the innermost frame inside 3dvnet_amd/.
    python scripts/trace_torch_ops.py [--stage3]"""
import collections, importlib, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
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
b = b.to(dev) if hasattr(b, 'to') else b
for _ in range(2):
    drv.process_scene(b, net, k, dev)
torch.cuda.synchronize()
# ---- host-side attribution: which lines of 3dvnet_amd/ make device copies (a result with new storage) ------------------
import inspect
copies = collections.Counter()
def _caller():
    for fr in inspect.stack()[2:12]:
        if '3dvnet_amd/' in fr.filename:
            return '%s:%d' % (fr.filename.split('3dvnet_amd/')[-1], fr.lineno)
    return '?'
def _wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **kw):
        r = orig(self, *a, **kw)
        if isinstance(self, torch.Tensor) and self.is_cuda or (isinstance(r, torch.Tensor) and r.is_cuda):
            if name in ('__setitem__', '__iadd__', 'copy_') or (isinstance(r, torch.Tensor) and r.numel() and r.data_ptr() != self.data_ptr()
                                                                  and name not in ('__getitem__', 'view', 'expand', 'unsqueeze')):
                copies[(name, _caller())] += 1
        return r
    setattr(torch.Tensor, name, f)
for nm in ('contiguous', 'reshape', 'to', 'clone', 'float', 'long', 'int', '__setitem__', '__iadd__', 'copy_', 'amin', 'max', 'cpu', 'item'):
    _wrap(nm)
_cat, _zeros, _stack = torch.cat, torch.zeros, torch.stack
def cat(*a, **kw):
    copies[('cat', _caller())] += 1; return _cat(*a, **kw)
def stack(*a, **kw):
    copies[('stack', _caller())] += 1; return _stack(*a, **kw)
torch.cat, torch.stack = cat, stack
drv.process_scene(b, net, k, dev); torch.cuda.synchronize()
print('host-side copy sites of one scene:')
for key, n in sorted(copies.items(), key=lambda kv: -kv[1])[:40]:
    print('%5d  %-12s %s' % (n, key[0], key[1]))
if not os.environ.get("V3D_TRACE_PROFILER"): sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    drv.process_scene(b, net, k, dev); torch.cuda.synchronize()
by = collections.Counter(); dur = collections.Counter()
def self_dev(ev):
    for a in ('self_device_time_total', 'self_cuda_time_total'):
        v = getattr(ev, a, None)
        if v:
            return v
    return 0
for ev in prof.events():
    t = self_dev(ev)
    if not ev.name.startswith('aten::') or t <= 0:
        continue
    stack = ev.stack or []
    frame = next((f for f in stack if '3dvnet_amd/' in f), stack[0] if stack else '?')
    key = (ev.name, frame.split('3dvnet_amd/')[-1][:90])
    by[key] += 1; dur[key] += t
print('%6s %9s  op @ frame' % ('calls', 'dev us'))
for key, n in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:45]:
    print('%6d %9.0f  %s @ %s' % (n, dur[key], key[0], key[1]))
print('total aten device us per scene: %.0f in %d launches' % (sum(dur.values()), sum(by.values())))
