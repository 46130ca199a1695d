#!/usr/bin/env python3
"""Developer fuzz: the fused backbone kernels (precision 'split_bf16') against the exact-fp32 per-layer kernels on random image
sizes (sides multiples of 8) and batch sizes; prints the worst difference per size (of the range of each map)."""
import importlib, os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bb = importlib.import_module('3dvnet_amd.backbone'); syn = importlib.import_module('3dvnet_amd.synthetic')
dev = torch.device('cuda:0')
fe, fs = bb.build_backbone(32)
sd_e, sd_s = syn.backbone_weights(32, seed=6); fe.load_state_dict(sd_e, strict=False); fs.load_state_dict(sd_s)
fe, fs = fe.eval().to(dev), fs.eval().to(dev)
a, b = bb.NativeBackbone(fe, fs, precision='split_bf16'), bb.NativeBackbone(fe, fs, precision='fp32')
rng = random.Random(int(os.environ.get('SEED', 1)))
sizes = [(32, 32), (8, 8), (16, 200), (200, 16), (40, 72), (264, 328)] + [(8 * rng.randint(1, 40), 8 * rng.randint(1, 48)) for _ in range(int(os.environ.get('N', 24)))]
worst = 0.0
for H, W in sizes:
    n = rng.randint(1, 3)
    img = syn.make_images(n, (max(H, 16), max(W, 16)), seed=H * 1000 + W)[:, :, :H, :W].contiguous().to(dev)
    with torch.no_grad():
        ya, yb = a(img), b(img)
    torch.cuda.synchronize()
    errs = [float((p - q).abs().max() / q.abs().max()) for p, q in zip(ya, yb)]
    ok = all(torch.isfinite(p).all() for p in ya)
    worst = max(worst, max(errs))
    print('%3d x %3d  n=%d  %s  %s' % (H, W, n, ' '.join('%.1e' % e for e in errs), '' if ok and max(errs) < 1e-4 else '  <-- CHECK'), flush=True)
print('worst %.2e' % worst)
