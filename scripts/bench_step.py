#!/usr/bin/env python3
"""Whole-step time of the cost-volume path (cfg2 by default) under sets of developer options, interleaved on one box.

    python scripts/bench_step.py --opts "tail_streams=1" "tail_streams=4,tail_from=3,tail_to=8" [--refs 64] [--iters 20] [--rounds 3]

Every option set is timed `rounds` times in alternation (clock management follows recent activity: DESIGN.md 4.1); the depth
maps of all sets are compared bit for bit with the first set's.
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--precision', default='split_bf16', choices=['split_bf16', 'fp32'])
    ap.add_argument('--opts', nargs='+', default=[''])
    args = ap.parse_args()
    libm = importlib.import_module('3dvnet_amd._lib')
    if os.environ.get('V3D_LIB_OVERRIDE'):
        libm.LIB_PATH = os.environ['V3D_LIB_OVERRIDE']
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    dev = torch.device('cuda:0')
    inp = syn.make_costvolume_inputs(args.config, n_ref=args.refs)
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(syn.costregnet_weights(seed=0, sharpen=200.0), strict=False)
    net.cnn_3d.precision = args.precision
    net = net.to(dev)
    b = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)
    d0, dd, D = inp['depth']

    def parse(o):
        return [(k.strip(), int(v)) for k, _, v in (it.partition('=') for it in o.split(',') if it.strip())]

    def run(o, iters):
        old = [(k, libm.set_option(k, v)) for k, v in parse(o)]
        try:
            with torch.no_grad():
                for _ in range(3):
                    out = net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=args.refs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    out = net.cost_volume_depth(feat, b, d0, dd, D, inp['plane_size'], n_ref=args.refs)
                e1.record()
                torch.cuda.synchronize()
            depth = out[0] if isinstance(out, (tuple, list)) else out
            return e0.elapsed_time(e1) / iters, depth.clone()
        finally:
            for k, v in old:
                libm.set_option(k, v)

    ref = None
    times = {o: [] for o in args.opts}
    for r in range(args.rounds):
        for o in args.opts:
            ms, depth = run(o, args.iters)
            times[o].append(ms)
            if ref is None:
                ref = depth
            elif not torch.equal(ref, depth):
                print('!! depth of [%s] differs from the first set: max abs %.3e' % (o, (ref - depth).abs().max().item()))
    for o in args.opts:
        t = times[o]
        print('%-50s  min %.3f  med %.3f ms/step  (%s)  %.0f maps/s' % ('[' + o + ']', min(t), sorted(t)[len(t) // 2],
              ' '.join('%.3f' % x for x in t), args.refs / min(t) * 1e3))


if __name__ == '__main__':
    main()
