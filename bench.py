#!/usr/bin/env python3
"""Headline benchmark: depth maps / second on the plane-sweep cost-volume path
(BASELINE.json: 256x320 images, 96 depth planes, 1 reference + 7 source views), MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)

One "step" = one pass of rows A1-A6 (fused warp+variance -> CostRegNet -> soft-argmin) over one
batch of `--refs` synthetic reference views (sliding-window scene, features already resident in
HBM).  Reference views are independent units, so for N>1 every rank processes its own batch
(weak scaling, no data-path collective); value = refs processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0, including
  "roofline":     achieved vs peak for the dominant kernel (HIP-event timed inside this script)
  "cpu_baseline": the oracle (CPU restatement of the reference's PyTorch path) timed on host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3  # dense fp32 MFMA == fp32 vector peak

# (Cin, Cout, divisor of D*h*w giving the voxel count the 27*Cin*Cout MACs are spent on): output voxels
# for the convs, INPUT voxels for the stride-2 transposed convs (SURVEY.md §8a row A5 MAC table)
COSTREG_LAYERS = [
    (32, 8, 1), (8, 16, 8), (16, 16, 8), (16, 32, 64), (32, 32, 64), (32, 64, 512), (64, 64, 512),
    (64, 32, 512), (32, 16, 64), (16, 8, 8)]


def kernel_roofline(name, avg_ms, shape):
    """Algorithmic work of one launch of `name` / measured duration."""
    n_img, n_ref, C, Hf, Wf, D, h, w = shape
    vox = D * h * w
    if name == 'psv_variance':
        nbytes = 4.0 * (n_img * C * Hf * Wf + n_ref * C * vox)
        a = nbytes / (avg_ms * 1e-3) / 1e9
        return dict(bound='hbm', achieved=a, peak=PEAK_HBM_GBS, unit='GB/s', frac=a / PEAK_HBM_GBS)
    if name == 'costreg_conv9_prob':
        # fused deconv9 + skip + prob: reads u8 (16 ch at half resolution) and the conv0 skip (8 ch), writes 1 ch
        nbytes = 4.0 * n_ref * (16 * (vox // 8) + 8 * vox + vox)
    elif name.startswith('costreg_conv'):
        layer = int(name[len('costreg_conv'):])
        ci, co, div = COSTREG_LAYERS[layer]
        work_vox = vox // div                      # output voxels (conv) / input voxels (transposed conv)
        if layer == 0:
            # conv0 runs on bf16 MFMAs with each fp32 product split into 3 bf16 products (hi*hi + hi*lo + lo*hi):
            # the ceiling in algorithmic FLOPs is the dense bf16 peak / 3
            flops = 2.0 * 27 * ci * co * work_vox * n_ref
            a = flops / (avg_ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            return dict(bound='mfma', achieved=a, peak=peak, unit='TFLOP/s', frac=a / peak)
        # conv1..conv8: a few hundred MFMAs per workgroup -- bound by moving their activations (4 bytes per value in the
        # split layout as in fp32): input + output (+ the fp32 copy conv2 / conv4 keep for the skips, + the skip read)
        if layer in (7, 8):
            nbytes = 4.0 * n_ref * (ci * work_vox + 2 * co * 8 * work_vox)
        else:
            vox_in = work_vox * (8 if layer in (1, 3, 5) else 1)
            nbytes = 4.0 * n_ref * (ci * vox_in + co * work_vox * (2 if layer in (2, 4) else 1))
    elif name == 'costreg_prob':
        nbytes = 4.0 * n_ref * vox * (8 + 1)
    elif name == 'soft_argmin':
        nbytes = 4.0 * n_ref * (vox + h * w)
    elif name == 'transpose_channel_last':
        nbytes = 8.0 * n_img * C * Hf * Wf
    else:
        return None
    a = nbytes / (avg_ms * 1e-3) / 1e9
    return dict(bound='hbm', achieved=a, peak=PEAK_HBM_GBS, unit='GB/s', frac=a / PEAK_HBM_GBS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--refs', type=int, default=64, help='reference views per step per GPU')
    ap.add_argument('--cpu-refs', type=int, default=4, help='reference views in the CPU-baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback on the product path)'
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    libm = importlib.import_module('3dvnet_amd._lib')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch

    inp = syn.make_costvolume_inputs('cfg2', n_ref=args.refs, seed=1236 + rank)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(sd, strict=False)
    net = net.to(dev)
    batch = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)

    def step():
        with torch.no_grad():
            return net.cost_volume_depth(feat, batch, d0, dd, D, inp['plane_size'])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        depth = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        depth = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(depth).all()
    value = world * args.refs * args.steps / elapsed

    # ---- per-kernel HIP-event timing (separate pass so the events do not perturb `value`) --------
    roofline, kernels = None, {}
    if rank == 0:
        libm.timing_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        stats = libm.timing_collect()
        libm.timing_enable(False)
        C, Hf, Wf = feat.shape[1:]
        shape = (inp['n_img'], args.refs, C, Hf, Wf, D, inp['plane_size'][0], inp['plane_size'][1])
        total = sum(ms for ms, _ in stats.values())
        for name, (ms, cnt) in stats.items():
            avg = ms / max(cnt, 1)
            r = kernel_roofline(name, avg, shape) or {}
            kernels[name] = dict(avg_ms=round(avg, 4), share=round(ms / total, 3),
                                 **{k: (round(v, 4) if isinstance(v, float) else v)
                                    for k, v in r.items() if k in ('bound', 'achieved', 'frac', 'unit')})
        dom = max(stats, key=lambda k: stats[k][0])
        roofline = kernel_roofline(dom, stats[dom][0] / stats[dom][1], shape)
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/r01_traffic.json; FETCH_SIZE + WRITE_SIZE, KB -> bytes per launch), if they were taken
        # at the same batch size
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')))
            if tj.get('refs_per_step_per_gpu') == args.refs and dom in tj['kernels']:
                traffic = (tj['kernels'][dom]['fetch_kb'] + tj['kernels'][dom]['write_kb']) * 1024.0
        except (OSError, ValueError, KeyError):
            pass
        roofline.update(kernel=dom, avg_ms=stats[dom][0] / stats[dom][1], traffic=traffic)

    # ---- CPU baseline: the oracle on the host cores (rank 0, N=1 only) --------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import costvolume as ocv   # checker / reported baseline only
        per = inp['edges'].shape[1] // args.refs

        def cpu_run(n):
            with torch.no_grad():
                return ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                        inp['edges'][:, :n * per], sd, d0, dd, D, inp['img_size'],
                                        inp['plane_size'])[0]
        # PyTorch's CPU kernels do not scale to every hardware thread of a big host: probe a few
        # thread counts on one view each and report the fastest (its thread count is `cores`)
        ncpu = os.cpu_count() or 1
        best = None
        for nt in sorted({min(ncpu, c) for c in (16, 32, 64, ncpu)}):
            torch.set_num_threads(nt)
            cpu_run(1)
            t0 = time.perf_counter()
            cpu_run(1)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        torch.set_num_threads(best[1])
        cpu_run(1)
        t0 = time.perf_counter()
        d_cpu = cpu_run(args.cpu_refs)
        t_cpu = time.perf_counter() - t0
        rel = float(((depth[:args.cpu_refs].cpu() - d_cpu).abs() / d_cpu).max())
        cpu_baseline = dict(value=args.cpu_refs / t_cpu, unit='depth maps/s',
                            cores=torch.get_num_threads(), kind='port',
                            sample='%d reference views of the same cfg2 batch (oracle: torch CPU '
                                   'grid_sample + scatter-mean + Conv3d), 1 warm-up view' % args.cpu_refs,
                            max_rel_depth_err_gpu_vs_cpu=rel)

    if rank == 0:
        print(json.dumps({
            'metric': 'depth maps/sec (256x320, 96 planes, 7 src)',
            'value': value, 'unit': 'depth maps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'cfg2: ScanNet-shape 256x320, 1 ref + 7 src (8 edges/ref), '
                                   '96 planes, 56x56 plane grid, 32-ch quarter features 64x80; fused '
                                   'warp+variance -> CostRegNet -> soft-argmin depth (rows A1-A6)',
                       'refs_per_step_per_gpu': args.refs, 'n_img_per_gpu': inp['n_img'],
                       'parallelism': 'ref-view sharding, no collective' if world > 1 else 'single GPU'},
            'roofline': roofline, 'cpu_baseline': cpu_baseline, 'kernels': kernels}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
