#!/usr/bin/env python3
"""Headline benchmark: depth maps / second on the plane-sweep cost-volume path
(BASELINE.json: 256x320 images, 96 depth planes, 1 reference + 7 source views), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg5|cfg3|cfg4]

`--gpus N` with N > 1 spawns its own N ranks (one process per GPU, torch.distributed / RCCL) when it is not already
running under torch.distributed.run, so both `python bench.py --gpus 8` and
`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8` work.

Configurations (BASELINE.json `configs`):
  cfg2 (default, the configuration the metric is quoted on): one "step" = one pass of rows A1-A6 (fused warp+variance
        -> CostRegNet -> soft-argmin) over one batch of `--refs` synthetic reference views, features resident in HBM.
        Reference views are independent units: for N > 1 every rank processes its own batch (weak scaling, no data-path
        collective); value = views processed by all ranks / max-over-ranks time.
  cfg5  the same path at 480x640 / 192 planes / 10 source views / 120x160 plane grid (the HBM-heavy point).
  cfg3  full pipeline on one 64-view scene at 4 cm voxels (stage A + scene model + 2x3 point-flow sweeps); one step =
        one scene.  cfg4 = cfg3 with the reference views sharded over the ranks and the feature-rich point cloud
        all-gathered over RCCL once per outer iteration (strong scaling of one scene).

The default invocation (cfg2, one GPU) also runs cfg5 (8 views) and the cfg3 scene once each, shorter, and appends their
figures as compact objects under "extra" (value, ms_per_step, dominant-kernel roofline, depth error against the oracle), so
that every configuration's number is in the driver-run line; `--no-extra` skips them.

Prints ONE JSON line on rank 0, including
  "value_fp32_exact": the same batch with exact-fp32 MFMA operands (precision='fp32'); `value` uses split-bf16 operands
  "roofline":     achieved vs peak for the dominant kernel (HIP events recorded by the library on the launch stream)
  "cpu_baseline": the oracle (CPU restatement of the reference's PyTorch path) timed on the host cores.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3     # dense fp32 MFMA == fp32 vector peak
DTYPE = 'f32 storage/accumulate; MFMA operands split-bf16x3 (hi*hi+hi*lo+lo*hi, 16 mantissa bits); warp/variance f32 VALU'
PROFILE_ROUND = 'r06'

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
    if name in ('costreg_conv9_prob', 'costreg_conv9_prob_f32'):
        # fused deconv9 + skip + prob: reads u8 (16 ch at half resolution) and the conv0 skip (8 ch), writes 1 ch
        nbytes = 4.0 * n_ref * (16 * (vox // 8) + 8 * vox + vox)
    elif name == 'costreg_conv12':
        # conv1 + conv2 fused (conv12z.hip): reads conv0's 8 channels at full resolution, writes conv2's 16 at half resolution
        nbytes = 4.0 * n_ref * (8 * vox + 16 * (vox // 8))
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
        # conv1..conv8: a few hundred MFMAs per workgroup -- priced against moving their activations (4 bytes per value in
        # the split layout as in fp32): input + output (+ the skip read of the transposed convolutions)
        if layer in (7, 8):
            nbytes = 4.0 * n_ref * (ci * work_vox + 2 * co * 8 * work_vox)
        else:
            vox_in = work_vox * (8 if layer in (1, 3, 5) else 1)
            nbytes = 4.0 * n_ref * (ci * vox_in + co * work_vox)
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


def cpu_info():
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return model


def _median_time(fn, warmups, runs):
    for _ in range(warmups):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def _oracle_views_job(job):
    """One worker process of `_oracle_views`: the cost-volume oracle on views [v0, v1) of the seeded synthetic batch."""
    cfg, refs, seed, v0, v1, chunk, pinned, nt = job
    torch.set_num_threads(nt)
    from oracle import costvolume as ocv   # checker only
    syn = importlib.import_module('3dvnet_amd.synthetic')
    inp = syn.make_costvolume_inputs(cfg, n_ref=refs, seed=seed)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d0, dd, D = inp['depth']
    per = inp['edges'].shape[1] // refs
    out = []
    with torch.no_grad():
        for v in range(v0, v1, chunk):
            out.append(ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                        inp['edges'][:, v * per:min(v + chunk, v1) * per], sd, d0, dd, D, inp['img_size'],
                                        inp['plane_size'], pinned=pinned)[0])
    return torch.cat(out).numpy()


def _oracle_views(cfg, refs, seed, legs, chunk, nt):
    """Checker legs only (never a timed figure): the oracle's depths of the first n views of the seeded batch for every
    (n, pinned) in `legs`, computed by a few worker processes of `nt` threads each -- torch's CPU kernels stop scaling at
    16-32 threads, the bench host has 256, and the pinned evaluation orders are Python loops.  The workers are spawned (no
    fork of a process that holds a HIP context), rebuild the inputs from the seed and return depth maps only.  Same bits as the
    in-process loop at `nt` threads (tests/test_driver.py)."""
    import multiprocessing
    workers = max(1, min(6, (os.cpu_count() or 1) // (2 * nt)))
    jobs, owner = [], []
    for li, (n, pinned) in enumerate(legs):
        parts = max(1, min(workers, (n + chunk - 1) // chunk))
        per_part = -(-((n + chunk - 1) // chunk) // parts) * chunk
        for v0 in range(0, n, per_part):
            jobs.append((cfg, refs, seed, v0, min(v0 + per_part, n), chunk, pinned, nt))
            owner.append(li)
    res = None
    if workers > 1:
        try:
            with multiprocessing.get_context('spawn').Pool(min(workers, len(jobs))) as pool:
                res = pool.map(_oracle_views_job, jobs, chunksize=1)
        except Exception as e:      # a host that cannot spawn workers: the same jobs in this process (slower, same numbers)
            sys.stderr.write('oracle worker pool failed (%s: %s): running the checker in-process\n' % (type(e).__name__, e))
    if res is None:
        res = [_oracle_views_job(j) for j in jobs]
    return [torch.from_numpy(np.concatenate([r for r, o in zip(res, owner) if o == li])) for li in range(len(legs))]


def traffic_for(kernel, refs, cfg_name):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>_traffic_<cfg>.json, made by profiles/make_traffic.py: WRITE_SIZE + FETCH_SIZE with the gfx950
    x2 correction for wide streaming reads), if they were taken at the same batch size."""
    try:
        tj = json.load(open(os.path.join(ROOT, 'profiles', '%s_traffic_%s.json' % (PROFILE_ROUND, cfg_name))))
        if tj.get('refs_per_step_per_gpu') == refs and kernel in tj['kernels']:
            return float(tj['kernels'][kernel]['hbm_bytes'])
    except (OSError, ValueError, KeyError):
        pass
    return None


# ------------------------------------------------------------------------------------------------------------------
# cfg2 / cfg5: the cost-volume path (rows A1-A6)
# ------------------------------------------------------------------------------------------------------------------
def ranks_seen(args, world, dev, dist):
    """The number of ranks that take part, counted by an all-reduce over the process group (RCCL): must equal --gpus."""
    if dist is None:
        assert world == 1 and args.gpus == 1
        return 1
    one = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(one)
    seen = int(one.item())
    assert seen == args.gpus == dist.get_world_size(), 'ranks seen by the all-reduce: %d, --gpus %d' % (seen, args.gpus)
    return seen


def bench_costvolume(args, rank, world, dev, dist):
    syn = importlib.import_module('3dvnet_amd.synthetic')
    mvs = importlib.import_module('3dvnet_amd.mvsnet')
    libm = importlib.import_module('3dvnet_amd._lib')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    cfg = args.config
    refs = args.refs or {'cfg2': 64, 'cfg5': 8}[cfg]

    seed = 1234 + int(cfg[-1]) + rank
    inp = syn.make_costvolume_inputs(cfg, n_ref=refs, seed=seed)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d0, dd, D = inp['depth']
    net = mvs.MVSNet(32, inp['img_size']).eval()
    net.cnn_3d.load_state_dict(sd, strict=False)
    net = net.to(dev)
    batch = Batch(None, inp['rotmats'], inp['tvecs'], inp['K'], None, inp['edges']).to(dev)
    feat = inp['feat'].to(dev)

    # The edge list -> per-reference tables (mvsnet.py:179: torch.unique + the scatter grouping) are rebuilt every step, as
    # in the reference's forward -- by the library's device kernel (v3d_edges_csr), which needs the number of reference
    # images of the batch (a property of the batch, `refs` here) instead of reading torch.unique's length back to the host.
    # Checked once against the torch construction before anything is timed.
    csr_dev = mvs.edges_to_csr(batch.ref_src_edges, n_ref=refs, n_img=feat.shape[0]).check()
    csr_ref = mvs.edges_to_csr(batch.ref_src_edges)
    assert all(torch.equal(a, b) for a, b in zip(csr_dev, csr_ref)), 'device CSR differs from torch.unique + stable sort'

    def step(precision=None):
        with torch.no_grad():
            return net.cost_volume_depth(feat, batch, d0, dd, D, inp['plane_size'], precision=precision, n_ref=refs)

    # `--graph`: the timed step replays a HIP graph of exactly these launches (mvsnet.CostVolumeGraph): same kernels, same
    # work, one graph launch per step instead of ~17 kernel launches.  Measured equal to eager launches (4.16 vs 4.13 ms per
    # step: the launch queue is never empty), so eager is the default.
    graphs = {}
    launch_mode = 'eager'
    if args.graph:
        try:
            for prec in (None, 'fp32'):
                graphs[prec] = mvs.CostVolumeGraph(net, feat, batch, d0, dd, D, inp['plane_size'], n_ref=refs, precision=prec)
            with torch.no_grad():
                assert torch.equal(graphs[None].replay(), step(None)), 'graph replay differs from the eager step'
            launch_mode = 'hip graph (one launch per step)'
        except Exception as e:      # capture unsupported on this stack: time the eager launches
            sys.stderr.write('graph capture failed (%s): timing eager launches\n' % e)
            graphs = {}

    def run(precision):
        return graphs[precision].replay() if precision in graphs else step(precision)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(precision, steps, warmup):
        for _ in range(warmup):
            out = run(precision)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run(precision)
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, out

    elapsed, depth = timed(None, args.steps, args.warmup)
    assert torch.isfinite(depth).all()
    value = world * refs * args.steps / elapsed
    # the same batch with exact-fp32 MFMA operands (the reference's arithmetic type), same step count
    elapsed32, depth32 = timed('fp32', args.steps, min(args.warmup, 2))
    value32 = world * refs * args.steps / elapsed32

    # ---- per-kernel HIP-event timing (separate pass so the events do not perturb `value`) --------
    roofline, kernels = None, {}
    if rank == 0:
        libm.timing_enable(True)
        for _ in range(min(args.steps, 20)):
            step()
        torch.cuda.synchronize()
        stats = libm.timing_collect()
        libm.timing_enable(False)
        C, Hf, Wf = feat.shape[1:]
        shape = (inp['n_img'], refs, C, Hf, Wf, D, inp['plane_size'][0], inp['plane_size'][1])
        total = sum(ms for ms, _ in stats.values())
        for name, (ms, cnt) in stats.items():
            avg = ms / max(cnt, 1)
            r = kernel_roofline(name, avg, shape) or {}
            kernels[name] = dict(avg_ms=round(avg, 4), share=round(ms / total, 3),
                                 **{k: (round(v, 4) if isinstance(v, float) else v)
                                    for k, v in r.items() if k in ('bound', 'achieved', 'frac', 'unit')})
        dom = max(stats, key=lambda k: stats[k][0])
        roofline = kernel_roofline(dom, stats[dom][0] / stats[dom][1], shape)
        roofline.update(kernel=dom, avg_ms=stats[dom][0] / stats[dom][1], traffic=traffic_for(dom, refs, cfg))

    n_seen = ranks_seen(args, world, dev, dist)
    # ---- CPU baseline: the oracle on the host cores (rank 0, N=1 only) --------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import costvolume as ocv   # checker / reported baseline only
        per = inp['edges'].shape[1] // refs
        ncpu = os.cpu_count() or 1

        def cpu_run(v0, v1, pinned=False):
            with torch.no_grad():
                return ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'],
                                        inp['edges'][:, v0 * per:v1 * per], sd, d0, dd, D, inp['img_size'],
                                        inp['plane_size'], pinned=pinned)[0]
        n_s = max(1, min(args.cpu_refs, refs))
        timing = getattr(args, 'cpu_timing', True)      # the "extra" legs only use the oracle as the checker
        # SURVEY §8d protocol: n = os.cpu_count() threads, 2 warm-ups, median of 5, plus a 1-thread figure.  PyTorch's
        # CPU kernels do not scale to every hardware thread of a big host (256 threads ran 6x SLOWER than one thread
        # on the round-2 box), so a few intermediate counts are probed too (1 view, 1 warm-up, median of 3) and the
        # protocol is repeated at the fastest one: `value` is the best figure (the most favourable to the CPU), `cores`
        # its thread count, `by_threads` lists everything that was measured.
        by_threads, probe = {}, {}
        if timing:
            torch.set_num_threads(ncpu)
            # (a run that takes more than 3 s -- every hardware thread of a 256-thread host, 13 s per view -- : 1 warm-up, 1 run)
            t0 = time.perf_counter()
            cpu_run(0, n_s)
            slow = time.perf_counter() - t0 > 3.0
            by_threads[ncpu] = n_s / (_median_time(lambda: cpu_run(0, n_s), 0, 1) if slow
                                      else _median_time(lambda: cpu_run(0, n_s), 1, 5))
            for nt in sorted({1, min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
                torch.set_num_threads(nt)
                probe[nt] = 1.0 / _median_time(lambda: cpu_run(0, 1), 1, 3)
            best_nt = max(probe, key=probe.get)
            by_threads[1] = probe[1]
            if best_nt not in (1, ncpu):
                torch.set_num_threads(best_nt)
                by_threads[best_nt] = n_s / _median_time(lambda: cpu_run(0, n_s), 2, 5)
            cores = max(by_threads, key=by_threads.get)
        else:
            best_nt = cores = min(32, ncpu)
        # accuracy of the timed GPU batch against the oracle: every view of the step (or --check-refs of them).  The
        # checker is the oracle with the pinned evaluation orders of the reference run behind the goldens
        # (oracle/pinned.py): torch.bmm's last bits depend on the host BLAS, and one ulp of a sample coordinate is worth
        # ~5e-5 of relative depth under the sharpened soft-argmin.  The plain torch oracle of THIS host is reported
        # beside it on a few views.
        torch.set_num_threads(best_nt)
        n_chk = refs if args.check_refs < 0 else min(args.check_refs, refs)
        chunk = max(1, min(2, n_chk))
        n_host = n_chk if args.host_check_refs < 0 else min(args.host_check_refs, n_chk)
        d_cpu, d_host = _oracle_views(cfg, refs, seed, [(n_chk, True), (n_host, False)], chunk, min(best_nt, 16))
        d_gpu, d_gpu32 = depth[:n_chk].cpu(), depth32[:n_chk].cpu()
        rel = float(((d_gpu - d_cpu).abs() / d_cpu).max())
        rel32 = float(((d_gpu32 - d_cpu).abs() / d_cpu).max())
        # ... and the plain torch oracle of this host (its BLAS's own evaluation order) on the same views: all of them at
        # cfg2 (a few seconds each at the best thread count), --host-check-refs of them otherwise
        rel_host = float(((d_gpu[:n_host] - d_host).abs() / d_host).max())
        rel_host32 = float(((d_gpu32[:n_host] - d_host).abs() / d_host).max())
        # reference-vs-reference spread: the two CPU evaluations of the SAME reference arithmetic (pinned orders of the
        # golden run vs this host's BLAS orders) on the same views -- what a 1e-4 gate against "the reference" can resolve
        spread = float(((d_cpu[:n_host] - d_host).abs() / d_host).max())
        abs_rel = float(((d_gpu - d_cpu).abs() / (d_cpu + 1e-7)).mean())   # eval/metricfunctions.py:41 with gt := oracle
        cpu_baseline = dict(value=by_threads.get(cores), unit='depth maps/s', cores=cores, kind='port',
                            sample='%d reference view(s) of the same %s batch (oracle: torch CPU grid_sample + '
                                   'scatter-mean + Conv3d), 2 warm-ups, median of 5 (at every hardware thread, when a run '
                                   'takes more than 3 s: 1 warm-up, 1 run)' % (n_s, cfg),
                            by_threads={str(k): round(v, 4) for k, v in sorted(by_threads.items())},
                            probe_1view_by_threads={str(k): round(v, 4) for k, v in sorted(probe.items())},
                            value_1thread=by_threads.get(1), value_all_threads=by_threads.get(ncpu), host_threads=ncpu,
                            cpu_model=cpu_info(),
                            parallel_info=' '.join(torch.__config__.parallel_info().split())[:400],
                            checked_views=n_chk, max_rel_depth_err_gpu_vs_cpu=rel,
                            max_rel_depth_err_gpu_fp32_exact_vs_cpu=rel32, abs_rel_gpu_vs_cpu=abs_rel,
                            checker='oracle with pinned evaluation orders (oracle/pinned.py)',
                            max_rel_depth_err_gpu_vs_host_blas_oracle=rel_host,
                            max_rel_depth_err_gpu_fp32_exact_vs_host_blas_oracle=rel_host32, host_blas_checked_views=n_host,
                            max_rel_depth_spread_pinned_vs_host_blas_oracle=spread)

    if rank != 0:
        return None
    e = inp['edges'].shape[1] // refs
    workload = {
        'cfg2': 'cfg2: ScanNet-shape 256x320, 1 ref + 7 src (8 edges/ref), 96 planes, 56x56 plane grid, 32-ch quarter '
                'features 64x80; fused warp+variance -> CostRegNet -> soft-argmin depth (rows A1-A6)',
        'cfg5': 'cfg5: 480x640, 1 ref + 10 src (11 edges/ref), 192 planes, 120x160 plane grid, 32-ch quarter features '
                '120x160; fused warp+variance -> CostRegNet -> soft-argmin depth (rows A1-A6)'}[cfg]
    return {
        'metric': 'depth maps/sec (256x320, 96 planes, 7 src)' if cfg == 'cfg2'
                  else 'depth maps/sec (480x640, 192 planes, 10 src)',
        'value': value, 'value_fp32_exact': value32, 'unit': 'depth maps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'ms_per_step_fp32_exact': elapsed32 / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE,
        'data': 'synthetic',
        'config': {'workload': workload, 'refs_per_step_per_gpu': refs, 'n_img_per_gpu': inp['n_img'],
                   # (the reference's arithmetic type beside the headline, inside `config` so that a record that keeps only the
                   # contract's keys still carries the VALUES: exact-fp32 MFMA operands, `value` uses split-bf16 x 3)
                   'value_fp32_exact': value32, 'ms_per_step_fp32_exact': elapsed32 / args.steps * 1e3,
                   'edges_per_ref': e,
                   'parallelism': 'ref-view sharding, no collective' if world > 1 else 'single GPU',
                   'launch': launch_mode,
                   'ranks_seen': n_seen},
        'roofline': roofline, 'cpu_baseline': cpu_baseline, 'kernels': kernels}


# ------------------------------------------------------------------------------------------------------------------
# cfg3 / cfg4: the full pipeline on one scene (rows A, B, C; H2 driver)
# ------------------------------------------------------------------------------------------------------------------
_ORACLE_CHAIN = {}        # oracle depths of the parity scene after every outer iteration (see bench_scene)


def bench_scene(args, rank, world, dev, dist):
    syn = importlib.import_module('3dvnet_amd.synthetic')
    lm = importlib.import_module('3dvnet_amd.lightningmodel')
    drv = importlib.import_module('3dvnet_amd.eval_3dvnet')
    libm = importlib.import_module('3dvnet_amd._lib')
    Batch = importlib.import_module('3dvnet_amd.batch').Batch
    refs = args.refs or 64
    cfg = syn.CONFIGS['cfg3']
    # cfg3/4 use 1 ref + 7 src = 8 edges per reference view (ref-4 .. ref+3), as cfg2 does.  (The reference's own
    # eval script runs its driver with 2 src on either side = 5 edges, eval/main.py:36: `--scene-window 2,2`.)
    nb, na = (int(v) for v in args.scene_window.split(','))
    win = (nb, na)
    stage3 = bool(getattr(args, 'stage3', False))       # + the three PropagationNet upsampling steps to 256 x 320

    def make(n_ref, seed):
        edges, n_img = syn.make_edges(n_ref, nb, na)
        rot, tv, K = syn.make_cameras(n_img, cfg['img_size'], seed=seed, yaw_step_deg=360.0 / max(n_img, 60))
        bb = Batch(None, rot, tv, K, None, edges)
        bb.features_quarter = syn.make_features(n_img, 32, *cfg['feat_size'], seed=seed)
        if stage3:      # the guides of stage 3 (eval-3dvnet.py:36,62,118): half-resolution features and the images
            bb.features_half = syn.make_features(n_img, 32, 2 * cfg['feat_size'][0], 2 * cfg['feat_size'][1], seed=seed + 1)
            bb.images = syn.make_images(n_img, cfg['img_size'], seed=seed + 2)
        # Random synthetic features give noise depths => a volume-filling point cloud.  Stage 1 runs and is timed, but
        # its output is then replaced by surface-like depths (analytic wall depth of the box room + 2 cm seeded noise,
        # SURVEY §8d) so that the scene model and the point-flow sweeps see a ScanNet-like voxel count.
        gt = syn.ray_box_depth(rot[nb:nb + n_ref], tv[nb:nb + n_ref], K[nb:nb + n_ref], cfg['img_size'],
                               drv.DEPTH_CONFIG['size'])
        gt = gt + 0.02 * torch.randn(gt.shape, generator=torch.Generator().manual_seed(7))
        return bb, gt
    b, gt = make(refs, 1237)
    b = b.to(dev)          # inputs resident in HBM before the timed region (features = the backbone's output, cameras, edges)
    gt = gt.to(dev)
    sds = dict(cr=syn.costregnet_weights(seed=0, sharpen=200.0), pn=syn.pointnet_weights(), un=syn.sparse_unet_weights(),
               dec=syn.decoder_weights(sharpen=50.0))
    sds_prop = [syn.propagation_weights(33, 32, 5), syn.propagation_weights(33, 32, 6), syn.propagation_weights(4, 32, 7)]

    def make_net(precision):
        n_ = lm.PL3DVNet(None, drv.DEPTH_CONFIG, cfg['edge_len'], feat_dim=32, img_size=cfg['img_size'], precision=precision).eval()
        n_.mvsnet.cnn_3d.load_state_dict(sds['cr'], strict=False)
        n_.pointnet.load_state_dict(sds['pn'])
        n_.sparse_conv.load_state_dict(sds['un'])
        n_.decoder.load_state_dict(sds['dec'], strict=False)
        for m, sdp in zip((n_.refine_quarter, n_.refine_half, n_.refine_full), sds_prop):
            m.load_state_dict(sdp, strict=False)
        return n_.to(dev)
    net = make_net('split_bf16')
    # the reference's arithmetic type: every matrix-core kernel of stages 1 and 2 on exact-fp32 operands
    # (cost volume: the fp32 chain of cfg2; PointNet / sparse U-Net: v3d_gemm_gather_f32 with V3D_PRECISION_FP32; hypothesis
    # decoder: the unfused interpolation + conv1d chain on fp32 operands -- the fused kernel is split-bf16 only; stage 3: the
    # row-marching PropagationNet kernel on v_mfma_f32_16x16x4_f32).
    net32 = make_net('fp32') if getattr(args, 'fp32_exact', True) else None
    group = None

    def step(n_=None):
        return drv.process_scene(b, n_ or net, win, dev, rank=rank, world=world, group=group, gather_depth=False,
                                 init_depth_override=gt, upsample=stage3)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        d = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(d).all()
    value = refs * args.steps / elapsed
    value32 = ms32 = None
    if net32 is not None:
        steps32 = max(2, args.steps // 2)
        step(net32)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps32):
            d32 = step(net32)
        fence()
        e32 = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([e32], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e32 = float(tt.item())
        assert torch.isfinite(d32).all()
        value32, ms32 = refs * steps32 / e32, e32 / steps32 * 1e3
    n_seen = ranks_seen(args, world, dev, dist)
    # the one collective of the path: the all-gather of the feature-rich point cloud, once per outer iteration -- its
    # payload and its time on this machine, measured on tensors of the scene's shapes between barriers
    collective = None
    if dist is not None:
        n_pix = drv.DEPTH_CONFIG['size'][0] * drv.DEPTH_CONFIG['size'][1]
        rows = [(drv.shard_range(refs, g, world)[1] - drv.shard_range(refs, g, world)[0]) * n_pix for g in range(world)]
        mine = rows[rank]
        pts, feat = torch.rand((mine, 3), device=dev), torch.rand((mine, 32), device=dev)
        pb = torch.zeros(mine, dtype=torch.long, device=dev)
        for _ in range(3):
            drv.gather_pointcloud(pts, feat, pb, rows, group)
        fence()
        t0 = time.perf_counter()
        for _ in range(20):
            drv.gather_pointcloud(pts, feat, pb, rows, group)
        fence()
        tg = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        collective = dict(op='all_gather_into_tensor of [pts | feat | batch id] rows (3 + 32 + 1 floats), rank order = view order',
                          per_outer_iteration=1, outer_iterations=len(drv.OFFSETS_LIST), bytes_per_rank=max(rows) * 36 * 4,
                          bytes_total=sum(rows) * 36 * 4, avg_ms=float(tg.item()) * 1e3,
                          note='includes packing the three tensors into one send buffer and unpacking the result')
    kernels, roofline = {}, None
    if rank == 0:
        libm.timing_enable(True)
        step()
        torch.cuda.synchronize()
        st = libm.timing_collect()
        libm.timing_enable(False)
        tot = sum(ms for ms, _ in st.values())
        for name, (ms, c) in sorted(st.items(), key=lambda kv: -kv[1][0]):
            kernels[name] = dict(total_ms=round(ms, 3), launches=c, share=round(ms / tot, 3))
        dom = max(st, key=lambda kk: st[kk][0])
        stage3_info = None
        if stage3:
            # stage 3's own figures: time of its launches inside the scene, and the four conv layers priced against the
            # split-bf16 MFMA peak: 2*9*(33*32 + 2*32*32 + 32*9) FLOP per pixel at 1/4 and 1/2 resolution, 4 instead of 33
            # input channels at full resolution (upsampling.py:17-20)
            H, W = cfg['img_size']
            per_px = lambda cin: 2.0 * 9 * (cin * 32 + 2 * 32 * 32 + 32 * 9)
            flops3 = (refs // world) * ((H // 4) * (W // 4) * per_px(33) + (H // 2) * (W // 2) * per_px(33) + H * W * per_px(4))
            fused = 'propagation_fused' in st       # round 6: one row-marching kernel per net (csrc/propz.hip)
            conv_ms = sum(ms for kk, (ms, _) in st.items() if kk.startswith('propagation_fused' if fused else 'propagation_conv'))
            all_ms = sum(ms for kk, (ms, _) in st.items() if kk.startswith('propagation_'))
            n_launch = sum(c for kk, (_, c) in st.items() if kk.startswith('propagation_fused' if fused else 'propagation_conv'))
            a3 = flops3 / (conv_ms * 1e-3) / 1e12
            stage3_info = dict(stage3_kernel_ms_per_scene=round(all_ms, 3), conv_kernel_ms_per_scene=round(conv_ms, 3),
                               roofline=dict(bound='mfma', achieved=a3, peak=PEAK_BF16_MFMA_TFLOPS / 3.0, unit='TFLOP/s',
                                             frac=a3 / (PEAK_BF16_MFMA_TFLOPS / 3.0),
                                             kernel=('propagation_fused (one row-marching kernel per net: 4 conv layers + softmax + '
                                                     '3x3 propagation, nearest resize in the addressing; all three resolutions)'
                                                     if fused else 'propagation_conv1..4 (all three resolutions)'),
                                             avg_ms=conv_ms / max(n_launch, 1),
                                             traffic=traffic_for('propagation_fused' if fused else 'propagation_conv', refs, 'cfg3')))
        if dom in ('conv1d_gemm', 'decoder_fused'):
            # decoder conv1d stack: 2*7*P*(3*352*128 + 2*3*128*128 + 3*128) FLOP per view per sweep, 6 sweeps
            P = drv.DEPTH_CONFIG['size'][0] * drv.DEPTH_CONFIG['size'][1]
            flops = 2.0 * 7 * P * (3 * 352 * 128 + 2 * 3 * 128 * 128) * (refs // world) * 6
            a = flops / (st[dom][0] * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            roofline = dict(bound='mfma', achieved=a, peak=peak, unit='TFLOP/s', frac=a / peak, kernel=dom,
                            avg_ms=st[dom][0] / st[dom][1], traffic=traffic_for(dom, refs, 'cfg3'))
    # ---- parity of the refinement leg: an 8-view scene of the same shapes / weights through the same driver, HIP against the
    # oracle-backed net (CPU; rows A1-A4 and the back-projections with the pinned orders)
    parity, cpu_baseline = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.net import OracleNet          # checker / reported baseline only
        n_chk = 8
        bs, gts = make(n_chk, 77)
        from oracle import scene as osc

        def cell_ids(p_):        # voxel cell of every point, from the run's own bounding box (utils.py:39-45)
            p_ = p_.double().cpu()
            return torch.floor((p_ - p_.min(0).values) / cfg['edge_len']).long()
        with torch.no_grad():
            nt = min(32, os.cpu_count() or 1)
            torch.set_num_threads(nt)
            onet = OracleNet(sds['cr'], sds['pn'], sds['un'], sds['dec'], cfg['img_size'], cfg['edge_len'], pinned=True)
            # Protocol ("cell flips").  The refinement leg is compared ONE OUTER ITERATION AT A TIME (scene model + 3
            # sweeps), each started from the oracle's depths: with identical depths the HIP back-projection equals the pinned
            # oracle's bit for bit, hence identical voxel cells, and the comparison sees the arithmetic of the networks.  The
            # free-running figure is reported beside it with the number of points that sit in different cells at the start of
            # the second iteration: one such point (1e-5 m of offset difference next to a cell face) adds or moves a voxel, and
            # the sparse U-Net's global receptive field turns that into centimetres for thousands of pixels -- in the reference
            # as much as here; at 8 views (25 088 points) every seed tried has 1-3 of them.
            d_free = drv.process_scene(bs, net, win, dev, init_depth_override=gts.to(dev)).cpu()
            state, errs, errs32, t_w, hip_after0, cpu_after0 = gts, [], [], 0.0, None, None
            # (the oracle's chain depends on the scene, the weights and the window only: the default bench line runs this leg for
            # cfg3 and again for cfg3 + stage 3 -- the second run takes the first one's oracle depths and its timing)
            memo_key = (n_chk, 77, win, cfg['edge_len'])
            memo = _ORACLE_CHAIN.get(memo_key)
            for it, offs in enumerate(drv.OFFSETS_LIST):
                d_it = drv.process_scene(bs, net, win, dev, init_depth_override=state.to(dev), offsets_list=[offs]).cpu()
                tp = time.perf_counter()
                if memo is not None:
                    nxt = memo['states'][it]
                else:
                    nxt = drv.process_scene(bs, onet, win, torch.device('cpu'), init_depth_override=state, offsets_list=[offs])
                    _ORACLE_CHAIN.setdefault(memo_key, {'states': [], 't_w': 0.0})['states'].append(nxt)
                t_w += time.perf_counter() - tp
                errs.append(float(((d_it - nxt).abs() / nxt).max()))
                if net32 is not None:      # the same teacher-forced iteration on exact-fp32 operands
                    d32 = drv.process_scene(bs, net32, win, dev, init_depth_override=state.to(dev), offsets_list=[offs]).cpu()
                    errs32.append(float(((d32 - nxt).abs() / nxt).max()))
                if it == 0:
                    hip_after0, cpu_after0 = d_it, nxt
                state = nxt
            oracle_states = [gts] + list(_ORACLE_CHAIN[memo_key]['states'])
            if memo is not None:
                t_w = memo['t_w']
            else:
                _ORACLE_CHAIN[memo_key]['t_w'] = t_w
            d_cpu = d_grid = state
            zb = torch.zeros(n_chk, dtype=torch.long)
            e_chk = bs.ref_src_edges
            p_hip = net.construct_feature_rich_pointcloud(hip_after0.to(dev), zb.to(dev), bs.features_quarter.to(dev), bs.rotmats.to(dev),
                                                          bs.tvecs.to(dev), bs.K.to(dev), e_chk.to(dev))[0]
            p_cpu = osc.feature_rich_pointcloud(cpu_after0, zb, bs.features_quarter, bs.rotmats, bs.tvecs, bs.K, e_chk,
                                                cfg['img_size'], pinned=True)[0]
            flips = int((cell_ids(p_hip) != cell_ids(p_cpu)).any(dim=1).sum())

            def free_run_stats(n_):
                """The FREE-RUNNING chain of net n_ (every outer iteration from its own depths) against the oracle's free-running
                chain: points in different voxel cells at the start of every outer iteration, and the final depths' max / median
                relative difference and the fraction of pixels within the 1e-4 gate."""
                cur, flips_it = gts, []
                for it, offs in enumerate(drv.OFFSETS_LIST):
                    ph = n_.construct_feature_rich_pointcloud(cur.to(dev), zb.to(dev), bs.features_quarter.to(dev), bs.rotmats.to(dev),
                                                              bs.tvecs.to(dev), bs.K.to(dev), e_chk.to(dev))[0]
                    pc = osc.feature_rich_pointcloud(oracle_states[it], zb, bs.features_quarter, bs.rotmats, bs.tvecs, bs.K, e_chk,
                                                     cfg['img_size'], pinned=True)[0]
                    flips_it.append(int((cell_ids(ph) != cell_ids(pc)).any(dim=1).sum()))
                    cur = drv.process_scene(bs, n_, win, dev, init_depth_override=cur.to(dev), offsets_list=[offs]).cpu()
                rel = ((cur - oracle_states[-1]).abs() / oracle_states[-1]).flatten()
                return dict(max_rel_depth_err_gpu_vs_cpu=float(rel.max()), median_rel_depth_err=float(rel.median()),
                            fraction_of_pixels_within_1e_4=float((rel <= 1e-4).float().mean()),
                            points_in_different_cells_per_outer_iteration=flips_it, pixels=int(rel.numel()))
            want_free = getattr(args, 'free_stats', True)      # (the default line's cfg3 + stage 3 leg does not repeat them)
            free_stats = free_run_stats(net) if want_free else None
            free_stats32 = free_run_stats(net32) if want_free and net32 is not None else None
            # CPU baseline (SURVEY 8d protocol within a time budget): the chain above is one pass over the scene (it is also the
            # checker); as many further timed passes as fit ~20 s (at most 5), median; the sample string says what was run
            n_timed = min(5, int(20.0 / max(t_w, 1e-3))) if getattr(args, 'cpu_timing', True) else 0
            passes = []
            for _ in range(n_timed):
                tp = time.perf_counter()
                drv.process_scene(bs, onet, win, torch.device('cpu'), init_depth_override=gts)
                passes.append(time.perf_counter() - tp)
            t_scene = sorted(passes)[len(passes) // 2] if passes else t_w
            t_cpu = time.perf_counter()
            d_hip = drv.process_scene(bs, net, win, dev, init_depth_override=gts.to(dev), upsample=stage3).cpu() if stage3 else d_free
            if stage3:      # the oracle's stage-3 chain (oracle/scene.py::propagation_net) on the refined plane-grid depths
                import torch.nn.functional as F
                for sdp, gd in zip(sds_prop, (bs.features_quarter[nb:nb + n_chk], bs.features_half[nb:nb + n_chk],
                                              bs.images[nb:nb + n_chk])):
                    d_cpu = osc.propagation_net(gd, F.interpolate(d_cpu.unsqueeze(1), gd.shape[-2:], mode='nearest'), sdp)
            t_cpu = t_scene + (time.perf_counter() - t_cpu)        # (+ the oracle's stage-3 chain when stage 3 is timed)
            stage3_err = None
            if stage3:      # stage 3 by itself: the HIP upsampling chain on the ORACLE's refined depths against the oracle's chain
                from importlib import import_module
                up = import_module('3dvnet_amd.upsampling').upsample_depth
                d_up = up(d_grid.to(dev), [(net.refine_quarter, bs.features_quarter[nb:nb + n_chk].to(dev)),
                                           (net.refine_half, bs.features_half[nb:nb + n_chk].to(dev)),
                                           (net.refine_full, bs.images[nb:nb + n_chk].to(dev))]).cpu()
                stage3_err = float(((d_up - d_cpu).abs() / d_cpu).max())
        # the reference's CPU path of the WHOLE pipeline (dense-formulation sparse convolutions, torch CPU
        # grid_sample / Conv3d) timed beside the GPU figure
        cpu_baseline = dict(value=n_chk / t_cpu, unit='depth maps/s', cores=nt, kind='port',
                            sample='one %d-view scene of the same shapes / weights through the same driver on the oracle-backed '
                                   'net (oracle/net.py; sample coordinates and back-projections with the pinned orders, i.e. '
                                   'elementwise torch ops instead of bmm): ' % n_chk +
                                   ('1 warm-up pass (%.1f s, also the checker), then median of %d timed pass(es) (%.1f s per scene)'
                                    % (t_w, n_timed, t_cpu) if n_timed else 'one pass of %.1f s (the checker), no warm-up' % t_cpu),
                            cpu_model=cpu_info())
        free_err = float(((d_hip - d_cpu).abs() / d_cpu).max())
        parity = dict(checked_views=n_chk, checker='oracle-backed scene driver (oracle/net.py: oracle/costvolume.py + '
                      'oracle/scene.py, pinned orders), same driver code, CPU',
                      protocol='every outer iteration (scene model + 3 sweeps) from the oracle\'s depths at its start: identical '
                               'depths give bit-identical points, hence identical voxel cells; max over the iterations',
                      max_rel_depth_err_gpu_vs_cpu=max(errs + ([stage3_err] if stage3_err is not None else [])),
                      per_outer_iteration=errs, per_outer_iteration_fp32_exact=errs32 or None,
                      max_rel_depth_err_gpu_fp32_exact_vs_cpu=max(errs32) if errs32 else None,
                      stage3_from_the_oracles_depths=stage3_err,
                      free_running=dict(max_rel_depth_err_gpu_vs_cpu=free_err, points_in_different_cells_at_iteration_2=flips,
                                        points=int(p_hip.shape[0]), stages_1_2=free_stats, stages_1_2_fp32_exact=free_stats32,
                                        note='free-running runs differ by <= 1e-5 m after the first iteration; a point that this '
                                             'moves across a cell face changes the voxel set, which the sparse U-Net\'s global '
                                             'receptive field amplifies to centimetres (discretisation of the algorithm, not '
                                             'arithmetic); 0 such points => the free-running figure is an arithmetic one'
                                             + (' (stage 3 included)' if stage3 else '')),
                      max_abs_refinement_m=float((d_grid - gts).abs().max()))
    if rank != 0:
        return None
    line_extra = {'stage3': stage3_info} if stage3 and rank == 0 else {}
    return {
        **line_extra,
        'metric': 'depth maps/sec (256x320, 96 planes, full 3DVNet pipeline: cost volume + scene model + 2x3 sweeps%s)'
                  % (' + stage-3 upsampling to 256x320' if stage3 else ''),
        'value': value, 'unit': 'depth maps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'value_fp32_exact': value32, 'ms_per_step_fp32_exact': ms32,
        'higher_is_better': True,
        'scaling': 'strong' if world > 1 else 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'value_fp32_exact': value32, 'ms_per_step_fp32_exact': ms32,
                   'fp32_exact_note': 'every matrix-core kernel on exact-fp32 operands (the hypothesis decoder as its unfused chain'
                                      + (', stage 3 as the same row-marching kernel on v_mfma_f32_16x16x4_f32)' if stage3 else ')'),
                   'workload': '%s: one %d-view scene of the 6x5x3 m box room, 256x320, %d edges/ref (ref-%d .. ref+%d), '
                               '96 planes, 56x56 plane grid, 4 cm voxels; stage A (timed) -> its depths replaced '
                               'by analytic wall depth + 2 cm noise -> 2 x (scene model + 3 point-flow sweeps)%s'
                               % (args.config, refs, nb + na + 1, nb, na,
                                  ' -> stage 3: nearest + PropagationNet at 64x80, 128x160, 256x320 (eval-3dvnet.py:101-125)'
                                  if stage3 else ''),
                   'refs_per_scene': refs, 'refs_per_gpu': refs // world, 'edges_per_ref': nb + na + 1,
                   'parallelism': ('ref-view sharding + RCCL all-gather of the feature-rich point cloud per outer '
                                   'iteration (the communicating mode)' if world > 1 else 'single GPU'),
                   'ranks_seen': n_seen, 'collective': collective},
        'roofline': roofline, 'cpu_baseline': cpu_baseline, 'parity': parity, 'kernels': kernels}


def bench_backbone(dev, n_img=71, iters=10):
    """the 2D MnasNet-1.0 + FPN backbone (mvsnet.py:55-105; fp32, seeded random weights) on the cfg2 batch's 71
    images of 256 x 320 -- the library's own kernels (csrc/backbone.hip through backbone.NativeBackbone), with the stock PyTorch-ROCm
    / MIOpen modules timed beside them and the agreement of the two.  Not part of `value`: the cost-volume benches start from
    quarter-resolution features, as BASELINE config 2 does; `from_images` adds the two."""
    bb = importlib.import_module('3dvnet_amd.backbone')
    syn = importlib.import_module('3dvnet_amd.synthetic')
    libm = importlib.import_module('3dvnet_amd._lib')
    fe, fs = bb.build_backbone(32)
    sd_e, sd_s = syn.backbone_weights(32, seed=6)
    fe.load_state_dict(sd_e, strict=False)
    fs.load_state_dict(sd_s)
    fe, fs = fe.eval().to(dev), fs.eval().to(dev)
    imgs = syn.make_images(n_img, (256, 320), seed=8).to(dev)
    nat = bb.NativeBackbone(fe, fs)                       # the package default: split-bf16 matrix operands, fused blocks
    nat32 = bb.NativeBackbone(fe, fs, precision='fp32')   # exact-fp32 matrix instructions, the per-layer kernels of round 5
    assert nat.supports(imgs)

    def timed(fn):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, out
    with torch.no_grad():
        ms_stock, out_s = timed(lambda: fs(*fe(imgs)))       # the explicit stock path (`native_backbone = False`): MIOpen / rocBLAS
        ms, out = timed(lambda: nat(imgs))
        ms32, out32 = timed(lambda: nat32(imgs))
        imgs240 = syn.make_images(n_img, (240, 320), seed=9).to(dev)      # the reference's default MVSNet(img_size=(240, 320))
        ms240, _ = timed(lambda: nat(imgs240))
        libm.timing_enable(True)
        nat(imgs)
        torch.cuda.synchronize()
        st = libm.timing_collect()
        libm.timing_enable(False)
    agree = max(float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(out, out_s))
    agree32 = max(float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(out32, out_s))
    kern = {k: dict(total_ms=round(v[0], 3), launches=v[1]) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}
    return {'workload': 'MnasNet-1.0 trunk + FPN + shrinker, %d images 256x320 -> half / quarter / eighth (/ 16th / 32nd) '
                        'features: hand-written HIP kernels on channels-last fp32 activations -- one kernel per inverted-residual '
                        'block, per fine pyramid level and for the stem (split-bf16 matrix operands, fp32 depthwise taps); '
                        '`fp32_exact`: the per-layer kernels on exact-fp32 matrix instructions' % n_img,
            'ms_per_batch': ms, 'images_per_s': n_img / ms * 1e3, 'ms_per_batch_fp32_exact': ms32,
            'kernel_ms_per_batch': round(sum(v[0] for v in st.values()), 3), 'launches_per_batch': sum(v[1] for v in st.values()),
            'kernels': kern, 'ms_per_batch_240x320': ms240, 'stock_pytorch_miopen_ms_per_batch': ms_stock,
            'max_diff_vs_stock_modules_of_range': agree, 'max_diff_vs_stock_modules_of_range_fp32_exact': agree32,
            'quarter_features': list(out[1].shape),
            'note': 'parity with torchvision unpinned (absent here); tests/test_backbone.py pins the kernels against oracle/backbone.py '
                    'on the CPU (256x320, 240x320, 248x328): 8e-5 of range split-bf16 (measured <= 4.0e-5), 2e-5 exact fp32'}


def compact(line):
    """The figures of a full bench line that go into the default line's "extra" object."""
    out = {k: line[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'value_fp32_exact',
                                'ms_per_step_fp32_exact') if k in line}
    out['workload'] = line['config']['workload']
    for k in ('refs_per_step_per_gpu', 'refs_per_scene', 'edges_per_ref'):
        if k in line['config']:
            out[k] = line['config'][k]
    out['roofline'] = line.get('roofline')
    cb = line.get('parity') or line.get('cpu_baseline') or {}
    out['parity'] = {k: cb[k] for k in ('checked_views', 'checker', 'max_rel_depth_err_gpu_vs_cpu',
                                        'max_rel_depth_err_gpu_fp32_exact_vs_cpu', 'abs_rel_gpu_vs_cpu',
                                        'max_rel_depth_err_gpu_vs_host_blas_oracle',
                                        'max_rel_depth_err_gpu_fp32_exact_vs_host_blas_oracle', 'host_blas_checked_views',
                                        'max_rel_depth_spread_pinned_vs_host_blas_oracle', 'max_abs_refinement_m', 'protocol',
                                        'per_outer_iteration', 'per_outer_iteration_fp32_exact', 'stage3_from_the_oracles_depths',
                                        'free_running') if k in cb}
    if line['config'].get('fp32_exact_note'):
        out['fp32_exact_note'] = line['config']['fp32_exact_note']
    if line.get('cpu_baseline') and line.get('parity'):
        out['cpu_baseline'] = line['cpu_baseline']
    top = sorted(line.get('kernels', {}).items(), key=lambda kv: -kv[1].get('share', 0))[:4]
    out['top_kernels'] = {k: {kk: v[kk] for kk in ('avg_ms', 'total_ms', 'share', 'frac') if kk in v} for k, v in top}
    return out


def rank_command(n, port, argv):
    """Command line that re-runs this script as n ranks on this node (what the driver itself uses for N > 1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: re-exec under it (one process per GPU)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(rank_command(n, port, sys.argv[1:]), env=env)


class _DryNet:
    """Stand-in for PL3DVNet in `--config cfg4 --dry-run`: the three methods eval_3dvnet.process_scene calls, with
    bookkeeping arithmetic only (a view's "depth" is its global image index, carried in tvecs[:, 0]) -- so that the REAL
    driver code (sharding, chunking, halos, gather_pointcloud with uneven shards, the final all-gather of the depths) runs
    over gloo on CPU and its result can be checked in closed form.  It is not on any product or measured path."""

    def __init__(self, k):
        self.k = k

    def make_initial_depth_predictions(self, sl, cfg):
        n_ref, (h, w) = sl.n_ref, cfg['size']
        ident = sl.tvecs[self.k:self.k + n_ref, 0].float()
        return ident.view(-1, 1, 1).expand(n_ref, h, w).clone(), None, None, sl.features_quarter, None, None

    def model_scene(self, depth, depth_batch, feats, rot, tv, K, edges, gather_fn=None, **_):
        pts = depth.reshape(-1, 1).repeat(1, 3)
        feat = torch.ones((pts.shape[0], 2), dtype=torch.float32)
        bid = torch.zeros(pts.shape[0], dtype=torch.long)
        if gather_fn is not None:
            pts, feat, bid = gather_fn(pts, feat, bid)
        return {'rows': pts.shape[0], 'sum': float(pts[:, 0].double().sum()), 'sorted': bool((pts[1:, 0] >= pts[:-1, 0]).all())}

    def run_pointflow(self, xs, depth, *_, **__):
        return torch.full_like(depth, 1e-3 * xs['rows'])


def bench_dry_run(args, rank, world):
    """`--dry-run`: the multi-rank plumbing of this script without a GPU (gloo, CPU tensors, a no-op step): rendezvous from
    the launcher's environment, barrier-bracketed timing, max over ranks, every rank counted, ONE JSON line from rank 0.
    Exercised by tests/test_driver.py so that a future `bench.py --gpus 8` cannot die on plumbing."""
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    refs = args.refs or 64
    scene_check = None
    if args.config == 'cfg4':
        # the communicating mode's host path: one scene sharded by reference view through the real driver (see _DryNet)
        drv = importlib.import_module('3dvnet_amd.eval_3dvnet')
        syn = importlib.import_module('3dvnet_amd.synthetic')
        Batch = importlib.import_module('3dvnet_amd.batch').Batch
        nb, na = (int(v) for v in args.scene_window.split(','))
        edges, n_img = syn.make_edges(refs, nb, na)
        rot, tv, K = syn.make_cameras(n_img, (64, 80), seed=3)
        tv = tv.clone()
        tv[:, 0] = torch.arange(n_img, dtype=tv.dtype)                 # image identity for _DryNet
        b = Batch(None, rot, tv, K, None, edges)
        b.features_quarter = torch.zeros((n_img, 32, 2, 2))
        cfg = {'depth_start': 0.5, 'depth_interval': 0.1, 'n_intervals': 8, 'size': (4, 4)}
        offsets = [[0.05, 0.025], [0.05]]
        d = drv.process_scene(b, _DryNet(nb), (nb, na), torch.device('cpu'), cfg, offsets, 3, 2, rank=rank, world=world,
                              group=None, gather_depth=True)
        n_pix, sweeps = 16, sum(len(o) for o in offsets)
        shard = [drv.shard_range(refs, g, world) for g in range(world)]
        want = (torch.arange(nb, nb + refs, dtype=torch.float32) + sweeps * 1e-3 * refs * n_pix).view(-1, 1, 1).expand(refs, 4, 4)
        scene_check = {'refs': refs, 'shard_views': [e - s0 for s0, e in shard], 'gathered_rows_per_outer_iteration': refs * n_pix,
                       'depths_equal_closed_form': bool(d.shape == want.shape and torch.allclose(d, want, rtol=0, atol=1e-4))}

    def fence():
        if dist is not None:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(1e-3)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank))            # uneven ranks: the reported time must be the slowest rank's
    fence()
    el = time.perf_counter() - t0
    seen = 1
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        cnt = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(cnt)
        seen = int(cnt.item())
    if rank == 0:
        print(json.dumps({'metric': 'dry run (no GPU work)', 'value': None, 'unit': 'depth maps/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE,
                          'data': 'none', 'dry_run': True,
                          'config': {'workload': 'dry run of the rank plumbing', 'refs_per_step_per_gpu': refs,
                                     'parallelism': 'ref-view sharding, no collective' if world > 1 else 'single process',
                                     'ranks_seen': seen, 'cfg4_scene_check': scene_check}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='cfg2', choices=('cfg2', 'cfg5', 'cfg3', 'cfg4'))
    ap.add_argument('--refs', type=int, default=0, help='reference views per step per GPU (cfg2: 64, cfg5: 8; '
                    'cfg3/cfg4: views of the scene, 64)')
    ap.add_argument('--cpu-refs', type=int, default=1, help='reference views in the timed CPU-baseline sample')
    ap.add_argument('--check-refs', type=int, default=-1, help='views of the timed GPU batch compared with the '
                    'oracle (-1 = all)')
    ap.add_argument('--host-check-refs', type=int, default=16, help='views also compared with the plain torch oracle of '
                    'this host (-1 = all checked views; the pinned oracle checks every view of the step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32', dest='fp32_exact', action='store_false', help='cfg3/cfg4: skip the exact-fp32 legs (profiling runs)')
    ap.add_argument('--scene-window', default='4,3', help='cfg3/cfg4: source views before,after each reference view '
                    '(4,3 = SURVEY 8d: 1 ref + 7 src; 2,2 = the reference eval script)')
    ap.add_argument('--stage3', action='store_true', help='cfg3/cfg4: include stage 3 (PropagationNet upsampling to full '
                    'resolution) in the timed scene')
    ap.add_argument('--no-extra', action='store_true', help='default cfg2 run: do not append the cfg5 / cfg3 figures')
    ap.add_argument('--extra', action='store_true', help='append the cfg5 / cfg3 figures also when --refs is given')
    ap.add_argument('--dry-run', action='store_true', help='rank plumbing only (gloo, no GPU, no-op step): see bench_dry_run')
    ap.add_argument('--graph', action='store_true', help='time HIP-graph replays of the step instead of eager launches (cfg2 / cfg5)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d' % (args.gpus, world)
    if args.dry_run:
        return bench_dry_run(args, rank, world)
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback on the product path)'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if args.config in ('cfg3', 'cfg4'):
        if args.config == 'cfg4' and world == 1:
            print('cfg4 is cfg3 sharded over ranks: run with --gpus N (N > 1); running the 1-GPU scene', file=sys.stderr)
        line = bench_scene(args, rank, world, dev, dist)
    else:
        t_leg = time.perf_counter()
        line = bench_costvolume(args, rank, world, dev, dist)
        wall = {'headline': round(time.perf_counter() - t_leg, 1)}
        if args.config == 'cfg2' and world == 1 and not args.no_extra and (args.extra or not args.refs):
            # every configuration's figure in the driver-run line: cfg5 (8 views per step) and the cfg3 scene, fewer steps,
            # the oracle only as the checker (its timing legs belong to the headline configuration)
            import copy
            extra = {}
            a5 = copy.copy(args)
            a5.config, a5.refs, a5.steps, a5.warmup = 'cfg5', 8, min(args.steps, 10), 2
            a5.check_refs, a5.host_check_refs, a5.cpu_timing, a5.graph = 8, 4, False, False
            t_leg = time.perf_counter()
            extra['cfg5'] = compact(bench_costvolume(a5, rank, world, dev, dist))
            wall['cfg5'] = round(time.perf_counter() - t_leg, 1)
            t_leg = time.perf_counter()
            a3 = copy.copy(args)
            a3.config, a3.refs, a3.steps, a3.warmup = 'cfg3', 64, min(args.steps, 10), 2
            extra['cfg3'] = compact(bench_scene(a3, rank, world, dev, dist))
            wall['cfg3'] = round(time.perf_counter() - t_leg, 1)
            t_leg = time.perf_counter()
            # ... and the same scene with stage 3 (full-resolution output): BASELINE config 3's "Full 3DVNet" end to end
            a3f = copy.copy(a3)
            a3f.stage3, a3f.steps, a3f.cpu_timing, a3f.free_stats = True, min(args.steps, 10), False, False
            line3f = bench_scene(a3f, rank, world, dev, dist)
            extra['cfg3_full'] = compact(line3f)
            extra['cfg3_full']['stage3'] = line3f.get('stage3')
            wall['cfg3_full'] = round(time.perf_counter() - t_leg, 1)
            t_leg = time.perf_counter()
            extra['backbone'] = bench_backbone(dev)
            wall['backbone'] = round(time.perf_counter() - t_leg, 1)
            extra['wall_s_per_leg'] = wall
            # cfg2 from IMAGES: the backbone's batch (71 images = 64 reference views + halo) + the cost-volume step it feeds
            ms_img = extra['backbone']['ms_per_batch'] + line['ms_per_step']
            extra['from_images'] = dict(ms_per_step=ms_img, value=line['config']['refs_per_step_per_gpu'] / ms_img * 1e3,
                                        unit='depth maps/s', note='backbone on 71 images (64 reference views + 7 halo images) + '
                                        'the timed cost-volume step; sum of the two separately timed stages')
            if line.get('ms_per_step_fp32_exact'):
                ms_img32 = extra['backbone']['ms_per_batch_fp32_exact'] + line['ms_per_step_fp32_exact']
                extra['from_images'].update(ms_per_step_fp32_exact=ms_img32,
                                            value_fp32_exact=line['config']['refs_per_step_per_gpu'] / ms_img32 * 1e3)
            line['extra'] = extra
            # the other configurations' figures also INSIDE `config` (a record that keeps the contract's keys keeps these values)
            line['config']['other_configs'] = {
                k: {kk: extra[k].get(kk) for kk in ('value', 'ms_per_step', 'value_fp32_exact', 'ms_per_step_fp32_exact', 'unit')}
                for k in ('cfg5', 'cfg3', 'cfg3_full', 'from_images')}
            line['config']['multi_gpu_note'] = ('`--gpus N` at this configuration runs N communication-free replicas (weak '
                                                'scaling: reference views are independent units); the communicating mode is '
                                                '`--config cfg4 --gpus N` (one scene sharded by reference view, one RCCL '
                                                'all-gather of the point cloud per outer iteration)')
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
