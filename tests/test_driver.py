"""Row H2 (mv3d/eval-3dvnet.py:26-129) and the multi-GPU partitioning.
CPU: driver logic with the oracle as backend (chunk invariance, 2-rank gloo == 1 rank, bit-exact).
GPU: the HIP driver against the oracle-backed driver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import v3d
from oracle_net import OracleNet

CFG = {'depth_start': 0.5, 'depth_interval': 0.15, 'n_intervals': 16, 'size': (8, 8)}
IMG, FEAT = (64, 80), (16, 20)
OFFSETS = [[0.05, 0.025], [0.05]]


def make_scene(n_ref=5, k=1, seed=41):
    syn = v3d('synthetic')
    Batch = v3d('batch').Batch
    edges, n_img = syn.make_edges(n_ref, k, k)
    rot, tv, K = syn.make_cameras(n_img, IMG, seed=seed)
    b = Batch(None, rot, tv, K, None, edges)
    b.features_quarter = syn.make_features(n_img, 32, *FEAT, seed=seed)
    return b


def weights():
    syn = v3d('synthetic')
    return (syn.costregnet_weights(seed=0, sharpen=200.0), syn.pointnet_weights(seed=1),
            syn.sparse_unet_weights(seed=2), syn.decoder_weights(seed=3, sharpen=50.0))


def run_oracle(rank=0, world=1, init_b=18, off_b=16, group=None):
    drv = v3d('eval_3dvnet')
    net = OracleNet(*weights(), IMG, 0.16)
    return drv.process_scene(make_scene(), net, 1, torch.device('cpu'), CFG, OFFSETS, init_b, off_b,
                             rank=rank, world=world, group=group)


def test_shard_range_partitions():
    drv = v3d('eval_3dvnet')
    for n in (1, 5, 8, 64):
        for w in (1, 2, 3, 8):
            r = [drv.shard_range(n, g, w) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_chunk_sizes_do_not_change_results():
    ref = run_oracle()
    # the CPU oracle's library kernels (oneDNN conv, bmm) pick batch-size dependent summation orders,
    # so chunking is equal to rounding (1e-5 relative), not bit-exact; the HIP path is bit-exact
    # (tests/test_scene_gpu.py, tests/test_costvolume_gpu.py)
    np.testing.assert_allclose(run_oracle(init_b=2, off_b=3).numpy(), ref.numpy(), rtol=1e-5, atol=0)
    assert float((ref - 0.5).abs().max()) > 0.1 and torch.isfinite(ref).all()


def test_one_sided_window_chunking():
    """process_scene with a (before, after) window (SURVEY 8d's ref-4 .. ref+3 convention, here 2 / 1): chunked stage 1 and
    chunked sweeps with their one-sided image halo give the unchunked result."""
    syn, drv = v3d('synthetic'), v3d('eval_3dvnet')
    Batch = v3d('batch').Batch

    def scene(nb, na, n_ref=5):
        edges, n_img = syn.make_edges(n_ref, nb, na)
        rot, tv, K = syn.make_cameras(n_img, IMG, seed=43)
        b = Batch(None, rot, tv, K, None, edges)
        b.features_quarter = syn.make_features(n_img, 32, *FEAT, seed=43)
        return b

    def run(win, **kw):
        net = OracleNet(*weights(), IMG, 0.16)
        return drv.process_scene(scene(*win), net, win, torch.device('cpu'), CFG, OFFSETS, **kw)
    ref = run((2, 1), init_depth_batch=18, offset_batch=16)
    np.testing.assert_allclose(run((2, 1), init_depth_batch=2, offset_batch=3).numpy(), ref.numpy(), rtol=1e-5, atol=0)
    assert ref.shape[0] == 5 and torch.isfinite(ref).all()
    # the one-sided halo is what the chunks slice: the same call with the reference views shifted by one image (a
    # (1, 2) window on the same images) is a different problem with different depths
    other = run((1, 2), init_depth_batch=2, offset_batch=3)
    assert other.shape == ref.shape and float((other - ref).abs().max()) > 1e-3


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    out = run_oracle(rank=rank, world=world)
    if rank == 0:
        q.put(out.numpy())
    dist.destroy_process_group()


def test_two_ranks_gloo_equal_single_process():
    """5 reference views over 2 ranks (3 + 2, uneven): the all-gathered point cloud must reproduce the
    single-process scene (view order preserved), hence the same depths up to the CPU oracle's
    batch-size dependent rounding."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(out, run_oracle().numpy(), rtol=1e-5, atol=0)


def _hip_net(dev):
    lm = v3d('lightningmodel')
    cr, pn, un, dec = weights()
    net = lm.PL3DVNet(None, CFG, 0.16, feat_dim=32, img_size=IMG).eval()
    net.mvsnet.cnn_3d.load_state_dict(cr, strict=False)
    net.pointnet.load_state_dict(pn)
    net.sparse_conv.load_state_dict(un)
    net.decoder.load_state_dict(dec, strict=False)
    return net.to(dev)


def _nccl_worker(rank, world, port, q):
    """One process per GPU over RCCL: (1) the all-gathered feature-rich point cloud must equal the single-process
    tensor bit for bit; (2) the sharded scene driver must reproduce the single-process depths bit for bit (the HIP path
    is batch-invariant, unlike the CPU oracle's library kernels)."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', rank)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    drv, utils = v3d('eval_3dvnet'), v3d('utils')
    net = _hip_net(dev)
    scene = make_scene()
    k, n_ref = 1, 5
    with torch.no_grad():
        # the exchange step in isolation, on analytic depths
        syn = v3d('synthetic')
        depth = syn.ray_box_depth(scene.rotmats[k:k + n_ref], scene.tvecs[k:k + n_ref], scene.K[k:k + n_ref], IMG,
                                  CFG['size']).to(dev)
        feats = scene.features_quarter.to(dev)
        rot, tv, K = scene.rotmats.to(dev), scene.tvecs.to(dev), scene.K.to(dev)
        edges = scene.ref_src_edges.to(dev)
        full = net.construct_feature_rich_pointcloud(depth, torch.zeros(n_ref, dtype=torch.long, device=dev), feats, rot,
                                                     tv, K, edges)
        r0, r1 = drv.shard_range(n_ref, rank, world)
        e_loc = utils.slice_edges(edges, r0 + k, r1 + k, 0) - r0
        loc = net.construct_feature_rich_pointcloud(depth[r0:r1], torch.zeros(r1 - r0, dtype=torch.long, device=dev),
                                                    feats[r0:r1 + 2 * k], rot[r0:r1 + 2 * k], tv[r0:r1 + 2 * k],
                                                    K[r0:r1 + 2 * k], e_loc)
        n_pix = CFG['size'][0] * CFG['size'][1]
        sizes = [(drv.shard_range(n_ref, g, world)[1] - drv.shard_range(n_ref, g, world)[0]) * n_pix
                 for g in range(world)]
        got = drv.gather_pointcloud(*loc, sizes=sizes)
        same_cloud = all(torch.equal(a, b) for a, b in zip(got, full))
        out = drv.process_scene(scene, net, k, dev, CFG, OFFSETS, 2, 3, rank=rank, world=world)
        single = drv.process_scene(scene, net, k, dev, CFG, OFFSETS, 2, 3)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((same_cloud, bool(torch.equal(out, single)), dist.get_world_size(), out.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_nccl_equal_single_process():
    """cfg4's exchange on real devices (RCCL all-gather over xGMI): needs two HIP devices, skipped on a 1-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 HIP devices (the gpurun / driver test box has one)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same_cloud, same_depth, world_seen, out = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert world_seen == 2 and same_cloud and same_depth
    np.testing.assert_allclose(out, run_oracle().numpy(), rtol=1e-4, atol=0)


@pytest.mark.gpu
def test_hip_driver_matches_oracle_driver(cuda):
    lm, drv = v3d('lightningmodel'), v3d('eval_3dvnet')
    cr, pn, un, dec = weights()
    net = lm.PL3DVNet(None, CFG, 0.16, feat_dim=32, img_size=IMG).eval()
    net.mvsnet.cnn_3d.load_state_dict(cr, strict=False)
    net.pointnet.load_state_dict(pn)
    net.sparse_conv.load_state_dict(un)
    net.decoder.load_state_dict(dec, strict=False)
    net = net.to(cuda)
    scene = make_scene()
    out = drv.process_scene(scene, net, 1, cuda, CFG, OFFSETS, 2, 3)
    ref = run_oracle()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=0)
    # Row 8f-4 on the device path: the HIP depths written as the reference's preds.npz record (eval/main.py:74-101),
    # reloaded, and fed to the reference's 2D metrics with the oracle's depths as ground truth.
    import tempfile
    res = v3d('results')
    scene.images = torch.zeros((scene.rotmats.shape[0], 3) + IMG)           # only the image SIZE enters (K rescale)
    ref_idx = torch.unique(scene.ref_src_edges[0])
    img_idx = np.arange(100, 100 + scene.rotmats.shape[0])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'preds.npz')
        res.write_preds(path, '/data/scans/scene0000_00', out.cpu().numpy(), scene, ref_idx, img_idx)
        rec = dict(np.load(path))
    assert sorted(rec) == ['K', 'depth_preds', 'img_idx', 'rotmats', 'scene', 'tvecs']
    assert str(rec['scene']) == 'scene0000_00' and rec['depth_preds'].shape == (5,) + CFG['size']
    assert np.array_equal(rec['depth_preds'], out.cpu().numpy()) and rec['depth_preds'].dtype == np.float32
    assert np.array_equal(rec['img_idx'], img_idx[ref_idx.numpy()])
    assert np.array_equal(rec['rotmats'], scene.rotmats[ref_idx].numpy()) and np.array_equal(rec['tvecs'], scene.tvecs[ref_idx].numpy())
    K0 = scene.K[ref_idx].numpy()
    sx, sy = CFG['size'][1] / IMG[1], CFG['size'][0] / IMG[0]
    np.testing.assert_allclose(rec['K'][:, 0], K0[:, 0] * sx, rtol=1e-6)
    np.testing.assert_allclose(rec['K'][:, 1], K0[:, 1] * sy, rtol=1e-6)
    assert np.array_equal(rec['K'][:, 2], K0[:, 2])
    m = res.depth_metrics_2d(torch.from_numpy(rec['depth_preds']), ref)
    assert float(m['abs_rel']) < 2e-5 and float(m['d_125']) == pytest.approx(1.0) and float(m['rmse']) < 1e-4


@pytest.mark.gpu
def test_deepcopy_and_pickle_of_a_net_after_a_forward(cuda):
    """The packed weight images (ctypes handles), scratch workspaces and the native backbone's kernel handles are caches, not
    state: ``copy.deepcopy`` and ``pickle`` of a net that has already run carry the parameters only, the copies' caches belong
    to the copies, and they reproduce the original's depths bit for bit; the scene's feature tensor is released afterwards."""
    import copy
    import pickle
    lm, drv = v3d('lightningmodel'), v3d('eval_3dvnet')
    cr, pn, un, dec = weights()
    net = lm.PL3DVNet(None, CFG, 0.16, feat_dim=32, img_size=IMG, backbone=True).eval()
    net.mvsnet.cnn_3d.load_state_dict(cr, strict=False)
    net.pointnet.load_state_dict(pn)
    net.sparse_conv.load_state_dict(un)
    net.decoder.load_state_dict(dec, strict=False)
    net = net.to(cuda)
    scene = make_scene()
    scene.images = v3d('synthetic').make_images(scene.rotmats.shape[0], IMG, seed=3)
    out = drv.process_scene(scene, net, 1, cuda, CFG, OFFSETS, 2, 3)
    assert net.mvsnet.cnn_3d._handle is not None and net.sparse_conv._cache._packs is not None
    assert net.mvsnet._native_backbone is not None and net._ws.tags.get('bp') is None
    clones = [copy.deepcopy(net), pickle.loads(pickle.dumps(net))]
    for c in clones:
        assert c.mvsnet.cnn_3d._handle is None and c.sparse_conv._cache._packs is None
        assert c.sparse_conv._cache._module is c.sparse_conv and c.decoder._cache._module is c.decoder
        assert c.mvsnet._native_backbone.fe is c.mvsnet.feat_extractor
        assert torch.equal(drv.process_scene(scene, c.to(cuda), 1, cuda, CFG, OFFSETS, 2, 3), out)
    assert torch.equal(drv.process_scene(scene, net, 1, cuda, CFG, OFFSETS, 2, 3), out)      # the original is untouched


@pytest.mark.gpu
def test_hip_driver_feat_dim_16_matches_oracle_driver(cuda):
    """The reference's signature default feat_dim = 16 (lightningmodel.py:18): CostRegNet(16, 8) -- conv0 on the volume
    zero-extended to 32 channels (split-bf16) / the 16-channel exact-fp32 conv0 --, PointNet(64, 32, 19), SparseUNet((32, 128,
    128)) with GroupNorm over 8-channel groups on its first level, the unfused HypothesisDecoder(304): the whole scene driver
    against the oracle-backed driver, both operand precisions."""
    lm, drv, syn = v3d('lightningmodel'), v3d('eval_3dvnet'), v3d('synthetic')
    Batch = v3d('batch').Batch
    cr = syn.costregnet_weights(in_channels=16, seed=0, sharpen=200.0)
    pn = syn.pointnet_weights(hidden=64, out_dim=32, in_dim=19, seed=1)
    un = syn.sparse_unet_weights(dims=(32, 128, 128), seed=2)
    dec = syn.decoder_weights(in_dim=304, seed=3, sharpen=50.0)

    def scene():
        edges, n_img = syn.make_edges(5, 1, 1)
        rot, tv, K = syn.make_cameras(n_img, IMG, seed=43)
        b = Batch(None, rot, tv, K, None, edges)
        b.features_quarter = syn.make_features(n_img, 16, *FEAT, seed=43)
        return b
    ref = drv.process_scene(scene(), OracleNet(cr, pn, un, dec, IMG, 0.16), 1, torch.device('cpu'), CFG, OFFSETS, 18, 16)
    for precision, rtol in (('split_bf16', 1e-4), ('fp32', 2e-5)):
        net = lm.PL3DVNet(None, CFG, 0.16, feat_dim=16, img_size=IMG, precision=precision).eval()
        net.mvsnet.cnn_3d.load_state_dict(cr, strict=False)
        net.pointnet.load_state_dict(pn)
        net.sparse_conv.load_state_dict(un)
        net.decoder.load_state_dict(dec, strict=False)
        out = drv.process_scene(scene(), net.to(cuda), 1, cuda, CFG, OFFSETS, 2, 3)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=rtol, atol=0)


def test_driver_rejects_an_edge_list_that_breaks_the_dataset_layout():
    """The driver addresses reference views as images k .. k + n - 1 (eval-3dvnet.py:42-52) and builds its device-side edge
    tables from a chunk's reference count: an edge list in which a reference view has no edges would be processed with the
    wrong count.  It is refused on the host before anything runs."""
    drv = v3d('eval_3dvnet')
    b = make_scene()
    b.ref_src_edges = b.ref_src_edges[:, b.ref_src_edges[0] != 3]
    with pytest.raises(ValueError, match='reference views'):
        drv.process_scene(b, OracleNet(*weights(), IMG, 0.16), 1, torch.device('cpu'), CFG, OFFSETS, 18, 16)


def test_driver_rejects_a_source_view_outside_the_window():
    """A source index outside a chunk's halo would make the device-side table builder write an EMPTY table (zero variance,
    plausible depth): the driver validates the whole list on the host -- window [ref - k, ref + ka] and the scene's image
    range -- before anything runs."""
    drv = v3d('eval_3dvnet')
    net = OracleNet(*weights(), IMG, 0.16)
    b = make_scene()
    e = b.ref_src_edges.clone()
    j = int((e[0] == 3).nonzero()[0])
    e[1, j] = 0                                   # image 0 is 3 views away from reference view 3; the window is +-1
    b.ref_src_edges = e
    with pytest.raises(ValueError, match='outside the source window'):
        drv.process_scene(b, net, 1, torch.device('cpu'), CFG, OFFSETS, 18, 16)
    b = make_scene()
    e = b.ref_src_edges.clone()
    e[1, -1] = b.rotmats.shape[0]                 # one past the last image of the scene
    b.ref_src_edges = e
    with pytest.raises(ValueError, match='outside the source window'):
        drv.process_scene(b, net, (1, 2), torch.device('cpu'), CFG, OFFSETS, 18, 16)


@pytest.mark.gpu
def test_bench_line_schema_and_checks(cuda):
    """`bench.py` end to end on a small batch (8 views, 2 steps): ONE JSON line with the contract's fields, both operand
    precisions, a roofline object for the dominant kernel, the CPU baseline object with the parity check of the timed batch
    against the pinned oracle, and per-kernel times that cover the step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--refs', '8',
                        '--cpu-refs', '1', '--check-refs', '2'], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'value_fp32_exact'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0 and d['value_fp32_exact'] > 0
    assert 'split-bf16' in d['dtype'] and 'workload' in d['config']
    rf = d['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and 0 < rf['frac'] < 1 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1
    assert cb['max_rel_depth_err_gpu_vs_cpu'] < 1e-4 and cb['max_rel_depth_err_gpu_fp32_exact_vs_cpu'] < 2e-5
    assert abs(sum(v['avg_ms'] for v in d['kernels'].values()) / d['ms_per_step'] - 1) < 0.5
    assert 'extra' not in d                   # --refs without --extra: the headline configuration only


@pytest.mark.gpu
def test_bench_line_carries_cfg5_and_cfg3_figures(cuda):
    """The default `bench.py` line also reports cfg5 (8 views) and the cfg3 scene under "extra", each with a value, the time
    per step, the dominant kernel and a depth error against the oracle inside the 1e-4 gate (here forced with --extra on a
    small headline batch so that the test stays short)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--refs', '4',
                        '--cpu-refs', '1', '--check-refs', '1', '--host-check-refs', '1', '--extra'],
                       capture_output=True, text=True, timeout=1800, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    e5, e3 = d['extra']['cfg5'], d['extra']['cfg3']
    assert e5['value'] > 0 and e5['value_fp32_exact'] > 0 and e5['refs_per_step_per_gpu'] == 8 and e5['edges_per_ref'] == 11
    assert e5['roofline']['kernel'] in e5['top_kernels'] and 0 < e5['roofline']['frac'] < 1
    assert e5['parity']['checked_views'] >= 1 and e5['parity']['max_rel_depth_err_gpu_vs_cpu'] < 1e-4
    assert e3['value'] > 0 and e3['refs_per_scene'] == 64 and e3['edges_per_ref'] == 8
    assert e3['parity']['checked_views'] == 8 and e3['parity']['max_rel_depth_err_gpu_vs_cpu'] < 1e-4
    assert len(e3['parity']['per_outer_iteration']) == 2 and 'points_in_different_cells_at_iteration_2' in e3['parity']['free_running']
    assert e3['parity']['max_abs_refinement_m'] > 0.01       # the sweeps really moved the depths
    assert 'replicas' in d['config']['multi_gpu_note']
    # the stage-3 leg prices its launches (round 6: one row-marching kernel per net) against the matrix peak and carries their HBM
    # traffic from the committed PMC passes (profiles/<round>_traffic_cfg3.json); the from-images figure is the sum of the two
    # timed stages
    s3 = d['extra']['cfg3_full']['stage3']
    assert s3['roofline']['bound'] == 'mfma' and 0 < s3['roofline']['frac'] < 1 and s3['roofline']['traffic'] > 1e7
    assert 'propagation_fused' in s3['roofline']['kernel']
    assert d['extra']['cfg3']['roofline']['traffic'] > 1e8
    assert d['extra']['from_images']['value'] > 0 and d['extra']['backbone']['ms_per_batch'] > 0
    assert d['extra']['backbone']['ms_per_batch_240x320'] > 0
    # round 6: the backbone in both arithmetic types (fused split-bf16 blocks by default, the exact-fp32 per-layer kernels beside them)
    bbk = d['extra']['backbone']
    assert 0 < bbk['ms_per_batch'] < bbk['ms_per_batch_fp32_exact'] and bbk['launches_per_batch'] <= 30
    assert bbk['max_diff_vs_stock_modules_of_range'] < 1e-4 and bbk['max_diff_vs_stock_modules_of_range_fp32_exact'] < 2e-5
    assert any('block' in k for k in bbk['kernels'])
    assert 0 < d['extra']['from_images']['value_fp32_exact'] < d['extra']['from_images']['value']
    # the reference's arithmetic type beside every figure, and inside `config` (a record that keeps only the contract's keys)
    assert d['config']['value_fp32_exact'] == d['value_fp32_exact'] and d['config']['ms_per_step_fp32_exact'] > d['ms_per_step']
    oc = d['config']['other_configs']
    for k in ('cfg5', 'cfg3', 'cfg3_full'):
        assert oc[k]['value'] > 0 and 0 < oc[k]['value_fp32_exact'] < oc[k]['value'], k
    fr = e3['parity']['free_running']
    assert fr['stages_1_2_fp32_exact']['fraction_of_pixels_within_1e_4'] > 0.995 or \
        sum(fr['stages_1_2_fp32_exact']['points_in_different_cells_per_outer_iteration']) > 0
    assert e3['parity']['max_rel_depth_err_gpu_fp32_exact_vs_cpu'] < 2e-5


def test_bench_rank_plumbing_dry_run():
    """`bench.py --gpus 2 --dry-run` on CPU: the script spawns its own two ranks through torch.distributed.run (the command
    line the driver uses), they rendezvous on 127.0.0.1 (gloo), only rank 0 prints, ONE JSON line, both ranks counted, and
    the reported time is the slower rank's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.rank_command(8, 29511, ['--gpus', '8', '--steps', '5'])
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29511' and cmd[-4:] == ['--gpus', '8', '--steps', '5']
    assert os.path.basename(cmd[cmd.index('29511') + 1]) == 'bench.py'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '1',
                        '--dry-run'], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['dry_run'] and d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['steps'] == 20
    assert d['ms_per_step'] >= 2.0          # rank 1 sleeps 2 ms per step, rank 0 1 ms: max over ranks
    # under an external launcher (the driver's form) the script must not spawn again
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    r = subprocess.run(bench.rank_command(2, port, ['--gpus', '2', '--steps', '3', '--warmup', '0', '--dry-run']),
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['config']['ranks_seen'] == 2


def test_bench_checker_processes_return_the_in_process_oracle(monkeypatch):
    """bench._oracle_views: the oracle of the parity legs computed by spawned worker processes (inputs rebuilt from the seed)
    returns the bits of the in-process loop at the same team size -- the checker's numbers must not depend on how the bench
    schedules it."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from oracle import costvolume as ocv
    syn = v3d('synthetic')
    inp = syn.make_costvolume_inputs('cfg1', n_ref=4, seed=3)
    sd = syn.costregnet_weights(seed=0, sharpen=200.0)
    d0, dd, D = inp['depth']
    per = inp['edges'].shape[1] // 4

    def run(v0, v1, pinned):
        with torch.no_grad():
            return ocv.mvsnet_depth(inp['feat'], inp['rotmats'], inp['tvecs'], inp['K'], inp['edges'][:, v0 * per:v1 * per],
                                    sd, d0, dd, D, inp['img_size'], inp['plane_size'], pinned=pinned)[0]
    nt0 = torch.get_num_threads()
    monkeypatch.setattr(os, 'cpu_count', lambda: 8)         # 8 // (2 * 2) = 2 worker processes of 2 threads each
    try:
        pinned, plain = bench._oracle_views('cfg1', 4, 3, [(4, True), (3, False)], 1, 2)
        torch.set_num_threads(2)                            # (MKL-DNN's blocking follows the team size: 8e-7 between 2 and 8)
        assert torch.equal(pinned, torch.cat([run(v, v + 1, True) for v in range(4)]))
        assert torch.equal(plain, torch.cat([run(v, v + 1, False) for v in range(3)]))
    finally:
        torch.set_num_threads(nt0)


def test_bench_cfg4_dry_run_shards_a_scene_through_the_real_driver():
    """`bench.py --config cfg4 --gpus 2 --dry-run`: the communicating mode's host path on CPU -- the real process_scene
    (ref-view sharding 3 + 2, chunking, one-sided halos, gather_pointcloud with uneven shards, the final all-gather of the
    depths) over gloo with a bookkeeping stand-in for the net; every rank sees the whole cloud each outer iteration and the
    gathered depths equal the closed form."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'cfg4', '--gpus', '2', '--dry-run',
                        '--refs', '5', '--scene-window', '1,2', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    chk = d['config']['cfg4_scene_check']
    assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2
    assert chk['shard_views'] == [3, 2] and chk['gathered_rows_per_outer_iteration'] == 5 * 16
    assert chk['depths_equal_closed_form'] is True


def test_pred_func_adapter_has_the_reference_entry_shape():
    """mv3d/eval-3dvnet.py:26,129 / mv3d/eval/main.py:59: ``depth_preds, init_prob, final_prob = pred_func(batch, scene, dset,
    net)`` -- a host numpy array of full-resolution depth maps (stage 3 included) and two Nones.  Oracle-backed net on the CPU:
    the adapter's result equals stages 1-2 of the driver followed by the oracle's stage-3 chain."""
    import types
    import torch.nn.functional as F
    from oracle import scene as osc
    syn, drv = v3d('synthetic'), v3d('eval_3dvnet')
    scene = make_scene()
    n_img = scene.rotmats.shape[0]
    scene.features_half = syn.make_features(n_img, 32, 2 * FEAT[0], 2 * FEAT[1], seed=7)
    scene.images = torch.rand((n_img, 3) + IMG, generator=torch.Generator().manual_seed(8))
    sds = [syn.propagation_weights(33, 32, 5), syn.propagation_weights(33, 32, 6), syn.propagation_weights(4, 32, 7)]

    class Net(OracleNet):
        hparams = types.SimpleNamespace(depth_test=CFG)

        def make_initial_depth_predictions(self, batch, cfg):
            d, b, _, fq, _, ref_idx = super().make_initial_depth_predictions(batch, cfg)
            return d, b, batch.features_half, fq, None, ref_idx

    net = Net(*weights(), IMG, 0.16)
    for name, sd in zip(('refine_quarter', 'refine_half', 'refine_full'), sds):
        setattr(net, name, lambda f, d, sd=sd: osc.propagation_net(f, d, sd))
    dset = types.SimpleNamespace(n_src_on_either_side=1)
    out = drv.pred_func(scene, '/data/scene0000_00', dset, net)
    assert isinstance(out, tuple) and len(out) == 3 and out[1] is None and out[2] is None
    depth = out[0]
    assert isinstance(depth, np.ndarray) and depth.dtype == np.float32 and depth.shape == (5,) + IMG
    ref = drv.process_scene(scene, net, 1, torch.device('cpu'), CFG, drv.OFFSETS_LIST)
    for sd, guide in zip(sds, (scene.features_quarter[1:6], scene.features_half[1:6], scene.images[1:6])):
        ref = osc.propagation_net(guide, F.interpolate(ref.unsqueeze(1), guide.shape[-2:], mode='nearest'), sd)
    np.testing.assert_allclose(depth, ref.numpy(), rtol=1e-5, atol=0)


def test_batch_sizes_default_to_the_reference_and_auto_on_cpu():
    """eval-3dvnet.py:12-13: INIT_DEPTH_BATCH = 18, OFFSET_BATCH = 16 are the module defaults; `auto_batches` only raises them
    on a HIP device with enough free memory."""
    drv = v3d('eval_3dvnet')
    assert (drv.INIT_DEPTH_BATCH, drv.OFFSET_BATCH) == (18, 16)
    assert drv.auto_batches(torch.device('cpu')) == (18, 16)
