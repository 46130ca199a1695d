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
    out = drv.process_scene(make_scene(), net, 1, cuda, CFG, OFFSETS, 2, 3)
    ref = run_oracle()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=0)
