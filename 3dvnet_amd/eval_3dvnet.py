"""Per-scene inference driver: counterpart of ``mv3d/eval-3dvnet.py::process_scene`` (:26-129), SURVEY.md
§8a row H2, with the multi-GPU partitioning of §8e added.

Single process: identical control flow to the reference -- chunked initial depth with a +-k image
halo (:41-63), ``len(offsets_list)`` outer iterations of ``model_scene`` followed by chunked
``run_pointflow`` sweeps with in-place ``+=`` (:73-99), and optionally stage 3, the three nearest +
PropagationNet upsampling steps to full resolution (:101-125; a "next" row, stock 2D convolutions).

Multi-GPU (one process per GPU, ``torch.distributed``): reference views are independent units for
the cost volume and the point-flow sweeps, so rank g owns a contiguous block of reference views and
the images within its halo; no data-path collective there.  The scene model needs every view's
points: each rank back-projects its own views and the feature-rich point cloud is all-gathered in
view order (bit-identical to the single-process tensor), after which voxelise / PointNet / sparse
U-Net run replicated on every rank.
"""
import torch

from .batch import Batch
from . import utils
from .mvsnet import edges_to_csr

# Reference views per call.  The reference's values (eval-3dvnet.py:12-14: INIT_DEPTH_BATCH = 18, OFFSET_BATCH = 16, "change these
# to scale eval script to your GPU") are the module defaults here too; results do not depend on the chunking
# (tests/test_driver.py).  `auto_batches(device)` returns what an MI355X-class device should use instead: both stages hold a
# whole 64-view scene at once (2.5 GB variance volume + regulariser workspace of 288 GB; 64 views per call is the batch the
# cost-volume kernels are tuned for, and one point-flow call per sweep fills the persistent decoder's tile walk evenly) when
# at least MI355X_BATCH_MIN_FREE bytes of device memory are free, else the reference's values.  `process_scene` takes
# `init_depth_batch=None / offset_batch=None` to mean "auto"; bench.py and `pred_func` use that.
INIT_DEPTH_BATCH = 18
OFFSET_BATCH = 16
MI355X_BATCH = 64
MI355X_BATCH_MIN_FREE = 24 << 30


def auto_batches(device):
    """(init_depth_batch, offset_batch) for `device`: 64 / 64 on a HIP device with >= 24 GB free, else the reference's 18 / 16."""
    device = torch.device(device)
    if device.type == 'cuda' and torch.cuda.is_available():
        free, _ = torch.cuda.mem_get_info(device)
        if free >= MI355X_BATCH_MIN_FREE:
            return MI355X_BATCH, MI355X_BATCH
    return INIT_DEPTH_BATCH, OFFSET_BATCH


DEPTH_CONFIG = {'depth_start': 0.5, 'depth_interval': 0.05, 'n_intervals': 96, 'size': (56, 56)}
OFFSETS_LIST = [[0.05, 0.05, 0.025], [0.05, 0.05, 0.025]]


def shard_range(n, rank, world):
    """Contiguous block partition of n reference views: [start, end) of `rank`."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_rows(x, sizes=None, group=None):
    """All-gather tensors that differ in dim 0, concatenated in rank order: ONE collective (all_gather_into_tensor on
    rows padded to the largest shard).  ``sizes`` = rows per rank; the scene driver knows them analytically
    (shard_range), so no size exchange and no host synchronisation is needed; when omitted they are exchanged first."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if sizes is None:
        n = torch.tensor([x.shape[0]], dtype=torch.long, device=x.device)
        got = torch.empty(world, dtype=torch.long, device=x.device)
        dist.all_gather_into_tensor(got, n, group=group)
        sizes = [int(v) for v in got.tolist()]
    assert len(sizes) == world and sizes[dist.get_rank(group)] == x.shape[0], (sizes, x.shape)
    m = max(sizes)
    if x.shape[0] == m:
        pad = x.contiguous()
    else:
        pad = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
    out = torch.empty((world * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if all(sz == m for sz in sizes):
        return out
    out = out.view((world, m) + tuple(x.shape[1:]))
    return torch.cat([out[r, :sz] for r, sz in enumerate(sizes)], dim=0)


def gather_pointcloud(pts, pts_feat, pts_batch, sizes=None, group=None):
    """The one exchange step of the path: the feature-rich point cloud [pts | feat | batch id] of
    every rank in ONE all-gather, rank order = view order, so the result equals the single-process tensor bit for
    bit (batch ids are small integers, exact in fp32)."""
    packed = all_gather_rows(torch.cat((pts, pts_feat, pts_batch.to(pts.dtype).unsqueeze(1)), dim=1), sizes, group)
    c = pts_feat.shape[1]
    return (packed[:, :3].contiguous(), packed[:, 3:3 + c].contiguous(),
            packed[:, 3 + c].to(pts_batch.dtype).contiguous())


def process_scene(batch, net, n_src_on_either_side, device, depth_config=None, offsets_list=None,
                  init_depth_batch=None, offset_batch=None, rank=0, world=1,
                  group=None, gather_depth=True, upsample=False, init_depth_override=None):
    """Returns the refined depth maps [n_ref, h, w] (all views when gather_depth, else this rank's).

    ``n_src_on_either_side``: the reference's k (eval/main.py:36; window ref-k .. ref+k, 2k+1 edges per reference view), or
    a pair ``(n_before, n_after)`` for the one-sided windows of SURVEY.md 8d (cfg2-4: ref-4 .. ref+3 = 1 ref + 7 src); the
    image halo of a chunk of reference views is n_before images in front of it and n_after behind it.
    ``batch``: images (or precomputed ``features_quarter`` [+ ``features_half`` for ``upsample``]), rotmats, tvecs, K,
    ref_src_edges for the whole scene with the reference's edge convention (dsets/dataset.py:133-137).
    ``init_depth_override`` (benchmark hook, [n_ref, h, w]): stage 1 still runs, then its depths are replaced by
    these (bench.py uses surface-like depths because random synthetic features give noise depths)."""
    depth_config = depth_config or DEPTH_CONFIG
    offsets_list = offsets_list or OFFSETS_LIST
    if init_depth_batch is None or offset_batch is None:       # "auto": by the device's free memory (auto_batches)
        auto_i, auto_o = auto_batches(device)
        init_depth_batch = auto_i if init_depth_batch is None else init_depth_batch
        offset_batch = auto_o if offset_batch is None else offset_batch
    if isinstance(n_src_on_either_side, (tuple, list)):
        k, ka = int(n_src_on_either_side[0]), int(n_src_on_either_side[1])
    else:
        k = ka = int(n_src_on_either_side)
    halo = k + ka          # images a chunk of reference views needs beyond its own (2k in the reference)
    with torch.no_grad():
        # the edge list is sliced per chunk with boolean masks: on a host copy (one transfer per scene when the batch already
        # lives on the device; masks on device tensors synchronise the host at every chunk)
        scene_edges = batch.ref_src_edges.cpu()
        ref_idx = torch.unique(scene_edges[0])
        n_ref_imgs = len(ref_idx)
        # The reference's layout (dsets/dataset.py:133-137): the reference views are images k .. k + n_ref - 1.  The chunking
        # below (like eval-3dvnet.py:42-52) addresses reference views by that rule, and a chunk [c0, c1) then holds exactly
        # c1 - c0 of them -- the count the device-side edge tables are built from (mvsnet.edges_to_csr) without a readback.
        if not torch.equal(ref_idx, torch.arange(k, k + n_ref_imgs, dtype=ref_idx.dtype)):
            raise ValueError('process_scene: the reference views of ref_src_edges must be images %d .. %d (dataset layout, '
                             'n_src_on_either_side = %d before); got %s' % (k, k + n_ref_imgs - 1, k, ref_idx.tolist()))
        # Every source view must lie inside its reference view's window [ref - k, ref + ka] and inside the scene: the chunk
        # slices below carry exactly that halo, and the device-side table builder (v3d_edges_csr) answers an index outside a
        # chunk with an EMPTY table -- a zero variance volume and a plausible-looking depth, not an error.  The list is on the
        # host already, so the check costs nothing.
        n_img_scene = int(batch.rotmats.shape[0])
        if scene_edges.shape[1]:
            delta = scene_edges[1] - scene_edges[0]
            if int(delta.min()) < -k or int(delta.max()) > ka or int(scene_edges[1].min()) < 0 \
                    or int(scene_edges[1].max()) >= n_img_scene:
                bad = ((delta < -k) | (delta > ka) | (scene_edges[1] < 0) | (scene_edges[1] >= n_img_scene)).nonzero()[0]
                raise ValueError('process_scene: edge %s -> %s lies outside the source window [ref - %d, ref + %d] / the %d '
                                 'images of the scene (n_src_on_either_side = %r)'
                                 % (int(scene_edges[0, bad[0]]), int(scene_edges[1, bad[0]]), k, ka, n_img_scene,
                                    n_src_on_either_side))
        r0, r1 = shard_range(n_ref_imgs, rank, world)
        n_local = r1 - r0
        has_feats = getattr(batch, 'features_quarter', None) is not None
        all_depth = torch.empty((n_local, *depth_config['size']), dtype=torch.float32, device=device)
        feats_local = None          # quarter features of images [r0, r1 + halo)
        half_local = None           # half-resolution features of the same images (stage 3 only, eval-3dvnet.py:36,62)
        # Precomputed features that already live on `device`: the slices ARE the tensors stage 2 / 3 need -- no per-chunk copy
        # into a second buffer (the reference fills all_feats_quarter / all_feats_half chunk by chunk because its features come
        # out of the backbone per chunk, eval-3dvnet.py:56-62)
        # (only when stage 1 takes the batch's features too: a net with its own backbone and a batch with images computes
        # features_quarter itself -- stages 2 / 3 must then see THOSE, as in the reference, not the batch's)
        own_backbone = getattr(getattr(net, 'mvsnet', None), 'feat_extractor', None) is not None and batch.images is not None
        dev_feats = (has_feats and not own_backbone and batch.features_quarter.device == torch.device(device)
                     and batch.features_quarter.dtype == torch.float32)
        half_is_view = False
        if dev_feats:
            feats_local = batch.features_quarter[r0:r1 + halo]
            fh = getattr(batch, 'features_half', None)
            if upsample and fh is not None and fh.device == torch.device(device) and fh.dtype == torch.float32:
                half_local, half_is_view = fh[r0:r1 + halo], True

        # ---- stage 1: initial depth, chunks of init_depth_batch reference views (:41-63) ----------
        for c0 in range(r0, r1, init_depth_batch):
            c1 = min(c0 + init_depth_batch, r1)
            ref_idx_start, ref_idx_end = c0 + k, c1 + k
            idx_start, idx_end = c0, c1 + halo
            edges = utils.slice_edges(scene_edges, ref_idx_start, ref_idx_end, 0) - idx_start
            sl = Batch(None if batch.images is None else batch.images[idx_start:idx_end],
                       batch.rotmats[idx_start:idx_end], batch.tvecs[idx_start:idx_end],
                       batch.K[idx_start:idx_end], None, edges)
            sl.images_batch = torch.zeros(idx_end - idx_start, dtype=torch.long)
            sl.n_ref = c1 - c0               # every image in [c0 + k, c1 + k) is a reference view of this chunk
            if has_feats:
                sl.features_quarter = batch.features_quarter[idx_start:idx_end]
                if getattr(batch, 'features_half', None) is not None:
                    sl.features_half = batch.features_half[idx_start:idx_end]
            sl.to(device)
            pred, _, feats_half, feats_quarter, _, _ = net.make_initial_depth_predictions(sl, depth_config)
            all_depth[c0 - r0:c1 - r0] = pred
            if not dev_feats:
                if feats_local is None:
                    feats_local = torch.empty((n_local + halo,) + tuple(feats_quarter.shape[1:]),
                                              dtype=torch.float32, device=device)
                feats_local[idx_start - r0:idx_end - r0] = feats_quarter
            if upsample and not half_is_view:
                # like all_feats_half of the reference (eval-3dvnet.py:36,62): kept from stage 1, whether the
                # features came from the injected backbone or were precomputed on the batch
                if feats_half is None:
                    raise ValueError('process_scene(upsample=True) needs half-resolution features: a backbone on '
                                     'net.mvsnet or batch.features_half')
                if half_local is None:
                    half_local = torch.empty((n_local + halo,) + tuple(feats_half.shape[1:]),
                                             dtype=torch.float32, device=device)
                half_local[idx_start - r0:idx_end - r0] = feats_half
        if init_depth_override is not None:
            all_depth = init_depth_override[r0:r1].to(device=device, dtype=torch.float32).clone()

        # ---- stage 2: volumetric refinement (:65-99) ------------------------------------------------
        # (contiguous once: the per-sweep calls below take leading-dimension slices, which then need no copy -- a rotation
        # stack that arrives as a transposed view was re-packed by every one of the 24 point-flow calls of a scene)
        rot = batch.rotmats[r0:r1 + halo].to(device).contiguous()
        tv = batch.tvecs[r0:r1 + halo].to(device).contiguous()
        K = batch.K[r0:r1 + halo].to(device).contiguous()
        edges_local_host = utils.slice_edges(scene_edges, r0 + k, r1 + k, 0) - r0
        edges_local = edges_local_host.to(device)
        depth_batch = torch.zeros(n_local, dtype=torch.long, device=device)
        n_pix = depth_config['size'][0] * depth_config['size'][1]
        shard_rows = [(shard_range(n_ref_imgs, g, world)[1] - shard_range(n_ref_imgs, g, world)[0]) for g in range(world)]
        gather_fn = (lambda p, f, b: gather_pointcloud(p, f, b, [n * n_pix for n in shard_rows], group)) \
            if world > 1 else None
        # the chunk edge lists (and their CSR form on the device) are the same for every sweep: build them once
        chunks = []
        for b0 in range(0, n_local, offset_batch):
            b1 = min(b0 + offset_batch, n_local)
            e = (utils.slice_edges(edges_local_host, b0 + k, b1 + k, 0) - b0).to(device)
            # every image in [b0 + k, b1 + k) of the chunk's slice is a reference view: the device kernel builds the tables
            chunks.append((b0, b1, e, edges_to_csr(e, n_ref=b1 - b0, n_img=b1 - b0 + halo) if e.is_cuda else None))
        scene_csr = edges_to_csr(edges_local, n_ref=n_local, n_img=n_local + halo) \
            if edges_local.is_cuda else None
        # what this driver knows and a bare ``net.model_scene`` would have to read back from the device: the edge tables, the
        # number of batch elements (one scene: depth_batch is all zeros); the hash-table range checks wait until the end.
        # (Only for nets that take these hints: the tests run this driver over a CPU net with the reference's plain signature.)
        unet = getattr(net, 'sparse_conv', None)
        hints = dict(csr=scene_csr, n_batches=1, defer_checks=True) if hasattr(unet, 'flush_checks') else {}
        for offsets in offsets_list:
            xs = net.model_scene(all_depth, depth_batch, feats_local, rot, tv, K, edges_local,
                                 gather_fn=gather_fn, **hints)
            for offset in offsets:
                for b0, b1, e, csr in chunks:
                    # `all_depth[b0:b1] += offset` (eval-3dvnet.py:99): inside the decoder kernel when the net offers it (the
                    # HIP net; the slice is a contiguous view), else here
                    if csr is not None:
                        net.run_pointflow(xs, all_depth[b0:b1], depth_batch[b0:b1], feats_local[b0:b1 + halo],
                                          rot[b0:b1 + halo], tv[b0:b1 + halo], K[b0:b1 + halo], e, offset, 3, csr=csr,
                                          add_to_depth=True)
                    else:
                        all_depth[b0:b1].add_(net.run_pointflow(xs, all_depth[b0:b1], depth_batch[b0:b1],
                                                                feats_local[b0:b1 + halo], rot[b0:b1 + halo],
                                                                tv[b0:b1 + halo], K[b0:b1 + halo], e, offset, 3))
        if hints:
            unet.flush_checks()             # hash-table range checks of both scene models: one wait here, not two in between
        if upsample:
            # ---- stage 3 (:101-125): plane grid -> 1/4 -> 1/2 -> full resolution, guided by the quarter /
            # half features and the image of each reference view (images k .. k + n_local of the halo'd slice)
            from .upsampling import upsample_depth
            imgs = batch.images[r0 + k:r1 + k].to(device)
            all_depth = upsample_depth(all_depth, [(net.refine_quarter, feats_local[k:k + n_local]),
                                                   (net.refine_half, half_local[k:k + n_local]),
                                                   (net.refine_full, imgs)])
        if world > 1 and gather_depth:
            all_depth = all_gather_rows(all_depth, shard_rows, group)
        # the back-projection's claim on its workspace (the channel-last copy of THIS scene's features, reused across the
        # sweeps) ends with the scene: it would keep the feature tensor alive until the next call
        ws = getattr(net, '_ws', None)
        if ws is not None and hasattr(ws, 'tags'):
            ws.tags.pop('bp', None)
        return all_depth


def pred_func(batch, scene, dset, net):
    """The reference's ``process_scene(batch, scene, dset, net)`` (mv3d/eval-3dvnet.py:26-129), i.e. the ``pred_func`` shape
    ``mv3d/eval/main.py:59`` calls: ``depth_preds, init_prob, final_prob = pred_func(batch, scene, dset, net)``.  Returns
    ``(all_depth.numpy() [n_ref, H, W], None, None)`` -- full-resolution depth maps after stage 3 (the reference always
    upsamples, :101-125), on the host, as ``eval/main.py:62-75`` consumes them (K rescale from ``depth_preds.shape[-2:]``,
    ``preds.npz``).  ``scene`` is unused here as in the reference; ``dset.n_src_on_either_side`` is the window (an int, or a
    ``(before, after)`` pair for this package's one-sided windows); the device is the net's.  Depth / offset configuration:
    ``net.hparams.depth_test`` when it carries the plane-sweep keys, else the script's DEPTH_CONFIG (eval-3dvnet.py:17-23)."""
    try:
        device = next(net.parameters()).device
    except (AttributeError, StopIteration):      # a parameter-free stand-in (the oracle-backed net of the tests)
        device = batch.rotmats.device
    cfg = getattr(getattr(net, 'hparams', None), 'depth_test', None)
    if not (isinstance(cfg, dict) and all(k in cfg for k in ('depth_start', 'depth_interval', 'n_intervals', 'size'))):
        cfg = DEPTH_CONFIG
    depth = process_scene(batch, net, dset.n_src_on_either_side, device, depth_config=cfg, offsets_list=OFFSETS_LIST,
                          upsample=True)
    return depth.detach().cpu().numpy(), None, None
