"""Host-side mirror of the inference methods of ``mv3d/lightningmodel.py`` (SURVEY.md §8a rows A7, B2,
B5, C1-C3; §8b): ``PL3DVNet`` with the reference's constructor arguments, sub-module names and method
signatures / return tuples.  Training-only members (losses, Lightning hooks, optimiser) are out of
scope -- the benchmark path runs under ``torch.no_grad()`` (mv3d/eval-3dvnet.py:27).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, utils
from .mvsnet import MVSNet, _Workspace, edges_to_csr
from .refinement import HypothesisDecoder
from .scenemodeling import PointNet, SparseUNet
from .upsampling import PropagationNet


def backproject_variance(depth_pred, img_feats, rotmats, tvecs, K, ref_src_edges, img_size, offset=0.0,
                         n=0, workspace=None, csr=None):
    """Rows B1-B2 / C1: -> (pts [n_ref*P, 2n+1, 3], var [n_ref*P, 2n+1, C])."""
    if not depth_pred.is_cuda:
        raise _lib.V3DLibraryError('backproject_variance: tensors must live on a HIP device (no CPU fallback)')
    lib = _lib.load()
    dev = depth_pred.device
    feat = img_feats.contiguous().float()
    n_img, C, Hf, Wf = feat.shape
    _, ref_img, edge_ofs, edge_src = csr if csr is not None else edges_to_csr(ref_src_edges.to(dev))
    n_ref, h, w = depth_pred.shape
    assert n_ref == ref_img.shape[0], 'one depth map per reference view'
    n_hyp = 2 * n + 1
    pts = torch.empty((n_ref * h * w, n_hyp, 3), dtype=torch.float32, device=dev)
    var = torch.empty((n_ref * h * w, n_hyp, C), dtype=torch.float32, device=dev)
    nbytes = lib.v3d_backproject_workspace_bytes(n_img, C, Hf, Wf)
    ws = (workspace or _Workspace()).get('bp', nbytes, dev)
    # the library keeps a channel-last copy of the features in the workspace: a caller that passes its workspace and the same
    # (unmodified) feature tensor again -- the scene driver, eight times per scene -- skips that copy.  The tag holds the
    # previous tensor alive, so its address cannot be handed to another tensor in between.
    tag = None
    if workspace is not None and not feat.is_inference():
        tag = (feat.data_ptr(), feat._version, tuple(feat.shape), tuple(feat.stride()), ws.data_ptr(), str(dev), _lib.stream_ptr(dev))
    prev = workspace.tags.get('bp') if workspace is not None else None
    reuse = tag is not None and prev is not None and prev[0] == tag
    Kc, Rc, tc = (x.to(dev).contiguous().float() for x in (K, rotmats, tvecs))
    d = depth_pred.contiguous().float()
    if workspace is not None:
        workspace.tags['bp'] = None              # (an exception below leaves no claim on the workspace's contents)
    rc = lib.v3d_backproject_variance_f32(d.data_ptr(), None if reuse else feat.data_ptr(), Kc.data_ptr(), Rc.data_ptr(),
                                          tc.data_ptr(), ref_img.data_ptr(), edge_ofs.data_ptr(),
                                          edge_src.data_ptr(), n_img, n_ref, edge_src.shape[0], C, Hf, Wf,
                                          int(img_size[0]), int(img_size[1]), h, w, float(offset), int(n),
                                          pts.data_ptr(), var.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr(dev))
    _lib.check(rc, 'v3d_backproject_variance_f32')
    if workspace is not None and tag is not None:
        workspace.tags['bp'] = (tag, feat)
    return pts, var


class PL3DVNet(nn.Module):
    """Reference ``PL3DVNet`` (lightningmodel.py:14-43), inference surface only."""

    def __init__(self, depth_train, depth_test, edge_len, feat_dim=16, img_size=(256, 320), hyp_ksize=3,
                 hyp_pad=1, lr=1e-3, lr_step=100, lr_gamma=0.1, finetune=False, feat_extractor=None,
                 feat_shrinker=None, precision='split_bf16', backbone=False):
        """Arguments up to ``finetune`` are the reference's (lightningmodel.py:18-20), defaults included: ``feat_dim`` defaults
        to 16 as in the reference's signature -- the same positional call builds the same network.  The reference's config
        (mv3d/config.py:42) and the hparams of every released checkpoint say 32, the width the fast kernels are specialised
        for (``load_from_checkpoint`` takes it from the checkpoint; bench.py and the tests pass it explicitly).  ``feat_dim=16``
        runs on the general entry points: the cost volume through the reference-layout entry points (conv0 on the volume
        zero-extended to 32 channels), GroupNorm over 8-channel groups on
        the first U-Net level, the unfused hypothesis decoder.  Extra keywords:
        ``feat_extractor`` / ``feat_shrinker`` inject the 2D backbone, ``backbone=True`` builds the MnasNet-1.0 + FPN
        one of the reference (``backbone.py``; random-init, there are no pretrained weights offline); ``precision``
        ('split_bf16' | 'fp32') selects the MFMA operand precision of the matrix-core kernels of all three stages (include/v3d.h;
        round 6: the PropagationNets of stage 3, ``refine_*``, included), on a HIP device, in eval mode."""
        super().__init__()
        if feat_dim not in (16, 32):
            raise ValueError('PL3DVNet: feat_dim=%d is not supported by the HIP path (16 -- the reference\'s signature default -- '
                             'and 32, the value of mv3d/config.py:42 and of the released checkpoints)' % feat_dim)
        if backbone and feat_extractor is None:
            from .backbone import build_backbone
            feat_extractor, feat_shrinker = build_backbone(feat_dim)
        self.depth_train, self.depth_test, self.edge_len = depth_train, depth_test, edge_len
        self.feat_dim, self.img_size = feat_dim, img_size
        self.hparams = SimpleNamespace(depth_train=depth_train, depth_test=depth_test, edge_len=edge_len,
                                       feat_dim=feat_dim, img_size=img_size, hyp_ksize=hyp_ksize,
                                       hyp_pad=hyp_pad, lr=lr, lr_step=lr_step, lr_gamma=lr_gamma,
                                       finetune=finetune)
        self.precision = precision
        self.mvsnet = MVSNet(feat_dim, img_size, feat_extractor, feat_shrinker, precision=precision)
        self.pointnet = PointNet(4 * feat_dim, 2 * feat_dim, feat_dim + 3, precision=precision)
        self.sparse_conv = SparseUNet(dims=(2 * feat_dim, 128, 128), n_groups=(4, 8, 8), n_res=(1, 2, 3),
                                      precision=precision)
        self.decoder = HypothesisDecoder(128 + 128 + 3 * feat_dim, 128, hyp_ksize, hyp_pad, precision=precision)
        # stage-3 upsamplers (lightningmodel.py:41-43), SURVEY.md 8f rank 2: the library's row-marching kernel (csrc/propz.hip)
        self.refine_quarter = PropagationNet(in_dim=feat_dim + 1, h_dim=32, precision=precision)
        self.refine_half = PropagationNet(in_dim=feat_dim + 1, h_dim=32, precision=precision)
        self.refine_full = PropagationNet(in_dim=3 + 1, h_dim=32, precision=precision)
        self._ws = _Workspace()
        self._offset_vals = {}
        self._pts_batch = {}

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
        """Counterpart of ``PL3DVNet.load_from_checkpoint(path)`` (mv3d/eval-3dvnet.py:134; pytorch-lightning 1.1.2): a Lightning
        checkpoint is a ``torch.save``d dict with ``'state_dict'`` (keys ``mvsnet.* / pointnet.* / sparse_conv.* / decoder.* /
        refine_*.*``) and ``'hyper_parameters'`` (the constructor arguments saved by ``save_hyperparameters``,
        lightningmodel.py:33).  ``kwargs`` override / complete the hyper-parameters (and carry this class's extra keywords,
        e.g. ``precision``).  The 2D backbone is built when the checkpoint carries ``mvsnet.feat_extractor.*`` keys.
        ``strict``: as ``nn.Module.load_state_dict``, except that BatchNorm ``num_batches_tracked`` counters may be absent.
        Sparse-convolution kernels are taken as MinkowskiEngine stores them (``.kernel`` [27, Cin, Cout], offset index with the
        first spatial dimension fastest, SURVEY.md Appendix A -- parity with a real ME checkpoint is unpinned)."""
        ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
        sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
        hp = dict(ckpt.get('hyper_parameters', {})) if isinstance(ckpt, dict) else {}
        hp.update(kwargs)
        for name in ('depth_train', 'depth_test', 'edge_len'):
            if name not in hp:
                raise KeyError('load_from_checkpoint: hyper-parameter %r is neither in the checkpoint nor given' % name)
        if any(k.startswith('mvsnet.feat_extractor.') for k in sd) and 'feat_extractor' not in hp:
            hp.setdefault('backbone', True)
        net = cls(**hp)
        res = net.load_state_dict(sd, strict=False)
        missing = [k for k in res.missing_keys if not k.endswith('num_batches_tracked')]
        if strict and (missing or res.unexpected_keys):
            raise RuntimeError('load_from_checkpoint: missing keys %s, unexpected keys %s' % (missing, list(res.unexpected_keys)))
        return net.eval()

    def make_initial_depth_predictions(self, batch, depth_config):
        """lightningmodel.py:124-130."""
        # `batch.n_ref` (optional, set by the scene driver, which knows how many reference views a chunk holds): the edge
        # tables are then built on the device and torch.unique -- a host synchronisation -- is not needed twice per chunk
        n_ref = getattr(batch, 'n_ref', None)
        depth_pred, feats_half, feats_quarter, features_eighth = self.mvsnet(
            batch, depth_config['depth_start'], depth_config['depth_interval'], depth_config['n_intervals'],
            depth_config['size'], n_ref=n_ref)
        if n_ref is not None and self.mvsnet.last_csr is not None:
            ref_idx = self.mvsnet.last_csr[0]                    # ascending distinct reference images = torch.unique's output
        else:
            ref_idx = torch.unique(batch.ref_src_edges[0])
        depth_batch = batch.images_batch.to(ref_idx.device)[ref_idx]
        return depth_pred, depth_batch, feats_half, feats_quarter, features_eighth, ref_idx

    def construct_feature_rich_pointcloud(self, depth_pred, depth_batch, img_feats, rotmats, tvecs, K,
                                          ref_src_edges, csr=None):
        """lightningmodel.py:132-174 -> (pts [Np,3], pts_feat [Np,C], pts_batch [Np]).  ``csr``: as in run_pointflow."""
        n_imgs = depth_pred.shape[0]
        pts, var = backproject_variance(depth_pred, img_feats, rotmats, tvecs, K, ref_src_edges,
                                        self.hparams.img_size, workspace=self._ws, csr=csr)
        pts_batch = depth_batch.unsqueeze(1).expand(n_imgs, depth_pred.shape[1] * depth_pred.shape[2]).reshape(-1)
        return pts.view(-1, 3), var.view(-1, var.shape[-1]), pts_batch

    def model_scene(self, depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges,
                    return_pts=False, gather_fn=None, csr=None, n_batches=None, defer_checks=False):
        """lightningmodel.py:176-185.  ``gather_fn`` (multi-GPU): all-gathers the local
        feature-rich point cloud in view order before the replicated voxelise / PointNet / U-Net.  ``csr``: the edge tables
        of ``ref_src_edges`` when the caller already holds them; ``n_batches``: the number of batch elements (scenes) when
        the caller knows it -- otherwise the U-Net reads ``max(batch) + 1`` back from the device (scenemodeling.py:221);
        ``defer_checks``: the caller will call ``self.sparse_conv.flush_checks()`` before it uses the results."""
        pts, pts_feat, pts_batch = self.construct_feature_rich_pointcloud(depth_pred, depth_batch, img_feats,
                                                                          rotmats, tvecs, K, ref_src_edges, csr=csr)
        if gather_fn is not None:
            pts, pts_feat, pts_batch = gather_fn(pts, pts_feat, pts_batch)
        anchor_pts, anchor_idx3d, anchor_batch, anchor_pts_edges = utils.voxelize(pts, pts_batch, self.edge_len)
        n_anchors = anchor_pts.shape[0]
        # x = cat(pts[e1] - anchor_pts[e0], pts_feat[e1]) (lightningmodel.py:180-183) in one kernel instead of five launches
        e0 = anchor_pts_edges[0].to(torch.int64).contiguous()
        e1 = anchor_pts_edges[1].to(torch.int64).contiguous()
        pts_c, anc_c, feat_c = pts.contiguous().float(), anchor_pts.contiguous().float(), pts_feat.contiguous().float()
        x = torch.empty((e0.shape[0], 3 + feat_c.shape[1]), dtype=torch.float32, device=pts.device)
        _lib.check(_lib.load().v3d_pointnet_input_f32(_lib.ptr(pts_c), _lib.ptr(anc_c), _lib.ptr(feat_c), _lib.ptr(e0), _lib.ptr(e1),
                                                      e0.shape[0], feat_c.shape[1], _lib.ptr(x), _lib.stream_ptr(pts.device)),
                   'v3d_pointnet_input_f32')
        x = self.pointnet(x, anchor_pts_edges[0], n_anchors)
        # (utils.voxelize shifts every batch element's minimum index to 0, utils.py:61-62: the U-Net needs no reduction for it)
        xs = self.sparse_conv(x, anchor_pts, anchor_idx3d, anchor_batch, self.edge_len, n_batches=n_batches,
                              defer_checks=defer_checks, idx_min_zero=True)
        return (xs, pts) if return_pts else xs

    def run_pointflow(self, xs, depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges,
                      offset, n, csr=None, add_to_depth=False):
        """lightningmodel.py:187-242 -> offset prediction [n_ref, h, w].  ``csr`` (optional) is the result of
        ``mvsnet.edges_to_csr(ref_src_edges)`` when the caller sweeps the same edge list repeatedly.  ``add_to_depth``: the
        caller's ``depth_pred += offset`` (eval-3dvnet.py:99) happens inside the decoder kernel (``depth_pred`` must be a
        contiguous fp32 tensor; it is modified in place and the offset is still returned)."""
        n_imgs = depth_pred.shape[0]
        n_pts = depth_pred.shape[1] * depth_pred.shape[2]
        pts_hyp, pts_feat = backproject_variance(depth_pred, img_feats, rotmats, tvecs, K, ref_src_edges,
                                                 self.hparams.img_size, offset=offset, n=n,
                                                 workspace=self._ws, csr=csr)
        # one batch index per point (lightningmodel.py:193-194); the scene driver sweeps the same chunks again and again: the
        # expanded tensor of the last few (storage, version) pairs is kept instead of being rebuilt per call
        # (inference-mode tensors carry no version counter: they are never cached)
        cacheable = not depth_batch.is_inference()
        pkey = (depth_batch.data_ptr(), depth_batch._version if cacheable else -1, tuple(depth_batch.stride()), str(depth_batch.dtype),
                n_imgs, n_pts, str(depth_batch.device))
        pts_batch = self._pts_batch.get(pkey) if cacheable else None
        if pts_batch is None:
            if len(self._pts_batch) >= 16:
                self._pts_batch.clear()
            pts_batch = depth_batch.unsqueeze(1).expand(n_imgs, n_pts).reshape(-1)
            if cacheable:
                self._pts_batch[pkey] = (pts_batch, depth_batch)   # (the source is kept alive: its address is the key)
        else:
            pts_batch = pts_batch[0]
        key = (float(offset), int(n), str(depth_pred.device))
        if key not in self._offset_vals:      # torch.linspace on the CPU then moved, like lightningmodel.py:238-240
            self._offset_vals[key] = torch.linspace(-n * offset, n * offset, 2 * n + 1).to(depth_pred.device)
        offset_vals = self._offset_vals[key]
        if self.decoder.can_fuse(xs, pts_hyp, pts_feat):
            fuse_add = add_to_depth and depth_pred.is_contiguous() and depth_pred.dtype == torch.float32
            _, expect = self.decoder.decode_fused(xs, pts_hyp, pts_feat, pts_batch, offset_vals,
                                                  depth_inout=depth_pred if fuse_add else None)
            if add_to_depth and not fuse_add:
                depth_pred.add_(expect.view(n_imgs, *depth_pred.shape[1:]))
        else:
            feats = self.decoder.features(xs, pts_hyp, pts_feat, pts_batch)
            _, expect = self.decoder.decode(feats, offset_vals)
            if add_to_depth:
                depth_pred.add_(expect.view(n_imgs, *depth_pred.shape[1:]))
        return expect.view(n_imgs, *depth_pred.shape[1:])
