"""Host-side mirror of ``mv3d/subnetworks/refinement.py``:
``HypothesisDecoder`` with the reference's constructor, ``forward`` signature and ``state_dict`` keys
(``net.{0,1,2}.{0.weight,1.*}``, ``net.3.{weight,bias}``).  ``MinkowskiInterpolation`` becomes the
hash-indexed trilinear kernel, the Conv1d+BN+ReLU stack a 3-segment gather-GEMM on the matrix cores (fp32
storage / accumulation, MFMA operands per ``precision``: 'split_bf16' default, 'fp32' exact), the last
conv + softmax a per-point wave kernel.  No CPU fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .mvsnet import _Workspace
from .scenemodeling import PackedGemm, _PackCache


def conv1d_bn_relu(in_channels, out_channels, kernel_size=3, stride=1, padding=1):
    """Parameter container with the reference's layout (refinement.py:8-13)."""
    return nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, bias=False),
                         nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True))


class HypothesisDecoder(nn.Module):
    """``forward(xs, pts[Nq,n_hyp,3], pts_feat[Nq,n_hyp,C]|None, pts_batch[Nq]) -> preds[Nq,n_hyp]``
    (refinement.py:28-44)."""

    def __init__(self, in_dim=128 + 128 + 64, h_dim=256, kernel_size=3, padding=1, precision='split_bf16'):
        super().__init__()
        _lib.precision_code(precision)
        self.precision = precision
        # True (default): interpolation + the three conv1d layers + head in ONE kernel (v3d_decoder_fused_f32) whenever the
        # configuration allows (can_fuse): nothing of size [Nq, in_dim, n_hyp] reaches HBM; 3.1 ms against 3.7 ms per
        # 64-view sweep at cfg3.  False: the 5-launch chain (sparse_interp + 3 x conv1d GEMM + head), also the fallback.
        self.fused = True
        assert kernel_size == 3 and padding == 1, 'the reference instantiates k=3, pad=1 (lightningmodel.py:39-40)'
        self.in_dim, self.h_dim = in_dim, h_dim
        self.net = nn.Sequential(conv1d_bn_relu(in_dim, h_dim), conv1d_bn_relu(h_dim, h_dim),
                                 conv1d_bn_relu(h_dim, h_dim), nn.Conv1d(h_dim, 1, kernel_size, 1, padding))
        self._cache = _PackCache(self)
        self._ws = _Workspace()

    def _build(self):
        packs = []
        for i in range(3):
            conv, bn = self.net[i][0], self.net[i][1]
            co, ci, _ = conv.weight.shape
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)          # eval-mode BatchNorm1d folded
            bias = bn.bias - bn.running_mean * scale
            packs.append(PackedGemm(conv.weight, 1, 3 * ci, 3, 3, co, ci, scale=scale, bias=bias, device=self._dev))
        dev = self._dev
        return packs, self.net[3].weight.detach().float().contiguous().to(dev), \
            self.net[3].bias.detach().float().contiguous().to(dev)

    def features(self, xs, pts, pts_feat, pts_batch):
        """Row C2a: [Nq, n_hyp, in_dim] = [finest level | ... | coarsest level | pts_feat]."""
        lib = _lib.load()
        dev = pts.device
        n_pts, n_hyp = pts.shape[:2]
        pts = pts.contiguous().float()
        pts_batch = pts_batch.contiguous().long()
        widths = [x['feats'].shape[1] for x in xs]
        cf = 0 if pts_feat is None else pts_feat.shape[2]
        total = sum(widths) + cf
        assert total == self.in_dim, (total, self.in_dim)
        feats = torch.empty((n_pts, n_hyp, total), dtype=torch.float32, device=dev)
        if pts_feat is not None:
            feats[:, :, total - cf:] = pts_feat
        col = total - cf
        wbytes = lib.v3d_sparse_interp_workspace_bytes(n_pts, n_hyp)
        ws = self._ws.get('interp', wbytes, dev)
        for x in xs:                                   # coarse -> fine, each level prepended (:41)
            lv = x['sparse']
            col -= x['feats'].shape[1]
            if '_min_pts' not in x:
                # scatter(x.pts, x.batch, reduce='min') (:33) as one reduction per batch element; cached
                # on the level dict (the levels are reused by every point-flow sweep of the scene)
                nb = int(x['batch'].max().item()) + 1
                x['_min_pts'] = torch.stack([x['pts'][x['batch'] == b].amin(dim=0) for b in range(nb)]) \
                    if nb > 1 else x['pts'].amin(dim=0, keepdim=True)
            min_pts = x['_min_pts'].contiguous()
            f = x['feats'].contiguous()
            rc = lib.v3d_sparse_interp_f32(lv.table.data_ptr(), lv.n, f.data_ptr(), f.shape[1],
                                           int(x['stride']), pts.data_ptr(), pts_batch.data_ptr(), n_pts,
                                           n_hyp, min_pts.data_ptr(), float(x['res']), feats.data_ptr(),
                                           total, col, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(rc, 'v3d_sparse_interp_f32')
        return feats

    def decode(self, feats, offset_vals=None):
        """Row C2b (+C3 when offset_vals is given): feats [Nq, n_hyp, in_dim] -> preds [Nq, n_hyp]
        (and the expected offset [Nq])."""
        assert not self.training, 'inference only: BatchNorm is folded with running statistics'
        lib = _lib.load()
        self._dev = feats.device
        packs, w_last, b_last = self._cache.get(self._build, feats.device)
        n_pts, n_hyp, _ = feats.shape
        x = feats.view(n_pts * n_hyp, -1)
        for pk in packs:
            x = pk(n_pts * n_hyp, [x, x, x], group_len=n_hyp, relu_out=True, precision=self.precision)
        preds = torch.empty((n_pts, n_hyp), dtype=torch.float32, device=feats.device)
        expect = None
        if offset_vals is not None:
            offset_vals = offset_vals.to(feats.device).float().contiguous()
            expect = torch.empty(n_pts, dtype=torch.float32, device=feats.device)
        rc = lib.v3d_decoder_head_f32(x.data_ptr(), n_pts, n_hyp, x.shape[1], w_last.data_ptr(),
                                      b_last.data_ptr(), _lib.ptr(offset_vals), preds.data_ptr(),
                                      _lib.ptr(expect), _lib.stream_ptr(feats.device))
        _lib.check(rc, 'v3d_decoder_head_f32')
        return preds if expect is None else (preds, expect)

    def can_fuse(self, xs, pts, pts_feat):
        """The fused kernel (v3d_decoder_fused_f32) covers the reference's configuration: three levels, 128 hidden
        channels, level / point-feature widths that are multiples of 32, at most 8 hypotheses, split-bf16 operands."""
        cf = 0 if pts_feat is None else pts_feat.shape[2]
        # (the kernel's own limits, mirrored so that an oversized call takes the unfused chain instead of raising: level rows
        # < 2^24 and < 2 GB of features per level -- 24-bit row indices / 32-bit byte offsets in the corner table; the number
        # of points per call is not a limit: decode_fused splits calls of 2^24 (point, hypothesis) columns or more)
        return (self.fused and self.precision == 'split_bf16' and self.h_dim == 128 and len(xs) == 3 and pts.shape[1] <= 8
                and cf % 16 == 0 and all(x['feats'].shape[1] % 16 == 0 for x in xs)
                and sum(x['feats'].shape[1] for x in xs) + cf == self.in_dim
                and all(x['feats'].shape[0] < (1 << 24) and x['feats'].numel() * 4 < (1 << 31) for x in xs))

    def decode_fused(self, xs, pts, pts_feat, pts_batch, offset_vals=None, depth_inout=None):
        """Rows C2a + C2b (+ C3) in one kernel: interpolation, the three conv1d layers, head and softmax
        (and the expected offset when offset_vals is given); nothing of size [Nq, in_dim, n_hyp] is materialised.
        ``depth_inout`` (contiguous fp32 [Nq], optional): the expected offset is also added to it in place."""
        assert not self.training, 'inference only: BatchNorm is folded with running statistics'
        lib = _lib.load()
        dev = pts.device
        self._dev = dev
        packs, w_last, b_last = self._cache.get(self._build, dev)
        n_pts, n_hyp = pts.shape[:2]
        pts = pts.contiguous().float()
        pts_batch = pts_batch.contiguous().long()
        cf = 0
        if pts_feat is not None:
            pts_feat = pts_feat.contiguous().float()
            cf = pts_feat.shape[2]
        max_pts = ((1 << 24) - 1) // max(n_hyp, 1)          # v3d_decoder_fused_f32: n_pts * n_hyp < 2^24 per call
        if n_pts > max_pts:
            # points are independent: pieces of the (contiguous) arguments, the same bits as one call would give
            outs = [self.decode_fused(xs, pts[s:s + max_pts], None if pts_feat is None else pts_feat[s:s + max_pts],
                                      pts_batch[s:s + max_pts], offset_vals,
                                      None if depth_inout is None else depth_inout.view(-1)[s:s + max_pts])
                    for s in range(0, n_pts, max_pts)]
            if offset_vals is None:
                return torch.cat(outs, dim=0)
            return torch.cat([o[0] for o in outs], dim=0), torch.cat([o[1] for o in outs], dim=0)
        keep = []
        levels = list(xs)[::-1]                      # xs is coarse -> fine; feature rows are finest first (:41)
        for x in levels:
            if '_min_pts' not in x:
                nb = int(x['batch'].max().item()) + 1
                x['_min_pts'] = torch.stack([x['pts'][x['batch'] == b].amin(dim=0) for b in range(nb)]) \
                    if nb > 1 else x['pts'].amin(dim=0, keepdim=True)
            keep.append((x['sparse'].table, x['feats'].contiguous(), x['_min_pts'].contiguous()))
        vp = ctypes.c_void_p
        layers = (vp * 3)(*[pk.handle for pk in packs])
        tables = (vp * 3)(*[k[0].data_ptr() for k in keep])
        n_in = (ctypes.c_int * 3)(*[x['sparse'].n for x in levels])
        feats = (vp * 3)(*[k[1].data_ptr() for k in keep])
        chans = (ctypes.c_int * 3)(*[k[1].shape[1] for k in keep])
        strides = (ctypes.c_int * 3)(*[int(x['stride']) for x in levels])
        mins = (vp * 3)(*[k[2].data_ptr() for k in keep])
        res = (ctypes.c_float * 3)(*[float(x['res']) for x in levels])
        preds = torch.empty((n_pts, n_hyp), dtype=torch.float32, device=dev)
        expect = None
        if offset_vals is not None:
            offset_vals = offset_vals.to(dev).float().contiguous()
            expect = torch.empty(n_pts, dtype=torch.float32, device=dev)
        ws = self._ws.get('fused', lib.v3d_decoder_fused_workspace_bytes(n_pts, n_hyp), dev)
        rc = lib.v3d_decoder_fused_f32(layers, w_last.data_ptr(), b_last.data_ptr(), tables, n_in, feats, chans, strides,
                                       mins, res, pts.data_ptr(), pts_batch.data_ptr(), _lib.ptr(pts_feat), cf, n_pts,
                                       n_hyp, _lib.ptr(offset_vals), preds.data_ptr(), _lib.ptr(expect), _lib.ptr(depth_inout),
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, 'v3d_decoder_fused_f32')
        return preds if expect is None else (preds, expect)

    def forward(self, xs, pts, pts_feat, pts_batch):
        if not pts.is_cuda:
            raise _lib.V3DLibraryError('HypothesisDecoder: tensors must live on a HIP device (no CPU fallback)')
        if self.can_fuse(xs, pts, pts_feat):
            return self.decode_fused(xs, pts, pts_feat, pts_batch)
        return self.decode(self.features(xs, pts, pts_feat, pts_batch))
