"""Depth upsampling by learned 3x3 propagation -- SURVEY.md §8f rank 2, the stage that follows the hot path in
``mv3d/eval-3dvnet.py:101-125``.  Host-side mirror of ``mv3d/subnetworks/upsampling.py::PropagationNet``: same
constructor, ``forward(features, depth)`` signature and ``state_dict`` keys (``conv{1..4}.{0.weight,1.*}``).

The arithmetic runs in the HIP library (``v3d_propagation_f32`` / ``v3d_propagation_up_f32``, csrc/propz.hip): ONE launch
per net marches down the image rows with one wave per layer -- the four 3x3 convolutions on matrix cores (split-bf16 or
exact fp32 operands) with eval-mode BatchNorm folded and ReLU in the epilogue, the activations between the layers in
LDS rings of four rows, then the 9-way softmax and the weighted sum over the replicate-padded 3x3 depth neighbourhood
in the last layer's epilogue (no unfold buffer, no intermediate in HBM).  No CPU fallback: tensors must live on a HIP
device.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .mvsnet import _Workspace, module_state_key


def _stage(c_in, c_out):
    """Parameter container with the reference's layout (upsampling.py:6-11)."""
    return nn.Sequential(nn.Conv2d(c_in, c_out, 3, 1, 1, bias=False), nn.BatchNorm2d(c_out), nn.ReLU(inplace=True))


class PropagationNet(nn.Module):
    """``forward(features[B,Cf,H,W], depth[B,1,H,W]) -> [B,H,W]``: each output depth is a convex combination (softmax
    over 9 logits predicted from features+depth) of its 3x3 neighbourhood (upsampling.py:23-36)."""

    def __init__(self, in_dim=4, h_dim=32, precision='split_bf16'):
        """``precision`` ('split_bf16' | 'fp32', an extra keyword of this package): the MFMA operand precision (include/v3d.h)."""
        super().__init__()
        _lib.precision_code(precision)
        self.precision = precision
        self.in_dim, self.h_dim = in_dim, h_dim
        widths = [in_dim, h_dim, h_dim, h_dim, 9]
        for i in range(4):
            setattr(self, 'conv%d' % (i + 1), _stage(widths[i], widths[i + 1]))
        self._handle, self._packed_key = None, None
        self._ws = _Workspace()

    def packed_handle(self, device):
        key = (str(device),) + module_state_key(self)
        if self._handle is not None and key == self._packed_key:
            return self._handle
        self.release()
        lib = _lib.load()
        keep = []

        def host(t):
            a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
            keep.append(a)
            return a.ctypes.data_as(_lib.c_float_p)

        def parray(ts):
            arr = (_lib.c_float_p * len(ts))(*[host(t) for t in ts])
            keep.append(arr)
            return arr
        stages = [getattr(self, 'conv%d' % i) for i in range(1, 5)]
        eps = {float(st[1].eps) for st in stages}
        assert len(eps) == 1
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):          # the library allocates the weight image on the current device
            rc = lib.v3d_propagation_pack(parray([st[0].weight for st in stages]), parray([st[1].weight for st in stages]),
                                          parray([st[1].bias for st in stages]), parray([st[1].running_mean for st in stages]),
                                          parray([st[1].running_var for st in stages]), self.in_dim, self.h_dim, eps.pop(),
                                          ctypes.byref(handle))
        _lib.check(rc, 'v3d_propagation_pack')
        self._handle, self._packed_key = handle, key
        return handle

    def release(self):
        if self._handle is not None:
            _lib.load().v3d_propagation_free(self._handle)
            self._handle = None

    def __getstate__(self):      # (as CostRegNet: the packed image is a cache; a copy re-packs)
        st = self.__dict__.copy()
        st['_handle'], st['_packed_key'] = None, None
        return st

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def forward(self, features, depth):
        if not (features.is_cuda and depth.is_cuda):
            raise _lib.V3DLibraryError('PropagationNet: tensors must live on a HIP device (no CPU fallback)')
        assert not self.training, 'inference only: BatchNorm is folded with running statistics'
        lib = _lib.load()
        dev = features.device
        features = features.contiguous().float()
        depth = depth.contiguous().float()
        B, Cf, H, W = features.shape
        assert depth.shape == (B, 1, H, W) and Cf + 1 == self.in_dim, (tuple(depth.shape), tuple(features.shape), self.in_dim)
        handle = self.packed_handle(dev)
        out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        # the conv kernels address a launch's image stack with 32-bit slot offsets (B * H * W < 2^24 slots, checked by the
        # library): larger stacks go through in pieces -- images are independent, the result is the same bit for bit
        step = max(1, ((1 << 24) - 1) // (H * W))
        for s in range(0, B, step):
            nb = min(step, B - s)
            ws = self._ws.get('prop', lib.v3d_propagation_workspace_bytes(handle, nb, H, W), dev)
            rc = lib.v3d_propagation_f32(handle, features[s:s + nb].data_ptr(), depth[s:s + nb].data_ptr(), nb, Cf, H, W,
                                         out[s:s + nb].data_ptr(), _lib.precision_code(self.precision), ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(dev))
            _lib.check(rc, 'v3d_propagation_f32')
        return out

    def forward_resized(self, features, depth_lo):
        """``forward(features, F.interpolate(depth_lo[:, None], features.shape[-2:], mode='nearest'))`` (eval-3dvnet.py:103-107)
        in one launch: depth_lo [B, h0, w0] -> [B, H, W]; the resize is two index tables in the kernel's addressing."""
        if not (features.is_cuda and depth_lo.is_cuda):
            raise _lib.V3DLibraryError('PropagationNet: tensors must live on a HIP device (no CPU fallback)')
        assert not self.training, 'inference only: BatchNorm is folded with running statistics'
        lib = _lib.load()
        dev = features.device
        features = features.contiguous().float()
        depth_lo = depth_lo.contiguous().float()
        B, Cf, H, W = features.shape
        h0, w0 = depth_lo.shape[-2:]
        assert depth_lo.shape == (B, h0, w0) and Cf + 1 == self.in_dim, (tuple(depth_lo.shape), tuple(features.shape), self.in_dim)
        iy, ix = nearest_tables((h0, w0), (H, W), dev)
        out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        rc = lib.v3d_propagation_up_f32(self.packed_handle(dev), features.data_ptr(), depth_lo.data_ptr(), B, Cf, H, W, h0, w0,
                                        iy.data_ptr(), ix.data_ptr(), out.data_ptr(), _lib.precision_code(self.precision),
                                        _lib.stream_ptr(dev))
        _lib.check(rc, 'v3d_propagation_up_f32')
        return out


_NEAREST_TABLES = {}


def nearest_tables(src_size, dst_size, device):
    """Source row / column of every output row / column of ``F.interpolate(x, dst_size, mode='nearest')`` for an input of
    ``src_size`` -- taken from torch's own rule (an interpolated ``arange``), so the kernel-side resize is the framework's bit
    for bit.  int32 tensors [H], [W] on ``device``, cached."""
    key = (tuple(src_size), tuple(dst_size), str(device))
    if key not in _NEAREST_TABLES:
        def table(n_src, n_dst):
            idx = torch.arange(n_src, dtype=torch.float32).view(1, 1, 1, n_src)
            return F.interpolate(idx, size=(1, n_dst), mode='nearest').view(-1).to(torch.int32)
        _NEAREST_TABLES[key] = (table(src_size[0], dst_size[0]).to(device), table(src_size[1], dst_size[1]).to(device))
    return _NEAREST_TABLES[key]


def upsample_depth(all_depth, stages, chunk=100):
    """Stage 3 of the scene driver (eval-3dvnet.py:101-125): for each (PropagationNet, guide tensor) pair,
    nearest-neighbour resize the depth to the guide's resolution and refine it.  With the package's PropagationNet on a HIP device
    the resize is folded into the kernel's addressing and a stage is ONE launch over all views (``forward_resized``; the
    reference's UPSAMPLE_BATCH = 100 chunking is a memory limit of its unfold buffer, which does not exist here); any other
    callable takes the reference's path, `chunk` views at a time."""
    for net, guide in stages:
        if isinstance(net, PropagationNet) and all_depth.is_cuda and guide.is_cuda:
            all_depth = net.forward_resized(guide, all_depth)
            continue
        all_depth = F.interpolate(all_depth.unsqueeze(1), guide.shape[-2:], mode='nearest').squeeze(1)
        for s in range(0, all_depth.shape[0], chunk):
            all_depth[s:s + chunk] = net(guide[s:s + chunk], all_depth[s:s + chunk].unsqueeze(1))
    return all_depth
