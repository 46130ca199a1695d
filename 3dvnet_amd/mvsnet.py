"""Host-side mirror of ``mv3d/subnetworks/mvsnet.py`` for the cost-volume path (SURVEY.md §8a rows
A1-A7, §8b).  Same class names, constructor arguments, ``forward`` signatures, return tuples,
tensor layouts and ``state_dict`` keys as the reference; the arithmetic runs in
``lib3dvnet_hip.so`` (hand-written gfx950 kernels) through the C ABI of ``include/v3d.h``.

There is NO CPU / eager-PyTorch fallback: without the built HIP library every forward raises
``V3DLibraryError``.  Inference only (the reference's benchmark path runs under
``torch.no_grad()``, mv3d/eval-3dvnet.py:27).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _Workspace:
    """Grow-only device scratch buffers keyed by name (the C ABI never allocates tensor memory)."""

    def __init__(self):
        self._bufs = {}
        self.tags = {}          # name -> what a caller left in the buffer (its own key), dropped when the buffer is replaced

    def get(self, name, nbytes, device):
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[name] = buf
            self.tags.pop(name, None)
        return buf

    # scratch is not state: a copied / unpickled module starts with empty buffers (and does not duplicate gigabytes of them)
    def __deepcopy__(self, memo):
        return _Workspace()

    def __reduce__(self):
        return (_Workspace, ())


def _psv_kernel_option():
    v = ctypes.c_int(0)
    _lib.check(_lib.load().v3d_get_option(b'psv_kernel', ctypes.byref(v)), 'v3d_get_option')
    return v.value


def module_state_key(module):
    """Cache key of a module's packed device weights.  ``tensor._version`` alone misses ``module.to(device)``,
    ``p.data = ...``, ``load_state_dict(assign=True)`` and ``swap_tensors`` (new storage, coinciding versions), so the
    identity, storage address, device and dtype of every parameter / buffer are part of the key."""
    return tuple((id(p), int(p._version), p.data_ptr(), str(p.device), str(p.dtype))
                 for p in list(module.parameters()) + list(module.buffers()))


def module_device(module):
    for p in module.parameters():
        return p.device
    return torch.device('cpu')


def _require_cuda(t, what):
    if not t.is_cuda:
        raise _lib.V3DLibraryError('%s: tensors must live on a HIP device (no CPU fallback)' % what)


class ConvBnRelu3d(nn.Module):
    """Parameter container with the reference's keys ``conv.weight`` / ``bn.*`` (mvsnet.py:18-25).
    The arithmetic is executed by the fused HIP layer kernel, not by these modules."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)


class DeconvBnRelu3d(nn.Module):
    """Parameter container with keys ``deconv.weight`` / ``bn.*`` (mvsnet.py:28-36)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, padding=1,
                 output_padding=1):
        super().__init__()
        self.deconv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride, padding,
                                         output_padding, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)


class CostRegNet(nn.Module):
    """Dense 3D-conv regulariser, reference ``CostRegNet(in_channels, base_channels)``
    (mvsnet.py:133-163).  ``forward(x[B,Cin,D,h,w]) -> [B,1,D,h,w]``."""

    def __init__(self, in_channels, base_channels, precision='split_bf16'):
        super().__init__()
        b = base_channels
        self.in_channels, self.base_channels = in_channels, base_channels
        # MFMA operand precision of every layer (include/v3d.h, V3D_PRECISION_*): 'split_bf16' (default: three bf16
        # products per fp32 product, 16 mantissa bits) or 'fp32' (exact-fp32 matrix cores, the reference's arithmetic)
        _lib.precision_code(precision)
        self.precision = precision
        self.conv0 = ConvBnRelu3d(in_channels, b)
        self.conv1 = ConvBnRelu3d(b, 2 * b, stride=2)
        self.conv2 = ConvBnRelu3d(2 * b, 2 * b)
        self.conv3 = ConvBnRelu3d(2 * b, 4 * b, stride=2)
        self.conv4 = ConvBnRelu3d(4 * b, 4 * b)
        self.conv5 = ConvBnRelu3d(4 * b, 8 * b, stride=2)
        self.conv6 = ConvBnRelu3d(8 * b, 8 * b)
        self.conv7 = DeconvBnRelu3d(8 * b, 4 * b, output_padding=1)
        self.conv8 = DeconvBnRelu3d(4 * b, 2 * b)
        self.conv9 = DeconvBnRelu3d(2 * b, b, output_padding=1)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1)
        self._handle = None
        self._packed_key = None
        self._ws = _Workspace()

    # -- weight packing (BN fold + MFMA fragment order happen inside the library) ---------------
    def _layers(self):
        return [getattr(self, 'conv%d' % i) for i in range(10)]

    def packed_handle(self, device=None):
        """BN-folded MFMA weight image on `device` (default: the parameters' device), re-packed whenever a
        parameter / buffer changed, was replaced or moved, or the input lives on another HIP device."""
        device = torch.device(device) if device is not None else module_device(self)
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

        layers = self._layers()
        convs = [(l.conv if hasattr(l, 'conv') else l.deconv).weight for l in layers]
        eps = {float(l.bn.eps) for l in layers}
        assert len(eps) == 1
        handle = ctypes.c_void_p()
        if device.type != 'cuda':
            raise _lib.V3DLibraryError('CostRegNet: weights must be packed for a HIP device (no CPU fallback)')
        with torch.cuda.device(device):          # the library allocates the weight image on the current device
            rc = lib.v3d_costreg_pack(parray(convs), parray([l.bn.weight for l in layers]),
                                      parray([l.bn.bias for l in layers]),
                                      parray([l.bn.running_mean for l in layers]),
                                      parray([l.bn.running_var for l in layers]),
                                      host(self.prob.weight), host(self.prob.bias),
                                      self.in_channels, self.base_channels, eps.pop(),
                                      ctypes.byref(handle))
        _lib.check(rc, 'v3d_costreg_pack')
        self._handle, self._packed_key = handle, key
        return handle

    def release(self):
        if self._handle is not None:
            _lib.load().v3d_costreg_free(self._handle)
            self._handle = None

    def __getstate__(self):
        # the packed device image (a ctypes handle) is a cache, not state: copy.deepcopy / pickle carry the parameters only and
        # the copy re-packs on its first forward
        st = self.__dict__.copy()
        st['_handle'], st['_packed_key'] = None, None
        return st

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # -- execution --------------------------------------------------------------------------------
    def regularize_depth(self, x, depth_vals, return_reg=False, precision=None):
        """Rows A5-A6 fused: x [B,Cin,D,h,w] variance volume, depth_vals [D] ->
        depth [B,h,w] (and x_reg [B,D,h,w] when return_reg).  ``precision`` overrides ``self.precision``."""
        cl8 = isinstance(x, Cl8Variance)
        split = isinstance(x, SplitVariance) and not cl8
        precision = precision or self.precision
        if split and precision != 'split_bf16':
            raise ValueError("a SplitVariance volume is the split_bf16 operand encoding; pass the fp32 volume for 'fp32'")
        if cl8 and precision != 'fp32':
            raise ValueError("a Cl8Variance volume is the input of the exact-fp32 chain; pass precision='fp32'")
        _require_cuda(x.data if (split or cl8) else x, 'CostRegNet')
        assert not self.training, 'inference only: BatchNorm is folded with running statistics'
        lib = _lib.load()
        if split or cl8:
            (B, C, D, h, w), x = x.shape, x.data
        else:
            x = x.contiguous().float()
            B, C, D, h, w = x.shape
        assert C == self.in_channels
        handle = self.packed_handle(x.device)
        depth = torch.empty((B, h, w), dtype=torch.float32, device=x.device)
        reg = torch.empty((B, D, h, w), dtype=torch.float32, device=x.device) if return_reg else None
        nbytes = lib.v3d_costreg_workspace_bytes(handle, B, D, h, w)
        ws = self._ws.get('costreg', nbytes, x.device)
        depth_vals = depth_vals.to(device=x.device, dtype=torch.float32).contiguous()
        if split or cl8:
            fn = lib.v3d_costreg_depth_cl8 if cl8 else lib.v3d_costreg_depth_split
            rc = fn(handle, _lib.ptr(x), _lib.ptr(depth_vals), B, D, h, w, _lib.ptr(depth),
                    _lib.ptr(reg), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        else:
            rc = lib.v3d_costreg_depth_f32(handle, _lib.ptr(x), _lib.ptr(depth_vals), B, D, h, w, _lib.ptr(depth),
                                           _lib.ptr(reg), _lib.precision_code(precision), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(x.device))
        _lib.check(rc, 'v3d_costreg_depth_cl8' if cl8 else 'v3d_costreg_depth_split' if split else 'v3d_costreg_depth_f32')
        return (depth, reg) if return_reg else depth

    def run_layer(self, layer, x, skip=None, split=False, precision='split_bf16'):
        """One conv/deconv + folded BN + ReLU (+ skip) layer, for per-layer parity tests.
        `split=True` (layers 1..8): the kernel the fused path uses for that layer (split-bf16 matrix
        cores, split activation layout); otherwise the fp32-layout per-layer kernel: exact fp32 for
        layers 1..9, and for layer 0 its split-bf16 product kernel unless ``precision='fp32'``."""
        _require_cuda(x, 'CostRegNet')
        lib = _lib.load()
        x = x.contiguous().float()
        n, _, Di, Hi, Wi = x.shape
        mod = self._layers()[layer]
        if hasattr(mod, 'deconv'):
            co, shape = mod.deconv.out_channels, (2 * Di, 2 * Hi, 2 * Wi)
        else:
            s = mod.conv.stride[0]
            co, shape = mod.conv.out_channels, tuple((d - 1) // s + 1 for d in (Di, Hi, Wi))
        out = torch.empty((n, co) + shape, dtype=torch.float32, device=x.device)
        if skip is not None:
            skip = skip.contiguous().float()
            assert skip.shape == out.shape
        if split:
            nbytes = lib.v3d_costreg_layer_split_workspace_bytes(n, x.shape[1], Di, Hi, Wi)
            ws = self._ws.get('layer_split', nbytes, x.device)
            rc = lib.v3d_costreg_layer_split_f32(self.packed_handle(x.device), layer, _lib.ptr(x), _lib.ptr(skip), n,
                                                 Di, Hi, Wi, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr(x.device))
            _lib.check(rc, 'v3d_costreg_layer_split_f32')
            return out
        rc = lib.v3d_costreg_layer_f32(self.packed_handle(x.device), layer, _lib.ptr(x), _lib.ptr(skip), n,
                                       Di, Hi, Wi, _lib.ptr(out), _lib.precision_code(precision),
                                       _lib.stream_ptr(x.device))
        _lib.check(rc, 'v3d_costreg_layer_f32')
        return out

    def forward(self, x):
        D = x.shape[2]
        zeros = torch.zeros(D, dtype=torch.float32, device=x.device)
        _, reg = self.regularize_depth(x, zeros, return_reg=True)
        return reg.unsqueeze(1)


class EdgeCsr(tuple):
    """(ref_idx int64, ref_img i32, edge_ofs i32, edge_src i32) of ``edges_to_csr``; ``status_ws`` is the device workspace
    of the native builder (None for the torch path) and ``check()`` reads its error word (synchronises)."""
    status_ws = None

    def check(self):
        if self.status_ws is not None:
            lib = _lib.load()
            ws = self.status_ws
            _lib.check(lib.v3d_edges_csr_status(_lib.ptr(ws), ws.numel(), _lib.stream_ptr(ws.device)), 'v3d_edges_csr')
        return self


def edges_to_csr(ref_src_edges, n_ref=None, n_img=None):
    """ref_src_edges [2,E] -> (ref_idx [n_ref] int64 sorted unique, ref_img i32, edge_ofs i32
    [n_ref+1], edge_src i32 [E] grouped per reference in original edge order).  Mirrors
    ``torch.unique(edges[0], return_inverse=True)`` + scatter-by-``gather_idx`` (mvsnet.py:179,
    214-215) so the per-reference sums run over the same edges in the same order.

    ``torch.unique`` makes the host wait for the device (its output length is data).  A caller that knows how many
    reference images the batch holds passes ``n_ref`` (and ``n_img``, the number of images the indices refer to): the
    tables are then built by one device kernel (include/v3d.h, v3d_edges_csr) with no synchronisation; a wrong ``n_ref``
    yields an empty CSR and an error reported by ``.check()``."""
    if n_ref is not None and ref_src_edges.is_cuda:
        if n_img is None:
            raise ValueError('edges_to_csr: n_img is required with n_ref')
        lib = _lib.load()
        dev = ref_src_edges.device
        e = ref_src_edges.to(torch.int64).contiguous()
        n_edges = e.shape[1]
        ref_img = torch.empty(n_ref, dtype=torch.int32, device=dev)
        edge_ofs = torch.empty(n_ref + 1, dtype=torch.int32, device=dev)
        edge_src = torch.empty(n_edges, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.v3d_edges_csr_workspace_bytes(int(n_img), int(n_ref)), dtype=torch.uint8, device=dev)
        rc = lib.v3d_edges_csr(_lib.ptr(e), n_edges, int(n_img), int(n_ref), _lib.ptr(ref_img), _lib.ptr(edge_ofs),
                               _lib.ptr(edge_src), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, 'v3d_edges_csr')
        csr = EdgeCsr((ref_img.to(torch.int64), ref_img, edge_ofs, edge_src))
        csr.status_ws = ws
        return csr
    ref_idx, gather_idx = torch.unique(ref_src_edges[0], return_inverse=True)
    n_ref = ref_idx.shape[0]
    order = torch.sort(gather_idx, stable=True).indices
    edge_src = ref_src_edges[1][order].to(torch.int32).contiguous()
    counts = torch.bincount(gather_idx, minlength=n_ref)
    edge_ofs = torch.zeros(n_ref + 1, dtype=torch.int32, device=ref_src_edges.device)
    edge_ofs[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return EdgeCsr((ref_idx, ref_idx.to(torch.int32).contiguous(), edge_ofs, edge_src))


class SplitVariance:
    """The variance volume in the regulariser's private input format (include/v3d.h,
    v3d_psv_variance_split): every fp32 value stored as a bf16 hi + bf16 lo pair, channel-last in
    16-byte slots.  Same bytes as the fp32 volume and the same numbers conv0 would derive from it;
    only `CostRegNet.regularize_depth` consumes it."""

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)

    @property
    def device(self):
        return self.data.device


class Cl8Variance(SplitVariance):
    """The variance volume as fp32 in the channel-last layout of the exact-fp32 chain's conv0 (include/v3d.h,
    v3d_psv_variance_cl8): [n_ref][4 channel groups][2 halves][D][h][w] slots of 4 floats.  The numbers of the
    reference-layout tensor, bit for bit; only `CostRegNet.regularize_depth(precision='fp32')` consumes it."""


def plane_sweep_variance(features_quarter, rotmats, tvecs, K, ref_src_edges, depth_start,
                         depth_interval, n_planes, img_size, depth_img_size, workspace=None,
                         csr=None, split=False, n_ref=None, cl8=False):
    """Rows A1-A4 (mvsnet.py:186-216): variance cost volume [n_ref, C, D, h, w]
    (`split=True`: the same volume as a `SplitVariance`, `cl8=True`: as a `Cl8Variance`; C == 32 only)."""
    _require_cuda(features_quarter, 'plane_sweep_variance')
    lib = _lib.load()
    feat = features_quarter.contiguous().float()
    dev = feat.device
    n_img, C, Hf, Wf = feat.shape
    if csr is None:
        csr = edges_to_csr(ref_src_edges.to(dev), n_ref=n_ref, n_img=n_img)
    _, ref_img, edge_ofs, edge_src = csr
    n_ref, n_edges = ref_img.shape[0], edge_src.shape[0]
    h, w = depth_img_size
    var = torch.empty((n_ref, C, n_planes, h, w), dtype=torch.float32, device=dev)
    nbytes = lib.v3d_psv_workspace_bytes(n_img, C, Hf, Wf)
    ws = (workspace or _Workspace()).get('psv', nbytes, dev)
    Kc, Rc, tc = (x.to(dev).contiguous().float() for x in (K, rotmats, tvecs))
    fn = lib.v3d_psv_variance_cl8 if cl8 else lib.v3d_psv_variance_split if split else lib.v3d_psv_variance_f32
    rc = fn(_lib.ptr(feat), _lib.ptr(Kc), _lib.ptr(Rc), _lib.ptr(tc), _lib.ptr(ref_img),
            _lib.ptr(edge_ofs), _lib.ptr(edge_src), n_img, n_ref, n_edges, C, Hf, Wf, int(img_size[0]),
            int(img_size[1]), float(depth_start), float(depth_interval), int(n_planes), int(h), int(w),
            _lib.ptr(var), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, 'v3d_psv_variance_cl8' if cl8 else 'v3d_psv_variance_split' if split else 'v3d_psv_variance_f32')
    return Cl8Variance(var, var.shape) if cl8 else SplitVariance(var, var.shape) if split else var


def plane_sweep_sample_positions(rotmats, tvecs, K, ref_src_edges, depth_start, depth_interval, n_planes, img_size,
                                 feat_size, depth_img_size, device):
    """Diagnostic (include/v3d.h, v3d_psv_sample_positions_f32): -> (pos [E, D*h*w, 2] = (ix, iy) in feature pixels in
    CSR edge order, world [n_ref, 3, D*h*w], csr) exactly as the warp kernels compute them."""
    lib = _lib.load()
    dev = torch.device(device)
    csr = edges_to_csr(ref_src_edges.to(dev))
    _, ref_img, edge_ofs, edge_src = csr
    n_ref, n_edges = ref_img.shape[0], edge_src.shape[0]
    h, w = depth_img_size
    n_vox = n_planes * h * w
    Kc, Rc, tc = (x.to(dev).contiguous().float() for x in (K, rotmats, tvecs))
    n_img = Kc.shape[0]
    pos = torch.empty((n_edges, n_vox, 2), dtype=torch.float32, device=dev)
    world = torch.empty((n_ref, 3, n_vox), dtype=torch.float32, device=dev)
    ws = torch.empty(n_img * 36 * 4 + 256, dtype=torch.uint8, device=dev)
    rc = lib.v3d_psv_sample_positions_f32(_lib.ptr(Kc), _lib.ptr(Rc), _lib.ptr(tc), _lib.ptr(ref_img), _lib.ptr(edge_ofs),
                                          _lib.ptr(edge_src), n_img, n_ref, n_edges, int(feat_size[0]), int(feat_size[1]),
                                          int(img_size[0]), int(img_size[1]), float(depth_start), float(depth_interval),
                                          int(n_planes), int(h), int(w), _lib.ptr(pos), _lib.ptr(world), _lib.ptr(ws),
                                          ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, 'v3d_psv_sample_positions_f32')
    return pos, world, csr


class MVSNet(nn.Module):
    """Reference ``MVSNet(feat_dim=32, img_size=(240, 320))`` (mvsnet.py:166-229).

    ``forward(batch, depth_start, depth_interval, n_planes, depth_img_size) ->
    (depth_img[n_ref,h,w], features_half, features_quarter, features_eighth)``.

    ``feat_extractor`` / ``feat_shrinker`` (the 2D MnasNet + FPN backbone, stock PyTorch modules in
    the reference) are injected; when they are ``None`` the batch must carry
    pre-computed ``features_half / features_quarter / features_eighth`` attributes."""

    def __init__(self, feat_dim=32, img_size=(240, 320), feat_extractor=None, feat_shrinker=None,
                 precision='split_bf16'):
        super().__init__()
        self.feat_dim = feat_dim
        self.img_size = img_size
        self.feat_extractor = feat_extractor
        self.feat_shrinker = feat_shrinker
        self.native_backbone = True          # False (explicit opt-in): the package's backbone containers run as stock PyTorch modules (MIOpen)
        self._native_backbone = None
        self.cnn_3d = CostRegNet(feat_dim, 8, precision=precision)
        self._ws = _Workspace()
        self._depth_vals = {}
        self.last_csr = None

    def depth_values(self, depth_start, depth_interval, n_planes, device):
        """torch.linspace(depth_start, depth_end, n_planes) built on the CPU then moved, exactly
        like mvsnet.py:223 (`.type_as(batch.images)`), cached per configuration."""
        key = (float(depth_start), float(depth_interval), int(n_planes), str(device))
        if key not in self._depth_vals:
            depth_end = depth_start + depth_interval * (n_planes - 1)
            self._depth_vals[key] = torch.linspace(depth_start, depth_end, n_planes).to(device)
        return self._depth_vals[key]

    def cost_volume_depth(self, features_quarter, batch, depth_start, depth_interval, n_planes,
                          depth_img_size, return_intermediates=False, csr=None, precision=None, n_ref=None):
        """Rows A1-A6 from quarter-resolution features.  ``precision`` ('split_bf16' | 'fp32') overrides the
        regulariser's ``cnn_3d.precision``.  ``n_ref`` (optional): the number of reference images in ``batch``; with it the
        edge tables are built on the device without the host synchronisation ``torch.unique`` implies (`edges_to_csr`).  With split-bf16 operands, unless the caller asks for the
        intermediates, the variance volume travels to the regulariser in its split-bf16 input format
        (identical depth, no conversion pass in conv0)."""
        precision = precision or self.cnn_3d.precision
        split = not return_intermediates and features_quarter.shape[1] == 32 and precision == 'split_bf16'
        # exact fp32: the volume in the channel-last fp32 layout conv0's depth march streams (same numbers as the reference layout).
        # Only the window kernel writes it: developer runs of the reuse / gather kernels (v3d_set_option "psv_kernel") and feature
        # stacks of 2 GB or more (which take the reuse kernel) keep the reference layout and the per-layer conv0.
        cl8 = (not return_intermediates and features_quarter.shape[1] == 32 and precision == 'fp32'
               and features_quarter.numel() * 4 < 2 ** 31 and _psv_kernel_option() == 0)
        if csr is None:
            # kept on the module: `check_edges()` reads the device builder's status word (a wrong n_ref gives an EMPTY edge
            # table, i.e. a zero variance volume and a plausible-looking depth, not an exception)
            csr = edges_to_csr(batch.ref_src_edges.to(features_quarter.device), n_ref=n_ref, n_img=features_quarter.shape[0])
        self.last_csr = csr
        var = plane_sweep_variance(features_quarter, batch.rotmats, batch.tvecs, batch.K,
                                   batch.ref_src_edges, depth_start, depth_interval, n_planes,
                                   self.img_size, depth_img_size, workspace=self._ws, csr=csr,
                                   split=split, n_ref=n_ref, cl8=cl8)
        vals = self.depth_values(depth_start, depth_interval, n_planes, var.device)
        if return_intermediates:
            depth, reg = self.cnn_3d.regularize_depth(var, vals, return_reg=True, precision=precision)
            return depth, var, reg
        return self.cnn_3d.regularize_depth(var, vals, precision=precision)

    def check_edges(self):
        """Raise if the edge tables of the most recent forward were rejected by the device builder (``n_ref`` did not match
        the edge list, or an image index was out of range).  Synchronises; the torch-built tables cannot fail."""
        if getattr(self, 'last_csr', None) is not None:
            self.last_csr.check()
        return self

    def forward(self, batch, depth_start, depth_interval, n_planes, depth_img_size, n_ref=None):
        if self.feat_extractor is not None:
            # The package's own MnasNet + FPN containers run on the library's backbone kernels (csrc/backbone.hip); a call that
            # does not qualify RAISES -- there is no silent second backend.  The stock PyTorch modules (MIOpen / rocBLAS) are an
            # explicit opt-in: `net.native_backbone = False`.  A foreign pair of modules (anything else the caller injected)
            # is the caller's own code and runs as given.
            if getattr(self, '_native_backbone', None) is None or self._native_backbone.fe is not self.feat_extractor \
                    or self._native_backbone.fs is not self.feat_shrinker or self._native_backbone.precision != self.cnn_3d.precision:
                from .backbone import NativeBackbone
                self._native_backbone = NativeBackbone(self.feat_extractor, self.feat_shrinker, precision=self.cnn_3d.precision)
            nb = self._native_backbone
            if self.native_backbone and nb.is_package_pair():
                why = nb.why_not(batch.images)
                if why is not None:
                    raise _lib.V3DLibraryError('MVSNet.forward: the HIP backbone cannot take this call (%s); set '
                                               '`native_backbone = False` on the MVSNet to run the stock PyTorch modules '
                                               'explicitly' % why)
                features_half, features_quarter, features_eighth, _, _ = nb(batch.images)
            else:
                features_half, features_quarter, features_eighth, _, _ = \
                    self.feat_shrinker(*self.feat_extractor(batch.images))
        else:
            features_half = getattr(batch, 'features_half', None)
            features_quarter = batch.features_quarter
            features_eighth = getattr(batch, 'features_eighth', None)
        depth_img = self.cost_volume_depth(features_quarter, batch, depth_start, depth_interval,
                                           n_planes, depth_img_size, n_ref=n_ref)
        return depth_img, features_half, features_quarter, features_eighth


class CostVolumeGraph:
    """Rows A1-A6 for FIXED shapes captured once into a HIP graph (``torch.cuda.CUDAGraph``) and replayed: the ~17 kernel
    launches of a step (edge tables, transpose, camera blocks, warp + variance, ten regulariser layers, soft-argmin) become
    one graph launch, which removes the host's per-launch cost and most of the gaps between dependent kernels.  Nothing in the
    step synchronises or allocates outside the graph's private pool: the edge tables come from ``v3d_edges_csr`` (``n_ref``
    is required), the workspaces are the module's cached buffers.

    The graph reads its inputs from the tensors given at capture time (kept alive here); ``update(...)`` copies new data of the
    same shapes into them, ``replay()`` runs the step and returns the depth tensor [n_ref, h, w] (the same storage each time).
    No reference counterpart (the reference runs eagerly); results are those of ``MVSNet.cost_volume_depth``, bit for bit."""

    def __init__(self, net, features_quarter, batch, depth_start, depth_interval, n_planes, depth_img_size, n_ref,
                 precision=None, warmup=2):
        _require_cuda(features_quarter, 'CostVolumeGraph')
        self.net = net
        self.features_quarter = features_quarter
        self.batch = batch
        self._args = (depth_start, depth_interval, n_planes, depth_img_size)
        self._kw = dict(precision=precision, n_ref=int(n_ref))
        dev = features_quarter.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up on a side stream: lazy one-time setup happens here
            for _ in range(max(1, warmup)):
                net.cost_volume_depth(features_quarter, batch, *self._args, **self._kw)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the edge list is validated once, eagerly: inside the graph nobody reads the builder's status word
        edges_to_csr(batch.ref_src_edges, n_ref=int(n_ref), n_img=features_quarter.shape[0]).check()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.depth = net.cost_volume_depth(features_quarter, batch, *self._args, **self._kw)
        # The captured launches carry raw device addresses.  Everything they point to is pinned here: the module's scratch
        # buffers (a later, larger eager call REPLACES the entries of the grow-only workspaces -- the tensors held below stay
        # alive and the graph keeps its own), the cached plane depths, and the identity of the packed weight image, which
        # `packed_handle()` frees as soon as a parameter of the regulariser changes: replay() refuses to run on a stale one.
        self._pinned = list(net._ws._bufs.values()) + list(net.cnn_3d._ws._bufs.values()) + list(net._depth_vals.values())
        self._weights_key = net.cnn_3d._packed_key
        self._weights_handle = net.cnn_3d._handle.value if net.cnn_3d._handle is not None else None

    def stale(self):
        """True when the regulariser's packed weight image the graph was captured with is gone: a parameter or buffer of
        ``net.cnn_3d`` was modified, replaced or moved (or the image was re-packed for another device) since capture."""
        c = self.net.cnn_3d
        cur = c._handle.value if c._handle is not None else None
        return cur != self._weights_handle or c._packed_key != self._weights_key or \
            c._packed_key != (c._packed_key[0],) + module_state_key(c)

    def update(self, features_quarter=None, rotmats=None, tvecs=None, K=None, ref_src_edges=None):
        if features_quarter is not None:
            self.features_quarter.copy_(features_quarter)
        for name, val in (('rotmats', rotmats), ('tvecs', tvecs), ('K', K), ('ref_src_edges', ref_src_edges)):
            if val is not None:
                getattr(self.batch, name).copy_(val)
        if ref_src_edges is not None:      # a new edge list must still hold exactly n_ref reference images
            edges_to_csr(self.batch.ref_src_edges, n_ref=self._kw['n_ref'], n_img=self.features_quarter.shape[0]).check()

    def replay(self):
        if self.stale():
            raise RuntimeError('CostVolumeGraph: the weights of net.cnn_3d changed after capture (the packed weight image '
                               'the graph points to was released); capture a new graph')
        self.graph.replay()
        return self.depth
