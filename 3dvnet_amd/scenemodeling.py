"""Host-side mirror of ``mv3d/subnetworks/scenemodeling.py``: ``PointNet``
and ``SparseUNet`` with the reference's constructor arguments, ``forward`` signatures, return
structures and ``state_dict`` keys (MinkowskiEngine parameter naming: ``.kernel``, ``.gn.weight``).
MinkowskiEngine is replaced by hash-indexed neighbour tables + the matrix-core gather-GEMM of
``lib3dvnet_hip.so`` (csrc/sparse.hip, csrc/gemm_gather.hip; fp32 storage and accumulation, MFMA operands per
the module's ``precision``: 'split_bf16' = three bf16 products per fp32 product (default), 'fp32' = exact-fp32
MFMA); torch_scatter's max is a per-voxel segment reduction (csrc/segment.hip; the GEMM epilogue's fused
scatter-max remains available through ``PackedGemm(pool=...)``).  No CPU fallback.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .mvsnet import _Workspace

_NEG_INF = float('-inf')


def _host_f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


class PackedGemm:
    """Device-resident packed weights of one gather-GEMM layer (v3d_gemm_pack)."""

    def __init__(self, w, stride_seg, stride_co, stride_k, n_seg, N, K, scale=None, bias=None,
                 gn_w=None, gn_b=None, device=None, gn_group=16):
        lib = _lib.load()
        assert gn_group in (8, 16), 'the GroupNorm epilogue handles 8- and 16-channel groups'
        self.gn_group = gn_group
        keep = [_host_f32(w)] + [None if x is None else _host_f32(x) for x in (scale, bias, gn_w, gn_b)]
        ptrs = [None if a is None else a.ctypes.data_as(_lib.c_float_p) for a in keep]
        self.handle = ctypes.c_void_p()
        if device is None:      # host-resident weights are packed for the current HIP device
            if not torch.cuda.is_available():
                raise _lib.V3DLibraryError('gather-GEMM: no HIP device to pack the weights for (no CPU fallback)')
            device = w.device if w.is_cuda else torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.V3DLibraryError('gather-GEMM: weights must be packed for a HIP device (no CPU fallback)')
        with torch.cuda.device(self.device):     # the library allocates the weight image on the current device
            rc = lib.v3d_gemm_pack(ptrs[0], stride_seg, stride_co, stride_k, n_seg, N, K, ptrs[1], ptrs[2],
                                   ptrs[3], ptrs[4], ctypes.byref(self.handle))
        _lib.check(rc, 'v3d_gemm_pack')
        self.n_seg, self.N, self.K = n_seg, N, K

    def __del__(self):
        try:
            if self.handle:
                _lib.load().v3d_gemm_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def __call__(self, M, srcs, idxs=None, lds=None, group_len=0, relu_in=False, use_gn=False,
                 gn_eps=1e-5, residual=None, relu_out=False, pool=None, pool_idx=None, out=True,
                 precision='split_bf16'):
        """srcs: list of n_seg source matrices [rows, ld] (float32, contiguous);
        idxs: list of int32 row maps (or None = identity) per segment."""
        lib = _lib.load()
        n = self.n_seg
        dev = srcs[0].device
        if not srcs[0].is_cuda:
            raise _lib.V3DLibraryError('gather-GEMM: tensors must live on a HIP device (no CPU fallback)')
        if dev != self.device:
            raise _lib.V3DLibraryError('gather-GEMM: weights were packed on %s, input lives on %s' % (self.device, dev))
        assert len(srcs) == n
        src_arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
        idx_arr = (ctypes.c_void_p * n)(*[(None if (idxs is None or idxs[i] is None) else
                                          (idxs[i] if isinstance(idxs[i], int) else idxs[i].data_ptr()))
                                         for i in range(n)])
        ld_arr = (ctypes.c_int * n)(*[(lds[i] if lds is not None else srcs[i].shape[-1]) for i in range(n)])
        y = None
        if out is True:
            y = torch.empty((M, self.N), dtype=torch.float32, device=dev)
        elif out is not None and out is not False:
            y = out
        rc = lib.v3d_gemm_gather_f32(
            self.handle, M, src_arr, idx_arr, ld_arr, group_len, int(relu_in), self.gn_group if use_gn else 0, gn_eps,
            _lib.ptr(residual), residual.shape[-1] if residual is not None else 0, int(relu_out),
            _lib.ptr(pool), _lib.ptr(pool_idx), pool.shape[-1] if pool is not None else 0,
            _lib.ptr(y), y.stride(0) if y is not None else 0, _lib.precision_code(precision), _lib.stream_ptr(dev))
        _lib.check(rc, 'v3d_gemm_gather_f32')
        return y


class _PackCache:
    """Re-packs when any parameter / buffer of the owning module changed, was replaced or moved to another device
    (see ``mvsnet.module_state_key``), or when the input lives on a different HIP device than the packed image."""

    def __init__(self, module):
        self._module = module
        self._key = None
        self._packs = None

    def get(self, builder, device=None):
        from .mvsnet import module_state_key
        key = (str(device),) + module_state_key(self._module)
        if self._packs is None or key != self._key:
            self._packs, self._key = builder(), key
        return self._packs

    # the packed images (ctypes handles) are a cache, not state: a deep copy / an unpickled object belongs to the COPIED module
    # and packs again on first use
    def __deepcopy__(self, memo):
        import copy
        return _PackCache(copy.deepcopy(self._module, memo))

    def __reduce__(self):
        return (_PackCache, (self._module,))


class PointNet(nn.Module):
    """Reference ``PointNet(hidden_dim, out_dim, in_dim=3)`` (scenemodeling.py:116-144):
    ``forward(pts[Np,in_dim], idx[Np], n_idx) -> [n_idx, out_dim]``."""

    def __init__(self, hidden_dim, out_dim, in_dim=3, precision='split_bf16'):
        super().__init__()
        _lib.precision_code(precision)
        self.precision = precision
        self.hidden_dim = hidden_dim
        self.fc_pos = nn.Linear(in_dim, hidden_dim)
        self.fc1 = nn.Linear(hidden_dim, hidden_dim)
        self.fc2 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.fc3 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.fc4 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.fc_out = nn.Linear(hidden_dim, out_dim)
        self._cache = _PackCache(self)
        self._ws = _Workspace()

    def _build(self):
        def lin(m, n_seg):
            N, Kt = m.weight.shape
            return PackedGemm(m.weight, Kt // n_seg, Kt, 1, n_seg, N, Kt // n_seg, bias=m.bias, device=self._dev)
        return dict(fc_pos=lin(self.fc_pos, 1), fc1=lin(self.fc1, 1), fc2=lin(self.fc2, 2),
                    fc3=lin(self.fc3, 2), fc4=lin(self.fc4, 2), fc_out=lin(self.fc_out, 1))

    def forward(self, pts, idx, n_idx):
        if not pts.is_cuda:
            raise _lib.V3DLibraryError('PointNet: tensors must live on a HIP device (no CPU fallback)')
        self._dev = dev = pts.device
        g = self._cache.get(self._build, dev)
        lib = _lib.load()
        pr = self.precision
        pts = pts.contiguous().float()
        Np, H = pts.shape[0], self.hidden_dim
        idx32 = idx.to(torch.int32).contiguous()
        stream = _lib.stream_ptr(dev)

        # Max-pool over the points of a voxel (scatter(x, idx, reduce='max'), :131,135,139): the rows are grouped by voxel once
        # (device radix sort -> row list + offsets), each layer's stored output is then reduced per voxel by
        # v3d_segment_max_f32 -- no atomics (the gather-GEMM's fused scatter-max was 80 % of a layer's time; csrc/segment.hip).
        perm = torch.empty(Np, dtype=torch.int32, device=dev)
        offs = torch.empty(n_idx + 1, dtype=torch.int32, device=dev)
        wb = lib.v3d_segment_csr_workspace_bytes(Np)
        ws = self._ws.get('segcsr', wb, dev)
        _lib.check(lib.v3d_segment_csr(idx32.data_ptr(), Np, int(n_idx), perm.data_ptr(), offs.data_ptr(), ws.data_ptr(),
                                       ws.numel(), stream), 'v3d_segment_csr')

        def pooled(x):
            pool = torch.empty((n_idx, H), dtype=torch.float32, device=dev)
            _lib.check(lib.v3d_segment_max_f32(x.data_ptr(), x.stride(0), perm.data_ptr(), offs.data_ptr(), int(n_idx), H,
                                               pool.data_ptr(), H, stream), 'v3d_segment_max_f32')
            return pool

        h = g['fc_pos'](Np, [pts], precision=pr)                              # fc_pos(pts)
        x = g['fc1'](Np, [h], relu_in=True, precision=pr)                     # fc1(relu(.))
        pool = pooled(x)
        for name in ('fc2', 'fc3', 'fc4'):
            x = g[name](Np, [x, pool], idxs=[None, idx32], relu_in=True, precision=pr)   # fcK(relu(cat(x, pool[idx])))
            pool = pooled(x)
        return g['fc_out'](n_idx, [pool], relu_in=True, precision=pr)


class _SparseConv(nn.Module):
    """Parameter container for a MinkowskiConvolution(-Transpose): ``kernel`` [27, Ci, Co] or [Ci, Co]."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        shape = (27, in_channels, out_channels) if kernel_size == 3 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        nn.init.kaiming_uniform_(self.kernel.view(-1, out_channels).t(), a=5 ** 0.5)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size


class MinkowskiGroupNorm(nn.Module):
    """Reference-defined wrapper (scenemodeling.py:78-113): ``gn = torch.nn.GroupNorm(G, C)`` applied to
    the [N, C] feature matrix => every voxel row is normalised on its own."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        super().__init__()
        self.gn = nn.GroupNorm(num_groups, num_channels, eps=eps, affine=affine)


class SparseResidual3d(nn.Module):
    def __init__(self, feat_dim, norm="gn", num_groups=None):
        super().__init__()
        assert norm == "gn", 'only the GroupNorm variant is instantiated by the reference (scenemodeling.py:154,174)'
        self.n1 = MinkowskiGroupNorm(num_groups, feat_dim)
        self.n2 = MinkowskiGroupNorm(num_groups, feat_dim)
        nn.init.constant_(self.n2.gn.weight, 0)
        self.conv1 = _SparseConv(feat_dim, feat_dim)
        self.conv2 = _SparseConv(feat_dim, feat_dim)


class SparseLevel:
    """Stand-in for the ``ME.SparseTensor`` stored under ``'sparse'`` in each level dict
    (scenemodeling.py:234): int32 coordinates + hash table + features + tensor stride."""

    def __init__(self, coords, stride):
        lib = _lib.load()
        self.coords = coords.contiguous()
        self.n = coords.shape[0]
        self.stride = stride
        nbytes = lib.v3d_hash_bytes(self.n)
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
        _lib.check(lib.v3d_hash_build(self.coords.data_ptr(), self.n, self.table.data_ptr(), nbytes,
                                      _lib.stream_ptr(coords.device)), 'v3d_hash_build')
        self.feats = None

    def check(self):
        """Raise if ``v3d_hash_build`` dropped rows (a coordinate outside the table's 16-bit key range): such voxels would
        silently lose their neighbours / interpolation corners.  Reads the table's status word (synchronises)."""
        _lib.check(_lib.load().v3d_hash_status(self.table.data_ptr(), self.n, _lib.stream_ptr(self.coords.device)),
                   'v3d_hash_build (stride %d)' % self.stride)
        return self

    def neighbors(self, out_coords, step):
        """[27, n_out] int32 row map: row of out_coords[p] + step * o_k in this level (or -1)."""
        lib = _lib.load()
        n_out = out_coords.shape[0]
        nbr = torch.empty((27, n_out), dtype=torch.int32, device=out_coords.device)
        _lib.check(lib.v3d_sparse_neighbors(self.table.data_ptr(), self.n, out_coords.data_ptr(), n_out,
                                            step, nbr.data_ptr(), _lib.stream_ptr(out_coords.device)),
                   'v3d_sparse_neighbors')
        return nbr


class LevelInfo(dict):
    """One level of ``SparseUNet.forward``'s result (scenemodeling.py:210-237): keys ``feats, pts, res, batch, idx, stride,
    sparse``.  ``pts`` (= idx * res + pts_min[batch], :225-226), ``idx`` and ``batch`` are derived from the level's coordinate
    map ON FIRST ACCESS: the HIP decoder consumes the hash table, the features and the level's minimum point only, so a scene
    sweep never launches the nine elementwise kernels that materialise them (they cost ~40 launches per scene)."""

    _LAZY = ('pts', 'idx', 'batch')

    def __init__(self, level, res, pts_min, like):
        super().__init__()
        self._level, self._res, self._pts_min, self._like = level, res, pts_min, like

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        c = self._level.coords
        if key == 'idx':
            v = c[:, 1:].type_as(self._like)
        elif key == 'batch':
            v = c[:, 0].type_as(self._like)
        else:
            v = self['idx'] * self._res + self._pts_min[self['batch']]
        self[key] = v
        return v

    def __contains__(self, key):
        return key in self._LAZY or super().__contains__(key)

    # Every view of the WHOLE dict materialises the lazy keys first, so that the object behaves like the reference's plain
    # seven-key dict for callers that copy or move a level (`{k: v.cpu() for k, v in x.items()}`, `dict(x)`, `x.get('pts')`).
    def _materialise(self):
        for k in self._LAZY:
            if not super().__contains__(k):
                self[k]

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._materialise()
        return super().keys()

    def items(self):
        self._materialise()
        return super().items()

    def values(self):
        self._materialise()
        return super().values()

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return dict.__len__(self) + sum(1 for k in self._LAZY if not dict.__contains__(self, k))

    def copy(self):
        return dict(self.items())


class SparseUNet(nn.Module):
    """Reference ``SparseUNet(dims, n_groups, n_res)`` (scenemodeling.py:147-237):
    ``forward(F, pts, idx, batch, res) -> list[dict(feats, pts, res, batch, idx, stride, sparse)]``,
    coarse -> fine."""

    def __init__(self, dims=(64, 128, 128), n_groups=(4, 8, 8), n_res=(1, 2, 3), precision='split_bf16'):
        super().__init__()
        _lib.precision_code(precision)
        self.precision = precision
        assert all(d % g == 0 and d // g in (8, 16) for d, g in zip(dims, n_groups)), \
            'the fused GroupNorm epilogue handles 8- and 16-channel groups (reference: 64/4, 128/8; 32/4 at feat_dim 16)'
        self.dims, self.n_groups, self.n_res = tuple(dims), tuple(n_groups), tuple(n_res)
        self.res_down = nn.ModuleList([nn.Sequential(*[SparseResidual3d(dims[i], "gn", n_groups[i])
                                                       for _ in range(n)]) for i, n in enumerate(n_res)])
        self.down = nn.ModuleList([nn.Sequential(_SparseConv(dims[i - 1], dims[i]),
                                                 MinkowskiGroupNorm(n_groups[i], dims[i]))
                                   for i in range(1, len(dims))])
        rn, rd, rg = n_res[::-1], dims[::-1], n_groups[::-1]
        self.res_up = nn.ModuleList([nn.Sequential(*[SparseResidual3d(rd[i + 1], "gn", rg[i + 1])
                                                     for _ in range(n)]) for i, n in enumerate(rn[1:])])
        self.up = nn.ModuleList([nn.Sequential(_SparseConv(rd[i - 1], rd[i]), MinkowskiGroupNorm(rg[i], rd[i]))
                                 for i in range(1, len(rd))])
        self.feat_adj = nn.ModuleList([nn.Sequential(_SparseConv(2 * rd[i], rd[i], kernel_size=1),
                                                     MinkowskiGroupNorm(rg[i], rd[i]))
                                       for i in range(1, len(rd))])
        self._cache = _PackCache(self)

    # -- weight packing ---------------------------------------------------------------------------
    def _pack3(self, conv, norm):
        _, ci, co = conv.kernel.shape
        return PackedGemm(conv.kernel, ci * co, 1, co, 27, co, ci, gn_w=norm.gn.weight, gn_b=norm.gn.bias,
                          device=self._dev, gn_group=norm.gn.num_channels // norm.gn.num_groups)

    def _build(self):
        g = {}
        for name, lst in (('res_down', self.res_down), ('res_up', self.res_up)):
            for i, seq in enumerate(lst):
                for l, blk in enumerate(seq):
                    g[(name, i, l)] = (self._pack3(blk.conv1, blk.n1), self._pack3(blk.conv2, blk.n2))
        for i, seq in enumerate(self.down):
            g[('down', i)] = self._pack3(seq[0], seq[1])
        for i, seq in enumerate(self.up):
            g[('up', i)] = self._pack3(seq[0], seq[1])
        for i, seq in enumerate(self.feat_adj):
            c2, co = seq[0].kernel.shape
            g[('feat_adj', i)] = PackedGemm(seq[0].kernel, (c2 // 2) * co, 1, co, 2, co, c2 // 2,
                                            gn_w=seq[1].gn.weight, gn_b=seq[1].gn.bias, device=self._dev,
                                            gn_group=seq[1].gn.num_channels // seq[1].gn.num_groups)
        return g

    # -- execution ----------------------------------------------------------------------------------
    @staticmethod
    def _strided_coords(level):
        """unique(floor(c / 2ts) * 2ts) in lexicographic (batch, x, y, z) order: packed 64-bit keys,
        radix sort + unique in the library, unpack."""
        from .utils import sort_unique_u64
        lib = _lib.load()
        dev, stream = level.coords.device, _lib.stream_ptr(level.coords.device)
        keys = torch.empty(level.n, dtype=torch.int64, device=dev)
        _lib.check(lib.v3d_strided_keys(level.coords.data_ptr(), level.n, level.stride, keys.data_ptr(), stream),
                   'v3d_strided_keys')
        uniq = sort_unique_u64(keys)
        out = torch.empty((uniq.shape[0], 4), dtype=torch.int32, device=dev)
        _lib.check(lib.v3d_unpack_coords(uniq.data_ptr(), uniq.shape[0], out.data_ptr(), stream), 'v3d_unpack_coords')
        return out

    def _conv(self, pack, x, nbr, n_out, eps, residual=None):
        # every offset reads `x` through column k of the neighbour table: one call with 4 pointers (v3d_sparse_conv_f32) instead
        # of three ctypes arrays of 27 entries per convolution -- the U-Net forward was bound by this host code, not the kernels
        if x.device != pack.device:
            raise _lib.V3DLibraryError('sparse convolution: weights were packed on %s, input lives on %s' % (pack.device, x.device))
        if not x.is_contiguous():
            x = x.contiguous()
        y = torch.empty((n_out, pack.N), dtype=torch.float32, device=x.device)
        rc = _lib.load().v3d_sparse_conv_f32(pack.handle, n_out, x.data_ptr(), x.shape[1], nbr.data_ptr(), nbr.shape[1],
                                           pack.gn_group, eps, residual.data_ptr() if residual is not None else None,
                                           residual.shape[1] if residual is not None else 0, 1, y.data_ptr(), pack.N,
                                           self._prec_code, self._stream)
        if rc:
            _lib.check(rc, 'v3d_sparse_conv_f32')
        return y

    def _residual(self, packs, blk, x, nbr):
        n = x.shape[0]
        y = self._conv(packs[0], x, nbr, n, blk.n1.gn.eps)
        return self._conv(packs[1], y, nbr, n, blk.n2.gn.eps, residual=x)

    def flush_checks(self):
        """Raise for every level built since the last call whose hash table dropped rows (``SparseLevel.check``)."""
        pending, self._pending_checks = getattr(self, '_pending_checks', []), []
        for lv in pending:
            lv.check()

    def forward(self, F, pts, idx, batch, res, n_batches=None, defer_checks=False, idx_min_zero=False):
        if not F.is_cuda:
            raise _lib.V3DLibraryError('SparseUNet: tensors must live on a HIP device (no CPU fallback)')
        self._dev = F.device
        # (plain integers: a module attribute must survive copy.deepcopy / pickling, a ctypes library handle would not)
        self._stream, self._prec_code = _lib.stream_ptr(F.device), _lib.precision_code(self.precision)
        g = self._cache.get(self._build, F.device)
        coords = torch.cat((batch.unsqueeze(1), idx), dim=1).int().contiguous()       # [N,4] (b,x,y,z)
        # coordinate maps: stride-2 conv output = unique(floor(c / 2ts) * 2ts), lexicographic order
        levels = [SparseLevel(coords, 1)]
        for i in range(1, len(self.dims)):
            levels.append(SparseLevel(self._strided_coords(levels[-1]), 2 * levels[-1].stride))
        same = [lv.neighbors(lv.coords, lv.stride) for lv in levels]                    # stride-1 maps
        x = F.contiguous().float()
        xs = []
        for i in range(len(self.dims)):
            if i > 0:
                nbr = levels[i - 1].neighbors(levels[i].coords, levels[i - 1].stride)   # stride-2 conv
                x = self._conv(g[('down', i - 1)], x, nbr, levels[i].n, self.down[i - 1][1].gn.eps)
            for l, blk in enumerate(self.res_down[i]):
                x = self._residual(g[('res_down', i, l)], blk, x, same[i])
            xs.append(x)
        lv_rev, xs_rev, same_rev = levels[::-1], xs[::-1], same[::-1]
        out = [(lv_rev[0], xs_rev[0])]
        x = xs_rev[0]
        for i in range(len(self.dims) - 1):
            tgt = lv_rev[i + 1]
            nbr = lv_rev[i].neighbors(tgt.coords, -tgt.stride)                          # transposed conv
            u = self._conv(g[('up', i)], x, nbr, tgt.n, self.up[i][1].gn.eps)
            x = g[('feat_adj', i)](tgt.n, [u, xs_rev[i + 1]], use_gn=True,
                                   gn_eps=self.feat_adj[i][1].gn.eps, relu_out=True,
                                   precision=self.precision)                           # 1x1 on ME.cat
            for l, blk in enumerate(self.res_up[i]):
                x = self._residual(g[('res_up', i, l)], blk, x, same_rev[i + 1])
            out.append((tgt, x))

        if n_batches is None:
            n_batches = int(torch.max(batch).item()) + 1                               # scenemodeling.py:221
        # rows the hash tables refused (range check): read the status words now (one readback each, after the last kernel of
        # the forward: the host cannot run ahead into the caller's next launches), or -- a driver that calls flush_checks()
        # before it uses the results -- at the next point where the host waits for the device anyway
        self._pending_checks = getattr(self, '_pending_checks', []) + list(levels)
        if not defer_checks:
            self.flush_checks()
        # scenemodeling.py:222-231 per batch element b: pts_min = pts[batch == b][0] - idx[batch == b][0] * res and
        # x_pts = x_idx * res + pts_min -- the same numbers without a boolean-mask gather (and its host synchronisation) per
        # level and batch element: the first row of every batch element.  One batch element (a scene): that is row 0.  A few:
        # the first True of a [n_batches, N] comparison (argmax returns the first maximum); a scatter-min of the row index
        # serialises its atomics on n_batches addresses (1.3 ms for 60 k rows of one scene) and is kept for many small batch
        # elements only
        if n_batches == 1:
            pts_min = pts[0:1] - idx[0:1] * res                                        # [1, 3]
        else:
            if n_batches <= 32:
                ids = torch.arange(n_batches, device=batch.device, dtype=batch.dtype)
                first = (batch[None, :] == ids[:, None]).to(torch.uint8).argmax(dim=1)
            else:
                rows = torch.arange(batch.shape[0], device=batch.device)
                first = torch.full((n_batches,), batch.shape[0], dtype=rows.dtype, device=batch.device) \
                    .scatter_reduce_(0, batch.to(rows.dtype), rows, 'amin')
            pts_min = pts[first] - idx[first] * res                                    # [n_batches, 3]
        # One batch element: x_pts is monotone in x_idx per axis (fl(fl(i * res) + pts_min), res > 0), so the minimum of a level's
        # x_pts is that formula at the level's minimum index, and a level of stride s holds floor(c / s) * s of the finest
        # coordinates c: ONE integer reduction over the finest level instead of a float reduction per level -- and none at all
        # when the caller vouches that every axis' minimum index is 0 (``idx_min_zero``: utils.voxelize shifts them there,
        # utils.py:61-62), in which case the minimum point of every level IS pts_min
        idx_min = None
        fine = min(levels, key=lambda l: l.stride)
        if n_batches == 1 and not idx_min_zero:
            idx_min = fine.coords[:, 1:].amin(dim=0, keepdim=True)                      # [1, 3] int32
        out_info = []
        for lv, xf in out:
            lv.feats = xf
            info = LevelInfo(lv, res, pts_min, batch)
            # scatter(x.pts, x.batch, reduce='min') of the interpolation (refinement.py:33), here where the number of batch
            # elements is known: the decoder would have to read it back from the device in front of its first launch
            if n_batches == 1 and idx_min_zero:
                min_pts = pts_min
            elif n_batches == 1:
                lv_min = torch.div(idx_min, lv.stride, rounding_mode='floor') * lv.stride if lv.stride != fine.stride else idx_min
                min_pts = lv_min.type_as(batch) * res + pts_min[0:1]
            else:
                sel = info['batch'][None, :, None] == torch.arange(n_batches, device=batch.device)[:, None, None]
                min_pts = torch.where(sel, info['pts'][None], info['pts'].new_full((), float('inf'))).amin(dim=1)
            info.update({'feats': xf, 'res': lv.stride * res, 'stride': lv.stride, 'sparse': lv, '_min_pts': min_pts})
            out_info.append(info)
        return out_info
