"""Host-side mirror of the hot-path helpers of ``mv3d/utils.py``.

``voxelize`` runs in ``lib3dvnet_hip.so`` (csrc/voxelize.hip): bounding box, voxel ids, radix sort +
unique, inverse map and decode are device kernels that restate ``utils.voxelize`` literally -- including
the reference's mix of a ceil-based grid size for decoding (utils.py:41) with torch_cluster's trunc+1
cell counts for encoding (``torch_geometric.nn.voxel_grid`` is an un-vendored third-party call).
"""
import ctypes

import torch

from . import _lib


def slice_edges(edges, index_start, index_end, slice_dim=0):
    """Row H1 (utils.py:32-35): keep edge columns whose ``edges[slice_dim]`` is in [start, end)."""
    keep = (edges[slice_dim] >= index_start) & (edges[slice_dim] < index_end)
    return edges[:, keep]


def sort_unique_u64(keys):
    """Ascending unique values of an int64 key tensor (device) -> int64 tensor [n_unique]."""
    lib = _lib.load()
    dev, n = keys.device, keys.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=dev)
    nbytes = lib.v3d_sort_unique_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cnt = ctypes.c_int(0)
    rc = lib.v3d_sort_unique_u64(keys.data_ptr(), n, out.data_ptr(), ctypes.byref(cnt), ws.data_ptr(), nbytes,
                                 _lib.stream_ptr(dev))
    _lib.check(rc, 'v3d_sort_unique_u64')
    return out[:cnt.value]


def voxelize(pts, pts_batch, edge_len):
    """Row B3 (utils.py:38-64): -> (anchor_pts [Nv,3] f32, anchor_idx3d [Nv,3] int32,
    anchor_batch [Nv] int64, anchor_pts_edges [2,Np] int64)."""
    if not pts.is_cuda:
        raise _lib.V3DLibraryError('voxelize: tensors must live on a HIP device (no CPU fallback)')
    lib = _lib.load()
    dev, n = pts.device, pts.shape[0]
    stream = _lib.stream_ptr(dev)
    pts = pts.contiguous().float()
    pts_batch = pts_batch.contiguous().long()
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    wbytes = lib.v3d_voxelize_workspace_bytes()
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.v3d_voxel_keys(pts.data_ptr(), pts_batch.data_ptr(), n, float(edge_len), keys.data_ptr(),
                                  ws.data_ptr(), wbytes, stream), 'v3d_voxel_keys')
    uniq = sort_unique_u64(keys)                                              # torch.unique (:48)
    # range checks of the fixed-size device tables (batch ids, cells per axis); the stream is already synchronised
    _lib.check(lib.v3d_voxelize_status(ws.data_ptr(), wbytes, stream), 'voxelize')
    nv = uniq.shape[0]
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(lib.v3d_lower_bound_u64(uniq.data_ptr(), nv, keys.data_ptr(), n, inv.data_ptr(), stream),
               'v3d_lower_bound_u64')
    anchor_pts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
    anchor_idx3d = torch.empty((nv, 3), dtype=torch.int32, device=dev)
    anchor_batch = torch.empty(nv, dtype=torch.int64, device=dev)
    _lib.check(lib.v3d_voxel_decode(uniq.data_ptr(), nv, float(edge_len), float(edge_len / 2.),
                                    anchor_pts.data_ptr(), anchor_idx3d.data_ptr(), anchor_batch.data_ptr(),
                                    ws.data_ptr(), wbytes, stream), 'v3d_voxel_decode')
    anchor_pts_edges = torch.stack((inv, torch.arange(n, dtype=torch.long, device=dev)), dim=0)
    return anchor_pts, anchor_idx3d, anchor_batch, anchor_pts_edges
