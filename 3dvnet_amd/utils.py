"""Host-side mirror of the hot-path helpers of ``mv3d/utils.py``.

``voxelize`` is index bookkeeping (bounding box, integer voxel ids, sorted unique, decode); it runs as
PyTorch device ops on the tensors' own device exactly as in the reference -- including the reference's
mix of a ceil-based grid size for decoding (utils.py:41) with torch_cluster's trunc+1 cell counts for
encoding.  ``torch_geometric.nn.voxel_grid`` and ``torch_scatter`` (un-vendored third-party packages)
are restated inline.
"""
import torch


def slice_edges(edges, index_start, index_end, slice_dim=0):
    """Row H1 (utils.py:32-35): keep edge columns whose ``edges[slice_dim]`` is in [start, end)."""
    keep = (edges[slice_dim] >= index_start) & (edges[slice_dim] < index_end)
    return edges[:, keep]


def _voxel_grid(pos, batch, size, start, end):
    """PyG 1.6.3 voxel_grid -> torch_cluster 1.5.8 grid: id = sum_d trunc((p_d - start_d)/size_d) *
    prod_{d'<d}(trunc((end_d' - start_d')/size_d') + 1), batch appended as 4th coordinate (size 1)."""
    pos = torch.cat([pos, batch.unsqueeze(-1).type_as(pos)], dim=-1)
    size_t = torch.tensor([float(size)] * 3 + [1.0], dtype=pos.dtype, device=pos.device)
    start_t = torch.cat([start.type_as(pos), pos.new_zeros(1)])
    end_t = torch.cat([end.type_as(pos), batch.max().type_as(pos).view(1)])
    num = ((end_t - start_t) / size_t).to(torch.long) + 1
    cum = num.cumprod(0)
    cum = torch.cat([cum.new_ones(1), cum[:-1]])
    return (((pos - start_t) / size_t).to(torch.long) * cum).sum(1)


def _scatter_min(src, index, dim_size):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce_(0, idx, src, 'amin', include_self=False)


def voxelize(pts, pts_batch, edge_len):
    """Row B3 (utils.py:38-64): -> (anchor_pts [Nv,3] f32, anchor_idx3d [Nv,3] int32,
    anchor_batch [Nv] int64, anchor_pts_edges [2,Np] int64)."""
    bbox_min = pts.min(dim=0)[0]
    bbox_max = pts.max(dim=0)[0]
    grid_size = torch.ceil((bbox_max - bbox_min) / edge_len).long()
    max_grid_idx = grid_size[0] * grid_size[1] * grid_size[2]
    voxel_idx = _voxel_grid(pts, pts_batch, edge_len, bbox_min, bbox_max)
    anchor_idx, inv_idx = torch.unique(voxel_idx, return_inverse=True)
    anchor_pts_edges = torch.stack((inv_idx, torch.arange(pts.shape[0], dtype=torch.long, device=pts.device)), dim=0)
    # scatter(pts_batch, inv, reduce='min') (:50): the voxel id encodes the batch, so every point of a
    # voxel carries the same batch id and a plain (non-atomic) scatter gives the same result
    anchor_batch = torch.empty(anchor_idx.shape[0], dtype=pts_batch.dtype, device=pts.device)
    anchor_batch.scatter_(0, inv_idx, pts_batch)
    anchor_idx = anchor_idx - anchor_batch * max_grid_idx
    anchor_idx3d = torch.zeros((anchor_idx.shape[0], 3), dtype=torch.int, device=pts.device)
    anchor_idx3d[:, 2] = anchor_idx // (grid_size[0] * grid_size[1])
    anchor_idx3d[:, 1] = (anchor_idx - anchor_idx3d[:, 2] * (grid_size[0] * grid_size[1])) // (grid_size[0])
    anchor_idx3d[:, 0] = (anchor_idx - anchor_idx3d[:, 2] * (grid_size[0] * grid_size[1])) % (grid_size[0])
    anchor_pts = anchor_idx3d * edge_len + bbox_min + edge_len / 2.
    n_batches = int(anchor_batch.max().item()) + 1
    min_idx3d = torch.stack([anchor_idx3d[anchor_batch == b].amin(dim=0) for b in range(n_batches)]) \
        if n_batches > 1 else anchor_idx3d.amin(dim=0, keepdim=True)              # scatter-min (:61)
    anchor_idx3d = anchor_idx3d - min_idx3d[anchor_batch]
    return anchor_pts, anchor_idx3d, anchor_batch, anchor_pts_edges
