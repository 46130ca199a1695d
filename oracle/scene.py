"""Oracle for SURVEY.md §8a rows B1-B6, C1-C3, H1, H4: back-projection + per-point variance,
voxelisation, PointNet, sparse 3D U-Net (MinkowskiEngine semantics restated), sparse trilinear
interpolation, hypothesis decoder.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: mv3d/lightningmodel.py:132-242, mv3d/utils.py:32-83, mv3d/subnetworks/scenemodeling.py,
mv3d/subnetworks/refinement.py, mv3d/eval/metricfunctions.py:26-41.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .costvolume import project_to_grid, scatter_mean


# ----------------------------------------------------------------------------------------------
# H1, H4
# ----------------------------------------------------------------------------------------------

def slice_edges(edges, index_start, index_end, slice_dim=0):
    """Row H1 (utils.py:32-35): keep edge columns with edges[slice_dim] in [start, end)."""
    keep = (edges[slice_dim] >= index_start) & (edges[slice_dim] < index_end)
    return edges[:, keep]


def abs_rel(depth_pred, depth_gt):
    """Row H4 (metricfunctions.py:29-41)."""
    valid = ((depth_gt >= 0.5) & (depth_gt < 65.)).float()
    denom = torch.sum(valid, dim=(1, 2)) + 1e-7
    abs_diff = torch.abs(depth_pred - depth_gt)
    return torch.mean(torch.sum((abs_diff / (depth_gt + 1e-7)) * valid, dim=(1, 2)) / denom)


# ----------------------------------------------------------------------------------------------
# B1, B2, C1, C3: back-projection, re-projection, per-point variance
# ----------------------------------------------------------------------------------------------

def build_img_pts(img_size, plane_size):
    """Row B1 (utils.py:67-77): homogeneous pixel grid [3, h*w], row-major (y outer)."""
    xs = np.linspace(0, img_size[1] - 1, plane_size[1], dtype=np.float32)
    ys = np.linspace(0, img_size[0] - 1, plane_size[0], dtype=np.float32)
    xx, yy = np.meshgrid(xs, ys)
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    return torch.from_numpy(np.stack((xx, yy, np.ones_like(xx))))


def _variance_over_edges(img_feats, pts, rotmats, tvecs, K, ref_src_edges, gather_idx, img_size):
    """Shared by B2 / C1 (lightningmodel.py:147-169, 207-229).  pts: [n_ref, 3, n_pts] world points.
    Returns x_var [n_ref, C, n_pts]."""
    grid = project_to_grid(pts[gather_idx], rotmats, tvecs, K, ref_src_edges[1], img_size)
    x = F.grid_sample(img_feats[ref_src_edges[1]], grid, mode='bilinear', align_corners=True)
    x = x.squeeze(3)
    n_ref = pts.shape[0]
    x_avg = scatter_mean(x, gather_idx, n_ref)
    x_avg_sq = scatter_mean(x ** 2, gather_idx, n_ref)
    return x_avg_sq - x_avg ** 2


def feature_rich_pointcloud(depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges,
                            img_size, pinned=False):
    """Row B2 (lightningmodel.py:132-174): -> pts [Np,3], pts_feat [Np,C], pts_batch [Np].  ``pinned``: the back-projection
    with the host-independent evaluation orders of oracle/pinned.py instead of this host's torch.bmm."""
    ref_idx, gather_idx = torch.unique(ref_src_edges[0], return_inverse=True)
    n_imgs = depth_pred.shape[0]
    if pinned:
        from oracle import pinned as opin
        pts = opin.backproject_points(K[ref_idx], rotmats[ref_idx], tvecs[ref_idx], depth_pred, img_size)
    else:
        K_inv = torch.inverse(K[ref_idx])
        R_T = rotmats[ref_idx].transpose(2, 1)
        pts_img = build_img_pts(img_size, depth_pred.shape[1:])[None].repeat(n_imgs, 1, 1)
        pts_img = pts_img * depth_pred.reshape(n_imgs, 1, -1)
        pts = torch.bmm(R_T, torch.bmm(K_inv, pts_img) - tvecs[ref_idx].unsqueeze(-1))
    x_var = _variance_over_edges(img_feats, pts, rotmats, tvecs, K, ref_src_edges, gather_idx, img_size)
    C = img_feats.shape[1]
    P = depth_pred.shape[1] * depth_pred.shape[2]
    pts_flat = pts.transpose(2, 1).reshape(-1, 3)
    pts_feat = x_var.transpose(2, 1).reshape(-1, C)
    pts_batch = depth_batch.unsqueeze(1).expand(n_imgs, P).reshape(-1)
    return pts_flat, pts_feat, pts_batch


def pointflow_hypotheses(depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges, offset,
                         n, img_size, pinned=False):
    """Row C1 (lightningmodel.py:187-235): hypothesis points depth + i*offset, i in [-n, n], and their
    multi-view variance features.  -> pts_hyp [n_ref*P, 2n+1, 3], pts_feat [n_ref*P, 2n+1, C],
    pts_batch [n_ref*P]."""
    n_imgs = depth_pred.shape[0]
    ref_idx, gather_idx = torch.unique(ref_src_edges[0], return_inverse=True)
    K_inv = torch.inverse(K[ref_idx])
    R_T = rotmats[ref_idx].transpose(2, 1)
    pts_img = build_img_pts(img_size, depth_pred.shape[1:])[None].repeat(n_imgs, 1, 1)
    n_pts = pts_img.shape[2]
    pts_batch = depth_batch.unsqueeze(1).expand(n_imgs, n_pts).reshape(-1)
    pts_hyp = torch.empty((n_imgs, 3, 2 * n + 1, n_pts), dtype=torch.float32)
    for i in range(-n, n + 1):
        if pinned:
            from oracle import pinned as opin
            pts_h = opin.backproject_points(K[ref_idx], rotmats[ref_idx], tvecs[ref_idx], depth_pred + i * offset, img_size)
        else:
            pts_h = pts_img * (depth_pred.reshape(n_imgs, 1, -1) + i * offset)
            pts_h = torch.bmm(R_T, torch.bmm(K_inv, pts_h) - tvecs[ref_idx].unsqueeze(-1))
        pts_hyp[..., i + n, :] = pts_h
    n_hpts = (2 * n + 1) * n_pts
    x_var = _variance_over_edges(img_feats, pts_hyp.view(n_imgs, 3, n_hpts), rotmats, tvecs, K,
                                 ref_src_edges, gather_idx, img_size)
    C = img_feats.shape[1]
    pts_feat = x_var.view(n_imgs, C, 2 * n + 1, n_pts).permute(0, 3, 2, 1).reshape(n_pts * n_imgs, 2 * n + 1, C)
    pts_hyp = pts_hyp.permute(0, 3, 2, 1).reshape(n_pts * n_imgs, 2 * n + 1, 3)
    return pts_hyp, pts_feat, pts_batch


def offset_expectation(preds, offset, n, out_shape):
    """Row C3 (lightningmodel.py:238-242)."""
    vals = torch.linspace(-n * offset, n * offset, 2 * n + 1).unsqueeze(0)
    return torch.sum(vals * preds, dim=1).view(out_shape)


# ----------------------------------------------------------------------------------------------
# B3: voxelize
# ----------------------------------------------------------------------------------------------

def voxel_grid_ids(pos, batch, size, start, end):
    """torch_geometric 1.6.3 voxel_grid -> torch_cluster 1.5.8 grid, restated (not vendored by the
    reference): batch appended as a 4th coordinate with cell size 1; per dimension
    i = trunc((p - start)/size), n = trunc((end - start)/size) + 1; id = sum_d i_d prod_{d'<d} n_d'."""
    pos = torch.cat([pos, batch.unsqueeze(-1).type_as(pos)], dim=-1)
    size_t = torch.tensor([float(size)] * 3 + [1.0], dtype=pos.dtype)
    start_t = torch.cat([start.type_as(pos), pos.new_zeros(1)])
    end_t = torch.cat([end.type_as(pos), batch.max().type_as(pos).view(1)])
    num = ((end_t - start_t) / size_t).to(torch.long) + 1
    cum = num.cumprod(0)
    cum = torch.cat([cum.new_ones(1), cum[:-1]])
    c = ((pos - start_t) / size_t).to(torch.long)
    return (c * cum).sum(1)


def _scatter_min(src, index, dim_size):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce_(0, idx, src, 'amin', include_self=False)


def voxelize(pts, pts_batch, edge_len):
    """Row B3 (utils.py:38-64), literally: the decode uses ceil-based grid_size (:41) while the
    encode uses torch_cluster's trunc+1 counts -- they agree unless an extent is an exact multiple
    of edge_len (synthetic scenes are jittered so it never is)."""
    bbox_min = pts.min(dim=0)[0]
    bbox_max = pts.max(dim=0)[0]
    grid_size = torch.ceil((bbox_max - bbox_min) / edge_len).long()
    max_grid_idx = grid_size[0] * grid_size[1] * grid_size[2]
    voxel_idx = voxel_grid_ids(pts, pts_batch, edge_len, bbox_min, bbox_max)
    anchor_idx, inv_idx = torch.unique(voxel_idx, return_inverse=True)
    anchor_pts_edges = torch.stack((inv_idx, torch.arange(pts.shape[0], dtype=torch.long)), dim=0)
    anchor_batch = _scatter_min(pts_batch, anchor_pts_edges[0], anchor_idx.shape[0])
    anchor_idx = anchor_idx - anchor_batch * max_grid_idx
    anchor_idx3d = torch.zeros((anchor_idx.shape[0], 3), dtype=torch.int)
    anchor_idx3d[:, 2] = anchor_idx // (grid_size[0] * grid_size[1])
    anchor_idx3d[:, 1] = (anchor_idx - anchor_idx3d[:, 2] * (grid_size[0] * grid_size[1])) // (grid_size[0])
    anchor_idx3d[:, 0] = (anchor_idx - anchor_idx3d[:, 2] * (grid_size[0] * grid_size[1])) % (grid_size[0])
    anchor_pts = anchor_idx3d * edge_len + bbox_min + edge_len / 2.
    min_idx3d = _scatter_min(anchor_idx3d, anchor_batch, int(anchor_batch.max()) + 1)
    anchor_idx3d = anchor_idx3d - min_idx3d[anchor_batch]
    return anchor_pts, anchor_idx3d, anchor_batch, anchor_pts_edges


# ----------------------------------------------------------------------------------------------
# B4: PointNet
# ----------------------------------------------------------------------------------------------

def _scatter_max(src, index, dim_size):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view(-1, 1).expand_as(src)
    return out.scatter_reduce_(0, idx, src, 'amax', include_self=False)


def pointnet(pts, idx, n_idx, sd):
    """Row B4 (scenemodeling.py:127-144).  pts [Np, in_dim], idx [Np] voxel of each point."""
    lin = lambda x, name: F.linear(x, sd[name + '.weight'], sd[name + '.bias'])
    x = lin(F.relu(lin(pts, 'fc_pos')), 'fc1')
    for name in ('fc2', 'fc3', 'fc4'):
        pool = _scatter_max(x, idx, n_idx)
        x = lin(F.relu(torch.cat((x, pool[idx]), dim=1)), name)
    pool = _scatter_max(x, idx, n_idx)
    return lin(F.relu(pool), 'fc_out')


# ----------------------------------------------------------------------------------------------
# B6: sparse 3D U-Net -- MinkowskiEngine 0.5 semantics restated.
# A sparse tensor is (coords [N,4] int64 = (batch, x, y, z), feats [N,C], tensor_stride).
# ----------------------------------------------------------------------------------------------

_KEY_M = 1 << 14          # coordinates (plus a 4-voxel guard) stay below this
_GUARD = 8


def _keys(c):
    return ((c[:, 0] * _KEY_M + c[:, 1] + _GUARD) * _KEY_M + c[:, 2] + _GUARD) * _KEY_M + c[:, 3] + _GUARD


def _lookup(coords, query):
    """Row index of each query coordinate in `coords` (unique rows), or -1."""
    k = _keys(coords)
    order = torch.argsort(k)
    ks = k[order]
    q = _keys(query)
    pos = torch.searchsorted(ks, q).clamp(max=ks.shape[0] - 1)
    hit = ks[pos] == q
    return torch.where(hit, order[pos], torch.full_like(pos, -1))


def kernel_offsets():
    """k = (ox+1) + 3 (oy+1) + 9 (oz+1): first spatial dimension fastest (Appendix A)."""
    o = []
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                o.append((ox, oy, oz))
    return torch.tensor(o, dtype=torch.long)


def strided_coords(coords, ts):
    """Output coordinate map of a stride-2 conv on a tensor of stride ts: unique(floor(c/(2ts))*2ts),
    sorted lexicographically by (batch, x, y, z)."""
    c = coords.clone()
    c[:, 1:] = torch.div(c[:, 1:], 2 * ts, rounding_mode='floor') * (2 * ts)
    return torch.unique(c, dim=0)


def sparse_conv(coords, feats, ts, kernel, stride=1):
    """MinkowskiConvolution(k=3, stride 1|2, no bias): out[p] = sum_{o: p + o*ts in C_in} in[p+o*ts] @ W[k(o)].
    Returns (out_coords, out_feats, ts_out)."""
    out_coords = coords if stride == 1 else strided_coords(coords, ts)
    out = torch.zeros((out_coords.shape[0], kernel.shape[-1]), dtype=feats.dtype)
    if kernel.dim() == 2:            # kernel_size 1
        return out_coords, feats @ kernel, ts
    for k, o in enumerate(kernel_offsets()):
        q = out_coords.clone()
        q[:, 1:] += o * ts
        src = _lookup(coords, q)
        m = src >= 0
        out[m] += feats[src[m]] @ kernel[k]
    return out_coords, out, ts * stride


def sparse_conv_transpose(coords, feats, ts, kernel, out_coords):
    """MinkowskiConvolutionTranspose(k=3, stride 2) onto the existing coordinate map `out_coords` at
    stride ts/2: out[p] = sum_{o: p - o*ts_out in C_in} in[p - o*ts_out] @ W[k(o)]."""
    ts_out = ts // 2
    out = torch.zeros((out_coords.shape[0], kernel.shape[-1]), dtype=feats.dtype)
    for k, o in enumerate(kernel_offsets()):
        q = out_coords.clone()
        q[:, 1:] -= o * ts_out
        src = _lookup(coords, q)
        m = src >= 0
        out[m] += feats[src[m]] @ kernel[k]
    return out, ts_out


def row_group_norm(x, sd, prefix, groups, eps=1e-5):
    """MinkowskiGroupNorm (scenemodeling.py:78-104): torch.nn.GroupNorm(G, C) on the [N, C] matrix,
    i.e. every row (voxel) is normalised independently over its C/G-channel groups."""
    return F.group_norm(x, groups, sd[prefix + '.gn.weight'], sd[prefix + '.gn.bias'], eps)


def sparse_residual(coords, feats, ts, sd, prefix, groups):
    """SparseResidual3d(norm='gn') (scenemodeling.py:16-44)."""
    _, y, _ = sparse_conv(coords, feats, ts, sd[prefix + '.conv1.kernel'])
    y = F.relu(row_group_norm(y, sd, prefix + '.n1', groups))
    _, y, _ = sparse_conv(coords, y, ts, sd[prefix + '.conv2.kernel'])
    y = row_group_norm(y, sd, prefix + '.n2', groups)
    return F.relu(y + feats)


def sparse_unet(feat, pts, idx, batch, res, sd, dims=(64, 128, 128), n_groups=(4, 8, 8), n_res=(1, 2, 3)):
    """Row B6 (scenemodeling.py:191-237).  feat [Nv, dims[0]], pts [Nv,3] voxel centres, idx [Nv,3]
    int voxel indices (per-batch min = 0), batch [Nv], res = voxel edge length.  Returns the list of
    3 level dicts, coarse -> fine (feats, pts, res, batch, idx, stride, coords)."""
    coords = torch.cat((batch.unsqueeze(1).long(), idx.long()), dim=1)
    levels = []
    x, c, ts = feat, coords, 1
    for i, n in enumerate(n_res):
        if i > 0:
            c, x, ts = sparse_conv(c, x, ts, sd['down.%d.0.kernel' % (i - 1)], stride=2)
            x = F.relu(row_group_norm(x, sd, 'down.%d.1' % (i - 1), n_groups[i]))
        for l in range(n):
            x = sparse_residual(c, x, ts, sd, 'res_down.%d.%d' % (i, l), n_groups[i])
        levels.append((c, x, ts))
    levels = levels[::-1]
    rd, rg, rn = dims[::-1], n_groups[::-1], n_res[::-1]
    out = [levels[0]]
    c, x, ts = levels[0]
    for i in range(len(rd) - 1):
        c_skip, x_skip, ts_skip = levels[i + 1]
        x, ts = sparse_conv_transpose(c, x, ts, sd['up.%d.0.kernel' % i], c_skip)
        x = F.relu(row_group_norm(x, sd, 'up.%d.1' % i, rg[i + 1]))
        x = torch.cat((x, x_skip), dim=1)                                    # ME.cat (:206)
        x = x @ sd['feat_adj.%d.0.kernel' % i]
        x = F.relu(row_group_norm(x, sd, 'feat_adj.%d.1' % i, rg[i + 1]))
        c = c_skip
        for l in range(rn[i + 1]):
            x = sparse_residual(c, x, ts, sd, 'res_up.%d.%d' % (i, l), rg[i + 1])
        out.append((c, x, ts))
    info = []
    n_batches = int(batch.max()) + 1
    for c, x, ts in out:
        x_idx, x_batch = c[:, 1:], c[:, 0]
        x_pts = torch.empty((c.shape[0], 3), dtype=torch.float32)
        for b in range(n_batches):
            bin_, bout = batch == b, x_batch == b
            pts_min = pts[bin_][0] - (idx[bin_][0] * res)                   # centre of voxel (0,0,0)
            x_pts[bout] = x_idx[bout] * res + pts_min
        info.append({'feats': x, 'pts': x_pts, 'res': ts * res, 'batch': x_batch, 'idx': x_idx,
                     'stride': ts, 'coords': c})
    return info


# ----------------------------------------------------------------------------------------------
# C2a / C2b: hypothesis decoder
# ----------------------------------------------------------------------------------------------

def sparse_interpolate(coords, feats, ts, query):
    """MinkowskiInterpolation restated (Appendix A): query [Nq, 4] float (batch, x, y, z) in base-voxel
    units; 8 corners floor(q/ts)*ts + {0,ts}^3, weights prod(1 - |q - c|/ts), absent corners add 0."""
    b = query[:, 0].long()
    q = query[:, 1:]
    lo = torch.floor(q / ts) * ts
    out = torch.zeros((query.shape[0], feats.shape[1]), dtype=feats.dtype)
    for dz in (0, ts):
        for dy in (0, ts):
            for dx in (0, ts):
                c = lo + torch.tensor([dx, dy, dz], dtype=q.dtype)
                w = torch.prod(1 - torch.abs(q - c) / ts, dim=1)
                src = _lookup(coords, torch.cat((b.unsqueeze(1), c.long()), dim=1))
                m = src >= 0
                out[m] += w[m].unsqueeze(1) * feats[src[m]]
    return out


def decoder_features(xs, pts, pts_feat, pts_batch):
    """Row C2a (refinement.py:28-41): [Nq, n_hyp, sum(C_level) (+ C_feat)], finest level first."""
    n_pts, n_hyp = pts.shape[:2]
    features = pts_feat
    for x in xs:
        min_pts = _scatter_min(x['pts'], x['batch'], int(x['batch'].max()) + 1)
        pts_idx = pts - min_pts[pts_batch].unsqueeze(1).expand(*pts.shape)
        pts_idx = (pts_idx / x['res']) * x['stride']
        b = pts_batch.unsqueeze(1).repeat(1, n_hyp).unsqueeze(2).float()
        q = torch.cat((b, pts_idx), dim=2).view(n_pts * n_hyp, 4)
        feats = sparse_interpolate(x['coords'], x['feats'], x['stride'], q).view(n_pts, n_hyp, -1)
        features = feats if features is None else torch.cat((feats, features), dim=2)
    return features


def decoder_net(features, sd, eps=1e-5):
    """Row C2b (refinement.py:16-25,42-43): features [Nq, n_hyp, C] -> softmax scores [Nq, n_hyp]."""
    x = features.transpose(2, 1)
    for i in range(3):
        x = F.conv1d(x, sd['net.%d.0.weight' % i], None, 1, 1)
        x = F.batch_norm(x, sd['net.%d.1.running_mean' % i], sd['net.%d.1.running_var' % i],
                         sd['net.%d.1.weight' % i], sd['net.%d.1.bias' % i], False, 0., eps)
        x = F.relu(x)
    x = F.conv1d(x, sd['net.3.weight'], sd['net.3.bias'], 1, 1)
    return F.softmax(x.squeeze(1), dim=1)


def run_pointflow(xs, depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges, offset, n,
                  sd_decoder, img_size, pinned=False):
    """Rows C1-C3 (lightningmodel.py:187-242)."""
    pts_hyp, pts_feat, pts_batch = pointflow_hypotheses(depth_pred, depth_batch, img_feats, rotmats,
                                                        tvecs, K, ref_src_edges, offset, n, img_size, pinned=pinned)
    preds = decoder_net(decoder_features(xs, pts_hyp, pts_feat, pts_batch), sd_decoder)
    return offset_expectation(preds, offset, n, depth_pred.shape)


def model_scene(depth_pred, depth_batch, img_feats, rotmats, tvecs, K, ref_src_edges, edge_len,
                sd_pointnet, sd_unet, img_size):
    """Row B5 (lightningmodel.py:176-185)."""
    pts, pts_feat, pts_batch = feature_rich_pointcloud(depth_pred, depth_batch, img_feats, rotmats,
                                                       tvecs, K, ref_src_edges, img_size)
    anchor_pts, anchor_idx3d, anchor_batch, edges = voxelize(pts, pts_batch, edge_len)
    x = torch.cat((pts[edges[1]] - anchor_pts[edges[0]], pts_feat[edges[1]]), dim=1)
    x = pointnet(x, edges[0], anchor_pts.shape[0], sd_pointnet)
    xs = sparse_unet(x, anchor_pts, anchor_idx3d, anchor_batch, edge_len, sd_unet)
    return xs, pts


# ----------------------------------------------------------------------------------------------
# "next" row: PropagationNet (mv3d/subnetworks/upsampling.py:14-36)
# ----------------------------------------------------------------------------------------------

def propagation_net(features, depth, sd, eps=1e-5):
    x = torch.cat((features, depth), dim=1)
    for i in range(1, 5):
        x = F.conv2d(x, sd['conv%d.0.weight' % i], None, 1, 1)
        x = F.relu(F.batch_norm(x, sd['conv%d.1.running_mean' % i], sd['conv%d.1.running_var' % i],
                                sd['conv%d.1.weight' % i], sd['conv%d.1.bias' % i], False, 0., eps))
    p = F.softmax(x, dim=1)
    unf = F.unfold(F.pad(depth, (1, 1, 1, 1), mode='replicate'), kernel_size=3)     # [B, 9, H*W]
    b, c, h, w = p.shape
    return torch.sum(p.view(b, c, h * w) * unf, dim=1).view(b, h, w)
