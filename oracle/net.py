"""A CPU 'net' with the PL3DVNet inference surface whose arithmetic is the oracle -- used by tests to
exercise the scene driver's chunking / sharding logic without a GPU and as the checker of the HIP
driver run (tests/test_driver.py; bench.py's cfg3 parity leg).  TEST INFRASTRUCTURE: never imported by the
product package.  ``pinned=True`` evaluates rows A1-A4 and the back-projections of rows B2 / C1 with the host-independent
orders of oracle/pinned.py.  (In rows B2 / C1 the variance FEATURES of the points still come from this host's torch ops --
continuous inputs of PointNet and the decoder; the point COORDINATES decide which voxel cell a point falls into.)"""
import torch

from oracle import costvolume as ocv
from oracle import scene as osc


class OracleNet:
    def __init__(self, sd_costreg, sd_pointnet, sd_unet, sd_decoder, img_size, edge_len, pinned=False):
        self.sd = dict(cr=sd_costreg, pn=sd_pointnet, un=sd_unet, dec=sd_decoder)
        self.img_size, self.edge_len, self.pinned = img_size, edge_len, pinned

    def make_initial_depth_predictions(self, batch, cfg):
        d, _, _ = ocv.mvsnet_depth(batch.features_quarter, batch.rotmats, batch.tvecs, batch.K,
                                   batch.ref_src_edges, self.sd['cr'], cfg['depth_start'],
                                   cfg['depth_interval'], cfg['n_intervals'], self.img_size, cfg['size'],
                                   pinned=self.pinned)
        ref_idx = torch.unique(batch.ref_src_edges[0])
        return d, batch.images_batch[ref_idx], None, batch.features_quarter, None, ref_idx

    def model_scene(self, depth, depth_batch, feats, rot, tv, K, edges, return_pts=False, gather_fn=None):
        pts, pf, pb = osc.feature_rich_pointcloud(depth, depth_batch, feats, rot, tv, K, edges, self.img_size,
                                                  pinned=self.pinned)
        if gather_fn is not None:
            pts, pf, pb = gather_fn(pts, pf, pb)
        a_pts, a_idx, a_batch, e = osc.voxelize(pts, pb, self.edge_len)
        x = torch.cat((pts[e[1]] - a_pts[e[0]], pf[e[1]]), dim=1)
        x = osc.pointnet(x, e[0], a_pts.shape[0], self.sd['pn'])
        xs = osc.sparse_unet(x, a_pts, a_idx, a_batch, self.edge_len, self.sd['un'])
        return (xs, pts) if return_pts else xs

    def run_pointflow(self, xs, depth, depth_batch, feats, rot, tv, K, edges, offset, n):
        return osc.run_pointflow(xs, depth, depth_batch, feats, rot, tv, K, edges, offset, n,
                                 self.sd['dec'], self.img_size, pinned=self.pinned)
