"""Data formats on the output side of the path -- SURVEY.md §8f rank 4 ("next" row): the ``preds.npz``
record the reference's evaluation tooling consumes (``mv3d/eval/main.py:74-101``) and its 2D depth metrics
(``mv3d/eval/metricfunctions.py:26-67``; ``abs_rel`` is the accuracy metric of BASELINE.json).  Plain
tensor/NumPy bookkeeping: nothing here is a kernel.
"""
import os

import numpy as np
import torch


def write_preds(path, scene, depth_preds, batch, ref_idx, img_idx, init_prob=None, final_prob=None):
    """``np.savez(preds.npz)`` with the reference's keys.  Intrinsics are rescaled from the image size
    they refer to (``batch.images``) to the size of the predicted depth maps (main.py:74-81)."""
    depth_preds = np.asarray(depth_preds)
    old_h, old_w = batch.images.shape[-2:]
    new_h, new_w = depth_preds.shape[-2:]
    K = batch.K[ref_idx].detach().cpu().clone()
    K[:, 0, :] *= float(new_w) / float(old_w)
    K[:, 1, :] *= float(new_h) / float(old_h)
    rec = dict(scene=os.path.basename(scene), depth_preds=depth_preds,
               rotmats=batch.rotmats[ref_idx].detach().cpu().numpy(),
               tvecs=batch.tvecs[ref_idx].detach().cpu().numpy(), K=K.numpy(),
               img_idx=np.asarray(img_idx)[np.asarray(ref_idx)])
    if init_prob is not None:
        rec['init_prob'] = init_prob
    if final_prob is not None:
        rec['final_prob'] = final_prob
    np.savez(path, **rec)
    return rec


def depth_metrics_2d(depth_pred, depth_gt, pred_valid=None):
    """The reference's per-batch 2D metrics (metricfunctions.py:26-67): averages over images of the masked
    per-image means, mask = 0.5 <= gt < 65 (and ``pred_valid`` when given)."""
    with torch.no_grad():
        mask = (depth_gt >= 0.5) & (depth_gt < 65.)
        out = {}
        if pred_valid is not None:
            mask = mask & pred_valid
            hw = pred_valid.shape[1] * pred_valid.shape[2]
            out['perc_valid'] = torch.mean(torch.sum(pred_valid, dim=(1, 2)) / hw)
        m = mask.float()
        n = m.sum(dim=(1, 2)) + 1e-7
        err = (depth_pred - depth_gt).abs()
        inv = (1. / depth_pred - 1. / depth_gt).abs()
        inv = torch.where(torch.isfinite(inv), inv, torch.zeros_like(inv))

        def per_image(x):
            return (x * m).sum(dim=(1, 2)) / n

        ratio = torch.maximum(depth_pred / depth_gt, depth_gt / depth_pred)
        out.update(abs_rel=per_image(err / (depth_gt + 1e-7)).mean(),
                   abs_diff=per_image(err).mean(), abs_inv=per_image(inv).mean(),
                   sq_rel=per_image(err ** 2 / (depth_gt + 1e-7)).mean(),
                   rmse=torch.sqrt(per_image(err ** 2)).mean(),
                   d_125=per_image((ratio < 1.25).float()).mean(),
                   d_125_2=per_image((ratio < 1.25 ** 2).float()).mean(),
                   d_125_3=per_image((ratio < 1.25 ** 3).float()).mean())
    return out
