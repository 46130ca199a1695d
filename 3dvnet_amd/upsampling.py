"""Depth upsampling by learned 3x3 propagation -- "next" row of SURVEY.md §8f (rank 2), the stage that
follows the hot path in ``mv3d/eval-3dvnet.py:101-125``.  Mirrors the interface and ``state_dict`` keys of
``mv3d/subnetworks/upsampling.py::PropagationNet`` (``conv{1..4}.{0.weight,1.*}``).

Like in the reference these are stock 2D convolutions executed by PyTorch-ROCm (MIOpen); nothing here is a
hand-written kernel.  Formulation: the 9-way softmax weights are applied to the nine shifted views of the
replicate-padded depth map (no im2col buffer).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _stage(c_in, c_out):
    return nn.Sequential(nn.Conv2d(c_in, c_out, 3, 1, 1, bias=False), nn.BatchNorm2d(c_out), nn.ReLU(inplace=True))


class PropagationNet(nn.Module):
    """``forward(features[B,Cf,H,W], depth[B,1,H,W]) -> [B,H,W]``: each output depth is a convex
    combination (softmax over 9 logits predicted from features+depth) of its 3x3 neighbourhood."""

    def __init__(self, in_dim=4, h_dim=32):
        super().__init__()
        widths = [in_dim, h_dim, h_dim, h_dim, 9]
        for i in range(4):
            setattr(self, 'conv%d' % (i + 1), _stage(widths[i], widths[i + 1]))

    def forward(self, features, depth):
        x = torch.cat((features, depth), dim=1)
        for i in range(1, 5):
            x = getattr(self, 'conv%d' % i)(x)
        w = F.softmax(x, dim=1)                                   # [B, 9, H, W], row-major 3x3 order
        padded = F.pad(depth, (1, 1, 1, 1), mode='replicate')[:, 0]
        H, W = depth.shape[-2:]
        out = torch.zeros_like(depth[:, 0])
        for k in range(9):
            dy, dx = divmod(k, 3)
            out = out + w[:, k] * padded[:, dy:dy + H, dx:dx + W]
        return out


def upsample_depth(all_depth, stages, chunk=100):
    """Stage 3 of the scene driver (eval-3dvnet.py:101-125): for each (PropagationNet, guide tensor) pair,
    nearest-neighbour resize the depth to the guide's resolution and refine it, `chunk` views at a time."""
    for net, guide in stages:
        all_depth = F.interpolate(all_depth.unsqueeze(1), guide.shape[-2:], mode='nearest').squeeze(1)
        for s in range(0, all_depth.shape[0], chunk):
            all_depth[s:s + chunk] = net(guide[s:s + chunk], all_depth[s:s + chunk].unsqueeze(1))
    return all_depth
