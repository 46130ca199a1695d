"""Import the upstream 3DVNet reference (``/root/reference``) with import-time stubs.

TEST INFRASTRUCTURE, build container only.  The reference is pure Python but depends on
packages absent from this image (torch_scatter, torch_geometric, MinkowskiEngine,
pytorch_lightning, torchvision, cv2, open3d).  This module registers minimal stand-in
modules in ``sys.modules`` *before* ``mv3d.*`` is imported so that the reference's own
source files run unmodified from where they lie.  Nothing here travels to the GPU box as
anything but dead code: ``/root/reference`` does not exist there and ``make_golden.py``
is the only caller.

Stub semantics:
  * ``torch_scatter.scatter(src, index, dim, out, dim_size, reduce)`` -- restated from the
    torch-scatter 2.0.5 documentation: sum / mean (= sum / clamp(count, 1)) / min / max
    with empty slots = 0.
  * ``torch_geometric.nn.voxel_grid(pos, batch, size, start, end)`` -- restated from
    PyG 1.6.3 ``voxel_grid`` -> torch_cluster 1.5.8 ``grid``: the batch index is appended
    as an extra coordinate with cell size 1; per dimension ``i = trunc((p - start)/size)``,
    ``n = trunc((end - start)/size) + 1``; id = sum_d i_d * prod_{d'<d} n_d'.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert out is None
    if dim < 0:
        dim += src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    if reduce in ("sum", "add"):
        return res.scatter_add_(dim, idx, src)
    if reduce == "mean":
        res.scatter_add_(dim, idx, src)
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp_(min=1).view(view)
        if src.dtype.is_floating_point:
            return res / cnt
        return torch.div(res, cnt, rounding_mode="floor")
    if reduce in ("min", "max"):
        return res.scatter_reduce_(dim, idx, src, "amin" if reduce == "min" else "amax",
                                   include_self=False)
    raise ValueError(reduce)


def _voxel_grid(pos, batch, size, start=None, end=None):
    pos = torch.cat([pos, batch.unsqueeze(-1).type_as(pos)], dim=-1)
    dim = pos.shape[1]
    size_t = torch.tensor([float(size)] * (dim - 1) + [1.0], dtype=pos.dtype, device=pos.device)
    start_t = torch.cat([start.type_as(pos), pos.new_zeros(1)])
    end_t = torch.cat([end.type_as(pos), batch.max().type_as(pos).view(1)])
    p = pos - start_t
    num = ((end_t - start_t) / size_t).to(torch.long) + 1
    cum = num.cumprod(0)
    cum = torch.cat([cum.new_ones(1), cum[:-1]])
    c = (p / size_t).to(torch.long)
    return (c * cum).sum(1)


def install_stubs():
    if "mv3d" in sys.modules:
        return
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("cv2")
    mod("open3d")
    mod("kornia")

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    tr = mod("torchvision.transforms", Compose=_Dummy, ToPILImage=_Dummy, Resize=_Dummy,
             ToTensor=_Dummy, Normalize=_Dummy)
    tvm = mod("torchvision.models")
    tvo = mod("torchvision.ops")
    mod("torchvision", transforms=tr, models=tvm, ops=tvo)

    mod("torch_scatter", scatter=_scatter)
    tgn = mod("torch_geometric.nn", voxel_grid=_voxel_grid)

    class _Data:
        def __init__(self, *a, **k):
            pass

    tgd = mod("torch_geometric.data", Data=_Data)
    mod("torch_geometric", nn=tgn, data=tgd)

    class _Interp(torch.nn.Module):
        def forward(self, *a, **k):
            raise RuntimeError("MinkowskiEngine is not available")

    mod("MinkowskiEngine", MinkowskiInterpolation=_Interp)

    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    mod("pytorch_lightning", LightningModule=_LM)


def reference():
    """Returns a namespace with the reference modules used to generate golden vectors."""
    install_stubs()
    from mv3d import utils as ref_utils
    from mv3d.subnetworks import mvsnet as ref_mvsnet
    from mv3d.subnetworks import scenemodeling as ref_scene
    from mv3d.subnetworks import refinement as ref_refine
    from mv3d.subnetworks import upsampling as ref_up
    from mv3d import lightningmodel as ref_lm
    from mv3d.eval import metricfunctions as ref_metrics
    return types.SimpleNamespace(utils=ref_utils, mvsnet=ref_mvsnet, scene=ref_scene,
                                 refine=ref_refine, up=ref_up, lm=ref_lm, metrics=ref_metrics)
