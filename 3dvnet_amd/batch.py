"""``Batch`` container with the field contract of ``mv3d/dsets/batch.py:6-17`` (the reference
subclasses torch_geometric's ``Data``; only attribute access and ``.to(device)`` are used on the
inference path, mv3d/eval-3dvnet.py:54-56)."""
import torch


class Batch:
    def __init__(self, images, rotmats, tvecs, K, depth_images, ref_src_edges):
        self.images = images
        self.rotmats = rotmats
        self.tvecs = tvecs
        self.K = K
        self.depth_images = depth_images
        self.ref_src_edges = ref_src_edges

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self
