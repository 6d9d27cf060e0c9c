"""SpaMat -- nn.Module with the reference's surface (modules/SparseMatching/modules/SpaMat.py:12-28)."""
from torch.nn.modules.module import Module

from ..functions.SpaMat import SpaMatFunction


class SpaMat(Module):
    def __init__(self):
        super(SpaMat, self).__init__()

    def forward(self, ref_feas, tar_feas, ref_mask, tar_mask, max_disp):
        """sparse matching while forwarding

        Args:
            ref_feas, tar_feas: feature map of left/right view, Batch*Channel*Height*Width;
            ref_mask, tar_mask: mask of left/right view, Batch*Height*Width;
            max_disp:           the maximum disparity in current scale;

        Returns:
            output: the computed disparity map, Batch*Height*Width;
        """
        return SpaMatFunction.apply(ref_feas, tar_feas, ref_mask, tar_mask, max_disp)
