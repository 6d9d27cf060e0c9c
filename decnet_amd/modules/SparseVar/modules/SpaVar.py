"""SpaVar -- nn.Module with the reference's surface (modules/SparseVar/modules/SpaVar.py:12-28)."""
from torch.nn.modules.module import Module

from ..functions.SpaVar import SpaVarFunction


class SpaVar(Module):
    def __init__(self):
        super(SpaVar, self).__init__()

    def forward(self, ref_feas, tar_feas, ref_mask, tar_mask, disparity, max_disp):
        """variance of the sparse matching distribution

        Args:
            ref_feas, tar_feas: feature map of left/right view, Batch*Channel*Height*Width;
            ref_mask, tar_mask: mask of left/right view, Batch*Height*Width;
            disparity:          disparity map the variance is taken around, Batch*Height*Width;
            max_disp:           the maximum disparity in current scale;

        Returns:
            output: the variance map, Batch*Height*Width;
        """
        return SpaVarFunction.apply(ref_feas, tar_feas, ref_mask, tar_mask, disparity, max_disp)
