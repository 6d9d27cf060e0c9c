"""SpaMatFunction -- same surface as the reference's modules/SparseMatching/functions/
SpaMat.py:8-50, running on the gfx950 kernels."""
import torch
from torch.autograd import Function

from .... import ops


class SpaMatFunction(Function):
    @staticmethod
    def forward(ctx, ref_feas, tar_feas, ref_mask, tar_mask, max_disp):
        """sparse matching while forwarding

        Args:
            ref_feas, tar_feas: feature map of left/right view, Batch*Channel*Height*Width;
            ref_mask, tar_mask: mask of left/right view, Batch*Height*Width;
            max_disp:           the maximum disparity in current scale (int or numpy.int64);

        Returns:
            output: the computed disparity map, Batch*Height*Width;
        """
        assert ref_feas.is_contiguous() and tar_feas.is_contiguous()        # SpaMat.py:21
        assert ref_mask.is_contiguous() and tar_mask.is_contiguous()        # SpaMat.py:22
        # the kernels write every element, so no zero fill (SpaMat.py:25-27 needs three)
        # (one allocation for the three planes: they live and die together in ctx)
        output, sum_similarities, max_cost = torch.empty((3,) + tuple(ref_mask.shape), dtype=ref_mask.dtype,
                                                         device=ref_mask.device).unbind(0)
        ops.spamat_forward(ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities,
                           max_cost, max_disp)
        ctx.save_for_backward(ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities,
                              max_cost)
        ctx.max_disp = int(max_disp)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities, max_cost = ctx.saved_tensors
        assert grad_output.is_contiguous()                                  # SpaMat.py:40
        grad_ref_feas = torch.empty_like(ref_feas)
        grad_tar_feas = torch.empty_like(tar_feas)
        ops.spamat_backward(ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities,
                            max_cost, grad_output, grad_ref_feas, grad_tar_feas, ctx.max_disp)
        # the reference returns dummy CPU tensors for the mask grads (SpaMat.py:50); None is
        # what autograd expects for non-differentiable inputs
        return grad_ref_feas, grad_tar_feas, None, None, None
