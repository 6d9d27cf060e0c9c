"""SpaVarFunction -- same surface as the reference's modules/SparseVar/functions/SpaVar.py:8-52."""
import torch
from torch.autograd import Function

from .... import ops


class SpaVarFunction(Function):
    @staticmethod
    def forward(ctx, ref_feas, tar_feas, ref_mask, tar_mask, disparity, max_disp):
        """variance of the sparse matching distribution around `disparity`

        Args:
            ref_feas, tar_feas: feature map of left/right view, Batch*Channel*Height*Width;
            ref_mask, tar_mask: mask of left/right view, Batch*Height*Width;
            disparity:          the disparity the variance is taken around, Batch*Height*Width;
            max_disp:           the maximum disparity in current scale;

        Returns:
            output: sum_d p_d (d - disparity)^2, Batch*Height*Width;
        """
        assert ref_feas.is_contiguous() and tar_feas.is_contiguous()        # SpaVar.py:21
        assert ref_mask.is_contiguous() and tar_mask.is_contiguous()        # SpaVar.py:22
        disparity = disparity.contiguous()
        output, sum_similarities, max_cost = torch.empty((3,) + tuple(ref_mask.shape), dtype=ref_mask.dtype,
                                                         device=ref_mask.device).unbind(0)
        ops.spavar_forward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                           sum_similarities, max_cost, max_disp)
        ctx.save_for_backward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                              sum_similarities, max_cost)
        ctx.max_disp = int(max_disp)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        (ref_feas, tar_feas, ref_mask, tar_mask, disparity, output, sum_similarities,
         max_cost) = ctx.saved_tensors
        assert grad_output.is_contiguous()                                  # SpaVar.py:39
        grad_ref_feas = torch.empty_like(ref_feas)
        grad_tar_feas = torch.empty_like(tar_feas)
        grad_disparity = torch.empty_like(disparity)
        ops.spavar_backward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                            sum_similarities, max_cost, grad_output, grad_ref_feas, grad_tar_feas,
                            grad_disparity, ctx.max_disp)
        return grad_ref_feas, grad_tar_feas, None, None, grad_disparity, None
