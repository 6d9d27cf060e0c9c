"""Drop-in stand-ins for the reference's compiled extension modules.

The reference imports `from ..build.lib import SpaMat` (modules/SparseMatching/functions/
SpaMat.py:4) and `from ..build.lib import SpaVar` (modules/SparseVar/functions/SpaVar.py:4),
pybind modules defined in SM_cuda.cpp:29-33 and SV_cuda.cpp:34-38.  `SpaMat` and `SpaVar`
below expose the same function names with the same positional arguments and the same return
value (1), so changing that one import line makes the reference's own autograd Functions run
on the MI355X kernels.
"""
from . import ops


class _SpaMatExt:
    __name__ = "SpaMat"

    @staticmethod
    def sparse_matching_cuda_forward(ref_feas, tar_feas, ref_mask, tar_mask, output,
                                     sum_similarities, max_cost, max_disp):
        """SM_cuda.cpp:7-15"""
        ops.spamat_forward(ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities,
                           max_cost, max_disp)
        return 1

    @staticmethod
    def sparse_matching_cuda_backward(ref_feas, tar_feas, ref_mask, tar_mask, output,
                                      sum_similarities, max_cost, grad_output, grad_ref_feas,
                                      grad_tar_feas, max_disp):
        """SM_cuda.cpp:17-27"""
        ops.spamat_backward(ref_feas, tar_feas, ref_mask, tar_mask, output, sum_similarities,
                            max_cost, grad_output, grad_ref_feas, grad_tar_feas, max_disp)
        return 1


class _SpaVarExt:
    __name__ = "SpaVar"

    @staticmethod
    def sparse_var_cuda_forward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                                sum_similarities, max_cost, max_disp):
        """SV_cuda.cpp:7-17"""
        ops.spavar_forward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                           sum_similarities, max_cost, max_disp)
        return 1

    @staticmethod
    def sparse_var_cuda_backward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                                 sum_similarities, max_cost, grad_output, grad_ref_feas,
                                 grad_tar_feas, grad_disparity, max_disp):
        """SV_cuda.cpp:19-32"""
        ops.spavar_backward(ref_feas, tar_feas, ref_mask, tar_mask, disparity, output,
                            sum_similarities, max_cost, grad_output, grad_ref_feas, grad_tar_feas,
                            grad_disparity, max_disp)
        return 1


SpaMat = _SpaMatExt()
SpaVar = _SpaVarExt()
