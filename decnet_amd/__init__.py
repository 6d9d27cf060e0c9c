"""decnet_amd -- MI355X (gfx950) implementation of DecNet's data-parallel hot path.

* SpaMat / SpaVar   (reference modules/SparseMatching, modules/SparseVar)
* stage-0 dense cost volume -> Conv3d aggregation -> soft-argmax
  (reference modules/submodule.py GetCostVolume / CostRegNetNoDown / disparity_regression)

Host side is Python on PyTorch-ROCm (device memory + streams only); all arithmetic runs in
hand-written HIP kernels behind the C ABI of include/decnet_hip.h.  No CPU fallback.
"""
from ._lib import DecnetHipError, version  # noqa: F401
from .modules.SparseMatching.modules.SpaMat import SpaMat  # noqa: F401
from .modules.SparseMatching.functions.SpaMat import SpaMatFunction  # noqa: F401
from .modules.SparseVar.modules.SpaVar import SpaVar  # noqa: F401
from .modules.SparseVar.functions.SpaVar import SpaVarFunction  # noqa: F401
from .ops import spamatvar_forward, spamatvar_forward_bits  # noqa: F401
from .stage0 import (CostRegNetNoDown, GetCostVolume, Stage0, disparity_regression,  # noqa: F401
                     get_disp_samples)

__all__ = ["SpaMat", "SpaVar", "SpaMatFunction", "SpaVarFunction", "spamatvar_forward", "spamatvar_forward_bits",
           "GetCostVolume", "CostRegNetNoDown", "disparity_regression", "get_disp_samples",
           "Stage0", "DecnetHipError", "version"]
