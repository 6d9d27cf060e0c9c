"""ctypes binding of libdecnet_hip.so (the C ABI declared in include/decnet_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes
import os

from .build import LIB_PATH as _DEFAULT_LIB

# DECNET_HIP_LIB points the loader at another build of the same ABI (diagnostic / ablation builds)
LIB_PATH = os.environ.get("DECNET_HIP_LIB", _DEFAULT_LIB)

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float

# name -> argtypes  (kept in the order of include/decnet_hip.h; tests check every symbol)
SIGNATURES = {
    "decnet_spamat_forward": [_P] * 7 + [_I] * 5 + [_P],
    "decnet_spamat_backward": [_P] * 10 + [_I] * 5 + [_P],
    "decnet_spavar_forward": [_P] * 8 + [_I] * 5 + [_P],
    "decnet_spavar_backward": [_P] * 12 + [_I] * 5 + [_P],
    "decnet_spamatvar_forward": [_P] * 8 + [_I] * 5 + [_P],
    "decnet_spamatvar_forward_bits": [_P] * 8 + [_I] * 5 + [_P],
    "decnet_costvol_forward": [_P] * 3 + [_I] * 5 + [_P],
    "decnet_costvol_forward_cf": [_P] * 3 + [_I] * 6 + [_P],
    "decnet_conv3d_pointwise": [_P] * 3 + [_I] * 6 + [_P],
    "decnet_conv3d_packed_cout": [_I],
    "decnet_conv3d_pack_weight": [_P, _P, _I, _I, _P],
    "decnet_conv3d_bn_act": [_P] * 6 + [_I] * 7 + [_P],
    "decnet_conv3d_wino_weight_floats": [_I, _I],
    "decnet_conv3d_wino_pack_weight": [_P, _P, _I, _I, _I, _P],
    "decnet_conv3d_wino_workspace_floats": [_I] * 7,
    "decnet_conv3d_wino_bn_act": [_P] * 7 + [_I] * 8 + [_P],
    "decnet_conv3d_wino_gemm": [_P] * 3 + [_I] * 4 + [_P],
    "decnet_conv3d_wino_stack_workspace_floats": [_I] * 6,
    "decnet_conv3d_wino_stack_bn_act": [_P] * 4 + [_I] * 3 + [_P] * 2 + [_I] * 6 + [_P],
    "decnet_costvol_wino_stack_bn_act": [_P] * 5 + [_I] * 3 + [_P] * 2 + [_I] * 6 + [_P],
    "decnet_costvol_wino_stack_bn_act_cf": [_P] * 5 + [_I] * 3 + [_P] * 2 + [_I] * 7 + [_P],
    "decnet_conv3d_cout1_softargmax": [_P, _P, _F, _F, _P, _P] + [_I] * 5 + [_P],
    "decnet_conv2d_packed_floats": [_I] * 4,
    "decnet_conv2d_pack_weight": [_P, _P] + [_I] * 4 + [_P],
    "decnet_conv2d_bn_act": [_P] * 5 + [_I] * 8 + [_P],
    "decnet_deconv2d_k3s3_bn_act": [_P] * 5 + [_I] * 6 + [_P],
    "decnet_conv2d_k3s3_bn_act": [_P] * 5 + [_I] * 6 + [_P],
    "decnet_bias_act_inplace": [_P, _P] + [_I] * 5 + [_P],
    "decnet_unfold3_cat": [_P] * 3 + [_I] * 4 + [_P],
    "decnet_s2d3_pad1": [_P, _P] + [_I] * 4 + [_P],
    "decnet_detail_mask": [_P] * 6 + [_F] * 3 + [_P] * 3 + [_I] * 3 + [_P],
    "decnet_conv2d_cat_bn_act": [_P, _P, _I, _P, _P, _P, _P] + [_I] * 7 + [_P],
    "decnet_conv2d_cat_epilogue": [_P, _P, _I, _P, _P, _P, _P] + [_I] * 7 + [_P, _P, _P],
    "decnet_conv2d_mfma_packed_bytes": [_I] * 3,
    "decnet_conv2d_mfma_pack_weight": [_P, _P] + [_I] * 3 + [_P],
    "decnet_conv2d_mfma_cat_bn_act": [_P, _P, _I, _P, _P, _P, _P] + [_I] * 7 + [_P],
    "decnet_deconv2d_mfma_packed_bytes": [_I] * 2,
    "decnet_deconv2d_mfma_pack_weight": [_P, _P] + [_I] * 2 + [_P],
    "decnet_deconv2d_mfma_k3s3_bn_act": [_P] * 5 + [_I] * 6 + [_P],
    "decnet_warp_disparity": [_P, _P, _P] + [_I] * 4 + [_P],
    "decnet_dynamic_upsample3": [_P, _P, _P] + [_I] * 3 + [_P],
    "decnet_tapconv_chunk_floats": [_I] * 4,
    "decnet_tapconv_to_chunks": [_P, _P] + [_I] * 4 + [_P],
    "decnet_tapconv_weight_floats": [_I, _I],
    "decnet_tapconv_pack_weight": [_P, _P] + [_I] * 4 + [_P],
    "decnet_tapconv_split_weight": [_P, _I, _I, _P],
    "decnet_tap_gemm": [_P, _P, _P] + [_I] * 5 + [_P],
    "decnet_tapconv_gather": [_P, _P, _P, _P] + [_I] * 5 + [_P, _P, _P, _I, _P],
    "decnet_conv3d_cout1_workspace_floats": [_I] * 4,
    "decnet_conv3d_cout1_softargmax_ws": [_P, _P, _F, _F, _P, _P, _P] + [_I] * 5 + [_P],
    "decnet_disparity_regression": [_P] * 3 + [_I] * 4 + [_P],
    "decnet_stage0_workspace_floats": [_I] * 6,
    "decnet_stage0_forward": [_P] * 6 + [_I] * 6 + [_P],
    "decnet_stage0_cf_workspace_floats": [_I] * 7,
    "decnet_stage0_forward_cf": [_P] * 7 + [_I] * 7 + [_P],
    "decnet_ncdhw_to_ndhwc": [_P, _P] + [_I] * 5 + [_P],
    "decnet_ndhwc_to_ncdhw": [_P, _P] + [_I] * 5 + [_P],
}



class Stage0Params(ctypes.Structure):
    """decnet_stage0_params of include/decnet_hip.h (plain device pointers)."""
    _fields_ = [("w", _P * 7), ("scale", _P * 7), ("shift", _P * 7), ("w_last", _P),
                ("scale_last", _F), ("shift_last", _F)]


ERRORS = {-1: "null pointer", -2: "bad shape", -3: "shape not supported by the gfx950 kernels",
          -4: "non-finite (NaN / Inf) element in a feature map (DECNET_CHECK_FINITE=1)"}

_lib = None


class DecnetHipError(RuntimeError):
    """``code``: the C ABI's return value (negative: argument error, DECNET_ERR_* of include/decnet_hip.h;
    positive: hipError_t; None: raised by the Python side)."""
    code = None


UNSUPPORTED = -3          # DECNET_ERR_UNSUPPORTED: valid arguments the gfx950 kernels of that entry do not cover
COST_FUNC = {"cor": 0, "ssd": 1, "cat": 2, "sum": 3}          # DECNET_COST_* of include/decnet_hip.h


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DecnetHipError(
                "%s not found: build it with `python -m decnet_amd.build` (hipcc, gfx950). "
                "decnet_amd has no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = ctypes.c_size_t if name.endswith(("_floats", "_bytes")) else _I
        h.decnet_version.restype = ctypes.c_char_p
        h.decnet_version.argtypes = []
        _lib = h
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        e = DecnetHipError("%s: %s (code %d)" % (what, ERRORS.get(rc, "error"), rc))
    else:
        e = DecnetHipError("%s: HIP launch failed, hipError_t=%d" % (what, rc))
    e.code = rc
    raise e


def version():
    return lib().decnet_version().decode()
