"""Tensor-level wrappers over the C ABI (include/decnet_hip.h).

PyTorch is used only for device memory and the current HIP stream; every wrapper checks
device / dtype / contiguity / shape (the reference checks none of them and would read out of
bounds -- SM_kernel.cu:369-376 takes raw data_ptr<float>()) and then passes raw pointers.
"""
import torch

from . import _lib


_F32 = torch.float32
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The current HIP stream of t's device as a raw handle (what the C ABI takes as void*)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(name, t, shape=None):
    # fast path first: this runs for every tensor of every call (SpaMatFunction's host time is what is
    # left of a training step at stages 1-2, bench.py `train`)
    if (type(t) is torch.Tensor or isinstance(t, torch.Tensor)) and t.is_cuda and t.dtype is _F32 and \
            t.is_contiguous() and (shape is None or t.shape == shape):
        return t
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.DecnetHipError(
            "%s is on %s: decnet_amd runs on the MI355X HIP path only (no CPU fallback)"
            % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise AssertionError("%s must be contiguous" % name)      # functions/SpaMat.py:21-22
    raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))


class _on_device:
    """`with _on_device(t):` = torch.cuda.device(t.device), free when that device is already current (the
    reference wraps its launches in torch.cuda.device_of, functions/SpaMat.py:24)."""
    __slots__ = ("idx", "ctx")

    def __init__(self, t):
        self.idx = t.device.index
        self.ctx = None

    def __enter__(self):
        if self.idx != torch.cuda.current_device():
            self.ctx = torch.cuda.device(self.idx)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


_FN = {}


def _fn(name):
    """ctypes entry point, looked up once."""
    f = _FN.get(name)
    if f is None:
        f = _FN[name] = getattr(_lib.lib(), name)
    return f


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise ValueError("all tensors must be on %s" % dev)
    return dev


def _feat_args(ref, tar, rmask, tmask, max_disp):
    _chk("ref_feas", ref)
    if ref.dim() != 4:
        raise ValueError("ref_feas must be [B,C,H,W]")
    B, C, H, W = ref.shape
    _chk("tar_feas", tar, (B, C, H, W))
    _chk("ref_mask", rmask, (B, H, W))
    _chk("tar_mask", tmask, (B, H, W))
    _same_device(ref, tar, rmask, tmask)
    D = int(max_disp)                       # numpy.int64 arrives here
    if D < 1:
        raise ValueError("max_disp must be >= 1")
    return B, C, H, W, D


def spamat_forward(ref, tar, rmask, tmask, output, sum_sim, max_cost, max_disp):
    B, C, H, W, D = _feat_args(ref, tar, rmask, tmask, max_disp)
    for n, t in (("output", output), ("sum_similarities", sum_sim), ("max_cost", max_cost)):
        _chk(n, t, (B, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spamat_forward")(
            ref.data_ptr(), tar.data_ptr(), rmask.data_ptr(), tmask.data_ptr(), output.data_ptr(),
            sum_sim.data_ptr(), max_cost.data_ptr(), B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spamat_forward")


def spamat_backward(ref, tar, rmask, tmask, output, sum_sim, max_cost, grad_out, grad_ref, grad_tar,
                    max_disp):
    B, C, H, W, D = _feat_args(ref, tar, rmask, tmask, max_disp)
    for n, t in (("output", output), ("sum_similarities", sum_sim), ("max_cost", max_cost),
                 ("grad_output", grad_out)):
        _chk(n, t, (B, H, W))
    _chk("grad_ref_feas", grad_ref, (B, C, H, W))
    _chk("grad_tar_feas", grad_tar, (B, C, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spamat_backward")(
            ref.data_ptr(), tar.data_ptr(), rmask.data_ptr(), tmask.data_ptr(), output.data_ptr(),
            sum_sim.data_ptr(), max_cost.data_ptr(), grad_out.data_ptr(), grad_ref.data_ptr(),
            grad_tar.data_ptr(), B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spamat_backward")


def spavar_forward(ref, tar, rmask, tmask, disparity, output, sum_sim, max_cost, max_disp):
    B, C, H, W, D = _feat_args(ref, tar, rmask, tmask, max_disp)
    for n, t in (("disparity", disparity), ("output", output), ("sum_similarities", sum_sim),
                 ("max_cost", max_cost)):
        _chk(n, t, (B, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spavar_forward")(
            ref.data_ptr(), tar.data_ptr(), rmask.data_ptr(), tmask.data_ptr(),
            disparity.data_ptr(), output.data_ptr(), sum_sim.data_ptr(), max_cost.data_ptr(),
            B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spavar_forward")


def spavar_backward(ref, tar, rmask, tmask, disparity, output, sum_sim, max_cost, grad_out,
                    grad_ref, grad_tar, grad_disp, max_disp):
    B, C, H, W, D = _feat_args(ref, tar, rmask, tmask, max_disp)
    for n, t in (("disparity", disparity), ("output", output), ("sum_similarities", sum_sim),
                 ("max_cost", max_cost), ("grad_output", grad_out), ("grad_disparity", grad_disp)):
        _chk(n, t, (B, H, W))
    _chk("grad_ref_feas", grad_ref, (B, C, H, W))
    _chk("grad_tar_feas", grad_tar, (B, C, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spavar_backward")(
            ref.data_ptr(), tar.data_ptr(), rmask.data_ptr(), tmask.data_ptr(),
            disparity.data_ptr(), output.data_ptr(), sum_sim.data_ptr(), max_cost.data_ptr(),
            grad_out.data_ptr(), grad_ref.data_ptr(), grad_tar.data_ptr(), grad_disp.data_ptr(),
            B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spavar_backward")


def spamatvar_forward_bits(ref, tar, rbits, tbits, max_disp, out=None):
    """spamatvar_forward with bit-packed masks: rbits / tbits int64 [B,H,ceil(W/64)], bit i of word w of a row = pixel
    64 w + i (what ``decnet_detail_mask`` writes beside the float plane).  Same results as the float-mask call."""
    _chk("ref_feas", ref)
    if ref.dim() != 4:
        raise ValueError("ref_feas must be [B,C,H,W]")
    B, C, H, W = ref.shape
    _chk("tar_feas", tar, (B, C, H, W))
    for n, t in (("ref_bits", rbits), ("tar_bits", tbits)):
        if (not isinstance(t, torch.Tensor) or t.dtype != torch.int64 or not t.is_contiguous() or
                tuple(t.shape) != (B, H, (W + 63) // 64)):
            raise ValueError("%s must be a contiguous int64 tensor [B,H,ceil(W/64)]" % n)
    _same_device(ref, tar, rbits, tbits)
    D = int(max_disp)
    if D < 1:
        raise ValueError("max_disp must be >= 1")
    if out is None:
        out = tuple(torch.empty((B, H, W), dtype=torch.float32, device=ref.device) for _ in range(4))
    o, v, s, m = out
    for n, t in (("output", o), ("variance", v), ("sum_similarities", s), ("max_cost", m)):
        _chk(n, t, (B, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spamatvar_forward_bits")(
            ref.data_ptr(), tar.data_ptr(), rbits.data_ptr(), tbits.data_ptr(), o.data_ptr(),
            v.data_ptr(), s.data_ptr(), m.data_ptr(), B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spamatvar_forward_bits")
    return o, v, s, m


def spamatvar_forward(ref, tar, rmask, tmask, max_disp, out=None):
    """Fused SpaMat + SpaVar forward (the model's only use of SpaVar,
    SparseDenseNetRefinementMask.py:183-192).  Returns (disparity, variance, sum_sim, max_cost),
    each [B,H,W].  Inference only (no autograd graph is recorded)."""
    B, C, H, W, D = _feat_args(ref, tar, rmask, tmask, max_disp)
    if out is None:
        out = tuple(torch.empty((B, H, W), dtype=torch.float32, device=ref.device) for _ in range(4))
    o, v, s, m = out
    for n, t in (("output", o), ("variance", v), ("sum_similarities", s), ("max_cost", m)):
        _chk(n, t, (B, H, W))
    with _on_device(ref):
        rc = _fn("decnet_spamatvar_forward")(
            ref.data_ptr(), tar.data_ptr(), rmask.data_ptr(), tmask.data_ptr(), o.data_ptr(),
            v.data_ptr(), s.data_ptr(), m.data_ptr(), B, C, H, W, D, _stream(ref))
    _lib.check(rc, "decnet_spamatvar_forward")
    return o, v, s, m
