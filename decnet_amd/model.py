"""Inference graph around the hot path: this repo's counterpart of
``modules/SparseDenseNetRefinementMask.py:102-236`` and ``modules/__init__.py:7-19``.

Only the per-stage loop matters here: it is what calls the MI355X kernels (stage 0:
``Stage0`` = cost volume + Conv3d aggregation + soft-argmax; stages 1..3: the fused
SpaMat/SpaVar launch).  The 2-D convolution blocks either side of the path (feature extractor,
mask generator, dynamic upsampling, soft attention, refinement -- ``modules/submodule.py:245-372,
566-604, 666-762``) are stock convolutions and run as PyTorch-ROCm (MIOpen) ops; they are declared
from small spec tables below with the reference's attribute names, so a reference checkpoint
(``--resume``, demo.py:124-133) loads key-for-key.

Inference only (``model.eval()``, ``torch.no_grad()``): the reference's training entry point is
not runnable as shipped.
"""
import math
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import DecnetHipError, UNSUPPORTED
from .ops import spamatvar_forward, spamatvar_forward_bits
from .stage0 import CostRegNetNoDown, Stage0


# bench.py's end-to-end accounting: a list that every Unit launch appends (kernel family, algorithmic flops, algorithmic
# bytes) to while it is set; None (the default) costs one comparison per layer.
TALLY = None


def _tally(unit, kind, x):
    """Algorithmic work of one Conv2dUnit / Deconv2dUnit launch: flops = 2 * outputs * Cin * taps per output, bytes = the
    input and output tensors once (fp32); the concatenated inputs count as what they are, never as a copy."""
    xs = x if isinstance(x, (tuple, list)) else (x,)
    B, _, H, W = xs[0].shape
    cin = sum(t.shape[1] for t in xs)
    c = unit.conv
    k, st, co = c.kernel_size[0], c.stride[0], c.out_channels
    if isinstance(c, nn.ConvTranspose2d):
        Ho, Wo = (H - 1) * st - 2 * c.padding[0] + k, (W - 1) * st - 2 * c.padding[0] + k
        taps = (k * k) / float(st * st)                  # k 3, stride 3: every output pixel has exactly one tap
    else:
        Ho = (H + 2 * c.padding[0] - c.dilation[0] * (k - 1) - 1) // st + 1
        Wo = (W + 2 * c.padding[1] - c.dilation[1] * (k - 1) - 1) // st + 1
        taps = k * k
    TALLY.append({"family": kind, "flops": 2.0 * B * Ho * Wo * cin * co * taps,
                  "bytes": 4.0 * B * (cin * H * W + co * Ho * Wo)})


class Unit(nn.Module):
    """conv / transposed conv -> optional BatchNorm2d -> optional ReLU; attributes ``conv`` and
    ``bn`` as in the reference's Conv2dUnit / Deconv2dUnit (submodule.py:15-87)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, relu=True, bn=True, momentum=0.1,
                 transposed=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose2d(cin, cout, k, stride=stride, padding=pad, bias=not bn)
        else:
            self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=not bn)
        self.bn = nn.BatchNorm2d(cout, momentum=momentum) if bn else None
        self.relu = relu

    # ---- fused HIP path for the full-resolution few-channel layers (csrc/conv2d_small.hip) ----
    def _hip_kind(self, x):
        """"conv" / "deconv" / "conv_s3" when this unit, in eval mode on the GPU, is one the
        small-channel kernels cover (they pay off where the tensors are large: >= 64 k pixels per
        image); else None.  x: a tensor, or a tuple of tensors standing for their channel concatenation."""
        if isinstance(x, (tuple, list)):
            if (len(x) > 6 or any(t.shape[0] != x[0].shape[0] or t.shape[2:] != x[0].shape[2:] or t.dtype != x[0].dtype
                                  or t.device != x[0].device for t in x)):
                return None
            kind = self._hip_kind(x[0])
            return kind if kind in ("conv", "mfma") else None
        if self.training or not x.is_cuda or x.dtype != torch.float32 or torch.is_grad_enabled():
            return None
        if os.environ.get("DECNET_CONV2D", "hip") != "hip":
            return None
        c = self.conv
        # many channels: the bf16x3 matrix-core kernel (csrc/conv2d_mfma.hip) where the image gives it enough
        # workgroups (>= 4096 pixels, e.g. the 60 x 108 level; at 20 x 36 the library's kernels win)
        # (round 3: also 9..23 outputs from >= 16 inputs.  The 20 x 36 level stays on the library: measured with the
        # pixel threshold at 512, 649 -> 81 takes 0.265 ms here against 0.107 ms, the 864 / 432 -> 216 1 x 1 layers
        # 0.131 / 0.081 against 0.053 / 0.042 -- 144 workgroups of a K = 5841 reduction each do not fill 256 CUs)
        if (isinstance(c, nn.Conv2d) and
                (c.out_channels >= 9 or c.in_channels >= 48) and
                c.in_channels >= 16 and c.kernel_size in ((1, 1), (3, 3)) and
                c.stride == (1, 1) and c.dilation[0] == c.dilation[1] and c.groups == 1 and c.padding_mode == "zeros" and
                c.padding == (c.dilation[0] * (c.kernel_size[0] // 2),) * 2 and
                x.shape[-1] * x.shape[-2] >= 4096 and
                c.dilation[0] <= 4 and os.environ.get("DECNET_CONV2D_MFMA", "1") == "1"):
            return "mfma"
        # Conv2d k 3, stride 3, padding 1 with more than 24 outputs: space-to-depth + the same kernel as a 1 x 1 convolution
        if (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.stride == (3, 3) and c.padding == (1, 1) and
                c.dilation == (1, 1) and c.groups == 1 and c.padding_mode == "zeros" and c.out_channels > 24 and
                c.in_channels >= 8 and x.shape[-1] * x.shape[-2] >= 4608 and
                os.environ.get("DECNET_CONV2D_MFMA", "1") == "1"):
            return "mfma_s3"
        # transposed convolution k = 3, stride 3 with more than 8 output channels: the same kernel, as a 1 x 1 convolution
        # to 9 Cout channels with a pixel-shuffle store
        if (isinstance(c, nn.ConvTranspose2d) and (c.out_channels > 8 or c.in_channels >= 64) and c.in_channels >= 16 and
                c.kernel_size == (3, 3) and
                c.stride == (3, 3) and c.padding == (0, 0) and c.output_padding == (0, 0) and c.dilation == (1, 1) and
                c.groups == 1 and x.shape[-1] * x.shape[-2] >= 512 and
                os.environ.get("DECNET_CONV2D_MFMA", "1") == "1"):
            return "mfma_deconv"
        # the few-channel kernels at every size (at 60 x 108 and 20 x 36 they do not fill the chip, but one launch
        # replaces the library's convolution + layout transposes + bias / ReLU passes)
        up = 9 if isinstance(c, nn.ConvTranspose2d) else 1
        if x.shape[-1] * x.shape[-2] * up < 256:
            return None
        if (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.stride == (3, 3) and c.padding == (1, 1) and
                c.dilation == (1, 1) and c.groups == 1 and c.padding_mode == "zeros" and c.out_channels <= 24):
            return "conv_s3"
        tr = isinstance(c, nn.ConvTranspose2d)
        # 9..24 output channels: only where the library is slow (dilated taps) or the input is thin
        if c.out_channels > (8 if tr else 24) or (c.out_channels > 8 and c.dilation[0] == 1 and c.in_channels > 12):
            return None
        if isinstance(c, nn.ConvTranspose2d):
            ok = (c.kernel_size == (3, 3) and c.stride == (3, 3) and c.padding == (0, 0) and
                  c.output_padding == (0, 0) and c.dilation == (1, 1) and c.groups == 1)
            return "deconv" if ok else None
        k = c.kernel_size[0]
        ok = (c.kernel_size in ((1, 1), (3, 3)) and c.stride == (1, 1) and c.dilation[0] == c.dilation[1] and
              c.padding == (c.dilation[0] * (k // 2),) * 2 and c.groups == 1 and c.padding_mode == "zeros")
        return "conv" if ok else None

    def _folded(self, neg_last=False):
        """Per-channel scale / shift of the eval-mode BatchNorm (or 1 / bias), cached per weight version.
        neg_last: the weights of the last input channel negated (a caller that feeds -x passes x instead)."""
        c, bn = self.conv, self.bn
        ts = [c.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else
                           ([c.bias] if c.bias is not None else []))
        key = tuple((t.data_ptr(), t._version) for t in ts) + (bool(neg_last),)
        if getattr(self, "_fold_key", None) != key:
            with torch.no_grad():
                co = c.out_channels
                if bn is not None:
                    scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                    shift = bn.bias.float() - bn.running_mean.float() * scale
                else:
                    scale = torch.ones(co, device=c.weight.device)
                    shift = c.bias.float() if c.bias is not None else torch.zeros(co, device=c.weight.device)
                from . import _lib
                L = _lib.lib()
                tr = 1 if isinstance(c, nn.ConvTranspose2d) else 0
                w = c.weight.detach().float().contiguous()
                if neg_last:
                    w = w.clone()
                    w[:, -1] = -w[:, -1]
                k = c.kernel_size[0]
                wp = torch.empty(L.decnet_conv2d_packed_floats(c.in_channels, co, k, tr), dtype=torch.float32,
                                 device=w.device)
                with torch.cuda.device(w.device):
                    _lib.check(L.decnet_conv2d_pack_weight(w.data_ptr(), wp.data_ptr(), c.in_channels, co, k, tr,
                                                           torch.cuda.current_stream(w.device).cuda_stream),
                               "decnet_conv2d_pack_weight")
                self._fold = (wp, scale.contiguous(), shift.contiguous())
            self._fold_key = key
        return self._fold

    def _folded_mfma(self):
        """Weights split into bf16 terms in the operand layout of csrc/conv2d_mfma.hip + folded BN, cached per version."""
        c, bn = self.conv, self.bn
        ts = [c.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else
                           ([c.bias] if c.bias is not None else []))
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if getattr(self, "_mfold_key", None) != key:
            from . import _lib
            L = _lib.lib()
            with torch.no_grad():
                co, ci, k = c.out_channels, c.in_channels, c.kernel_size[0]
                if bn is not None:
                    scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                    shift = bn.bias.float() - bn.running_mean.float() * scale
                else:
                    scale = torch.ones(co, device=c.weight.device)
                    shift = c.bias.float() if c.bias is not None else torch.zeros(co, device=c.weight.device)
                w = c.weight.detach().float().contiguous()
                st = torch.cuda.current_stream(w.device).cuda_stream
                if isinstance(c, nn.Conv2d) and c.stride == (3, 3):      # stride-3: [Cout, Cin, 3, 3] read as [Cout, 9 Cin]
                    wp = torch.empty(L.decnet_conv2d_mfma_packed_bytes(9 * ci, co, 1), dtype=torch.uint8, device=w.device)
                    with torch.cuda.device(w.device):
                        _lib.check(L.decnet_conv2d_mfma_pack_weight(w.data_ptr(), wp.data_ptr(), 9 * ci, co, 1, st),
                                   "decnet_conv2d_mfma_pack_weight")
                elif isinstance(c, nn.ConvTranspose2d):
                    wp = torch.empty(L.decnet_deconv2d_mfma_packed_bytes(ci, co), dtype=torch.uint8, device=w.device)
                    with torch.cuda.device(w.device):
                        _lib.check(L.decnet_deconv2d_mfma_pack_weight(w.data_ptr(), wp.data_ptr(), ci, co, st),
                                   "decnet_deconv2d_mfma_pack_weight")
                else:
                    wp = torch.empty(L.decnet_conv2d_mfma_packed_bytes(ci, co, k), dtype=torch.uint8, device=w.device)
                    with torch.cuda.device(w.device):
                        _lib.check(L.decnet_conv2d_mfma_pack_weight(w.data_ptr(), wp.data_ptr(), ci, co, k, st),
                                   "decnet_conv2d_mfma_pack_weight")
                self._mfold = (wp, scale.contiguous(), shift.contiguous())
            self._mfold_key = key
        return self._mfold

    def _forward_mfma(self, x):
        import ctypes
        from . import _lib
        from .ops import _stream
        wp, scale, shift = self._folded_mfma()
        xs = [t.contiguous() for t in (x if isinstance(x, (tuple, list)) else (x,))]
        B, _, H, W = xs[0].shape
        c = self.conv
        assert sum(t.shape[1] for t in xs) == c.in_channels
        y = torch.empty((B, c.out_channels, H, W), dtype=torch.float32, device=xs[0].device)
        ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        cins = (ctypes.c_int * len(xs))(*[int(t.shape[1]) for t in xs])
        with torch.cuda.device(y.device):
            rc = _lib.lib().decnet_conv2d_mfma_cat_bn_act(ptrs, cins, len(xs), wp.data_ptr(), scale.data_ptr(),
                                                          shift.data_ptr(), y.data_ptr(), B, c.out_channels, H, W,
                                                          c.kernel_size[0], c.dilation[0], 1 if self.relu else 0,
                                                          _stream(y))
        _lib.check(rc, "decnet_conv2d_mfma_cat_bn_act")
        return y

    def _forward_mfma_s3(self, x):
        """Conv2d k 3, stride 3, padding 1: decnet_s2d3_pad1 + the matrix-core kernel as a 1 x 1 convolution."""
        import ctypes
        from . import _lib
        from .ops import _stream
        wp, scale, shift = self._folded_mfma()
        x = x.contiguous()
        B, Cin, H, W = x.shape
        c = self.conv
        Ho, Wo = (H - 1) // 3 + 1, (W - 1) // 3 + 1
        t = torch.empty((B, 9 * Cin, Ho, Wo), dtype=torch.float32, device=x.device)
        y = torch.empty((B, c.out_channels, Ho, Wo), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            _lib.check(L.decnet_s2d3_pad1(x.data_ptr(), t.data_ptr(), B, Cin, H, W, _stream(x)), "decnet_s2d3_pad1")
            ptrs = (ctypes.c_void_p * 1)(t.data_ptr())
            cins = (ctypes.c_int * 1)(9 * Cin)
            rc = L.decnet_conv2d_mfma_cat_bn_act(ptrs, cins, 1, wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                 y.data_ptr(), B, c.out_channels, Ho, Wo, 1, 1, 1 if self.relu else 0,
                                                 _stream(y))
        _lib.check(rc, "decnet_conv2d_mfma_cat_bn_act")
        return y

    def _forward_mfma_deconv(self, x):
        from . import _lib
        from .ops import _stream
        wp, scale, shift = self._folded_mfma()
        x = x.contiguous()
        B, Cin, H, W = x.shape
        c = self.conv
        y = torch.empty((B, c.out_channels, 3 * H, 3 * W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().decnet_deconv2d_mfma_k3s3_bn_act(x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                             y.data_ptr(), B, Cin, c.out_channels, H, W,
                                                             1 if self.relu else 0, _stream(x))
        _lib.check(rc, "decnet_deconv2d_mfma_k3s3_bn_act")
        return y

    def _forward_hip(self, x, kind, out=None, epi=0, ea=None, eb=None, neg_last=False):
        """out: write into this [B,Cout,H,W] buffer (kind "conv"); epi / ea / eb: decnet_conv2d_cat_epilogue's fused tail
        of a single-output layer; neg_last: see _folded."""
        if TALLY is not None:
            n_tally = len(TALLY)
            _tally(self, kind, x)
            try:
                return self._forward_hip_kind(x, kind, out, epi, ea, eb, neg_last)
            except DecnetHipError:                       # forward() falls back to the library path and tallies THAT
                del TALLY[n_tally:]
                raise
        return self._forward_hip_kind(x, kind, out, epi, ea, eb, neg_last)

    def _forward_hip_kind(self, x, kind, out, epi, ea, eb, neg_last):
        if kind == "mfma":
            return self._forward_mfma(x)
        if kind == "mfma_deconv":
            return self._forward_mfma_deconv(x)
        if kind == "mfma_s3":
            return self._forward_mfma_s3(x)
        from . import _lib
        from .ops import _stream
        w, scale, shift = self._folded(neg_last)
        if isinstance(x, (tuple, list)) or epi:         # concatenated input, never materialised
            import ctypes
            xs = [t.contiguous() for t in (x if isinstance(x, (tuple, list)) else (x,))]
            B, _, H, W = xs[0].shape
            Co = self.conv.out_channels
            assert sum(t.shape[1] for t in xs) == self.conv.in_channels
            y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=xs[0].device)
            ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
            cins = (ctypes.c_int * len(xs))(*[int(t.shape[1]) for t in xs])
            with torch.cuda.device(y.device):
                if epi:
                    assert Co == 1 and ea.is_contiguous() and (eb is None or eb.is_contiguous())
                    rc = _lib.lib().decnet_conv2d_cat_epilogue(ptrs, cins, len(xs), w.data_ptr(), scale.data_ptr(),
                                                               shift.data_ptr(), y.data_ptr(), B, H, W,
                                                               self.conv.kernel_size[0], self.conv.dilation[0],
                                                               1 if self.relu else 0, int(epi), ea.data_ptr(),
                                                               eb.data_ptr() if eb is not None else None, _stream(y))
                    _lib.check(rc, "decnet_conv2d_cat_epilogue")
                    return y
                rc = _lib.lib().decnet_conv2d_cat_bn_act(ptrs, cins, len(xs), w.data_ptr(), scale.data_ptr(),
                                                         shift.data_ptr(), y.data_ptr(), B, Co, H, W,
                                                         self.conv.kernel_size[0], self.conv.dilation[0],
                                                         1 if self.relu else 0, _stream(y))
            _lib.check(rc, "decnet_conv2d_cat_bn_act")
            return y
        x = x.contiguous()
        B, Cin, H, W = x.shape
        Co = self.conv.out_channels
        L = _lib.lib()
        with torch.cuda.device(x.device):
            if kind == "conv_s3":
                y = torch.empty((B, Co, (H - 1) // 3 + 1, (W - 1) // 3 + 1), dtype=torch.float32, device=x.device)
                rc = L.decnet_conv2d_k3s3_bn_act(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                 y.data_ptr(), B, Cin, Co, H, W, 1 if self.relu else 0, _stream(x))
            elif kind == "deconv":
                y = torch.empty((B, Co, 3 * H, 3 * W), dtype=torch.float32, device=x.device)
                rc = L.decnet_deconv2d_k3s3_bn_act(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                   y.data_ptr(), B, Cin, Co, H, W, 1 if self.relu else 0,
                                                   _stream(x))
            else:
                y = out if out is not None else torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
                assert y.is_contiguous()
                rc = L.decnet_conv2d_bn_act(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                            y.data_ptr(), B, Cin, Co, H, W, self.conv.kernel_size[0],
                                            self.conv.dilation[0], 1 if self.relu else 0, _stream(x))
        _lib.check(rc, {"conv": "decnet_conv2d_bn_act", "deconv": "decnet_deconv2d_k3s3_bn_act",
                        "conv_s3": "decnet_conv2d_k3s3_bn_act"}[kind])
        return y

    def _folded_torch(self):
        """Eval-mode BatchNorm folded into the convolution itself (w * scale per output channel, bias =
        shift) for the layers that stay on MIOpen: one kernel instead of conv + batch-norm."""
        c, bn = self.conv, self.bn
        key = tuple((t.data_ptr(), t._version) for t in (c.weight, bn.weight, bn.bias, bn.running_mean,
                                                         bn.running_var))
        if getattr(self, "_tfold_key", None) != key:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                shift = bn.bias - bn.running_mean * scale
                shape = (1, -1, 1, 1) if isinstance(c, nn.ConvTranspose2d) else (-1, 1, 1, 1)
                self._tfold = ((c.weight * scale.view(shape)).contiguous(), shift.contiguous())
            self._tfold_key = key
        return self._tfold

    def forward(self, x):
        kind = self._hip_kind(x)
        if kind is not None:
            try:
                return self._forward_hip(x, kind)
            except DecnetHipError as e:                 # e.g. LDS budget / grid limits of the matrix-core kernel
                if e.code != UNSUPPORTED:
                    raise
        if TALLY is not None:
            _tally(self, "library", x)
        if isinstance(x, (tuple, list)):
            x = torch.cat(tuple(x), 1)
        if self.bn is not None and not self.training and not torch.is_grad_enabled() and x.is_cuda:
            w, b = self._folded_torch()
            c = self.conv
            if isinstance(c, nn.ConvTranspose2d):
                x = F.conv_transpose2d(x, w, None, c.stride, c.padding, c.output_padding, c.groups, c.dilation)
            else:
                x = F.conv2d(x, w, None, c.stride, c.padding, c.dilation, c.groups)
            # bias + ReLU in one in-place pass (the library would add the bias in a kernel of its own)
            from . import _lib
            from .ops import _stream
            B, Co, H, W = x.shape
            if x.is_contiguous() and B * Co <= 65535:
                with torch.cuda.device(x.device):
                    _lib.check(_lib.lib().decnet_bias_act_inplace(x.data_ptr(), b.data_ptr(), B, Co, H, W,
                                                                  1 if self.relu else 0, _stream(x)),
                               "decnet_bias_act_inplace")
                return x
            x = x + b.view(1, -1, 1, 1)
            return torch.relu_(x) if self.relu else x
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x) if self.relu else x


_SIDE = threading.local()


def _side_stream(device):
    """One extra HIP stream per (host thread, device): DataParallel drives replicas from worker threads."""
    d = getattr(_SIDE, "streams", None)
    if d is None:
        d = _SIDE.streams = {}
    key = torch.device(device).index
    if key not in d:
        d[key] = torch.cuda.Stream(device=device)
    return d[key]


def _seq(*units):
    return nn.Sequential(*units)


def _c3(cin, cout, **kw):
    return Unit(cin, cout, 3, pad=kw.pop("pad", 1), **kw)


class UpBlock(nn.Module):
    """Deconv2dBlock (submodule.py:162-178): x3 transposed conv, concat with the skip, two 3x3."""

    def __init__(self, cin, cout):
        super().__init__()
        self.deconv = Unit(cin, cout, 3, stride=3, transposed=True)
        self.conv = _seq(_c3(2 * cout, cout), _c3(cout, cout))

    def forward(self, skip, x):
        up = self.deconv(x)
        return self.conv((up, skip)), up                # Unit takes the concatenation as a tuple


class ASPP(nn.Module):
    """submodule.py:225-241: a 1x1 branch and three dilated 3x3 branches, concatenated."""

    def __init__(self, cin, cout, rates):
        super().__init__()
        self.stages = nn.Module()
        self.stages.add_module("c0", Unit(cin, cout, 1))
        for i, r in enumerate(rates):
            self.stages.add_module("c%d" % (i + 1), Unit(cin, cout, 3, pad=r, dil=r))

    # ---- fused path (csrc/tapconv.hip): one V, one batched per-tap GEMM, one gather for all branches ----
    def _hip_ok(self, x):
        if (self.training or not x.is_cuda or x.dtype != torch.float32 or torch.is_grad_enabled() or
                os.environ.get("DECNET_CONV2D", "hip") != "hip"):
            return False
        units = list(self.stages.children())
        c0 = units[0].conv
        if len(units) > 4 or c0.in_channels % 4 or c0.out_channels > 224 or x.shape[-1] * x.shape[-2] > 16384:
            return False
        for u in units:
            c = u.conv
            k = c.kernel_size[0]
            if (not isinstance(c, nn.Conv2d) or u.bn is None or not u.relu or c.kernel_size not in ((1, 1), (3, 3)) or
                    c.stride != (1, 1) or c.dilation[0] != c.dilation[1] or c.groups != 1 or
                    c.padding != (c.dilation[0] * (k // 2),) * 2 or c.out_channels != c0.out_channels or
                    c.in_channels != c0.in_channels):
                return False
        return True

    def _packed(self):
        from . import _lib
        units = list(self.stages.children())
        ts = [t for u in units for t in (u.conv.weight, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var)]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if getattr(self, "_pk_key", None) != key:
            L = _lib.lib()
            dev = units[0].conv.weight.device
            ci, co = units[0].conv.in_channels, units[0].conv.out_channels
            ks = [u.conv.kernel_size[0] for u in units]
            tap0 = [sum(k * k for k in ks[:i]) for i in range(len(ks))]
            ntaps = sum(k * k for k in ks)
            with torch.no_grad(), torch.cuda.device(dev):
                st = torch.cuda.current_stream(dev).cuda_stream
                u_all = torch.empty(L.decnet_tapconv_weight_floats(ci, ntaps), dtype=torch.float32, device=dev)
                scale, shift = [], []
                for u, k, t0 in zip(units, ks, tap0):
                    w = u.conv.weight.detach().float().contiguous()
                    _lib.check(L.decnet_tapconv_pack_weight(w.data_ptr(), u_all.data_ptr(), co, ci, k, t0, st),
                               "decnet_tapconv_pack_weight")
                    sc = u.bn.weight.float() / torch.sqrt(u.bn.running_var.float() + u.bn.eps)
                    scale.append(sc)
                    shift.append(u.bn.bias.float() - u.bn.running_mean.float() * sc)
                _lib.check(L.decnet_tapconv_split_weight(u_all.data_ptr(), ci, ntaps, st), "decnet_tapconv_split_weight")
                torch.cuda.current_stream(dev).synchronize()          # w temporaries may go now
            self._pk = dict(u=u_all, scale=torch.cat(scale).contiguous(), shift=torch.cat(shift).contiguous(),
                            ks=ks, tap0=tap0, ntaps=ntaps, dil=[u.conv.dilation[0] for u in units])
            self._pk_key = key
        return self._pk

    def _forward_hip(self, x):
        import ctypes
        from . import _lib
        from .ops import _stream
        pk = self._packed()
        L = _lib.lib()
        x = x.contiguous()
        B, Ci, H, W = x.shape
        Co, nb = list(self.stages.children())[0].conv.out_channels, len(pk["ks"])
        P = B * H * W
        V = torch.empty(L.decnet_tapconv_chunk_floats(B, Ci, H, W), dtype=torch.float32, device=x.device)
        T = torch.empty(pk["ntaps"] * ((Co + 15) // 16) * 16 * P, dtype=torch.float32, device=x.device)
        y = torch.empty((B, nb * Co, H, W), dtype=torch.float32, device=x.device)
        arr = lambda v: (ctypes.c_int * len(v))(*v)
        with torch.cuda.device(x.device):
            st = _stream(x)
            _lib.check(L.decnet_tapconv_to_chunks(x.data_ptr(), V.data_ptr(), B, Ci, H, W, st), "decnet_tapconv_to_chunks")
            _lib.check(L.decnet_tap_gemm(V.data_ptr(), pk["u"].data_ptr(), T.data_ptr(), P, Ci, Co, pk["ntaps"], 1, st),
                       "decnet_tap_gemm")
            _lib.check(L.decnet_tapconv_gather(T.data_ptr(), pk["scale"].data_ptr(), pk["shift"].data_ptr(),
                                               y.data_ptr(), B, Co, H, W, nb, arr(pk["tap0"]), arr(pk["ks"]),
                                               arr(pk["dil"]), 1, st), "decnet_tapconv_gather")
        return y

    def forward(self, x):
        if self._hip_ok(x):
            return self._forward_hip(x)
        return torch.cat([s(x) for s in self.stages.children()], 1)


class FeatExtNetChannelPlus(nn.Module):
    """submodule.py:245-343 for num_stage=4, down_scale in {3}: 1, 1/3, 1/9, 1/27 resolution with
    C, 3C, 9C, 27C channels; ``out_channels`` is coarse-to-fine like the reference's."""

    def __init__(self, base_channels, num_stage=4, down_scale=3):
        super().__init__()
        assert num_stage == 4 and down_scale == 3, "the shipped configuration (demo.sh / eval.sh)"
        c, s = base_channels, down_scale
        c1, c2, c3 = c * s, c * s * s, c * s ** 3
        self.conv0 = _seq(_c3(3, c), _c3(c, c))
        self.addition_trans0 = Unit(c, c, 1)
        self.conv1 = _seq(_c3(c, c1, stride=s), _c3(c1, c1), _c3(c1, c1))
        self.addition_trans1 = Unit(c1, c1, 1)
        self.deconv1 = UpBlock(c1, c)
        self.conv2 = _seq(_c3(c1, c2, stride=s), _c3(c2, c2), _c3(c2, c2))
        self.addition_trans2 = Unit(c2, c2, 1)
        self.deconv2 = UpBlock(c2, c1)
        self.conv3_1 = _c3(c2, c3, stride=s)
        self.conv3_2 = _seq(_c3(c3, c3), _c3(c3, c3))
        self.addition_ctx_collection = _seq(ASPP(c3, c3, [4, 8, 12]), Unit(4 * c3, c3, 1))
        self.addition_fusion = Unit(2 * c3, c3, 1)
        self.deconv3 = UpBlock(c3, c2)
        self.out_channels = [c3, c2, c1, c]

    def forward(self, x, x2=None):
        """x2: a second image batch (the right views); the result is that of forward(cat(x, x2)) -- every op is
        per-sample in eval mode -- without materialising the concatenation: the first convolution writes the two
        halves of one output buffer."""
        if x2 is not None:
            u0 = self.conv0[0]
            kind = u0._hip_kind(x)
            if kind == "conv" and x.shape == x2.shape:
                nb = x.shape[0]
                y = torch.empty((2 * nb, u0.conv.out_channels) + tuple(x.shape[-2:]), dtype=torch.float32, device=x.device)
                u0._forward_hip(x, kind, out=y[:nb])
                u0._forward_hip(x2, kind, out=y[nb:])
                f0 = self.conv0[1](y)
            else:
                f0 = self.conv0(torch.cat([x, x2]))
        else:
            f0 = self.conv0(x)
        f1 = self.conv1(f0)
        f2 = self.conv2(f1)
        f3a = self.conv3_1(f2)
        f3 = self.addition_fusion((self.conv3_2(f3a), self.addition_ctx_collection(f3a)))    # (the concatenation is the unit's)
        s1, _ = self.deconv3(self.addition_trans2(f2), f3)
        s2, _ = self.deconv2(self.addition_trans1(f1), s1)
        s3, _ = self.deconv1(self.addition_trans0(f0), s2)
        return {"stage0": f3, "stage1": s1, "stage2": s2, "stage3": s3}


class GenerateSparseMask(nn.Module):
    """submodule.py:347-372: squared difference between the level's features and the upsampled
    previous level's, reduced to one logit per pixel."""

    def __init__(self, in_channels, down_scale):
        super().__init__()
        self.deconv = _seq(Unit(in_channels * down_scale, 8, 3, stride=3, bn=False, transposed=True),
                           _c3(8, 3, relu=False))
        self.conv_sub = _seq(_c3(in_channels, 8, bn=False), _c3(8, 3, relu=False))
        self.conv = _seq(_c3(3, 3, relu=False), Unit(3, 1, 1, relu=False))

    def forward(self, cur, pre):
        d = self.conv_sub(cur) - self.deconv(pre)
        return self.conv(d * d).squeeze(1)

    def _host_params(self):
        """The 3x3 and 1x1 units of ``conv`` with BatchNorm folded, as host arrays (90 floats; one device ->
        host copy per weight version)."""
        import ctypes
        u3, u1 = self.conv[0], self.conv[1]
        ts = [t for u in (u3, u1) for t in (u.conv.weight, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var)]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if getattr(self, "_hp_key", None) != key:
            with torch.no_grad():
                def fold(u):
                    sc = u.bn.weight.float() / torch.sqrt(u.bn.running_var.float() + u.bn.eps)
                    return sc, u.bn.bias.float() - u.bn.running_mean.float() * sc
                s3, b3 = fold(u3)
                s1, b1 = fold(u1)
                flat = torch.cat([u3.conv.weight.float().reshape(-1), s3, b3, u1.conv.weight.float().reshape(-1),
                                  s1, b1]).cpu().tolist()
            arr = lambda v: (ctypes.c_float * len(v))(*v)
            self._hp = (arr(flat[0:81]), arr(flat[81:84]), arr(flat[84:87]), arr(flat[87:90]), flat[90], flat[91])
            self._hp_key = key
        return self._hp

    def mask(self, cur, pre, thold, want_bits=False):
        """``(sigmoid(self(cur, pre)) > thold)`` as a float 0/1 plane [B,H,W] (SparseDenseNetRefinementMask.py:
        148-170).  On the GPU in eval mode the squared difference, both convolutions of ``conv``, the sigmoid
        and the threshold are one kernel (csrc/maskgen.hip)."""
        if (self.training or not cur.is_cuda or cur.dtype != torch.float32 or torch.is_grad_enabled() or
                os.environ.get("DECNET_CONV2D", "hip") != "hip" or cur.shape[-2] > 65535):
            m = (torch.sigmoid(self(cur, pre)) > thold).to(cur.dtype)
            return (m, None) if want_bits else m
        from . import _lib
        from .ops import _stream
        a, b = self.conv_sub(cur).contiguous(), self.deconv(pre).contiguous()
        B, _, H, W = a.shape
        w3, s3, b3, w1, s1, b1 = self._host_params()
        out = torch.empty((B, H, W), dtype=torch.float32, device=a.device)
        # want_bits: also the bit-packed copy the SpaMat kernels read (64 pixels per int64 word)
        bits = torch.empty((B, H, (W + 63) // 64), dtype=torch.int64, device=a.device) if want_bits else None
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().decnet_detail_mask(a.data_ptr(), b.data_ptr(), w3, s3, b3, w1, s1, b1, float(thold),
                                                     out.data_ptr(), None, bits.data_ptr() if want_bits else None,
                                                     B, H, W, _stream(a)),
                       "decnet_detail_mask")
        return (out, bits) if want_bits else out


class DynamicUpsampling(nn.Module):
    """submodule.py:566-589: x3 upsampling of a disparity map with per-pixel softmax weights over
    the 3x3 neighbourhood, predicted from the finer level's features."""

    def __init__(self, in_channels, down_scale):
        super().__init__()
        self.s = down_scale
        k = down_scale ** 2 * 9
        self.pad = nn.ReplicationPad2d(1)
        self.weight_learning = _seq(_c3(in_channels * down_scale ** 2 + 1, k), _c3(k, k),
                                    _c3(k, k, relu=False))

    def forward(self, disp, fea):
        B, h, w = disp.shape
        s2 = self.s ** 2
        hip = (self.s == 3 and fea.is_cuda and fea.dtype == torch.float32 and not torch.is_grad_enabled() and
               h <= 65535 and B * (fea.shape[1] + 1) <= 65535 and os.environ.get("DECNET_CONV2D", "hip") == "hip" and
               tuple(fea.shape[-2:]) == (3 * h, 3 * w))
        if hip:                                         # cat(disp, unfold(fea)) as one pass (csrc/unfold.hip)
            from . import _lib
            from .ops import _stream
            f, dp = fea.contiguous(), disp.contiguous()
            wts = torch.empty((B, 9 * f.shape[1] + 1, h, w), dtype=torch.float32, device=f.device)
            with torch.cuda.device(f.device):
                _lib.check(_lib.lib().decnet_unfold3_cat(f.data_ptr(), dp.data_ptr(), wts.data_ptr(), B, f.shape[1], h, w,
                                                         _stream(f)), "decnet_unfold3_cat")
        else:
            wts = torch.cat((disp.unsqueeze(1), F.unfold(fea, self.s, stride=self.s).view(B, -1, h, w)), 1)
        logits = self.weight_learning(wts)
        if (self.s == 3 and logits.is_cuda and logits.dtype == torch.float32 and not torch.is_grad_enabled() and
                h <= 65535 and os.environ.get("DECNET_CONV2D", "hip") == "hip"):
            from . import _lib
            from .ops import _stream
            lg, dp = logits.contiguous(), disp.contiguous()
            out = torch.empty((B, 3 * h, 3 * w), dtype=torch.float32, device=lg.device)
            with torch.cuda.device(lg.device):
                _lib.check(_lib.lib().decnet_dynamic_upsample3(lg.data_ptr(), dp.data_ptr(), out.data_ptr(), B, h, w,
                                                               _stream(lg)), "decnet_dynamic_upsample3")
            return out
        wts = F.softmax(logits.view(B, s2, 9, h * w), 2)
        nb = F.unfold(self.pad(disp.unsqueeze(1)), 3).unsqueeze(1)
        up = (nb * wts).sum(2).view(B, s2, h, w)
        return (F.pixel_shuffle(up, self.s) * self.s).squeeze(1)


class SoftAttention(nn.Module):
    """submodule.py:593-604"""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        self.conv = _seq(_c3(in_channels, base_channels), _c3(base_channels, base_channels),
                         _c3(base_channels, 1, relu=False))

    def forward(self, x):
        return torch.sigmoid(self.conv(x))

    def fuse(self, fea, dense, sparse, mask, var):
        """The attention and the fusion of the stage loop (SparseDenseNetRefinementMask.py:195-202) in one go:
        soft = self(cat(fea, dense, sparse, mask, -var)); dense * (1 - soft) + soft * sparse.  On the GPU the
        concatenation, the negation (folded into the first layer's weights), the sigmoid and the blend are part
        of the three convolution launches."""
        parts = (fea, dense.unsqueeze(1), sparse.unsqueeze(1), mask.unsqueeze(1), var.unsqueeze(1))
        u0, u1, u2 = self.conv[0], self.conv[1], self.conv[2]
        k0 = u0._hip_kind(parts)
        if k0 in ("conv", "mfma") and u2.conv.out_channels == 1 and not u2.relu:
            if k0 == "conv":
                t = u0._forward_hip(parts, "conv", neg_last=True)
            else:                                       # many input channels (1/9, 1/3 resolution): matrix-core kernel
                t = u0._forward_hip(parts[:4] + (-var.unsqueeze(1),), "mfma")
            t = u1(t)
            if u2._hip_kind(t) == "conv":
                return u2._forward_hip(t, "conv", epi=1, ea=dense.contiguous(), eb=sparse.contiguous()).squeeze(1)
            soft = torch.sigmoid(u2(t)).squeeze(1)
            return dense * (1 - soft) + soft * sparse
        soft = self(parts[:4] + (-var.unsqueeze(1),)).squeeze(1)
        return dense * (1 - soft) + soft * sparse


def warp_by_disparity(right, disp):
    """Refinement.get_warped_feats_by_homgrp (submodule.py:719-745): the same stretched,
    half-pixel-shifted bilinear warp as stage 0, one disparity per pixel."""
    B, C, H, W = right.shape
    if (right.is_cuda and right.dtype == torch.float32 and not torch.is_grad_enabled() and H > 1 and W > 1 and
            H <= 65535 and os.environ.get("DECNET_CONV2D", "hip") == "hip"):
        from . import _lib
        from .ops import _stream
        r, d = right.contiguous(), disp.contiguous()
        out = torch.empty_like(r)
        with torch.cuda.device(r.device):
            _lib.check(_lib.lib().decnet_warp_disparity(r.data_ptr(), d.data_ptr(), out.data_ptr(), B, C, H, W,
                                                        _stream(r)), "decnet_warp_disparity")
        return out
    ys, xs = torch.meshgrid(torch.arange(H, dtype=right.dtype, device=right.device),
                            torch.arange(W, dtype=right.dtype, device=right.device), indexing="ij")
    cx = (xs.unsqueeze(0) - disp) / ((W - 1.0) / 2.0) - 1.0
    cy = (ys / ((H - 1.0) / 2.0) - 1.0).unsqueeze(0).expand_as(cx)
    return F.grid_sample(right, torch.stack((cx, cy), 3), mode="bilinear", padding_mode="zeros",
                         align_corners=False)


class Refinement(nn.Module):
    """submodule.py:666-762: residual disparity from (left, warped right, disparity)."""
    _DIL = {0: (1, 1, 1), 1: (1, 1, 1), 2: (2, 4, 6), 3: (3, 6, 9)}

    def __init__(self, in_channels, base_channels, stage_id=-1, down_scale=3):
        super().__init__()
        c, h = in_channels, in_channels // 2
        d0, d2, d4 = self._DIL[stage_id]
        self.conv = _seq(_c3(2 * c + 1, c, pad=d0, dil=d0), _c3(c, c), _c3(c, c, pad=d2, dil=d2),
                         _c3(c, h), _c3(h, h, pad=d4, dil=d4), _c3(h, h),
                         _c3(h, 1, relu=False, bn=False))

    def forward(self, left, right, disp):
        x = (left, warp_by_disparity(right, disp), disp.unsqueeze(1))
        last = self.conv[-1]
        if last.conv.out_channels == 1:
            t = x
            for u in list(self.conv)[:-1]:
                t = u(t)
            if last._hip_kind(t) == "conv":             # disp + res as the last layer's epilogue (reference :716)
                return last._forward_hip(t, "conv", epi=2, ea=disp.contiguous()).squeeze(1), None
            res = last(t).squeeze(1)
            return disp + res, res
        res = self.conv(x).squeeze(1)
        return disp + res, res


class SparseDenseNetRefinementMask(nn.Module):
    """Same constructor arguments and ``forward`` signature as the reference class
    (SparseDenseNetRefinementMask.py:17-99, 102); inference returns ``[pred]`` (:236)."""

    def __init__(self, max_disp=192, base_channels=8, num_stage=3, down_scale=3, step=(1, 2, 3),
                 samp_num=8, sample_spa_size_list=(-1, 3, 3, 3), down_func_name="bilinear",
                 weights=(0.2, 0.6, 1.8), grad_method="detach", cost_func="cat", if_overmask=False,
                 skip_stage_id=3, use_detail=False, thold=0.5, alpha=0.1):
        super().__init__()
        assert max_disp % down_scale ** (num_stage - 1) == 0, \
            "the max_disp({}) should be divisible by down_scale({})^num_stage({})".format(
                max_disp, down_scale, num_stage)                      # reference :42
        assert cost_func in ("ssd", "cor", "cat"), "no such cost_func: {}".format(cost_func)     # submodule.py:447
        self.max_disp, self.num_stage, self.down_scale = max_disp, num_stage, down_scale
        self.skip_stage_id, self.use_detail, self.thold = skip_stage_id, use_detail, thold
        self.feature_extractor = FeatExtNetChannelPlus(base_channels, num_stage, down_scale)
        ch = self.feature_extractor.out_channels
        self.cost_regularizer = CostRegNetNoDown(ch[0], ch[0] * 2, cost_func, down_scale)
        n = num_stage - 1
        self.detail_detection = nn.ModuleList(GenerateSparseMask(ch[i + 1], down_scale) for i in range(n))
        self.dynamic_upsampling = nn.ModuleList(DynamicUpsampling(ch[i + 1], down_scale) for i in range(n))
        self.soft_attention = nn.ModuleList(SoftAttention(ch[i + 1] + 4, base_channels) for i in range(n))
        self.refinement = nn.ModuleList(Refinement(ch[i + 1], base_channels // 2 ** i, i + 1, down_scale)
                                        for i in range(n))
        self._initialize_weights()

    def _initialize_weights(self):
        """SparseDenseNetRefinementMask.py:239-257: He-normal Conv2d / Conv3d weights (fan-out), zero conv
        biases, unit BatchNorm.  ConvTranspose2d is not a Conv2d: transposed convolutions keep PyTorch's
        default init, as in the reference.  The sub-nets above are built in the reference's order with
        the reference's layer shapes, so after ``torch.manual_seed(17)`` (demo.py:70) this yields the
        reference's from-scratch tensors bit for bit (tests/test_init_cpu.py)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Conv3d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, left, right, disparity=None, left_mask_list=None, right_mask_list=None,
                is_check=False, is_eval=False):
        if self.training:
            raise NotImplementedError("inference only: call .eval() (SURVEY.md S11)")
        f2 = None
        if left.is_cuda and left.shape == right.shape:
            # both views in one pass (per-sample ops, eval BN: same result; the 1/9 and 1/27 layers
            # are too small at B pairs to fill 256 CUs)
            f2 = self.feature_extractor(left, right)
            nb = left.shape[0]
            lf = {k: v[:nb] for k, v in f2.items()}
            rf = {k: v[nb:] for k, v in f2.items()}
        else:
            lf = self.feature_extractor(left)
            rf = self.feature_extractor(right)
        nstage = self.num_stage

        def max_disp_of(stage):
            return self.max_disp // self.down_scale ** (nstage - stage - 1)

        def masks_of(stage):
            """reference :148-170 -> (lmask, rmask, lbits, rbits)"""
            L, R = lf["stage%d" % stage], rf["stage%d" % stage]
            if not self.use_detail:
                return left_mask_list[stage - 1], right_mask_list[stage - 1], None, None
            gen = self.detail_detection[stage - 1]
            # both views as one batch of 2 B samples where the levels are launch-bound (1/9, 1/3 resolution); at full
            # resolution a [16,8,540,972] tensor is 268 MB -- more than the 256 MiB Infinity Cache that keeps a
            # view's 134 MB intermediates on chip between producer and consumer (measured: 656 vs 608 us)
            if f2 is not None and 2 * L.numel() * 4 <= (96 << 20):
                m2, b2 = gen.mask(f2["stage%d" % stage], f2["stage%d" % (stage - 1)], self.thold, want_bits=True)
                nb = L.shape[0]
                return m2[:nb], m2[nb:], (b2[:nb] if b2 is not None else None), (b2[nb:] if b2 is not None else None)
            lmask, lbits = gen.mask(L, lf["stage%d" % (stage - 1)], self.thold, want_bits=True)
            rmask, rbits = gen.mask(R, rf["stage%d" % (stage - 1)], self.thold, want_bits=True)
            return lmask, rmask, lbits, rbits

        def sparse_of(stage, masks, out=None):
            """SpaMat + (no_grad) SpaVar around its output, reference :183-192, one launch; the masks as the bit-packed
            copies the mask kernel wrote where there are any (the float planes stay what SoftAttention reads)."""
            L, R = lf["stage%d" % stage], rf["stage%d" % stage]
            lmask, rmask, lbits, rbits = masks
            D = max_disp_of(stage)
            res = None
            if lbits is not None and rbits is not None and os.environ.get("DECNET_SPAMAT_BITS", "1") == "1":
                try:
                    res = spamatvar_forward_bits(L.contiguous(), R.contiguous(), lbits, rbits, D, out=out)
                except DecnetHipError as e:             # shapes only the float-mask entry's fallback kernels cover
                    if e.code != UNSUPPORTED:
                        raise
            if res is None:
                res = spamatvar_forward(L.contiguous(), R.contiguous(), lmask.contiguous(), rmask.contiguous(), D, out=out)
            return res[0], res[1]

        # The masks and the SpaMat / SpaVar pass of EVERY level depend on the feature maps only -- not on the coarser
        # level's prediction (reference :148-192).  On the GPU they run ahead on a second HIP stream, beside the stage-0
        # Conv3d stack and the DynamicUpsampling convolutions (fp32-issue- and HBM-bound kernels beside bf16 matrix-core
        # GEMMs); the main stream picks a level's results up behind an event.
        stages = [st for st in range(1, nstage) if st < self.skip_stage_id]
        ahead = {}
        overlap = left.is_cuda and stages
        if overlap:
            cur = torch.cuda.current_stream(left.device)
            side = _side_stream(left.device)
            side.wait_stream(cur)                          # the feature maps are ready
            with torch.cuda.stream(side):
                for st in stages:
                    masks = masks_of(st)
                    sp = sparse_of(st, masks)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    for t in masks + sp:                    # allocated on the side stream, consumed on the main one
                        if t is not None:
                            t.record_stream(cur)
                    ahead[st] = (masks, sp, ev)

        pred = None
        for stage in range(nstage):
            L, R = lf["stage%d" % stage], rf["stage%d" % stage]
            if stage == 0:
                # get_disp_samples -> GetCostVolume -> CostRegNetNoDown -> disparity_regression
                # (reference :127-137) as one channels-last pipeline on the matrix cores
                pred = self.cost_regularizer.stage0(L, R, max_disp_of(0))
                continue
            if stage >= self.skip_stage_id:                               # reference :143-144
                pred = F.interpolate(pred.unsqueeze(1) * self.down_scale, L.shape[-2:],
                                     mode="bicubic").squeeze(1)
                continue
            dense = self.dynamic_upsampling[stage - 1](pred, L)           # reference :178
            if overlap:
                masks, (sparse, var), ev = ahead.pop(stage)
                cur.wait_event(ev)
            else:
                masks = masks_of(stage)
                sparse, var = sparse_of(stage, masks)
            lmask = masks[0]
            fused = self.soft_attention[stage - 1].fuse(L, dense, sparse, lmask, var)     # reference :195-202
            pred, _ = self.refinement[stage - 1](L, R, fused)             # reference :207
        return [pred]


def get_model(**params):
    """modules/__init__.py:7-19"""
    if params["name"].lower() != "sparsedensenetrefinementmask":
        raise Exception("No such model: {}".format(params["name"]))
    keys = ("max_disp", "base_channels", "cost_func", "num_stage", "down_scale", "step", "samp_num",
            "sample_spa_size_list", "down_func_name", "weights", "grad_method", "if_overmask",
            "skip_stage_id", "use_detail", "thold")
    return SparseDenseNetRefinementMask(**{k: params[k] for k in keys})


def load_reference_checkpoint(model, state, strict=True):
    """demo.py:124-133: strip DataParallel's ``module.`` prefix and load.  The reference merges the checkpoint
    into the model's own state_dict, so a checkpoint with foreign key names silently loads nothing and the net
    runs on its random init; here (``strict=True``) every parameter and BatchNorm statistic of the model must
    come from the checkpoint and every checkpoint key must be consumed, else RuntimeError names the keys.
    Allowed: a missing ``num_batches_tracked`` (checkpoints of older PyTorch), and keys of the reference's
    parameter-free loss modules (``train_loss_func.*``, ``test_mask_loss_func.*``)."""
    own = model.state_dict()
    ckpt = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
    missing = [k for k in own if k not in ckpt and not k.endswith("num_batches_tracked")]
    extra = [k for k in ckpt if k not in own and not k.startswith(("train_loss_func.", "test_mask_loss_func."))]
    if strict and (missing or extra):
        raise RuntimeError("checkpoint does not match the network: %d model keys missing (e.g. %s), %d checkpoint "
                           "keys unused (e.g. %s)" % (len(missing), missing[:3], len(extra), extra[:3]))
    own.update({k: v for k, v in ckpt.items() if k in own})
    model.load_state_dict(own)
    return model
