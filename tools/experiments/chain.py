"""EXPERIMENT, not part of the product (tools/experiments/README.md).
Host side of tools/experiments/chain2d.hip: chains of few-channel Conv2dUnit layers as one launch
(tools/experiments/decnet_chain2d.h, ``decnet_chain2d_forward``; built by tools/experiments/build.sh).  Builds the C descriptor from ``model.Unit`` modules (eval mode, BatchNorm folded) and
caches the packed weights per weight version.  No CPU fallback."""
import ctypes

import torch

import os

from decnet_amd import _lib as _prod

_HERE = os.path.dirname(os.path.abspath(__file__))


class _ChainLib:
    """ctypes handle of tools/experiments/libdecnet_chain2d.so with decnet_amd._lib's error convention."""
    _h = None
    check = staticmethod(_prod.check)
    DecnetHipError = _prod.DecnetHipError
    UNSUPPORTED = _prod.UNSUPPORTED

    @classmethod
    def lib(cls):
        if cls._h is None:
            path = os.path.join(_HERE, "libdecnet_chain2d.so")
            if not os.path.exists(path):
                raise _prod.DecnetHipError("%s not built: run tools/experiments/build.sh" % path)
            h = ctypes.CDLL(path)
            h.decnet_chain2d_packed_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
            h.decnet_chain2d_packed_bytes.restype = ctypes.c_size_t
            h.decnet_chain2d_pack_weight.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
            h.decnet_chain2d_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            cls._h = h
        return cls._h


_lib = _ChainLib

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
MAX_LAYERS, MAX_PARTS = 3, 6
PLAIN, DECONV, WARP = 0, 1, 2
EPI_AFFINE, EPI_SUBSQ = 0, 1
SINK_STORE, SINK_BLEND, SINK_ADD, SINK_MASK = 0, 1, 2, 3


class Part(ctypes.Structure):
    _fields_ = [("p", _P), ("p2", _P), ("aux", _P), ("scale", _P), ("shift", _P),
                ("c", _I), ("kind", _I), ("cp", _I), ("relu", _I)]


class Layer(ctypes.Structure):
    _fields_ = [("scale", ctypes.POINTER(_F)), ("shift", ctypes.POINTER(_F)), ("aux", _P),
                ("cin", _I), ("cout", _I), ("k", _I), ("dilation", _I), ("relu", _I), ("epilogue", _I),
                ("aux_channels", _I)]


class Desc(ctypes.Structure):
    _fields_ = [("parts", Part * MAX_PARTS), ("layers", Layer * MAX_LAYERS), ("w_packed", _P), ("out", _P),
                ("bits", _P), ("sink_a", _P), ("sink_b", _P), ("mask_w", _F * 3), ("mask_scale", _F),
                ("mask_shift", _F), ("thold", _F), ("n_parts", _I), ("n_layers", _I), ("bsplit", _I), ("B", _I),
                ("H", _I), ("W", _I), ("sink", _I), ("force_tw", _I), ("force_rows", _I), ("debug", _I)]


def _fold(unit):
    """(scale, shift) of a Unit in eval mode: BatchNorm running statistics, or 1 / conv bias."""
    c, bn = unit.conv, unit.bn
    co = c.out_channels
    if bn is not None:
        scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
        shift = bn.bias.float() - bn.running_mean.float() * scale
    else:
        scale = torch.ones(co, device=c.weight.device)
        shift = c.bias.float() if c.bias is not None else torch.zeros(co, device=c.weight.device)
    return scale, shift


def units_ok(units):
    """The layers the kernel covers: Conv2d k in {1, 3}, stride 1, padding = dilation * (k // 2), <= 8 outputs."""
    import torch.nn as nn
    if not 1 <= len(units) <= MAX_LAYERS:
        return False
    for u in units:
        c = u.conv
        if not isinstance(c, nn.Conv2d) or c.out_channels > 8 or c.kernel_size not in ((1, 1), (3, 3)):
            return False
        k = c.kernel_size[0]
        if (c.stride != (1, 1) or c.dilation[0] != c.dilation[1] or c.groups != 1 or c.padding_mode != "zeros" or
                c.padding != (c.dilation[0] * (k // 2),) * 2):
            return False
    return True


class Chain:
    """Packed weights + folded constants of a list of Units, cached on the first unit (per weight version).
    ``sign``: optional list of +-1 per input channel of the first layer (a negated input folded into its weights)."""

    def __init__(self, units, sign=None):
        L = _lib.lib()
        dev = units[0].conv.weight.device
        self.units = units
        sizes = [L.decnet_chain2d_packed_bytes(u.conv.in_channels, u.conv.kernel_size[0]) for u in units]
        self.wp = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        self.host = []
        off = 0
        with torch.no_grad(), torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            keep = []
            for i, (u, nb) in enumerate(zip(units, sizes)):
                c = u.conv
                w = c.weight.detach().float().contiguous()
                sg = None
                if i == 0 and sign is not None:
                    sg = torch.tensor(sign, dtype=torch.float32, device=dev)
                keep += [w, sg]
                _lib.check(L.decnet_chain2d_pack_weight(w.data_ptr(), sg.data_ptr() if sg is not None else None,
                                                        self.wp.data_ptr() + off, c.in_channels, c.out_channels,
                                                        c.kernel_size[0], st), "decnet_chain2d_pack_weight")
                off += nb
                sc, sh = _fold(u)
                sc, sh = sc.cpu().tolist(), sh.cpu().tolist()
                self.host.append(((_F * len(sc))(*sc), (_F * len(sh))(*sh)))
            torch.cuda.current_stream(dev).synchronize()         # the temporaries may go now

    @staticmethod
    def key(units, extra=()):
        ts = []
        for u in list(units) + list(extra):
            ts += [u.conv.weight] + ([u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var] if u.bn is not None
                                     else ([u.conv.bias] if u.conv.bias is not None else []))
        return tuple((t.data_ptr(), t._version) for t in ts)

    def desc(self, parts, B, H, W, out, sink=SINK_STORE, bsplit=1 << 30, epi=None, sink_a=None, sink_b=None, bits=None,
             mask=None, force_tw=0, force_rows=0):
        """parts: list of dicts(kind, p [, p2, aux, scale, shift, c, cp, relu]); epi: {layer index: aux tensor}."""
        d = Desc()
        assert len(parts) <= MAX_PARTS
        for i, pt in enumerate(parts):
            q = d.parts[i]
            q.p = pt["p"].data_ptr()
            q.p2 = pt["p2"].data_ptr() if pt.get("p2") is not None else None
            q.aux = pt["aux"].data_ptr() if pt.get("aux") is not None else None
            q.scale = pt["scale"].data_ptr() if pt.get("scale") is not None else None
            q.shift = pt["shift"].data_ptr() if pt.get("shift") is not None else None
            q.c, q.kind, q.cp, q.relu = int(pt["c"]), int(pt.get("kind", PLAIN)), int(pt.get("cp", 0)), int(pt.get("relu", 0))
        for i, u in enumerate(self.units):
            c, ly = u.conv, d.layers[i]
            ly.scale, ly.shift = self.host[i]
            ly.cin, ly.cout, ly.k, ly.dilation, ly.relu = (c.in_channels, c.out_channels, c.kernel_size[0], c.dilation[0],
                                                           1 if u.relu else 0)
            if epi and i in epi:
                ly.epilogue, ly.aux, ly.aux_channels = EPI_SUBSQ, epi[i].data_ptr(), int(epi[i].shape[1])
        d.w_packed, d.out = self.wp.data_ptr(), out.data_ptr()
        d.bits = bits.data_ptr() if bits is not None else None
        d.sink_a = sink_a.data_ptr() if sink_a is not None else None
        d.sink_b = sink_b.data_ptr() if sink_b is not None else None
        if mask is not None:
            w1, s1, b1, thold = mask
            d.mask_w[0], d.mask_w[1], d.mask_w[2] = w1
            d.mask_scale, d.mask_shift, d.thold = s1, b1, thold
        d.n_parts, d.n_layers, d.bsplit, d.B, d.H, d.W, d.sink = len(parts), len(self.units), int(bsplit), B, H, W, sink
        d.force_tw, d.force_rows = int(force_tw), int(force_rows)
        d.debug = int(__import__('os').environ.get('DECNET_CHAIN_DEBUG', '0'))
        return d

    def run(self, desc, stream_of):
        from .ops import _stream
        with torch.cuda.device(stream_of.device):
            _lib.check(_lib.lib().decnet_chain2d_forward(ctypes.byref(desc), _stream(stream_of)), "decnet_chain2d_forward")


def cached(owner, name, units, sign=None, extra=()):
    """The Chain of ``units`` stored on ``owner`` under ``name``, rebuilt when a weight changes."""
    key = Chain.key(units, extra)
    got = getattr(owner, name, None)
    if got is None or got[0] != key:
        got = (key, Chain(units, sign))
        object.__setattr__(owner, name, got)
    return got[1]


def conv_chain(units, xs, out=None, sign=None, owner=None, force_tw=0, force_rows=0):
    """y = units[-1](...units[0](cat(xs, 1))) as one launch.  xs: tensor or list of [B,c,H,W] tensors."""
    xs = [t.contiguous() for t in (xs if isinstance(xs, (list, tuple)) else [xs])]
    B, _, H, W = xs[0].shape
    ch = cached(owner if owner is not None else units[0], "_chain_" + str(len(units)), units, sign)
    if out is None:
        out = torch.empty((B, units[-1].conv.out_channels, H, W), dtype=torch.float32, device=xs[0].device)
    d = ch.desc([dict(p=t, c=t.shape[1]) for t in xs], B, H, W, out, force_tw=force_tw, force_rows=force_rows)
    ch.run(d, xs[0])
    return out
