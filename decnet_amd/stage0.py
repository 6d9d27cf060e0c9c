"""Stage-0 dense path (coarsest level) behind the reference's own names.

Reference (modules/submodule.py): get_disp_samples :376-424, GetCostVolume :428-562,
Conv3dUnit :90-123, CostRegNetNoDown :608-662, disparity_regression :766-777; used by
SparseDenseNetRefinementMask.forward :127-137.

The classes keep the reference's constructor arguments, attribute names and state_dict
keys (``conv0.0.conv.weight``, ``conv0.0.bn.running_mean`` ...) so a checkpoint of the
reference loads unchanged.  Activations between the HIP kernels are channels-last
[B,D,H,W,C]; tensors handed back to callers have the reference's logical shapes
([B,C,D,H,W] cost volume is returned as a permuted view of the channels-last buffer).

Inference (eval mode, no autograd) only: the reference's training path through these
modules is not runnable as shipped and is outside the hot-path scope.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from .ops import _chk, _stream

BN_EPS = 1e-5


WINO_VARIANT = {"winograd": 0, "winograd4": 1, "winograd444": 2}


def conv_algo(D=None):
    """The Conv3d algorithm of the 216-channel layers.  DECNET_CONV_ALGO = "winograd" (F(2,3) on D, H, W:
    3.4x fewer multiplications than the 27-tap sum), "winograd4" (F(2,3) on D, F(4,3) on H, W: 6x),
    "winograd444" (F(4,3) on all three: 8x) or "direct" (27-tap implicit GEMM).  Unset: winograd444
    where the depth tiles of 4 pay (D = 8 or 10 at stage 0 of the shipped configurations), winograd4
    for shallow volumes.  All are fp32 on the matrix cores and differ by fp32 rounding only
    (tests/conv_numerics.py: 1.7e-6 / 3.0e-6 / 5.7e-6 / 2.9e-6 of max|y| after the 8 layers)."""
    a = os.environ.get("DECNET_CONV_ALGO", "auto").lower()
    if a == "auto":
        if D is None:
            return "winograd444"
        return "winograd444" if 216 * ((D + 3) // 4) <= 0.92 * 144 * ((D + 1) // 2) else "winograd4"
    if a not in ("winograd", "winograd4", "winograd444", "direct"):
        raise ValueError("DECNET_CONV_ALGO must be 'winograd', 'winograd4', 'winograd444' or 'direct'")
    return a


def get_disp_samples(max_dis, feature_map, stage_id=0, disprity_map=None, step=1, samp_num=9,
                     sample_spa_size=None):
    """submodule.py:376-424, stage-0 branch (:389-390): arange(max_dis) broadcast to
    [B,max_dis,H,W] (an expanded view, nothing is materialised)."""
    if not (disprity_map is None or step == -1 or stage_id == 0):
        raise NotImplementedError(
            "only the stage-0 sampling (arange) is on the hot path; the neighbourhood sampler "
            "(submodule.py:391-411) is never reached by SparseDenseNetRefinementMask")
    B, _, H, W = feature_map.size()
    return torch.arange(int(max_dis), dtype=feature_map.dtype,
                        device=feature_map.device).expand(B, H, W, -1).permute(0, 3, 1, 2)


def _ndhwc_view(x):
    """If x ([B,C,D,H,W]) is a permuted view of a contiguous [B,D,H,W,C] buffer return that
    buffer, else None."""
    y = x.permute(0, 2, 3, 4, 1)
    return y if y.is_contiguous() else None


def _to_ndhwc(x):
    y = _ndhwc_view(x)
    if y is not None:
        return y
    x = x.contiguous()
    B, C, D, H, W = x.shape
    out = torch.empty((B, D, H, W, C), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().decnet_ncdhw_to_ndhwc(x.data_ptr(), out.data_ptr(), B, C, D, H, W, _stream(x))
    _lib.check(rc, "decnet_ncdhw_to_ndhwc")
    return out


def costvol_ndhwc(left, right, max_disp, out=None, cost_func="cor"):
    """[B,C,H,W] x2 -> channels-last cost volume [B,D,H,W,C] ([B,D,H,W,2C] for cost_func="cat")."""
    _chk("left_feature_map", left)
    B, C, H, W = left.shape
    _chk("right_feature_map", right, (B, C, H, W))
    D = int(max_disp)
    CO = 2 * C if cost_func == "cat" else C
    if out is None:
        out = torch.empty((B, D, H, W, CO), dtype=torch.float32, device=left.device)
    with torch.cuda.device(left.device):
        rc = _lib.lib().decnet_costvol_forward_cf(left.data_ptr(), right.data_ptr(), out.data_ptr(),
                                                  B, C, H, W, D, _lib.COST_FUNC[cost_func], _stream(left))
    _lib.check(rc, "decnet_costvol_forward_cf")
    return out


class GetCostVolume(nn.Module):
    """forward: compute the cost volume with warped features  (submodule.py:428-562)

    warp_ope="homgrp" with disp_samples = get_disp_samples(stage 0) -- the one call site of the reference
    (SparseDenseNetRefinementMask.py:66, 131) -- and every cost_func: "cor" (demo.sh / eval.sh), "ssd" (demo.py:31's
    default), "cat" (2C channels).  warp_ope="shift" cannot run in the reference either (submodule.py:463 names an
    undefined variable) and is refused.
    return: cost volume, N*C*S*H*W (a view of the channels-last buffer)."""

    def __init__(self, warp_ope="homgrp", cost_func="ssd"):
        super(GetCostVolume, self).__init__()
        assert cost_func in ["ssd", "cor", "cat"], "no such cost_func: {}".format(cost_func)
        self.warp_ope = warp_ope
        self.cost_func = cost_func

    def forward(self, left_feature_map, right_feature_map, **kargs):
        if self.warp_ope != "homgrp":
            raise NotImplementedError("the gfx950 path implements warp_ope='homgrp' (the reference's 'shift' raises "
                                      "NameError, submodule.py:463)")
        if "disp_samples" in kargs and kargs["disp_samples"] is not None:
            ds = kargs["disp_samples"]
            D = int(ds.size(1))
            # the kernel warps by d = 0 .. D-1 (get_disp_samples, stage 0); any other hypotheses would
            # silently give a different volume than submodule.py:479-510, so refuse them (one host
            # sync: this reference-shaped module is not the fast path, Stage0 is)
            ar = torch.arange(D, dtype=ds.dtype, device=ds.device).view(1, D, 1, 1)
            if ds.dim() != 4 or not bool((ds == ar).all()):
                raise NotImplementedError("GetCostVolume on gfx950 takes the stage-0 samples "
                                          "arange(max_disp) only (submodule.py:389-390)")
        else:
            D = int(kargs["max_disp"])
        cv = costvol_ndhwc(left_feature_map.contiguous(), right_feature_map.contiguous(), D, cost_func=self.cost_func)
        return cv.permute(0, 4, 1, 2, 3)


class Conv3dUnit(nn.Module):
    """Parameter container with the reference's layout (submodule.py:90-123): .conv
    (Conv3d k3 s1 p1, no bias) and .bn (BatchNorm3d); executed by CostRegNetNoDown."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True,
                 bn_momentum=0.1, **kwargs):
        super(Conv3dUnit, self).__init__()
        assert kernel_size == 3 and stride == 1 and bn and kwargs.get("padding", 1) == 1
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=False,
                              padding=1)
        self.bn = nn.BatchNorm3d(out_channels, momentum=bn_momentum)
        self.relu = relu


class CostRegNetNoDown(nn.Module):
    """forward: regularise the cost volume  (submodule.py:608-662)
    args:    x: cost volume, N*C*S*H*W
    return:  regularised cost volume, N*S*H*W"""

    def __init__(self, in_channels, base_channels, cost_func, down_scale=3):
        super(CostRegNetNoDown, self).__init__()
        self.cost_func = cost_func
        C = in_channels
        if self.cost_func == "cat":                      # submodule.py:618-619
            self.conv_pre = nn.Conv3d(C * 2, C, 1, stride=1, padding=0, bias=False)
        self.conv0 = nn.Sequential(Conv3dUnit(C, C, padding=1), Conv3dUnit(C, C, padding=1))
        self.conv1 = nn.Sequential(Conv3dUnit(C, C, padding=1), Conv3dUnit(C, C, padding=1),
                                   Conv3dUnit(C, C, padding=1))
        self.conv2 = nn.Sequential(Conv3dUnit(C, C, padding=1), Conv3dUnit(C, C, padding=1),
                                   Conv3dUnit(C, 1, padding=1, relu=False))
        self._packed = None
        self._packed_key = None
        self._ws = {}

    def units(self):
        return list(self.conv0) + list(self.conv1) + list(self.conv2)

    def _replicate_for_data_parallel(self):
        # torch.nn.DataParallel (eval.py:145-146) shallow-copies __dict__ per replica: the packed weights may be
        # shared (they are keyed by the parameters' pointers and versions), the scratch buffers may not -- replicas
        # run on worker threads at the same time
        r = super()._replicate_for_data_parallel()
        r._ws = {}
        return r

    # ---- parameter preparation (once per weight version): repack + BN folding -----------
    def _key(self):
        k = [(self.conv_pre.weight.data_ptr(), self.conv_pre.weight._version)] if self.cost_func == "cat" else []
        for u in self.units():
            for t in (u.conv.weight, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var):
                k.append((t.data_ptr(), t._version))
        return tuple(k)

    def prepare(self, D=None):
        """Repack the 7 wide Conv3d weights to [27,Ci,CoP] on the device and fold eval-mode
        BatchNorm into per-channel scale/shift.  Cached until a parameter (or, through the choice
        of algorithm, the depth D of the volume) changes."""
        algo = conv_algo(D)
        key = (algo,) + self._key()
        if self._packed is not None and key == self._packed_key:
            return self._packed
        units = self.units()
        dev = units[0].conv.weight.device
        if dev.type != "cuda":
            raise _lib.DecnetHipError("CostRegNetNoDown parameters are on %s: move the module to "
                                      "the MI355X (no CPU fallback)" % dev)
        L = _lib.lib()
        packed = []
        with torch.no_grad(), torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for i, u in enumerate(units):
                w = u.conv.weight.detach().float().contiguous()
                Co, Ci = int(w.shape[0]), int(w.shape[1])
                # the kernels move channels in 16-byte groups: channel counts that are not a
                # multiple of 4 (never the case for the shipped 216-channel net) run zero padded
                cip = (Ci + 3) & ~3
                cop = (Co + 3) & ~3 if i < 7 else 1
                if cip != Ci or cop != Co:
                    wpad = torch.zeros((cop, cip) + tuple(w.shape[2:]), dtype=w.dtype, device=dev)
                    wpad[:Co, :Ci] = w
                    w = wpad
                bn = u.bn
                eps = float(bn.eps)
                scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + eps))
                shift = bn.bias.detach().float() - bn.running_mean.float() * scale
                if i < 7 and cop != Co:                 # padded output channels stay exactly 0
                    scale = torch.cat((scale, torch.ones(cop - Co, device=dev)))
                    shift = torch.cat((shift, torch.zeros(cop - Co, device=dev)))
                Co, Ci = cop, cip
                if i < 7:
                    CoP = L.decnet_conv3d_packed_cout(Co)
                    if CoP < 0:
                        raise _lib.DecnetHipError("Conv3d with %d output channels is not supported "
                                                  "(<= 224)" % Co)
                    wp = torch.empty((27, Ci, CoP), dtype=torch.float32, device=dev)
                    _lib.check(L.decnet_conv3d_pack_weight(w.data_ptr(), wp.data_ptr(), Co, Ci,
                                                           stream), "decnet_conv3d_pack_weight")
                    wu = None
                    if algo in WINO_VARIANT:
                        var = WINO_VARIANT[algo]
                        wu = torch.empty(L.decnet_conv3d_wino_weight_floats(Ci, var), dtype=torch.float32,
                                         device=dev)
                        _lib.check(L.decnet_conv3d_wino_pack_weight(w.data_ptr(), wu.data_ptr(), Co, Ci, var,
                                                                    stream), "decnet_conv3d_wino_pack_weight")
                    packed.append(dict(w=wp, u=wu, scale=scale.contiguous(), shift=shift.contiguous(),
                                       Ci=Ci, Co=Co, relu=1 if u.relu else 0, keep=w))
                else:
                    assert Co == 1
                    packed.append(dict(w=w, scale=float(scale.item()), shift=float(shift.item()),
                                       Ci=Ci, Co=1, relu=0))
            if self.cost_func == "cat":
                # conv_pre.weight [C,2C,1,1,1] as [Cp][2 Cp] for the zero-padded feature maps of stage0() (left half in
                # columns [0, C), right half in [Cp, Cp + C)), and as it is for a 2C-channel volume (forward())
                wpre = self.conv_pre.weight.detach().float().reshape(self.conv_pre.weight.shape[0], -1).contiguous()
                C = int(wpre.shape[0])
                cp = (C + 3) & ~3
                if cp != C:
                    wp = torch.zeros((cp, 2 * cp), dtype=torch.float32, device=dev)
                    wp[:C, :C] = wpre[:, :C]
                    wp[:C, cp:cp + C] = wpre[:, C:]
                else:
                    wp = wpre
                packed[0]["w_pre"], packed[0]["w_pre_true"] = wp, wpre
        self._packed, self._packed_key = packed, key
        return packed

    def _workspace(self, dev, n):
        ws = self._ws.get(dev)
        if ws is None or ws[0].numel() < n:
            ws = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3)]
            self._ws[dev] = ws
        return ws

    def _scratch(self, dev, n):
        t = self._ws.get(("scratch", dev))
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=torch.float32, device=dev)
            self._ws[("scratch", dev)] = t
        return t

    def run_ndhwc(self, x, want_reg=True, want_pred=True):
        """x: channels-last cost volume [B,D,H,W,C].  Returns (reg [B,D,H,W] or None,
        pred [B,H,W] or None).  The input buffer is not modified."""
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            raise NotImplementedError("CostRegNetNoDown on gfx950 is inference-only: call "
                                      ".eval() and run under torch.no_grad()")
        _chk("cost volume", x)
        B, D, H, W, C = x.shape
        algo = conv_algo(D)
        P = self.prepare(D)
        c_true = int(self.units()[0].conv.weight.shape[1])
        if self.cost_func == "cat":                      # submodule.py:651-652: x = conv_pre(x), 2C -> C channels
            if C != 2 * c_true:
                raise ValueError("cost volume has %d channels, module expects %d" % (C, 2 * c_true))
            y = torch.empty((B, D, H, W, c_true), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = _lib.lib().decnet_conv3d_pointwise(x.data_ptr(), P[0]["w_pre_true"].data_ptr(), y.data_ptr(), B, C,
                                                        c_true, D * H * W, C, 1, _stream(x))
            _lib.check(rc, "decnet_conv3d_pointwise")
            x, C = y, c_true
        if C == c_true and P[0]["Ci"] != C:             # channel count not a multiple of 4: zero pad
            x = torch.nn.functional.pad(x, (0, P[0]["Ci"] - C))
            C = P[0]["Ci"]
        if P[0]["Ci"] != C:
            raise ValueError("cost volume has %d channels, module expects %d" % (C, c_true))
        L = _lib.lib()
        dev = x.device
        a, b, c = self._workspace(dev, B * D * H * W * C)
        reg = torch.empty((B, D, H, W), dtype=torch.float32, device=dev) if want_reg else None
        pred = torch.empty((B, H, W), dtype=torch.float32, device=dev)

        wino = algo in WINO_VARIANT
        wsp = None
        if wino:
            var = WINO_VARIANT[algo]
            n = L.decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, var)
            wsp = self._ws.get(("wino", dev))
            if wsp is None or wsp.numel() < n:
                wsp = torch.empty(n, dtype=torch.float32, device=dev)
                self._ws[("wino", dev)] = wsp

        def conv(i, src, dst, res=None):
            p = P[i]
            r = res.data_ptr() if res is not None else None
            if wino:
                rc = L.decnet_conv3d_wino_bn_act(src.data_ptr(), p["u"].data_ptr(), p["scale"].data_ptr(),
                                                 p["shift"].data_ptr(), r, dst.data_ptr(), wsp.data_ptr(),
                                                 B, D, H, W, p["Ci"], p["Co"], p["relu"], var, st)
                _lib.check(rc, "decnet_conv3d_wino_bn_act[%d]" % i)
            else:
                rc = L.decnet_conv3d_bn_act(src.data_ptr(), p["w"].data_ptr(), p["scale"].data_ptr(),
                                            p["shift"].data_ptr(), r, dst.data_ptr(), B, D, H, W,
                                            p["Ci"], p["Co"], p["relu"], st)
                _lib.check(rc, "decnet_conv3d_bn_act[%d]" % i)

        with torch.cuda.device(dev):
            st = _stream(x)
            # CostRegNetNoDown.forward submodule.py:650-662
            if not self._run_stack(L, P, x, c, B, D, H, W, C, algo, st):
                conv(0, x, a)
                conv(1, a, c)                 # c = output0
                conv(2, c, a)
                conv(3, a, b)
                conv(4, b, a, res=c)          # conv1(output0) + output0
                conv(5, a, b)
                conv(6, b, c)
            p = P[7]
            regp = reg.data_ptr() if reg is not None else None
            need = L.decnet_conv3d_cout1_workspace_floats(B, D, H, W)
            if p["Ci"] <= 256 and D <= 256:
                t = a if a.numel() >= need else self._scratch(dev, need)     # a is free by now
                rc = L.decnet_conv3d_cout1_softargmax_ws(c.data_ptr(), p["w"].data_ptr(), p["scale"], p["shift"],
                                                         regp, pred.data_ptr(), t.data_ptr(), B, D, H, W,
                                                         p["Ci"], st)
                _lib.check(rc, "decnet_conv3d_cout1_softargmax_ws")
            else:
                rc = L.decnet_conv3d_cout1_softargmax(c.data_ptr(), p["w"].data_ptr(), p["scale"], p["shift"],
                                                      regp, pred.data_ptr(), B, D, H, W, p["Ci"], st)
                _lib.check(rc, "decnet_conv3d_cout1_softargmax")
        return reg, (pred if want_pred else None)

    def _run_stack(self, L, P, x, out, B, D, H, W, C, algo, st):
        """The seven C -> C units as ONE fused stack (decnet_conv3d_wino_stack_bn_act: the activations between the
        layers stay on chip); False when the shape is not covered (the caller then runs the layers one by one).
        DECNET_WINO_STACK=0 turns it off."""
        if algo not in WINO_VARIANT or os.environ.get("DECNET_WINO_STACK", "1") == "0":
            return False
        if any(not p["relu"] or p["Ci"] != C or p["Co"] != C for p in P[:7]):
            return False
        var = WINO_VARIANT[algo]
        n = L.decnet_conv3d_wino_stack_workspace_floats(B, D, H, W, C, var)
        if n == 0:
            return False
        dev = x.device
        wsp = self._ws.get(("wino", dev))
        if wsp is None or wsp.numel() < n:
            wsp = torch.empty(n, dtype=torch.float32, device=dev)
            self._ws[("wino", dev)] = wsp
        import ctypes
        arr = ctypes.c_void_p * 7
        u, sc, sh = (arr(*[P[i][k].data_ptr() for i in range(7)]) for k in ("u", "scale", "shift"))
        rc = L.decnet_conv3d_wino_stack_bn_act(x.data_ptr(), u, sc, sh, 7, 1, 4, out.data_ptr(), wsp.data_ptr(),
                                               B, D, H, W, C, var, st)
        if rc == _lib.UNSUPPORTED:
            return False
        _lib.check(rc, "decnet_conv3d_wino_stack_bn_act")
        return True

    def forward(self, x):
        reg, _ = self.run_ndhwc(_to_ndhwc(x), want_reg=True, want_pred=False)
        return reg

    def costvol_buffer(self, dev, B, D, H, W, C):
        """The channels-last cost-volume buffer of this module for one shape on one device."""
        key = ("cv", dev)
        cv = self._ws.get(key)
        if cv is None or tuple(cv.shape) != (B, D, H, W, C):
            cv = torch.empty((B, D, H, W, C), dtype=torch.float32, device=dev)
            self._ws[key] = cv
        return cv

    def stage0(self, left_feature_map, right_feature_map, max_disp, return_reg=False):
        """SparseDenseNetRefinementMask.forward :127-137 in one call: cost volume (stage-0 ``arange``
        samples) -> this regulariser -> soft-argmax, through the single C entry ``decnet_stage0_forward_cf``
        (one workspace, one ctypes call; the module's cost_func picks the volume).  [B,C,H,W] x2 -> pred [B,H,W] (, reg [B,D,H,W])."""
        if self.training or (torch.is_grad_enabled() and (left_feature_map.requires_grad or
                                                          right_feature_map.requires_grad)):
            raise NotImplementedError("CostRegNetNoDown on gfx950 is inference-only: call "
                                      ".eval() and run under torch.no_grad()")
        left = left_feature_map.contiguous()
        right = right_feature_map.contiguous()
        _chk("left_feature_map", left)
        if left.shape[1] % 4:                           # see prepare()
            padc = (0, 0, 0, 0, 0, 4 - left.shape[1] % 4)
            left = torch.nn.functional.pad(left, padc)
            right = torch.nn.functional.pad(right, padc)
        B, C, H, W = left.shape
        _chk("right_feature_map", right, (B, C, H, W))
        D = int(max_disp)
        algo = conv_algo(D)
        P = self.prepare(D)
        if P[0]["Ci"] != C:
            raise ValueError("feature maps have %d channels, module expects %d"
                             % (left_feature_map.shape[1], int(self.units()[0].conv.weight.shape[1])))
        if any(not p["relu"] for p in P[:7]):
            # decnet_stage0_forward applies ReLU behind each of the seven C -> C units, as the reference's network
            # does (submodule.py:624-648); a unit built with relu=False needs the per-layer path (forward())
            raise _lib.DecnetHipError("decnet_stage0_forward: a Conv3dUnit without ReLU is not covered by the "
                                      "single-entry stage-0 path; use CostRegNetNoDown.forward")
        L = _lib.lib()
        dev = left.device
        variant = WINO_VARIANT.get(algo, 3)
        pk = self._ws.get(("s0params", dev))
        if pk is None or pk[0] is not P or pk[1] != variant:
            sp = _lib.Stage0Params()
            for i in range(7):
                sp.w[i] = P[i]["u"].data_ptr() if variant <= 2 else P[i]["w"].data_ptr()
                sp.scale[i], sp.shift[i] = P[i]["scale"].data_ptr(), P[i]["shift"].data_ptr()
            sp.w_last, sp.scale_last, sp.shift_last = P[7]["w"].data_ptr(), P[7]["scale"], P[7]["shift"]
            pk = (P, variant, sp)
            self._ws[("s0params", dev)] = pk
        cf = _lib.COST_FUNC[self.cost_func]
        n = L.decnet_stage0_cf_workspace_floats(B, C, H, W, D, variant, cf)
        if n == 0:
            raise _lib.DecnetHipError("decnet_stage0_forward_cf: shape not supported")
        ws = self._ws.get(("s0", dev))
        if ws is None or ws.numel() < n:
            ws = torch.empty(n, dtype=torch.float32, device=dev)
            self._ws[("s0", dev)] = ws
        reg = torch.empty((B, D, H, W), dtype=torch.float32, device=dev) if return_reg else None
        pred = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        import ctypes
        with torch.cuda.device(dev):
            rc = L.decnet_stage0_forward_cf(left.data_ptr(), right.data_ptr(), ctypes.byref(pk[2]),
                                            P[0]["w_pre"].data_ptr() if self.cost_func == "cat" else None, ws.data_ptr(),
                                            reg.data_ptr() if reg is not None else None, pred.data_ptr(),
                                            B, C, H, W, D, variant, cf, _stream(left))
        _lib.check(rc, "decnet_stage0_forward_cf")
        return (pred, reg) if return_reg else pred

    def stage0_buffers(self, dev, B, C, H, W, D):
        """Views of the single-entry workspace (cost volume, first activation buffer, Winograd scratch): what
        bench.py's roofline leg times the dominant kernel on."""
        n = B * D * H * W * C
        act = (n + 63) // 64 * 64
        ws = self._ws[("s0", dev)]
        return ws[:n].view(B, D, H, W, C), ws[act:act + n], ws[4 * act:]


def disparity_regression(cost_vol, disp_samples):
    """submodule.py:766-777: softmax over dim 1, expectation of disp_samples.  N*S*H*W -> N*H*W"""
    cost_vol = cost_vol.contiguous()
    _chk("cost_vol", cost_vol)
    B, S, H, W = cost_vol.shape
    disp_samples = disp_samples.expand(B, S, H, W).contiguous()
    _chk("disp_samples", disp_samples, (B, S, H, W))
    pred = torch.empty((B, H, W), dtype=torch.float32, device=cost_vol.device)
    with torch.cuda.device(cost_vol.device):
        rc = _lib.lib().decnet_disparity_regression(cost_vol.data_ptr(), disp_samples.data_ptr(),
                                                    pred.data_ptr(), B, S, H, W, _stream(cost_vol))
    _lib.check(rc, "decnet_disparity_regression")
    return pred


class Stage0(nn.Module):
    """The whole stage-0 branch of SparseDenseNetRefinementMask.forward (:127-137) as one
    call: get_disp_samples -> GetCostVolume -> CostRegNetNoDown -> disparity_regression,
    with no [B,C,D,H,W] tensor ever leaving the channels-last workspace.  A thin wrapper of
    ``CostRegNetNoDown.stage0`` (all caches live on the regulariser, keyed by device)."""

    def __init__(self, cost_regularizer):
        super(Stage0, self).__init__()
        self.cost_regularizer = cost_regularizer

    def forward(self, left_feature_map, right_feature_map, max_disp, return_reg=False):
        return self.cost_regularizer.stage0(left_feature_map, right_feature_map, max_disp, return_reg)
