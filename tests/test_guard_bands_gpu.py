"""No SpaMat / SpaVar entry writes outside its output planes (GPU AddressSanitizer is not available on this pool, so the
check is made with guard bands): every output is a window of a larger buffer whose margins hold a sentinel, pre-filled with
NaN inside; after the call the margins are intact and every element inside is written (the C ABI's contract: outputs are
fully written by the callee).  Shapes hit every route: sparse / mid / dense rows, band kernels (SpaVar, C > 24, narrow rows),
the one-pass dense-row backward at one / two / three channel blocks, max_disp above 272 (band by band), ragged widths and
unaligned planes (the windows start at odd offsets).  -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = 2048 + 3                      # odd margin: the planes are NOT 16-byte aligned
SENT = 12345.0

CASES = [  # B, C, H, W, D, density
    (2, 8, 3, 250, 216, 1.0),     # stage-3 class, dense rows, one-pass backward (one channel block)
    (1, 8, 2, 999, 216, 0.5),     # mid rows, ragged width
    (2, 8, 3, 333, 216, 0.1),     # sparse rows
    (1, 24, 3, 190, 72, 1.0),     # three channel blocks
    (1, 16, 2, 181, 72, 0.9),     # two channel blocks, ragged
    (2, 72, 3, 61, 24, 1.0),      # stage-1 class: band kernels, both sides in one launch
    (1, 8, 2, 20, 216, 1.0),      # row narrower than the band
    (1, 8, 2, 700, 405, 0.7),     # max_disp > 272: two bands
    (1, 5, 1, 1, 3, 1.0),         # one pixel
]


def guarded(n, dev):
    t = torch.full((n + 2 * G,), float("nan"), device=dev)
    t[:G] = SENT
    t[G + n:] = SENT
    return t, t[G:G + n]


def check(name, t, n):
    assert bool((t[:G] == SENT).all()) and bool((t[G + n:] == SENT).all()), name + ": write outside the buffer"
    assert not bool(torch.isnan(t[G:G + n]).any()), name + ": element not written"


@pytest.mark.parametrize("B,C,H,W,D,p", CASES)
def test_outputs_stay_inside_their_planes(B, C, H, W, D, p):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from decnet_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(B * 1000 + W)
    nf, npx = B * C * H * W, B * H * W
    # inputs too live at odd offsets of larger buffers
    def place(x):
        t, w = guarded(x.numel(), dev)
        w.copy_(x.reshape(-1).to(dev))
        return t, w
    tl, left = place(torch.relu(torch.randn(nf, generator=g)))
    tr, right = place(torch.relu(torch.randn(nf, generator=g)))
    trm, rm = place((torch.rand(npx, generator=g) < p).float())
    ttm, tm = place((torch.rand(npx, generator=g) < p).float())
    tg, go = place(torch.randn(npx, generator=g))
    outs = {k: guarded(npx, dev) for k in ("o", "s", "m", "v", "s2", "m2", "fo", "fv", "fs", "fm", "gd")}
    grads = {k: guarded(nf, dev) for k in ("gl", "gr", "vgl", "vgr")}
    P = lambda k: (outs.get(k) or grads[k])[1].data_ptr()
    a = (left.data_ptr(), right.data_ptr(), rm.data_ptr(), tm.data_ptr())
    assert L.decnet_spamat_forward(*a, P("o"), P("s"), P("m"), B, C, H, W, D, st) == 0
    assert L.decnet_spamatvar_forward(*a, P("fo"), P("fv"), P("fs"), P("fm"), B, C, H, W, D, st) == 0
    assert L.decnet_spavar_forward(*a, P("o"), P("v"), P("s2"), P("m2"), B, C, H, W, D, st) == 0
    assert L.decnet_spamat_backward(*a, P("o"), P("s"), P("m"), go.data_ptr(), P("gl"), P("gr"), B, C, H, W, D, st) == 0
    assert L.decnet_spavar_backward(*a, P("o"), P("v"), P("s2"), P("m2"), go.data_ptr(), P("vgl"), P("vgr"), P("gd"),
                                    B, C, H, W, D, st) == 0
    torch.cuda.synchronize()
    for k, (t, _) in outs.items():
        check(k, t, npx)
    for k, (t, _) in grads.items():
        check(k, t, nf)
    for name, t, n in (("left", tl, nf), ("right", tr, nf), ("rmask", trm, npx), ("tmask", ttm, npx), ("grad_out", tg, npx)):
        assert bool((t[:G] == SENT).all()) and bool((t[G + n:] == SENT).all()), name + ": an INPUT's margin was written"
    # the fused call and the two separate ones agree (same kernels, unaligned planes)
    assert torch.equal(outs["fo"][1], outs["o"][1]) or float((outs["fo"][1] - outs["o"][1]).abs().max()) < 1e-4


@pytest.mark.parametrize("shape", [(1, 216, 5, 9, 8), (2, 216, 20, 36, 8), (1, 24, 7, 11, 10), (2, 8, 3, 5, 3)])
def test_stage0_entry_stays_inside_workspace_and_outputs(shape):
    """decnet_stage0_forward_cf with a workspace of EXACTLY decnet_stage0_cf_workspace_floats floats between guard bands,
    every cost function and every Conv3d variant (fused stack where it applies, per-layer paths elsewhere)."""
    import ctypes
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import stage0 as o0
    from test_stage0_gpu import load_reg
    from decnet_amd import _lib
    B, C, H, W, D = shape
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(C + H)
    left = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    right = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    params = o0.random_params(C, 9)
    reg = load_reg(C, params, dev)
    wpre = o0.random_w_pre(C, 9).reshape(C, 2 * C).contiguous().to(dev)
    for variant, algo in ((2, "winograd444"), (1, "winograd4"), (3, "direct")):
        os.environ["DECNET_CONV_ALGO"] = algo
        try:
            P = reg.prepare(D)
        finally:
            del os.environ["DECNET_CONV_ALGO"]
        sp = _lib.Stage0Params()
        for i in range(7):
            sp.w[i] = (P[i]["u"] if variant <= 2 else P[i]["w"]).data_ptr()
            sp.scale[i], sp.shift[i] = P[i]["scale"].data_ptr(), P[i]["shift"].data_ptr()
        sp.w_last, sp.scale_last, sp.shift_last = P[7]["w"].data_ptr(), P[7]["scale"], P[7]["shift"]
        for cf in (0, 1, 2):
            n = L.decnet_stage0_cf_workspace_floats(B, C, H, W, D, variant, cf)
            assert n > 0
            tw, ws = guarded(n, dev)
            tp, pred = guarded(B * H * W, dev)
            trg, rg = guarded(B * D * H * W, dev)
            rc = L.decnet_stage0_forward_cf(left.data_ptr(), right.data_ptr(), ctypes.byref(sp),
                                            wpre.data_ptr() if cf == 2 else None, ws.data_ptr(), rg.data_ptr(),
                                            pred.data_ptr(), B, C, H, W, D, variant, cf, st)
            assert rc == 0, (algo, cf, rc)
            torch.cuda.synchronize()
            assert bool((tw[:G] == SENT).all()) and bool((tw[G + n:] == SENT).all()), ("workspace overrun", algo, cf)
            check("pred", tp, B * H * W)
            check("reg", trg, B * D * H * W)
