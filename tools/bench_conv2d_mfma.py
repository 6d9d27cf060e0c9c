#!/usr/bin/env python
"""conv2d_mfma (bf16x3 matrix-core Conv2dUnit) against the library convolution at the trunk's layer shapes:

    python tools/bench_conv2d_mfma.py [--iters 20] [--only SUBSTR]

prints ms per call for both, direct-equivalent TFLOP/s, and the max / mean abs difference."""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decnet_amd import _lib  # noqa: E402

# name, B, cins, Cout, H, W, k, dil
LAYERS = [
    ("dynup2.wl1 81->81", 8, (81,), 81, 180, 324, 3, 1),
    ("dynup2.wl0 73->81", 8, (73,), 81, 180, 324, 3, 1),
    ("deconv2.conv0 24+24->24", 16, (24, 24), 24, 180, 324, 3, 1),
    ("conv1.1 24->24", 16, (24,), 24, 180, 324, 3, 1),
    ("deconv3.conv0 72+72->72", 16, (72, 72), 72, 60, 108, 3, 1),
    ("conv2.1 72->72", 16, (72,), 72, 60, 108, 3, 1),
    ("refine0.conv0 72+72+1->72", 8, (72, 72, 1), 72, 60, 108, 3, 1),
    ("refine0.conv3 72->36", 8, (72,), 36, 60, 108, 3, 1),
    ("refine0.conv4 36->36", 8, (36,), 36, 60, 108, 3, 1),
    ("dynup1.wl0 217->81", 8, (217,), 81, 60, 108, 3, 1),
    ("dynup1.wl1 81->81", 8, (81,), 81, 60, 108, 3, 1),
    ("conv3_2 216->216", 16, (216,), 216, 20, 36, 3, 1),
    ("dynup0.wl0 649->81", 8, (649,), 81, 20, 36, 3, 1),
    ("dynup0.wl1 81->81", 8, (81,), 81, 20, 36, 3, 1),
    ("refine1.conv0 24+24+1->24 d2", 8, (24, 24, 1), 24, 180, 324, 3, 2),
    ("refine1.conv2 24->24 d4", 8, (24,), 24, 180, 324, 3, 4),
    ("trans1 24->24 1x1", 16, (24,), 24, 180, 324, 1, 1),
    ("ctx 864->216 1x1", 16, (864,), 216, 20, 36, 1, 1),
    ("detail0.conv_sub0 72->8", 8, (72,), 8, 60, 108, 3, 1),
    ("softatt0.conv0 72+4x1->8", 8, (72, 1, 1, 1, 1), 8, 60, 108, 3, 1),
    ("detail1.conv_sub0 24->8", 8, (24,), 8, 180, 324, 3, 1),
    ("softatt1.conv0 24+4x1->8", 8, (24, 1, 1, 1, 1), 8, 180, 324, 3, 1),
    ("refine0.conv6 36->1", 8, (36,), 1, 60, 108, 3, 1),
    ("detail0.deconv0 216->8 as 1x1->72", 8, (216,), 72, 20, 36, 1, 1),
]


def run_mfma(L, xs, wp, scale, shift, y, Cout, k, dil, relu):
    B, _, H, W = xs[0].shape
    ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
    cins = (ctypes.c_int * len(xs))(*[int(t.shape[1]) for t in xs])
    rc = L.decnet_conv2d_mfma_cat_bn_act(ptrs, cins, len(xs), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                         y.data_ptr(), B, Cout, H, W, k, dil, relu,
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(3)
    for name, B, cins, Cout, H, W, k, dil in LAYERS:
        if a.only not in name:
            continue
        Cin = sum(cins)
        xs = [torch.randn(B, c, H, W, device=dev, generator=g) for c in cins]
        w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
        scale = torch.rand(Cout, device=dev, generator=g) + 0.5
        shift = torch.randn(Cout, device=dev, generator=g) * 0.1
        nbytes = L.decnet_conv2d_mfma_packed_bytes(Cin, Cout, k)
        wp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = L.decnet_conv2d_mfma_pack_weight(w.data_ptr(), wp.data_ptr(), Cin, Cout, k,
                                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        y = torch.empty(B, Cout, H, W, device=dev)
        x = torch.cat(xs, 1)

        def lib_conv():
            return F.conv2d(x, w, None, 1, dil * (k // 2), dil)

        def mine():
            run_mfma(L, xs, wp, scale, shift, y, Cout, k, dil, 1)

        ref = torch.relu(lib_conv() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        mine()
        torch.cuda.synchronize()
        diff = (y - ref).abs()
        times = []
        for fn in (lib_conv, mine):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) / a.iters)
        flop = 2.0 * B * H * W * Cout * Cin * k * k
        print("%-32s lib %.3f ms  mfma %.3f ms (%.0f TFLOP/s direct-eq)  max|d| %.2e mean|d| %.2e  ref mean %.3f"
              % (name, times[0], times[1], flop / times[1] / 1e9, diff.max().item(), diff.mean().item(),
                 ref.abs().mean().item()), flush=True)


if __name__ == "__main__":
    main()
