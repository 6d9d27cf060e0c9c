#!/usr/bin/env python
"""Time one fused small-channel Conv2dUnit (csrc/conv2d_small.hip) at full resolution.
    python tools/bench_conv2d.py [--cin 8 --cout 8 --k 3 --dil 1 --batch 8 --h 540 --w 972]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from decnet_amd.model import Unit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("cin", 8), ("cout", 8), ("k", 3), ("dil", 1), ("batch", 8), ("h", 540), ("w", 972), ("iters", 20)):
        ap.add_argument("--" + k, type=int, default=v)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    u = Unit(a.cin, a.cout, a.k, pad=a.dil * (a.k // 2), dil=a.dil).to(dev).eval()
    x = torch.randn(a.batch, a.cin, a.h, a.w, device=dev)
    with torch.no_grad():
        for _ in range(3):
            u(x)
        beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        beg.record()
        for _ in range(a.iters):
            u(x)
        end.record()
        end.synchronize()
    ms = beg.elapsed_time(end) / a.iters
    px = a.batch * a.h * a.w
    print("conv %d->%d k%d dil %d on [%d,%d,%d]: %.4f ms  %.2f TB/s of tensor traffic, %.1f TFLOP/s"
          % (a.cin, a.cout, a.k, a.dil, a.batch, a.h, a.w, ms, 4.0 * px * (a.cin + a.cout) / ms / 1e9,
             2.0 * px * a.cin * a.cout * a.k * a.k / ms / 1e9))


if __name__ == "__main__":
    main()
