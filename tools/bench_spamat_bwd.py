#!/usr/bin/env python
"""Micro-benchmark of SpaMat forward+backward through the autograd.Function (BASELINE config 5
shapes: 4 pairs per GPU).  python tools/bench_spamat_bwd.py [--stage 3] [--density 1.0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import decnet_amd  # noqa: E402

SHAPES = {1: (72, 60, 108, 24), 2: (24, 180, 324, 72), 3: (8, 540, 972, 216)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--density", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    C, H, W, D = SHAPES[a.stage]
    B = a.batch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    L = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g)).requires_grad_()
    R = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g)).requires_grad_()
    rm = (torch.rand(B, H, W, device=dev, generator=g) < a.density).float()
    tm = (torch.rand(B, H, W, device=dev, generator=g) < a.density).float()
    go = torch.randn(B, H, W, device=dev, generator=g)
    mod = decnet_amd.SpaMat()

    def fwd_bwd():
        L.grad = R.grad = None
        out = mod(L, R, rm, tm, D)
        out.backward(go)

    for _ in range(2):
        fwd_bwd()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(a.iters):
        L.grad = R.grad = None
        e[0].record()
        out = mod(L, R, rm, tm, D)
        e[1].record()
        out.backward(go)
        e[2].record()
        e[2].synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    tf /= a.iters
    tb /= a.iters
    nb = 4.0 * B * H * W * (4 * C + 6)
    print("stage %d B=%d density %.2f: forward %.3f ms, backward %.3f ms (%.1f GB/s algorithmic, %.2f%% of 8 TB/s)"
          % (a.stage, B, a.density, tf, tb, nb / tb / 1e6, nb / tb / 1e6 / 80))


if __name__ == "__main__":
    main()
