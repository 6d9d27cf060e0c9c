#!/usr/bin/env python
"""Micro-benchmark of the SpaMat/SpaVar forward kernels at the BASELINE stage shapes.

    python tools/bench_spamat.py [--stage 3] [--mode fused|mat|var] [--density 1.0] [--iters 20]

Prints ms per launch and achieved algorithmic GB/s (SURVEY.md 8d byte counts).  Use
DECNET_SPAMAT_KERNEL=rowtile|mfma to pin the kernel variant.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import decnet_amd  # noqa: E402
from decnet_amd import ops  # noqa: E402

SHAPES = {1: (72, 60, 108, 24), 2: (24, 180, 324, 72), 3: (8, 540, 972, 216)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--mode", default="fused")
    ap.add_argument("--density", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shape", type=str, default=None, help="C,H,W,D override")
    ap.add_argument("--bits", action="store_true", help="fused mode with bit-packed masks (decnet_spamatvar_forward_bits)")
    a = ap.parse_args()
    C, H, W, D = SHAPES[a.stage] if a.shape is None else map(int, a.shape.split(","))
    B = a.batch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    L = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
    R = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
    rm = (torch.rand(B, H, W, device=dev, generator=g) < a.density).float()
    tm = (torch.rand(B, H, W, device=dev, generator=g) < a.density).float()
    outs = [torch.empty(B, H, W, device=dev) for _ in range(4)]
    mu = torch.rand(B, H, W, device=dev) * D

    if a.bits:
        def pack(m):                                      # 64 pixels per int64 word, bit i = pixel 64 w + i
            wpr = (W + 63) // 64
            z = torch.zeros(B, H, wpr * 64, dtype=torch.int64, device=dev)
            z[:, :, :W] = (m != 0).long()
            sh = torch.arange(64, device=dev, dtype=torch.int64)
            return (z.view(B, H, wpr, 64) << sh).sum(-1)   # two's complement wrap of bit 63 is the bit pattern wanted
        rb, tb = pack(rm).contiguous(), pack(tm).contiguous()

    def run():
        if a.bits:
            decnet_amd.spamatvar_forward_bits(L, R, rb, tb, D, out=tuple(outs))
        elif a.mode == "fused":
            decnet_amd.spamatvar_forward(L, R, rm, tm, D, out=tuple(outs))
        elif a.mode == "mat":
            ops.spamat_forward(L, R, rm, tm, outs[0], outs[2], outs[3], D)
        else:
            ops.spavar_forward(L, R, rm, tm, mu, outs[1], outs[2], outs[3], D)

    for _ in range(3):
        run()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(a.iters):
        run()
    end.record()
    end.synchronize()
    ms = beg.elapsed_time(end) / a.iters
    planes = {"fused": 6, "mat": 5, "var": 6}[a.mode]
    nbytes = 4.0 * B * H * W * (2 * C + planes)
    print("stage %d %s C=%d H=%d W=%d D=%d B=%d density=%.2f : %.4f ms  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)"
          % (a.stage, a.mode, C, H, W, D, B, a.density, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))


if __name__ == "__main__":
    main()
