"""Timing of the fused conv chains (csrc/chain2d.hip) against the same layers one kernel each (conv2d_small), at the
full-resolution shapes of the graph.  python tools/bench_chain.py [--tw N] [--rows N]"""
import argparse
import os
import sys

import torch

_H = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_H, "..", ".."), _H]
import bench  # noqa: E402
import chain  # noqa: E402
from decnet_amd.model import Unit  # noqa: E402

CASES = [
    ("8->8->8 (B=8)", [(8, 8, 1), (8, 8, 1)], (8,), 8),
    ("8->8 (B=8)", [(8, 8, 1)], (8,), 8),
    ("3->8->8 conv0 (B=16)", [(3, 8, 1), (8, 8, 1)], (3,), 16),
    ("12->8->8->1 SA (B=8)", [(12, 8, 1), (8, 8, 1), (8, 1, 1)], (8, 1, 1, 1, 1), 8),
    ("17->8(d3)->8 refine head (B=8)", [(17, 8, 3), (8, 8, 1)], (8, 8, 1), 8),
    ("8->8(d6)->4 refine mid (B=8)", [(8, 8, 6), (8, 4, 1)], (8,), 8),
    ("8->8->3 detail a (B=16)", [(8, 8, 1), (8, 3, 1)], (8,), 16),
    ("16->8->8 upblock (B=16)", [(16, 8, 1), (8, 8, 1)], (8, 8), 16),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tw", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W = bench.PAD_H, bench.PAD_W
    for i, (name, specs, split, B) in enumerate(CASES):
        if a.only >= 0 and i != a.only:
            continue
        torch.manual_seed(i)
        units = [Unit(ci, co, 3, pad=d, dil=d).to(dev).eval() for ci, co, d in specs]
        xs = [torch.randn(B, c, H, W, device=dev) for c in split]
        with torch.no_grad():
            out = torch.empty(B, specs[-1][1], H, W, device=dev)
            ch = chain.cached(units[0], "_chain_bench", units)
            dsc = ch.desc([dict(p=t, c=t.shape[1]) for t in xs], B, H, W, out, force_tw=a.tw, force_rows=a.rows)
            t_chain = bench.time_kernel(lambda: ch.run(dsc, xs[0]), 20)       # the launch alone: descriptor prebuilt

            def layerwise():
                y = tuple(xs) if len(xs) > 1 else xs[0]
                for u in units:
                    y = u(y)
                return y
            t_layers = bench.time_kernel(layerwise, 20)
            ref = layerwise()
            err = float((ref - out).abs().max())
        mb = 4.0 * B * H * W * (sum(split) + specs[-1][1]) / 1e6
        print("%-34s chain %.3f ms (%.2f TB/s of in+out)   layer by layer %.3f ms   max diff %.1e" % (
            name, t_chain, mb / t_chain / 1e6, t_layers, err))


if __name__ == "__main__":
    main()
