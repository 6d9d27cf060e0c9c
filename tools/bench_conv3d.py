#!/usr/bin/env python
"""Micro-benchmark of the Conv3d(216->216, 3^3)+BN+ReLU implicit-GEMM kernel.

    python tools/bench_conv3d.py [--batch 8] [--shape D,H,W] [--ci 216] [--co 216] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decnet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--shape", default="8,20,36")
    ap.add_argument("--ci", type=int, default=216)
    ap.add_argument("--co", type=int, default=216)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--algo", default="direct", choices=["direct", "winograd", "winograd4", "winograd444"])
    a = ap.parse_args()
    D, H, W = map(int, a.shape.split(","))
    B, Ci, Co = a.batch, a.ci, a.co
    dev = torch.device("cuda:0")
    L = _lib.lib()
    x = torch.randn(B, D, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.02
    wp = torch.empty(27, Ci, L.decnet_conv3d_packed_cout(Co), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.decnet_conv3d_pack_weight(w.data_ptr(), wp.data_ptr(), Co, Ci, st), "pack")
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    y = torch.empty(B, D, H, W, Co, device=dev)

    var = {"winograd": 0, "winograd4": 1, "winograd444": 2}.get(a.algo, -1)
    if a.algo != "direct":
        u = torch.empty(L.decnet_conv3d_wino_weight_floats(Ci, var), device=dev)
        _lib.check(L.decnet_conv3d_wino_pack_weight(w.data_ptr(), u.data_ptr(), Co, Ci, var, st), "wpack")
        ws = torch.empty(L.decnet_conv3d_wino_workspace_floats(B, D, H, W, Ci, Co, var), device=dev)

    def run():
        if a.algo != "direct":
            _lib.check(L.decnet_conv3d_wino_bn_act(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), None,
                                                   y.data_ptr(), ws.data_ptr(), B, D, H, W, Ci, Co, 1, var, st), "wino")
        else:
            _lib.check(L.decnet_conv3d_bn_act(x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None,
                                              y.data_ptr(), B, D, H, W, Ci, Co, 1, st), "conv")
    for _ in range(3):
        run()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(a.iters):
        run()
    end.record()
    end.synchronize()
    ms = beg.elapsed_time(end) / a.iters
    M = B * D * H * W
    flop = 2.0 * 27 * Ci * Co * M
    print(a.algo, end=" ")
    print("conv3d B=%d D=%d H=%d W=%d Ci=%d Co=%d (M=%d): %.4f ms  %.1f TFLOP/s (%.1f%% of 157.3)"
          % (B, D, H, W, Ci, Co, M, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573))


if __name__ == "__main__":
    main()
