"""Time the Winograd batched-GEMM stage alone (decnet_conv3d_wino_gemm).
    python tools/bench_wino_gemm.py [--variant 0|1] [--nt tiles] [--c 216]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from decnet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--nt", type=int, default=0, help="tiles (default: bench stage 0, B=8)")
    ap.add_argument("--c", type=int, default=216)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "relu", "small"])
    a = ap.parse_args()
    npts = (64, 144, 216)[a.variant]
    nt = a.nt or (5760, 1440, 720)[a.variant]
    C = a.c
    dev = torch.device("cuda:0")
    L = _lib.lib()
    CP = (C + 15) // 16 * 16
    V = torch.randn(npts * nt * CP, device=dev)
    U = torch.randn(L.decnet_conv3d_wino_weight_floats(C, a.variant), device=dev)
    if a.data == "zeros":
        V.zero_(), U.zero_()
    elif a.data == "relu":
        V.relu_()
    elif a.data == "small":          # few mantissa bits set
        V = V.to(torch.bfloat16).float()
        U = U.to(torch.bfloat16).float()
    M = torch.empty(npts * nt * CP, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(L.decnet_conv3d_wino_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), nt, C, C, a.variant, st), "gemm")
    for _ in range(3):
        run()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(a.iters):
        run()
    end.record()
    end.synchronize()
    ms = beg.elapsed_time(end) / a.iters
    flop = 2.0 * npts * nt * C * C
    print("wino_gemm variant=%d points=%d nt=%d C=%d: %.4f ms  %.1f TFLOP/s (%.1f%% of 157.3)  data=%s lib=%s"
          % (a.variant, npts, nt, C, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573, a.data,
             os.path.basename(os.environ.get("DECNET_HIP_LIB", "default"))))


if __name__ == "__main__":
    main()
