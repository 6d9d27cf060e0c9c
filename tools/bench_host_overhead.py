#!/usr/bin/env python
"""Host time per call of the operator wrappers (ctypes marshalling, checks, allocations, autograd bookkeeping).
Tiny tensors, so the kernels are a few microseconds and the loop is host-bound: wall time / calls = host time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import decnet_amd  # noqa: E402
from decnet_amd import ops  # noqa: E402


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


def main():
    dev = torch.device("cuda:0")
    B, C, H, W, D = 1, 8, 4, 64, 16
    L, R = torch.rand(B, C, H, W, device=dev), torch.rand(B, C, H, W, device=dev)
    m = torch.ones(B, H, W, device=dev)
    o, s, mc = (torch.empty(B, H, W, device=dev) for _ in range(3))
    go = torch.ones(B, H, W, device=dev)
    gl, gr = torch.empty_like(L), torch.empty_like(R)
    print("ops.spamat_forward (preallocated)      %6.1f us" % timeit(lambda: ops.spamat_forward(L, R, m, m, o, s, mc, D)))
    print("ops.spamat_backward (preallocated)     %6.1f us" % timeit(
        lambda: ops.spamat_backward(L, R, m, m, o, s, mc, go, gl, gr, D)))
    print("spamatvar_forward (allocating)         %6.1f us" % timeit(lambda: decnet_amd.spamatvar_forward(L, R, m, m, D)))
    mod = decnet_amd.SpaMat()
    with torch.no_grad():
        print("SpaMat module, no_grad                 %6.1f us" % timeit(lambda: mod(L, R, m, m, D)))
    Lg, Rg = L.clone().requires_grad_(), R.clone().requires_grad_()
    print("SpaMat module, forward with grad       %6.1f us" % timeit(lambda: mod(Lg, Rg, m, m, D)))

    def step():
        Lg.grad = Rg.grad = None
        mod(Lg, Rg, m, m, D).backward(go)
    print("SpaMat forward + backward (autograd)   %6.1f us" % timeit(step, 1000))


if __name__ == "__main__":
    main()
