"""The NaN contract of include/decnet_hip.h.  The reference propagates a NaN feature into every output whose candidate
set touches it (fmaxf / expf, SM_kernel.cu:46-58); the forward kernels here are built with -fno-honor-nans and give an
unspecified value there.  DECNET_CHECK_FINITE=1 turns that silent difference into an error at the C ABI
(DECNET_ERR_NONFINITE = -4): DecnetHipError through the ctypes path, RuntimeError through the compiled modules.
The knob is read once per process, hence the child interpreter."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import torch, decnet_amd
    from decnet_amd import _lib
    from decnet_amd.modules.SparseMatching.build.lib import SpaMat
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    L = torch.relu(torch.randn(2, 8, 6, 333, generator=g)).to(dev)       # 333: the float4 sweep has a tail
    R = torch.relu(torch.randn(2, 8, 6, 333, generator=g)).to(dev)
    m = torch.ones(2, 6, 333, device=dev)
    o, v, s, mx = decnet_amd.spamatvar_forward(L, R, m, m, 216)          # finite inputs pass
    assert torch.isfinite(o).all()
    for bad, where in ((float("nan"), (1, 3, 2, 100)), (float("inf"), (0, 0, 0, 0)), (float("nan"), (1, 7, 5, 332))):
        for side in (0, 1):
            L2, R2 = L.clone(), R.clone()
            (L2 if side == 0 else R2)[where] = bad
            before = o.clone()
            try:
                decnet_amd.spamatvar_forward(L2, R2, m, m, 216, out=(o, v, s, mx))
                raise SystemExit("non-finite input accepted: %r side %d" % (where, side))
            except _lib.DecnetHipError as e:
                assert e.code == -4, e.code
            torch.cuda.synchronize()
            assert torch.equal(o, before)                               # nothing was launched
            try:
                SpaMat.sparse_matching_cuda_forward(L2, R2, m, m, o, s, mx, 216)
                raise SystemExit("compiled module accepted a non-finite input")
            except RuntimeError as e:
                assert "-4" in str(e), str(e)
    # feature maps that are not 16-byte aligned (a view into a larger buffer) are checked too
    big = torch.zeros(2 * 8 * 6 * 333 + 1, device=dev)
    Lu = big[1:].view(2, 8, 6, 333); Lu.copy_(L); Lu[1, 2, 3, 4] = float("inf")
    assert Lu.data_ptr() % 16 != 0
    try:
        decnet_amd.spamatvar_forward(Lu, R, m, m, 216, out=(o, v, s, mx))
        raise SystemExit("non-finite element of an unaligned map accepted")
    except _lib.DecnetHipError as e:
        assert e.code == -4, e.code
    # under stream capture the check (which has to wait for the stream) is skipped, not an error
    gph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        decnet_amd.spamatvar_forward(L, R, m, m, 216, out=(o, v, s, mx))
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=st):
            decnet_amd.spamatvar_forward(L, R, m, m, 216, out=(o, v, s, mx))
    gph.replay()
    torch.cuda.synchronize()
    print("nan-contract ok")
""")


@pytest.mark.gpu
def test_check_finite_knob_rejects_nan_and_inf():
    env = dict(os.environ, DECNET_CHECK_FINITE="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "nan-contract ok" in r.stdout


@pytest.mark.gpu
def test_without_the_knob_a_nan_never_faults_or_spreads():
    """Default build: unspecified value at the pixels whose candidate set holds the NaN, every other pixel untouched."""
    import torch
    import decnet_amd
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    L = torch.relu(torch.randn(1, 8, 4, 400, generator=g)).to(dev)
    R = torch.relu(torch.randn(1, 8, 4, 400, generator=g)).to(dev)
    m = torch.ones(1, 4, 400, device=dev)
    o0, v0, s0, m0 = decnet_amd.spamatvar_forward(L, R, m, m, 64)
    R2 = R.clone()
    R2[0, 3, 2, 200] = float("nan")                  # right pixel 200 of row 2 is a candidate of left pixels 200..263
    o1, v1, s1, m1 = decnet_amd.spamatvar_forward(L, R2, m, m, 64)
    torch.cuda.synchronize()
    keep = torch.ones(1, 4, 400, dtype=torch.bool, device=dev)
    keep[0, 2, 200:264] = False
    for a, b in ((o0, o1), (v0, v1), (s0, s1), (m0, m1)):
        assert torch.equal(a[keep], b[keep])
    # the row's FIRST group is what staging lanes left of the row read on the aligned path (then zeroed): a NaN in right
    # pixel 3 of channel 0 belongs to left pixels 3 .. 66 only -- pixels 0 .. 2 keep their values (round 5 advice); the same
    # for the row's first and last elements of either view, dense rows (max_disp 216) and the fp32 band path (64)
    for D in (64, 216):
        o0, v0, s0, m0 = decnet_amd.spamatvar_forward(L, R, m, m, D)
        for side, x in ((1, 3), (1, 0), (0, 0), (0, 399), (1, 399)):
            L2, R2 = L.clone(), R.clone()
            (L2 if side == 0 else R2)[0, 0, 1, x] = float("nan")
            o1, v1, s1, m1 = decnet_amd.spamatvar_forward(L2, R2, m, m, D)
            torch.cuda.synchronize()
            keep = torch.ones(1, 4, 400, dtype=torch.bool, device=dev)
            if side == 0:
                keep[0, 1, x] = False                       # a left feature belongs to its own pixel only
            else:
                keep[0, 1, x:min(400, x + D)] = False       # right pixel x is a candidate of left pixels x .. x + D - 1
            for a, b in ((o0, o1), (v0, v1), (s0, s1), (m0, m1)):
                assert torch.equal(a[keep], b[keep]), (D, side, x)
