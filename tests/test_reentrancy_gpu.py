"""include/decnet_hip.h:14-15 -- "re-entrant: no global mutable state; safe to call from one host thread per
GPU".  The reference is driven that way by DataParallel (eval.py:146).  Two host threads call
decnet_spamat_forward / decnet_spamatvar_forward concurrently, each on its own HIP stream with its own
tensors, many times; every result must equal the single-threaded one bit for bit.  -m gpu."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_threads_two_streams():
    import decnet_amd
    from decnet_amd import ops
    dev = torch.device("cuda:0")
    cases = []
    for seed, (B, C, H, W, D, dens) in enumerate([(2, 8, 40, 500, 216, 1.0), (2, 24, 30, 324, 72, 0.3)]):
        g = torch.Generator(device=dev).manual_seed(seed)
        L = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
        R = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
        rm = (torch.rand(B, H, W, device=dev, generator=g) < dens).float()
        tm = (torch.rand(B, H, W, device=dev, generator=g) < dens).float()
        ref = [t.clone() for t in decnet_amd.spamatvar_forward(L, R, rm, tm, D)]
        o, s, m = (torch.empty(B, H, W, device=dev) for _ in range(3))
        ops.spamat_forward(L, R, rm, tm, o, s, m, D)
        cases.append((L, R, rm, tm, D, ref, o.clone()))
    torch.cuda.synchronize()
    errors = []

    def worker(case, n):
        L, R, rm, tm, D, ref, o_ref = case
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(n):
                    out = decnet_amd.spamatvar_forward(L, R, rm, tm, D)
                    o, s, m = (torch.empty_like(out[0]) for _ in range(3))
                    ops.spamat_forward(L, R, rm, tm, o, s, m, D)
                    st.synchronize()
                    if not all(torch.equal(a, b) for a, b in zip(out, ref)) or not torch.equal(o, o_ref):
                        errors.append("result changed under concurrency")
                        return
        except Exception as e:                      # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(c, 25)) for c in cases]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
