"""Random shapes, this repo's SpaMat / SpaVar kernels against the REFERENCE'S OWN kernels (oracle/_ref, built unmodified for
gfx950 by oracle/ref_build.sh) on the same GPU: forward, fused forward, SpaMat backward, SpaVar backward.  Tolerances of
tests/test_spamat_ref.py.  python tools/fuzz_vs_ref.py [first_seed [n_seeds [seconds [wide]]]]   (wide: max_disp 273 .. 700)"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_spamat_ref as t  # noqa: E402
from oracle import ref  # noqa: E402
import decnet_amd  # noqa: E402
from decnet_amd.ext import SpaMat as SM, SpaVar as SV  # noqa: E402

assert ref.available(), "oracle/_ref/*.so missing"


def pack_bits(mk):
    """float 0/1 mask [B,H,W] -> int64 [B,H,ceil(W/64)], bit i of word w = pixel 64 w + i."""
    B, H, W = mk.shape
    wpr = (W + 63) // 64
    z = torch.zeros(B, H, wpr * 64, dtype=torch.int64, device=mk.device)
    z[:, :, :W] = (mk != 0).long()
    sh = torch.arange(64, device=mk.device, dtype=torch.int64)
    return (z.view(B, H, wpr, 64) << sh).sum(-1).contiguous()


def truth_errors(L, Rt, rm, tm, D, ro, o, rmx, m):
    """mean |disparity - float64| and max |max_cost - float64| over the active left pixels, reference and HIP."""
    Ld, Rd = L.double().cpu().numpy(), Rt.double().cpu().numpy()
    rmn, tmn = rm.cpu().numpy() != 0, tm.cpu().numpy() != 0
    B, C, H, W = Ld.shape
    e_ref = e_hip = 0.0
    m_ref = m_hip = 0.0
    cnt = 0
    ron, on, rmxn, mn = ro.cpu().numpy(), o.cpu().numpy(), rmx.cpu().numpy(), m.cpu().numpy()
    for b in range(B):
        for y in range(H):
            for x in range(W):
                if not rmn[b, y, x]:
                    continue
                ds = np.arange(0, min(D, x + 1))
                ds = ds[tmn[b, y, x - ds]]
                c = (Ld[b, :, y, x][:, None] * Rd[b, :, y][:, x - ds]).sum(0) if len(ds) else np.zeros(0)
                mx = max(1e-6, c.max()) if len(ds) else 1e-6
                e = np.exp(c - mx)
                tv = (1e-6 + (e * ds).sum()) / (1e-6 + e.sum())
                e_ref += abs(ron[b, y, x] - tv); e_hip += abs(on[b, y, x] - tv); cnt += 1
                m_ref = max(m_ref, abs(rmxn[b, y, x] - mx)); m_hip = max(m_hip, abs(mn[b, y, x] - mx))
    cnt = max(cnt, 1)
    return e_ref / cnt, e_hip / cnt, m_ref, m_hip

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
WIDE = len(sys.argv) > 4 and sys.argv[4] == "wide"
dev = torch.device("cuda:0")
t0, bad, judged = time.time(), 0, 0
done = 0
for seed in range(first, first + n):
    if time.time() - t0 > budget:
        break
    done += 1
    rng = np.random.RandomState(50000 + seed)
    C = int(rng.choice([3, 8, 8, 8, 12, 24, 24, 40, 72]))
    W = int(rng.choice([rng.randint(5, 64), rng.randint(64, 400), 4 * rng.randint(20, 250), rng.randint(400, 1100)]))
    D = int(rng.choice([rng.randint(2, 30), 24, 72, 216, min(270, W + rng.randint(0, 40))]))
    if WIDE:                            # max_disp above the band kernels' 272: csrc/spamat_wide.hip (bands + per-pixel merges)
        D = int(rng.choice([rng.randint(273, 700), 405, 621, 273, 544, 545]))
    kD = max(1.0, D / 216.0)            # absolute tolerances on disparity-sized values grow with the range
    B, H = int(rng.randint(1, 3)), int(rng.randint(1, 9))
    pr, pt = (float(rng.choice([0.0, 0.03, 0.1, 0.25, 0.5, 0.9, 1.0])) for _ in range(2))
    signed = bool(rng.randint(2))
    tag = dict(seed=seed, B=B, C=C, H=H, W=W, D=D, pr=pr, pt=pt, signed=signed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    L, Rt = (torch.randn(B, C, H, W, generator=g) * 0.5 for _ in range(2))
    if not signed:
        L, Rt = torch.relu(L), torch.relu(Rt)
    L, Rt = L.to(dev), Rt.to(dev)
    rm = (torch.rand(B, H, W, generator=g) < pr).float().to(dev)
    tm = (torch.rand(B, H, W, generator=g) < pt).float().to(dev)
    go = torch.randn(B, H, W, generator=g).to(dev)
    try:
        ro, rs, rmx = ref.spamat_forward(L, Rt, rm, tm, D)
        rv, rvs, rvm = ref.spavar_forward(L, Rt, rm, tm, ro, D)
        o, v, s, m = decnet_amd.spamatvar_forward(L, Rt, rm, tm, D)
        fx = dict(out=ro.cpu().numpy(), ssum=rs.cpu().numpy(), mx=rmx.cpu().numpy())
        try:
            t.check_forward(fx, o.cpu().numpy(), s.cpu().numpy(), m.cpu().numpy())
        except AssertionError:
            # the fixed gates assume both sides are within fp32 rounding of each other.  Where they are not (long flat
            # softmaxes: C = 3, D = 270 -- the reference sums 270 terms in sequence; cancelling signed costs), decide by
            # the float64 value: this repo's result may not be farther from it than the reference's own
            e_ref, e_hip, m_ref, m_hip = truth_errors(L, Rt, rm, tm, D, ro, o, rmx, m)
            judged += 1
            assert e_hip <= 1.05 * e_ref + 1e-6, "disparity farther from float64 than the reference: %.3e vs %.3e" % (e_hip, e_ref)
            # (the primary gate on max_cost is 1e-6 relative, tests/test_spamat_ref.py: dense rows form the costs as bf16x3
            # products, exact to ~3e-7 of a cost; the same allowance here)
            m_tol = 1e-6 * max(1.0, float(rmx.abs().max()))
            assert m_hip <= 1.05 * m_ref + 2e-7 + m_tol, "max_cost farther from float64 than the reference: %.3e vs %.3e" % (m_hip, m_ref)
            np.testing.assert_allclose(s.cpu().numpy(), fx["ssum"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(v.cpu().numpy(), rv.cpu().numpy(), rtol=2e-4, atol=2e-3 * kD * kD)
        # the bit-packed mask entry (what the graph's mask kernel feeds): the same four planes, bit for bit
        try:
            bo = decnet_amd.spamatvar_forward_bits(L, Rt, pack_bits(rm), pack_bits(tm), D)
            for a, b_, nm in zip(bo, (o, v, s, m), ("out", "var", "sum", "max")):
                assert torch.equal(a, b_), "bit-mask entry differs in " + nm
        except decnet_amd._lib.DecnetHipError as e:
            if e.code != decnet_amd._lib.UNSUPPORTED:
                raise
        rgl, rgr = ref.spamat_backward(L, Rt, rm, tm, ro, rs, rmx, go, D)
        gl, gr = torch.empty_like(L), torch.empty_like(Rt)
        assert SM.sparse_matching_cuda_backward(L, Rt, rm, tm, ro, rs, rmx, go, gl, gr, D) == 1
        torch.cuda.synchronize()
        sc = t.gscale(rgl.cpu().numpy(), rgr.cpu().numpy())
        assert float((gl - rgl).abs().max()) < 5e-5 * sc, "grad_ref"
        assert float((gr - rgr).abs().max()) < 5e-5 * sc, "grad_tar"
        # SpaVar backward around a disparity that is NOT the layer's own output (no cancellation in grad_disparity)
        mu = (ro + 0.25).contiguous()
        v2, s2, m2 = ref.spavar_forward(L, Rt, rm, tm, mu, D)
        vgl, vgr, vgd = ref.spavar_backward(L, Rt, rm, tm, mu, v2, s2, m2, go, D)
        hl, hr, hd = torch.empty_like(L), torch.empty_like(Rt), torch.empty_like(mu)
        assert SV.sparse_var_cuda_backward(L, Rt, rm, tm, mu, v2, s2, m2, go, hl, hr, hd, D) == 1
        torch.cuda.synchronize()
        sc = max(t.gscale(vgl.cpu().numpy(), vgr.cpu().numpy()), float(vgd.abs().max()))
        for a, b, nm in ((hl, vgl, "var grad_ref"), (hr, vgr, "var grad_tar"), (hd, vgd, "var grad_disparity")):
            assert float((a - b).abs().max()) < 6e-5 * sc * kD, nm
    except AssertionError as e:
        bad += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("FAILED", tag, "line %d: %s" % (tb.lineno, tb.line), str(e)[:300].replace("\n", " | "), flush=True)
print("%d cases against oracle/_ref (seeds %d ..), %d failed, %d decided by the float64 value, %.0f s" % (done, first, bad, judged, time.time() - t0))
sys.exit(1 if bad else 0)
