"""SpaMat / SpaVar against the REFERENCE'S OWN KERNELS.

tests/golden/spamat_ref_*.npz were produced by running modules/SparseMatching/src/SM_kernel.cu and
modules/SparseVar/src/SV_kernel.cu -- compiled unmodified for gfx950 by oracle/ref_build.sh -- on an
MI355X (tests/golden/make_spamat_ref_golden.py).  They pin

  * the CPU oracle (oracle/spamat_oracle.c), CPU tests below, and
  * the HIP product path, `-m gpu` tests below (through the reference-shaped ext entry points,
    i.e. through the C ABI), plus a LIVE comparison against oracle/_ref/*.so at BASELINE config 2's
    full row shapes when those libraries travelled to the GPU box.

Tolerances.  Two fp32 implementations of the same sums differ by the summation order and by the
expf implementation (the reference's device expf vs glibc / v_exp_f32): sum_similarities 2e-5
relative, disparity 2e-4 px max abs (values reach 216 px, one ulp = 1.5e-5) and 5e-5 px mean abs,
variance 2e-4 relative + 2e-3 abs, gradients 5e-5 * max|grad| (grad_disparity: + its cancellation
floor, gd_tol below).  max_cost is the same c-ordered
fmaf chain everywhere: 1e-6 relative.
"""
import os

import numpy as np
import pytest

import sys

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import spamat_ref_cases as K  # noqa: E402
_CACHE = {}


def fixture(name):
    group = next(g for g, names in K.GROUPS.items() if name in names)
    if group not in _CACHE:
        _CACHE[group] = np.load(os.path.join(GOLDEN, f"spamat_ref_{group}.npz"))
    z = _CACHE[group]
    pre = name + "/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def inputs(name, fx):
    x = K.make_inputs(name)
    assert K.crc(x["L"], x["R"], x["rm"], x["tm"], x["g"], x["mu_noise"]) == fx["crc"], "input stream changed"
    assert int(fx["max_disp"]) == x["max_disp"]
    return x


def gscale(*a):
    return max(1.0, *(float(np.abs(v).max()) for v in a))


def check_forward(fx, out, ssum, mx):
    np.testing.assert_allclose(mx, fx["mx"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ssum, fx["ssum"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(out, fx["out"], rtol=1e-5, atol=2e-4)
    assert np.abs(out - fx["out"]).mean() < 5e-5


def check_backward(fx, gl, gr, pre=""):
    sc = gscale(fx[pre + "gl"], fx[pre + "gr"])
    assert np.abs(gl - fx[pre + "gl"]).max() < 5e-5 * sc
    assert np.abs(gr - fx[pre + "gr"]).max() < 5e-5 * sc


def gd_tol(fx, tag, x):
    """grad_disparity = -2 g sum_d e_d (d - mu) / S (SV_kernel.cu:275-325).  Around mu = SpaMat's own output (tag v0, what
    the net feeds) the sum is zero up to rounding: terms of magnitude <= max_disp cancel, so two summation orders differ
    by ~ulp(max_disp) * |g| whatever the (tiny) result is."""
    return 5e-5 * gscale(fx[tag + "_gd"]) + 2.0 ** -22 * x["max_disp"] * float(np.abs(x["g"]).max())


# ------------------------------------------------------------------ fixtures themselves
def test_provenance_names_the_reference_build():
    z = np.load(os.path.join(GOLDEN, "spamat_ref_quirks.npz"))
    prov = dict(s.split("=", 1) for s in z["provenance"])
    assert prov["arch"].startswith("gfx950")
    assert "sha256_SpaMat.so" in prov and "sha256_SpaVar.so" in prov


def test_known_answers_in_the_reference_outputs():
    """SURVEY S6, now read off the reference's own outputs rather than asserted about it."""
    fx = fixture("all_tar_off")
    assert (fx["out"] == 1.0).all() and np.allclose(fx["ssum"], 1e-6) and np.allclose(fx["mx"], 1e-6)
    fx = fixture("all_ref_off")
    assert not fx["out"].any() and not fx["ssum"].any() and not fx["mx"].any() and not fx["gl"].any()
    fx = fixture("negative_costs")
    assert np.allclose(fx["mx"], 1e-6)            # floor (SM_kernel.cu:45), not the true (negative) maximum
    fx = fixture("left_edge")
    assert fx["out"][..., 0].max() < 1e-3          # x = 0: only d = 0 -> (1e-6 + 0) / (1e-6 + e_0)
    assert not fx["out"][..., 4:].any()


# ------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("name", K.all_names())
def test_oracle_matches_reference_kernels(name):
    fx = fixture(name)
    x = inputs(name, fx)
    D = x["max_disp"]
    out, ssum, mx = oracle.spamat_forward(x["L"], x["R"], x["rm"], x["tm"], D)
    check_forward(fx, out, ssum, mx)
    # backward from the REFERENCE's saved tensors (functions/SpaMat.py:31,37)
    gl, gr = oracle.spamat_backward(x["L"], x["R"], x["rm"], x["tm"], fx["out"], fx["ssum"], fx["mx"], x["g"], D)
    check_backward(fx, gl, gr)
    for tag, mu in (("v0", fx["out"]), ("v1", fx["out"] + x["mu_noise"])):
        v, vs, vm = oracle.spavar_forward(x["L"], x["R"], x["rm"], x["tm"], mu, D)
        np.testing.assert_allclose(v, fx[tag + "_var"], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(vs, fx[tag + "_ssum"], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(vm, fx[tag + "_mx"], rtol=1e-6, atol=1e-7)
        gl, gr, gd = oracle.spavar_backward(x["L"], x["R"], x["rm"], x["tm"], mu, fx[tag + "_var"],
                                            fx[tag + "_ssum"], fx[tag + "_mx"], x["g"], D)
        check_backward(fx, gl, gr, tag + "_")
        assert np.abs(gd - fx[tag + "_gd"]).max() < gd_tol(fx, tag, x)


# ------------------------------------------------------------------ GPU: the HIP path
@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import decnet_amd  # noqa: F401  (fails loudly if libdecnet_hip.so is missing)
    return torch.device("cuda:0")


def _t(dev, *arrays):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


def providers(kind):
    """The two bindings of the same C ABI behind the reference's ext-level names: the ctypes shim (decnet_amd.ext) and
    the COMPILED pybind modules built by `python -m decnet_amd.build --pybind` into the reference's own layout
    (modules/Sparse*/build/lib/), imported with the reference's own import statement."""
    if kind == "ctypes_shim":
        from decnet_amd.ext import SpaMat as SM, SpaVar as SV
    else:
        from decnet_amd.modules.SparseMatching.build.lib import SpaMat as SM     # functions/SpaMat.py:4
        from decnet_amd.modules.SparseVar.build.lib import SpaVar as SV          # functions/SpaVar.py:4
        # this repository's modules, not the reference's build of the same names (oracle/ref.py explains)
        assert hasattr(SM, "decnet_version") and hasattr(SV, "decnet_version")
        assert "decnet_amd" in SM.__file__ and "decnet_amd" in SV.__file__
    return SM, SV


@pytest.mark.gpu
@pytest.mark.parametrize("provider", ["ctypes_shim", "compiled_module"])
@pytest.mark.parametrize("name", K.all_names())
def test_hip_matches_reference_kernels(dev, name, provider):
    import torch
    SM, SV = providers(provider)
    fx = fixture(name)
    x = inputs(name, fx)
    D = x["max_disp"]
    L, R, rm, tm, g = _t(dev, x["L"], x["R"], x["rm"], x["tm"], x["g"])
    out, ssum, mx = (torch.full_like(rm, 7.0) for _ in range(3))
    assert SM.sparse_matching_cuda_forward(L, R, rm, tm, out, ssum, mx, D) == 1
    check_forward(fx, out.cpu().numpy(), ssum.cpu().numpy(), mx.cpu().numpy())
    rout, rssum, rmx = _t(dev, fx["out"], fx["ssum"], fx["mx"])
    gl, gr = torch.full_like(L, 7.0), torch.full_like(R, 7.0)
    assert SM.sparse_matching_cuda_backward(L, R, rm, tm, rout, rssum, rmx, g, gl, gr, D) == 1
    check_backward(fx, gl.cpu().numpy(), gr.cpu().numpy())
    for tag, mu_np in (("v0", fx["out"]), ("v1", fx["out"] + x["mu_noise"])):
        (mu,) = _t(dev, mu_np)
        v, vs, vm = (torch.full_like(rm, 7.0) for _ in range(3))
        assert SV.sparse_var_cuda_forward(L, R, rm, tm, mu, v, vs, vm, D) == 1
        np.testing.assert_allclose(v.cpu().numpy(), fx[tag + "_var"], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(vs.cpu().numpy(), fx[tag + "_ssum"], rtol=2e-5, atol=1e-9)
        rv, rvs, rvm = _t(dev, fx[tag + "_var"], fx[tag + "_ssum"], fx[tag + "_mx"])
        gl, gr, gd = torch.full_like(L, 7.0), torch.full_like(R, 7.0), torch.full_like(mu, 7.0)
        assert SV.sparse_var_cuda_backward(L, R, rm, tm, mu, rv, rvs, rvm, g, gl, gr, gd, D) == 1
        check_backward(fx, gl.cpu().numpy(), gr.cpu().numpy(), tag + "_")
        assert np.abs(gd.cpu().numpy() - fx[tag + "_gd"]).max() < gd_tol(fx, tag, x)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg2_s3", "cfg3_s2", "cfg4_s1", "signed_s3"])
def test_hip_fused_forward_matches_reference_kernels(dev, name):
    """decnet_spamatvar_forward (SpaMat + SpaVar around its own output, one launch) vs the
    reference's two modules run back to back (SparseDenseNetRefinementMask.py:183-192)."""
    import decnet_amd
    fx = fixture(name)
    x = inputs(name, fx)
    L, R, rm, tm = _t(dev, x["L"], x["R"], x["rm"], x["tm"])
    o, v, s, m = decnet_amd.spamatvar_forward(L, R, rm, tm, x["max_disp"])
    check_forward(fx, o.cpu().numpy(), s.cpu().numpy(), m.cpu().numpy())
    np.testing.assert_allclose(v.cpu().numpy(), fx["v0_var"], rtol=2e-4, atol=2e-3)


LIVE = [  # BASELINE config 2 / 5 per-GPU shapes, B = 2 of the 8 / 4 samples
    (2, 72, 60, 108, 24, 1.0), (2, 72, 60, 108, 24, 0.3),
    (2, 24, 180, 324, 72, 1.0), (2, 24, 180, 324, 72, 0.3),
    (2, 8, 540, 972, 216, 1.0), (2, 8, 540, 972, 216, 0.5), (2, 8, 540, 972, 216, 0.1),
    (1, 8, 1026, 1512, 270, 0.5),       # config 4 stage 3, full size
    (2, 8, 378, 1242, 216, 0.25),       # config 3 stage 3
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,D,p", LIVE)
def test_hip_vs_reference_live_full_size(dev, B, C, H, W, D, p):
    """The reference's kernels and this repo's, same inputs, same GPU, full-size planes: forward,
    fused forward and backward."""
    import torch
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/*.so not built (oracle/ref_build.sh needs /root/reference)")
    import decnet_amd
    from decnet_amd.ext import SpaMat as SM
    g = torch.Generator(device="cpu").manual_seed(1000 + C + int(p * 10))
    L = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    R = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    rm = (torch.rand(B, H, W, generator=g) < p).float().to(dev)
    tm = (torch.rand(B, H, W, generator=g) < p).float().to(dev)
    go = torch.randn(B, H, W, generator=g).to(dev)
    ro, rs, rmx = ref.spamat_forward(L, R, rm, tm, D)
    rv, _, _ = ref.spavar_forward(L, R, rm, tm, ro, D)
    o, v, s, m = decnet_amd.spamatvar_forward(L, R, rm, tm, D)
    fx = dict(out=ro.cpu().numpy(), ssum=rs.cpu().numpy(), mx=rmx.cpu().numpy())
    check_forward(fx, o.cpu().numpy(), s.cpu().numpy(), m.cpu().numpy())
    np.testing.assert_allclose(v.cpu().numpy(), rv.cpu().numpy(), rtol=2e-4, atol=2e-3)
    rgl, rgr = ref.spamat_backward(L, R, rm, tm, ro, rs, rmx, go, D)
    gl, gr = torch.empty_like(L), torch.empty_like(R)
    assert SM.sparse_matching_cuda_backward(L, R, rm, tm, ro, rs, rmx, go, gl, gr, D) == 1
    torch.cuda.synchronize()
    sc = gscale(rgl.cpu().numpy(), rgr.cpu().numpy())
    assert float((gl - rgl).abs().max()) < 5e-5 * sc
    assert float((gr - rgr).abs().max()) < 5e-5 * sc


@pytest.mark.gpu
def test_long_flat_softmax_is_judged_by_float64(dev):
    """tools/fuzz_vs_ref.py (40 000 random shapes against oracle/_ref) found ONE family outside the fixed gates above:
    C = 3, max_disp = 270, dense rows -- 270 nearly equal exponentials per pixel.  The reference sums them in sequence in
    float32 (SM_kernel.cu:100-122) and is 5.6e-5 px (mean) from the float64 value; this repo's tile-wise sums are 6e-6 px
    from it, so the two differ by the reference's own rounding.  Gate here: not farther from float64 than the reference
    (how far the reference itself is depends on the inputs: 1e-5 .. 6e-5 px over seeds)."""
    import torch
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/*.so not built (oracle/ref_build.sh needs /root/reference)")
    import decnet_amd
    B, C, H, W, D = 2, 3, 1, 636, 270
    rs = np.random.RandomState(1670)                  # numpy: the same stream on every host (torch's CPU randn is not)
    L, R = (torch.from_numpy(np.maximum(rs.standard_normal((B, C, H, W)) * 0.5, 0).astype(np.float32)) for _ in range(2))
    rm = tm = torch.ones(B, H, W)
    ro, _, rmx = ref.spamat_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
    o, _, _, m = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
    Ld, Rd = L.double().numpy(), R.double().numpy()
    truth = np.zeros((B, H, W))
    for b in range(B):
        for x in range(W):
            ds = np.arange(0, min(D, x + 1))
            c = (Ld[b, :, 0, x][:, None] * Rd[b, :, 0][:, x - ds]).sum(0)
            e = np.exp(c - max(1e-6, c.max()))
            truth[b, 0, x] = (1e-6 + (e * ds).sum()) / (1e-6 + e.sum())
    e_ref = np.abs(ro.cpu().numpy() - truth).mean()
    e_hip = np.abs(o.cpu().numpy() - truth).mean()
    print("long flat softmax: mean |ref - f64| = %.3g px, mean |hip - f64| = %.3g px" % (e_ref, e_hip))
    assert e_hip <= e_ref + 1e-6, (e_hip, e_ref)
    assert e_hip < 2e-5                                # and close to float64 in absolute terms
    np.testing.assert_allclose(m.cpu().numpy(), rmx.cpu().numpy(), rtol=1e-6, atol=1e-7)
