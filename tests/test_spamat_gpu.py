"""Parity of the HIP SpaMat / SpaVar kernels against the CPU oracle.  -m gpu.

Tolerances (fp32 path; north_star allows 1e-3 px MEAN abs diff on disparity maps):
  max_cost          : 1e-5 relative  (same c-ordered fmaf chain -> normally bit-equal)
  sum_similarities  : 2e-5 relative  (expf implementations differ by ~1 ulp per term)
  disparity output  : max abs 2e-4 px + 1e-5 relative, and mean abs < 5e-5 px (values reach
                      ~200 px where one fp32 ulp is 1.5e-5; the sums run in a different order)
  variance          : 2e-4 relative + 2e-3 abs (values reach D^2 ~ 5e4)
  gradients         : 2e-5 * max|grad| abs
"""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import decnet_amd  # noqa: F401  (fails loudly if libdecnet_hip.so is missing)
    return torch.device("cuda:0")


def make_case(seed, B, C, H, W, p_ref, p_tar, relu=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    L = torch.randn(B, C, H, W, generator=g) * scale
    R = torch.randn(B, C, H, W, generator=g) * scale
    if relu:
        L, R = torch.relu(L), torch.relu(R)
    rm = (torch.rand(B, H, W, generator=g) < p_ref).float()
    tm = (torch.rand(B, H, W, generator=g) < p_tar).float()
    return L, R, rm, tm


def check_fwd(o, s, m, out, ssum, mx, D):
    out, ssum, mx = (t.cpu().numpy() for t in (out, ssum, mx))
    np.testing.assert_allclose(mx, m, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ssum, s, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(out, o, rtol=1e-5, atol=2e-4)
    assert np.abs(out - o).mean() < 5e-5


CASES = [
    # B, C, H,  W,   D,  p_ref, p_tar
    (2, 8, 5, 300, 216, 1.0, 1.0),      # stage-3 like: W > 256 (2 tiles), D < W
    (1, 8, 4, 243, 216, 0.5, 0.5),
    (2, 24, 6, 81, 72, 0.9, 0.9),       # stage-2 like
    (1, 72, 6, 27, 24, 0.4, 0.4),       # stage-1 like
    (1, 8, 3, 100, 216, 1.0, 1.0),      # max_disp > W
    (1, 3, 2, 7, 5, 1.0, 0.5),          # tiny, odd C
    (2, 8, 7, 515, 40, 0.1, 0.1),       # sparse, ragged tiles
    (1, 8, 2, 64, 64, 1.0, 0.0),        # no valid candidate anywhere -> 1.0 (S6)
    (1, 8, 2, 64, 64, 0.0, 1.0),        # ref all off -> zeros
]


@pytest.mark.parametrize("B,C,H,W,D,pr,pt", CASES)
def test_spamat_forward_vs_oracle(dev, B, C, H, W, D, pr, pt):
    import decnet_amd
    L, R, rm, tm = make_case(11, B, C, H, W, pr, pt)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    mod = decnet_amd.SpaMat()
    out = mod(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), np.int64(D))      # numpy.int64: S13
    # reach the saved intermediates through the ext-level entry point
    from decnet_amd.ext import SpaMat as ext
    o2, s2, m2 = (torch.full((B, H, W), 7.0, device=dev) for _ in range(3))    # NOT zero filled
    assert ext.sparse_matching_cuda_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), o2, s2,
                                            m2, D) == 1
    assert torch.equal(out, o2)
    check_fwd(o, s, m, o2, s2, m2, D)
    if pr == 0.0:
        assert float(o2.abs().max()) == 0 and float(s2.abs().max()) == 0
    if pt == 0.0:
        assert (o2.cpu().numpy()[rm.numpy() != 0] == 1.0).all()


@pytest.mark.parametrize("B,C,H,W,D,pr,pt", CASES[:7])
def test_spavar_and_fused_forward_vs_oracle(dev, B, C, H, W, D, pr, pt):
    import decnet_amd
    L, R, rm, tm = make_case(12, B, C, H, W, pr, pt)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    g = torch.Generator().manual_seed(3)
    mu = torch.from_numpy(o) + torch.randn(B, H, W, generator=g)
    v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
    dL, dR, drm, dtm = (t.to(dev) for t in (L, R, rm, tm))
    var = decnet_amd.SpaVar()(dL, dR, drm, dtm, mu.to(dev), D)
    np.testing.assert_allclose(var.cpu().numpy(), v, rtol=2e-4, atol=2e-3)
    # fused: disparity = SpaMat output of the same launch
    fo, fv, fs, fm = decnet_amd.spamatvar_forward(dL, dR, drm, dtm, D)
    check_fwd(o, s, m, fo, fs, fm, D)
    v_o, _, _ = oracle.spavar_forward(L, R, rm, tm, o, D)
    # the variance is taken around the GPU's own disparity (differs from the oracle's by
    # <= 2e-4 px), d(var)/d(mu) = 2 (mu - mean) = 0 at mu = mean -> second order only
    np.testing.assert_allclose(fv.cpu().numpy(), v_o, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("B,C,H,W,D,pr,pt", [(2, 8, 4, 130, 48, 0.7, 0.7), (1, 24, 3, 81, 72, 0.9, 0.8),
                                             (1, 72, 4, 27, 24, 0.5, 0.5), (1, 8, 2, 300, 216, 1.0, 1.0),
                                             (1, 5, 2, 9, 12, 1.0, 1.0),
                                             # the one-pass dense-row kernel with 2 / 3 channel blocks (round 6): odd widths
                                             # (4-byte stores), channel counts that fill a block partly, the 16-tile ring
                                             (1, 24, 3, 131, 72, 1.0, 1.0), (2, 13, 2, 101, 40, 1.0, 0.9),
                                             (1, 20, 2, 210, 150, 1.0, 1.0), (1, 17, 3, 324, 72, 0.95, 0.95),
                                             (1, 3, 2, 133, 30, 1.0, 1.0)])
def test_backward_vs_oracle(dev, B, C, H, W, D, pr, pt):
    import decnet_amd
    L, R, rm, tm = make_case(13, B, C, H, W, pr, pt, relu=False, scale=0.5)
    g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(5))
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
    dL, dR = L.to(dev).requires_grad_(), R.to(dev).requires_grad_()
    out = decnet_amd.SpaMatFunction.apply(dL, dR, rm.to(dev), tm.to(dev), D)
    out.backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 2e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 2e-5 * sc
    # SpaVar: three gradients, disparity differentiable
    mu = torch.from_numpy(o) + 0.25
    v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
    gl, gr, gd = oracle.spavar_backward(L, R, rm, tm, mu, v, s2, m2, g, D)
    dL, dR = L.to(dev).requires_grad_(), R.to(dev).requires_grad_()
    dmu = mu.to(dev).requires_grad_()
    var = decnet_amd.SpaVarFunction.apply(dL, dR, rm.to(dev), tm.to(dev), dmu, D)
    var.backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()), float(np.abs(gd).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 5e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 5e-5 * sc
    assert np.abs(dmu.grad.cpu().numpy() - gd).max() < 5e-5 * sc


@pytest.mark.parametrize("B,C,H,W,D,p", [
    (1, 8, 2, 1242, 216, 1.0),     # BASELINE config 3 (KITTI) stage 3: row split into segments
    (1, 8, 2, 1242, 216, 0.15),    #   ... sparse rows -> compaction path with segments
    (1, 8, 2, 1512, 270, 1.0),     # config 4 (Middlebury half-res) stage 3: 18-tile band
    (1, 8, 2, 1512, 270, 0.1),
    (1, 24, 2, 504, 90, 0.6),      # config 4 stage 2
    (1, 72, 3, 168, 30, 1.0),      # config 4 stage 1
    (1, 8, 1, 1100, 300, 0.5),     # band wider than the MFMA kernel covers -> row-tile fallback
])
def test_wide_rows_and_other_configs(dev, B, C, H, W, D, p):
    """Shapes of BASELINE configs 3 and 4 (full width, a few rows): forward (fused) and backward."""
    import decnet_amd
    L, R, rm, tm = make_case(31, B, C, H, W, p, p)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    v, _, _ = oracle.spavar_forward(L, R, rm, tm, o, D)
    dL, dR, drm, dtm = (t.to(dev) for t in (L, R, rm, tm))
    fo, fv, fs, fm = decnet_amd.spamatvar_forward(dL, dR, drm, dtm, D)
    check_fwd(o, s, m, fo, fs, fm, D)
    np.testing.assert_allclose(fv.cpu().numpy(), v, rtol=2e-4, atol=3e-3)
    g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(6))
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
    dL.requires_grad_(); dR.requires_grad_()
    decnet_amd.SpaMatFunction.apply(dL, dR, drm, dtm, D).backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 5e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 5e-5 * sc


@pytest.mark.parametrize("C,W,D", [(8, 972, 216), (8, 640, 216), (24, 324, 72),
                                   (8, 1242, 216),      # KITTI stage 3: 8 pixels per thread, 2 marker segments
                                   (8, 1512, 270)])     # Middlebury half-res stage 3
def test_mixed_row_densities_sparse_kernel_and_handover(dev, C, W, D):
    """Rows of every kind in one call: very sparse (sparse-row kernel), dense and medium (handed to
    the band kernel through the -1 marker), > 256 active pixels on one side only, a dense cluster
    of active right pixels inside an otherwise sparse row (windows too full even for 16-pixel
    spans -> handed over after the span search), empty rows.  SpaMat, SpaVar and the fused call."""
    import decnet_amd
    H = 12
    g = torch.Generator().manual_seed(77)
    L = torch.relu(torch.randn(1, C, H, W, generator=g))
    R = torch.relu(torch.randn(1, C, H, W, generator=g))
    dens = [(0.02, 0.02), (1.0, 1.0), (0.1, 0.1), (0.5, 0.5), (0.05, 0.9), (0.9, 0.05), (0.2, 0.2),
            (0.0, 0.3), (0.3, 0.0), (0.26, 0.26), (0.6, 0.6), (0.03, 0.03)]    # 0.6: > 512 active pixels per side, < 45 % of the pairs
    rm = torch.stack([(torch.rand(W, generator=g) < p).float() for p, _ in dens]).view(1, H, W)
    tm = torch.stack([(torch.rand(W, generator=g) < p).float() for _, p in dens]).view(1, H, W)
    tm[0, 10, 100:300] = 1.0                       # cluster: 200 consecutive active right pixels
    rm[0, 11, W - 40:] = 1.0                       # cluster of active left pixels at the right edge
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    v, sv, mv = oracle.spavar_forward(L, R, rm, tm, o, D)
    dL, dR, drm, dtm = (t.to(dev) for t in (L, R, rm, tm))
    fo, fv, fs, fm = decnet_amd.spamatvar_forward(dL, dR, drm, dtm, D)
    check_fwd(o, s, m, fo, fs, fm, D)
    np.testing.assert_allclose(fv.cpu().numpy(), v, rtol=2e-4, atol=3e-3)
    from decnet_amd import ops
    o2, s2, m2 = (torch.full((1, H, W), -3.0, device=dev) for _ in range(3))
    ops.spamat_forward(dL, dR, drm, dtm, o2, s2, m2, D)
    check_fwd(o, s, m, o2, s2, m2, D)
    v3, s3, m3 = (torch.full((1, H, W), -3.0, device=dev) for _ in range(3))
    ops.spavar_forward(dL, dR, drm, dtm, torch.from_numpy(o).to(dev), v3, s3, m3, D)
    np.testing.assert_allclose(v3.cpu().numpy(), v, rtol=2e-4, atol=3e-3)
    np.testing.assert_allclose(s3.cpu().numpy(), sv, rtol=2e-5, atol=1e-9)
    # backward through the same mix of rows: sparse-row backward kernel + marker hand-over
    g = torch.randn(1, H, W, generator=torch.Generator().manual_seed(9))
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
    aL, aR = dL.clone().requires_grad_(), dR.clone().requires_grad_()
    decnet_amd.SpaMatFunction.apply(aL, aR, drm, dtm, D).backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
    assert np.abs(aL.grad.cpu().numpy() - gl).max() < 5e-5 * sc
    assert np.abs(aR.grad.cpu().numpy() - gr).max() < 5e-5 * sc
    mu = torch.from_numpy(o) + 0.25
    v2, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
    gl, gr, gd = oracle.spavar_backward(L, R, rm, tm, mu, v2, s2, m2, g, D)
    aL, aR, amu = dL.clone().requires_grad_(), dR.clone().requires_grad_(), mu.to(dev).requires_grad_()
    decnet_amd.SpaVarFunction.apply(aL, aR, drm, dtm, amu, D).backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()), float(np.abs(gd).max()))
    assert np.abs(aL.grad.cpu().numpy() - gl).max() < 5e-5 * sc
    assert np.abs(aR.grad.cpu().numpy() - gr).max() < 5e-5 * sc
    assert np.abs(amu.grad.cpu().numpy() - gd).max() < 5e-5 * sc


def test_net_callsite_golden(dev, golden_dir):
    """Arrays recorded at the sparse_matching / sparse_var call sites of the reference graph."""
    import decnet_amd
    d = np.load(os.path.join(golden_dir, "net_54x243.npz"))
    for i in (1, 2, 3):
        args = [torch.from_numpy(d["sm%d_%s" % (i, n)]).to(dev) for n in ("ref", "tar", "rmask", "tmask")]
        D = d["sm%d_max_disp" % i][()]
        assert isinstance(D, np.int64)
        out = decnet_amd.SpaMat()(*args, D)
        ref_out = d["sm%d_out" % i]
        assert np.abs(out.cpu().numpy() - ref_out).max() < 2e-4
        assert np.abs(out.cpu().numpy() - ref_out).mean() < 2e-5
        var = decnet_amd.SpaVar()(*args, torch.from_numpy(d["sv%d_disparity" % i]).to(dev), D)
        np.testing.assert_allclose(var.cpu().numpy(), d["sv%d_out" % i], rtol=2e-4, atol=2e-3)


def test_full_size_properties(dev):
    """BASELINE config 2 size (B=8, 972x540, D=216, stage 3): size-independent properties."""
    import decnet_amd
    B, C, H, W, D = 8, 8, 540, 972, 216
    g = torch.Generator(device=dev).manual_seed(20)
    L = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
    R = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
    rm = (torch.rand(B, H, W, device=dev, generator=g) < 0.1).float()
    tm = (torch.rand(B, H, W, device=dev, generator=g) < 0.1).float()
    o, v, s, m = decnet_amd.spamatvar_forward(L, R, rm, tm, D)
    assert float(o[rm == 0].abs().max()) == 0 and float(v[rm == 0].abs().max()) == 0
    on = rm != 0
    assert float(o[on].min()) >= 0 and float(o[on].max()) <= D - 1 + 1e-3      # expectation of d
    assert float(v[on].min()) >= 0
    assert float(s[on].min()) >= np.float32(1e-6) and float(m[on].min()) >= np.float32(1e-6)
    # batch independence (the property the multi-GPU sharding relies on)
    o1, v1, _, _ = decnet_amd.spamatvar_forward(L[3:4].contiguous(), R[3:4].contiguous(),
                                                rm[3:4].contiguous(), tm[3:4].contiguous(), D)
    assert torch.equal(o1[0], o[3]) and torch.equal(v1[0], v[3])
    # a random sample of rows against the oracle
    rows = [(0, 0), (7, 539), (4, 271)]
    for b, y in rows:
        oo, ss, mm = oracle.spamat_forward(L[b:b + 1, :, y:y + 1].cpu(), R[b:b + 1, :, y:y + 1].cpu(),
                                           rm[b:b + 1, y:y + 1].cpu(), tm[b:b + 1, y:y + 1].cpu(), D)
        assert np.abs(o[b, y].cpu().numpy() - oo[0, 0]).max() < 2e-4


def test_rejects_bad_arguments(dev):
    import decnet_amd
    L, R, rm, tm = (t.to(dev) for t in make_case(1, 1, 4, 2, 8, 1.0, 1.0))
    with pytest.raises(decnet_amd.DecnetHipError):
        decnet_amd.SpaMat()(L.cpu(), R.cpu(), rm.cpu(), tm.cpu(), 4)            # no CPU fallback
    with pytest.raises(AssertionError):
        decnet_amd.SpaMat()(L.transpose(2, 3), R, rm, tm, 4)                    # SpaMat.py:21
    with pytest.raises(TypeError):
        decnet_amd.SpaMat()(L.double(), R.double(), rm, tm, 4)
    with pytest.raises(ValueError):
        decnet_amd.SpaMat()(L, R, rm[:, :1], tm, 4)


@pytest.mark.parametrize("seed", range(6))
def test_randomized_shapes_forward_and_backward(dev, seed):
    """Random (B, C, H, W, D, densities) around every dispatch boundary of the forward and backward kernels
    (sparse-row / band, aligned and ragged widths, D above and below W, C of the three compiled K depths
    and odd ones), several cases per seed, forward (SpaMat, fused SpaVar) and both backward passes."""
    import decnet_amd
    rng = np.random.RandomState(1000 + seed)
    for _ in range(5):
        C = int(rng.choice([3, 8, 8, 8, 12, 24, 24, 40, 72]))
        W = int(rng.choice([rng.randint(5, 64), rng.randint(64, 400), 4 * rng.randint(20, 140), rng.randint(400, 700)]))
        D = int(rng.choice([rng.randint(2, 30), 24, 72, 216, min(270, W + rng.randint(0, 40))]))
        B, H = int(rng.randint(1, 3)), int(rng.randint(1, 5))
        pr, pt = (float(rng.choice([0.0, 0.03, 0.1, 0.25, 0.5, 0.9, 1.0])) for _ in range(2))
        L, R, rm, tm = make_case(int(rng.randint(1 << 30)), B, C, H, W, pr, pt, relu=bool(rng.randint(2)), scale=0.5)
        tag = dict(B=B, C=C, H=H, W=W, D=D, pr=pr, pt=pt)
        o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
        dL, dR = L.to(dev).requires_grad_(), R.to(dev).requires_grad_()
        out = decnet_amd.SpaMatFunction.apply(dL, dR, rm.to(dev), tm.to(dev), D)
        np.testing.assert_allclose(out.detach().cpu().numpy(), o, rtol=1e-5, atol=3e-4, err_msg=str(tag))
        g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(seed))
        gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
        out.backward(g.to(dev))
        sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
        assert np.abs(dL.grad.cpu().numpy() - gl).max() < 3e-5 * sc, tag
        assert np.abs(dR.grad.cpu().numpy() - gr).max() < 3e-5 * sc, tag
        mu = torch.from_numpy(o) + 0.25
        v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
        fo, fv, fs, fm = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
        np.testing.assert_allclose(fo.cpu().numpy(), o, rtol=1e-5, atol=3e-4, err_msg=str(tag))
        vo, _, _ = oracle.spavar_forward(L, R, rm, tm, torch.from_numpy(o), D)
        np.testing.assert_allclose(fv.cpu().numpy(), vo, rtol=2e-4, atol=2e-2, err_msg=str(tag))
        gl, gr, gd = oracle.spavar_backward(L, R, rm, tm, mu, v, s2, m2, g, D)
        dL, dR, dmu = L.to(dev).requires_grad_(), R.to(dev).requires_grad_(), mu.to(dev).requires_grad_()
        decnet_amd.SpaVarFunction.apply(dL, dR, rm.to(dev), tm.to(dev), dmu, D).backward(g.to(dev))
        sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()), float(np.abs(gd).max()))
        assert np.abs(dL.grad.cpu().numpy() - gl).max() < 6e-5 * sc, tag
        assert np.abs(dR.grad.cpu().numpy() - gr).max() < 6e-5 * sc, tag
        assert np.abs(dmu.grad.cpu().numpy() - gd).max() < 6e-5 * sc, tag


@pytest.mark.parametrize("B,C,H,W,D,p", [(1, 72, 2, 300, 270, 0.5), (1, 216, 2, 280, 270, 0.3), (1, 100, 1, 150, 400, 1.0)])
def test_shapes_beyond_any_lds_tile(dev, B, C, H, W, D, p):
    """C x max_disp too large for the band and the row-tile kernels (outside the shipped configurations,
    inside what the reference's global-memory kernels accept): the global-memory fallbacks, forward and
    backward, SpaMat and SpaVar."""
    import decnet_amd
    L, R, rm, tm = make_case(21, B, C, H, W, p, p, relu=False, scale=0.25)
    g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(6))
    o, s_, m = oracle.spamat_forward(L, R, rm, tm, D)
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s_, m, g, D)
    dL, dR = L.to(dev).requires_grad_(), R.to(dev).requires_grad_()
    out = decnet_amd.SpaMatFunction.apply(dL, dR, rm.to(dev), tm.to(dev), D)
    np.testing.assert_allclose(out.detach().cpu().numpy(), o, rtol=1e-5, atol=3e-4)
    out.backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 3e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 3e-5 * sc
    mu = torch.from_numpy(o) + 0.25
    v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
    gl, gr, gd = oracle.spavar_backward(L, R, rm, tm, mu, v, s2, m2, g, D)
    dL, dR, dmu = L.to(dev).requires_grad_(), R.to(dev).requires_grad_(), mu.to(dev).requires_grad_()
    var = decnet_amd.SpaVarFunction.apply(dL, dR, rm.to(dev), tm.to(dev), dmu, D)
    np.testing.assert_allclose(var.detach().cpu().numpy(), v, rtol=2e-4, atol=2e-2)
    var.backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()), float(np.abs(gd).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 6e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 6e-5 * sc
    assert np.abs(dmu.grad.cpu().numpy() - gd).max() < 6e-5 * sc


@pytest.mark.parametrize("C,W,D,scale,relu", [(8, 500, 216, 12.0, True), (8, 400, 216, 6.0, False), (24, 324, 72, 8.0, True),
                                              (8, 972, 216, 1e-3, True)])
def test_large_and_small_magnitudes(dev, C, W, D, scale, relu):
    """Costs of ~1e3 (features x 12: the exponent arguments reach -1e3 and most candidates underflow to 0, as in
    the reference) and of ~1e-5 (every candidate at the 1e-6 floor of max_cost, SM_kernel.cu:45): no overflow, no
    NaN, same soft-argmax.  Dense rows (bf16x3 cost tiles at C = 8) and sparse rows."""
    import decnet_amd
    for dens in (1.0, 0.15):
        L, R, rm, tm = make_case(77, 1, C, 3, W, dens, dens, relu=relu, scale=scale)
        o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
        fo, fv, fs, fm = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
        for t in (fo, fv, fs, fm):
            assert bool(torch.isfinite(t).all())
        np.testing.assert_allclose(fm.cpu().numpy(), m, rtol=2e-5, atol=1e-7)
        # a sharp softmax amplifies cost rounding by the cost scale: disparities agree where the maximum is
        # separated, and on average
        err = np.abs(fo.cpu().numpy() - o)
        assert np.median(err) < 1e-4 and err.mean() < (2e-3 if scale > 1 else 1e-4), (dens, err.mean(), err.max())
        np.testing.assert_allclose(fs.cpu().numpy(), s, rtol=2e-3 if scale > 1 else 2e-5, atol=1e-9)


def pack_mask_bits(mask):
    """float 0/1 [B,H,W] -> int64 [B,H,ceil(W/64)], bit i of word w = pixel 64 w + i (decnet_detail_mask's layout)."""
    B, H, W = mask.shape
    wpr = (W + 63) // 64
    m = np.zeros((B, H, wpr * 64), dtype=np.uint64)
    m[:, :, :W] = (mask.cpu().numpy() != 0)
    words = (m.reshape(B, H, wpr, 64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
    return torch.from_numpy(words.view(np.int64))


@pytest.mark.parametrize("C,H,W,D", [(8, 12, 972, 216), (8, 12, 1242, 216), (24, 12, 324, 72), (72, 5, 108, 24),
                                     (8, 3, 61, 40), (12, 4, 130, 100), (8, 12, 1512, 270)])
def test_fused_forward_with_bit_packed_masks_equals_float_masks(dev, C, H, W, D):
    """decnet_spamatvar_forward_bits (masks as 64 pixels per word, what decnet_detail_mask writes) against the
    float-mask call on rows of every kind (sparse-row kernel, hand-over, dense, compact, empty): bit for bit."""
    import decnet_amd
    g = torch.Generator().manual_seed(C * 1000 + W)
    L = torch.relu(torch.randn(1, C, H, W, generator=g))
    R = torch.relu(torch.randn(1, C, H, W, generator=g))
    dens = [(0.02, 0.02), (1.0, 1.0), (0.1, 0.1), (0.5, 0.5), (0.05, 0.9), (0.9, 0.05), (0.2, 0.2),
            (0.0, 0.3), (0.3, 0.0), (0.26, 0.26), (0.03, 0.03), (0.4, 0.6)][:H]
    rm = torch.stack([(torch.rand(W, generator=g) < p).float() for p, _ in dens]).view(1, H, W)
    tm = torch.stack([(torch.rand(W, generator=g) < p).float() for _, p in dens]).view(1, H, W)
    dL, dR, drm, dtm = (t.to(dev) for t in (L, R, rm, tm))
    want = decnet_amd.spamatvar_forward(dL, dR, drm, dtm, D)
    if os.environ.get("DECNET_SPAMAT_KERNEL", "") == "rowtile":
        # the row-tile kernels have no bit-mask variant: the entry honours the pin by reporting UNSUPPORTED (the graph
        # then falls back to the float-mask entry, decnet_amd/model.py)
        from decnet_amd._lib import DecnetHipError, UNSUPPORTED
        with pytest.raises(DecnetHipError) as ei:
            decnet_amd.spamatvar_forward_bits(dL, dR, pack_mask_bits(rm).to(dev), pack_mask_bits(tm).to(dev), D)
        assert ei.value.code == UNSUPPORTED
        return
    got = decnet_amd.spamatvar_forward_bits(dL, dR, pack_mask_bits(rm).to(dev), pack_mask_bits(tm).to(dev), D)
    for a, b_, name in zip(got, want, ("output", "variance", "sum_similarities", "max_cost")):
        assert torch.equal(a, b_), name
    with pytest.raises(ValueError):
        decnet_amd.spamatvar_forward_bits(dL, dR, drm, dtm, D)            # float planes are not bit words


def test_mask_kernel_bits_feed_the_bit_mask_call(dev):
    """GenerateSparseMask.mask(want_bits=True): the bit-packed copy is the float plane, and the model's forward is the
    same with DECNET_SPAMAT_BITS=0 (float planes into SpaMat) and 1 (bits)."""
    from decnet_amd.model import GenerateSparseMask
    torch.manual_seed(3)
    gen = GenerateSparseMask(8, 3).to(dev).eval()
    cur, pre = torch.randn(2, 8, 66, 135, device=dev), torch.randn(2, 24, 22, 45, device=dev)
    with torch.no_grad():
        z = gen(cur, pre)
        thold = float(torch.sigmoid(z).median())
        m, bits = gen.mask(cur, pre, thold, want_bits=True)
    assert bits is not None and bits.dtype == torch.int64 and tuple(bits.shape) == (2, 66, 3)
    assert torch.equal(pack_mask_bits(m), bits.cpu())
    assert 0.2 < float(m.mean()) < 0.8


@pytest.mark.parametrize("env", [{"DECNET_SPAMAT_MID": "0"}])
def test_mid_density_rows_on_the_band_kernels_own_paths(env):
    """Rows of 257 - 640 active pixels per side on the band kernel's compact / dense paths -- what they take wherever the
    640-slot body does not fit LDS (wide rows, several segments) -- forced at the shipped shapes: the mixed-density and
    randomized cases in a child process (the switch is read once per process)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "mixed_row_densities or randomized_shapes or bit_packed"], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_randomized_densities_around_the_path_boundaries(dev):
    """130 random (C, H, W, D, left / right density) cases with the densities that decide a row's path (sparse-row kernel
    <= 256 active pixels, the 640-slot body, compact, dense): fused forward against the oracle."""
    import decnet_amd
    for seed in range(200, 330):
        rng = np.random.default_rng(seed)
        C = int(rng.choice([8, 8, 8, 24, 72]))
        W = int(rng.integers(40, 1000)) if C == 8 else int(rng.integers(40, 400))
        D = int(rng.integers(3, min(W, 230)))
        H = int(rng.integers(1, 4))
        pr, pt = [float(rng.choice([0.05, 0.2, 0.3, 0.45, 0.6, 0.7, 1.0])) for _ in range(2)]
        g = torch.Generator().manual_seed(seed)
        L = torch.relu(torch.randn(1, C, H, W, generator=g))
        R = torch.relu(torch.randn(1, C, H, W, generator=g))
        rm = (torch.rand(1, H, W, generator=g) < pr).float()
        tm = (torch.rand(1, H, W, generator=g) < pt).float()
        o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
        v, _, _ = oracle.spavar_forward(L, R, rm, tm, o, D)
        fo, fv, fs, fm = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
        tag = str(dict(seed=seed, C=C, H=H, W=W, D=D, pr=pr, pt=pt))
        np.testing.assert_allclose(fo.cpu().numpy(), o, rtol=1e-5, atol=3e-4, err_msg=tag)
        np.testing.assert_allclose(fs.cpu().numpy(), s, rtol=2e-5, atol=1e-9, err_msg=tag)
        np.testing.assert_allclose(fv.cpu().numpy(), v, rtol=2e-4, atol=2e-2, err_msg=tag)


def test_graphed_step_replays_forward_and_backward(dev):
    """decnet_amd.graphs.GraphedStep: SpaMatFunction forward + backward of two stages captured once into a HIP graph;
    a replay after changing the inputs IN PLACE gives the gradients of an eager run on the new values."""
    import decnet_amd
    from decnet_amd.graphs import GraphedStep
    mod = decnet_amd.SpaMat()
    items = []
    for seed, (B, C, H, W, D, p) in enumerate([(2, 8, 6, 300, 216, 0.6), (2, 24, 5, 81, 72, 1.0)]):
        L, R, rm, tm = make_case(40 + seed, B, C, H, W, p, p, relu=False, scale=0.5)
        g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(seed))
        items.append([L.to(dev).requires_grad_(), R.to(dev).requires_grad_(), rm.to(dev), tm.to(dev), D, g.to(dev)])

    def whole():
        return [mod(a, b, c, d, D).backward(g) for (a, b, c, d, D, g) in items]
    step = GraphedStep(whole, grads_of=[t for it in items for t in it[:2]])
    with torch.no_grad():                                 # new values, same storage
        for it in items:
            it[0].mul_(0.75)
            it[1].add_(0.1)
    step()
    torch.cuda.synchronize()
    got = [(it[0].grad.clone(), it[1].grad.clone()) for it in items]
    for it, (gl, gr) in zip(items, got):
        a, b = it[0].detach().clone().requires_grad_(), it[1].detach().clone().requires_grad_()
        mod(a, b, it[2], it[3], it[4]).backward(it[5])
        assert torch.equal(a.grad, gl) and torch.equal(b.grad, gr)


@pytest.mark.gpu
@pytest.mark.parametrize("C,W,D,p", [(7, 972, 216, 1.0), (5, 640, 216, 1.0), (6, 326, 216, 0.9), (8, 970, 216, 1.0),
                                     (8, 972, 216, 1.0), (8, 648, 100, 1.0)])
def test_dense_rows_batched_staging_loads(dev, C, W, D, p):
    """Round 5: a dense row's staging item is ONE batch of raw 16-byte loads behind a workgroup-uniform flag (rows on
    16-byte boundaries, W % 4 == 0) and four guarded loads otherwise.  Both forms, partial channel groups (C = 5 .. 7:
    the missing channels must read as zeros), rows whose left edge lies inside the first halo, against the oracle; H = 3
    rows so that W % 4 != 0 gives rows of every alignment."""
    import decnet_amd
    B, H = 1, 3
    L, R, rm, tm = make_case(31 + C, B, C, H, W, p, p)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    outs = decnet_amd.spamatvar_forward(L.to(dev), R.to(dev), rm.to(dev), tm.to(dev), D)
    check_fwd(o, s, m, outs[0], outs[2], outs[3], D)
    v, _, _ = oracle.spavar_forward(L, R, rm, tm, o, D)
    np.testing.assert_allclose(outs[1].cpu().numpy(), v, rtol=2e-4, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("C,W,D", [(24, 324, 72), (24, 322, 72), (72, 108, 24), (72, 106, 24), (20, 200, 72)])
def test_backward_band_kernel_batched_loads(dev, C, W, D):
    """The backward band kernel's staging (features, mask, per-pixel planes) as batches of raw loads on aligned rows and
    as guarded loads on ragged ones (W % 4 != 0: rows of every alignment), C below the staged channel count (20 of 24)."""
    import decnet_amd
    B, H = 1, 3
    L, R, rm, tm = make_case(41, B, C, H, W, 0.9, 0.9, relu=False, scale=0.5)
    g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(6))
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
    dL, dR = L.to(dev).requires_grad_(), R.to(dev).requires_grad_()
    out = decnet_amd.SpaMatFunction.apply(dL, dR, rm.to(dev), tm.to(dev), D)
    out.backward(g.to(dev))
    sc = max(1.0, float(np.abs(gl).max()), float(np.abs(gr).max()))
    assert np.abs(dL.grad.cpu().numpy() - gl).max() < 2e-5 * sc
    assert np.abs(dR.grad.cpu().numpy() - gr).max() < 2e-5 * sc
