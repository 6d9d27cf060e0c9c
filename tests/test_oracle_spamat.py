"""oracle/spamat_oracle.c against an independent vectorised restatement, autograd, and
the known-answer quirks of the reference kernels.  CPU only."""
import numpy as np
import pytest
import torch

import _vectorised as V
import oracle


def _case(seed, B, C, H, W, p_ref=0.6, p_tar=0.6, relu=True):
    g = torch.Generator().manual_seed(seed)
    L = torch.randn(B, C, H, W, generator=g)
    R = torch.randn(B, C, H, W, generator=g)
    if relu:
        L, R = torch.relu(L), torch.relu(R)
    rm = (torch.rand(B, H, W, generator=g) < p_ref).float()
    tm = (torch.rand(B, H, W, generator=g) < p_tar).float()
    return L, R, rm, tm


@pytest.mark.parametrize("shape,D", [((2, 8, 5, 40), 24), ((1, 24, 3, 17), 30), ((1, 3, 2, 9), 9),
                                     ((1, 72, 2, 12), 5)])
@pytest.mark.parametrize("fma", [True, False])
def test_forward_vs_vectorised_fp64(shape, D, fma):
    L, R, rm, tm = _case(1, *shape)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D, fma=fma)
    vo, vs, vm = V.spamat(L.double(), R.double(), rm.double(), tm.double(), D)
    np.testing.assert_allclose(m, vm.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(s, vs.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o, vo.numpy(), rtol=1e-5, atol=2e-5)
    mu = torch.from_numpy(o)
    v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D, fma=fma)
    vv, _, _ = V.spavar(L.double(), R.double(), rm.double(), tm.double(), mu.double(), D)
    np.testing.assert_allclose(v, vv.numpy(), rtol=2e-5, atol=1e-4)
    np.testing.assert_array_equal(s2, s)
    np.testing.assert_array_equal(m2, m)


def test_fma_and_nofma_agree():
    """Whichever contraction nvcc chose, outputs agree far inside the 1e-3 px budget."""
    L, R, rm, tm = _case(2, 2, 8, 6, 60)
    a = oracle.spamat_forward(L, R, rm, tm, 48, fma=True)
    b = oracle.spamat_forward(L, R, rm, tm, 48, fma=False)
    assert np.abs(a[0] - b[0]).max() < 1e-4


@pytest.mark.parametrize("shape,D", [((2, 8, 4, 30), 16), ((1, 24, 2, 20), 25)])
def test_backward_equals_autograd_with_max_cost_constant(shape, D):
    """S7: the hand-written backward == autograd of the forward, max_cost detached."""
    L, R, rm, tm = _case(3, *shape, relu=False)
    L, R = L * 0.5, R * 0.5
    g = torch.randn(shape[0], shape[2], shape[3], generator=torch.Generator().manual_seed(9))
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    gl, gr = oracle.spamat_backward(L, R, rm, tm, o, s, m, g, D)
    Ld, Rd = L.double().requires_grad_(), R.double().requires_grad_()
    vo, _, _ = V.spamat(Ld, Rd, rm.double(), tm.double(), D)
    vo.backward(g.double())
    scale = max(1.0, float(Ld.grad.abs().max()))
    assert np.abs(gl - Ld.grad.numpy()).max() < 2e-5 * scale
    assert np.abs(gr - Rd.grad.numpy()).max() < 2e-5 * scale
    # SpaVar: three gradients
    mu = torch.from_numpy(o) + 0.3
    v, s2, m2 = oracle.spavar_forward(L, R, rm, tm, mu, D)
    gl, gr, gd = oracle.spavar_backward(L, R, rm, tm, mu, v, s2, m2, g, D)
    Ld, Rd = L.double().requires_grad_(), R.double().requires_grad_()
    mud = mu.double().requires_grad_()
    vv, _, _ = V.spavar(Ld, Rd, rm.double(), tm.double(), mud, D)
    vv.backward(g.double())
    scale = max(1.0, float(Ld.grad.abs().max()), float(Rd.grad.abs().max()))
    assert np.abs(gl - Ld.grad.numpy()).max() < 2e-5 * scale
    assert np.abs(gr - Rd.grad.numpy()).max() < 2e-5 * scale
    assert np.abs(gd - mud.grad.numpy()).max() < 2e-5 * scale
    # masked-off entries keep the caller's zero fill
    assert (gl[:, :, :, :][np.broadcast_to(rm.numpy()[:, None] == 0, gl.shape)] == 0).all()
    assert (gr[np.broadcast_to(tm.numpy()[:, None] == 0, gr.shape)] == 0).all()


def test_quirks_known_answers():
    """S6: floor 1e-6; sums start at 1e-6; ref-off -> 0; no valid candidate -> exactly 1.0."""
    B, C, H, W, D = 1, 4, 1, 6, 3
    L = torch.ones(B, C, H, W)
    R = torch.ones(B, C, H, W)
    rm = torch.tensor([[[1., 0., 1., 1., 1., 1.]]])
    tm = torch.zeros(B, H, W)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, D)
    assert o[0, 0, 1] == 0 and s[0, 0, 1] == 0 and m[0, 0, 1] == 0            # ref-off
    on = rm.numpy()[0, 0] != 0
    assert (o[0, 0][on] == 1.0).all()                                         # 1e-6/1e-6
    assert (s[0, 0][on] == np.float32(1e-6)).all() and (m[0, 0][on] == np.float32(1e-6)).all()
    # all candidate costs negative -> max_cost stays at the 1e-6 floor, not the true max
    tm = torch.ones(B, H, W)
    o, s, m = oracle.spamat_forward(L, -R, torch.ones(B, H, W), tm, D)
    assert (m == np.float32(1e-6)).all()
    e = np.exp(np.float32(-4.0) - np.float32(1e-6))
    # x=0 has one candidate (d=0); x>=2 has three (d=0,1,2)
    np.testing.assert_allclose(s[0, 0, 0], 1e-6 + e, rtol=1e-6)
    np.testing.assert_allclose(o[0, 0, 5], (1e-6 + e * 3) / (1e-6 + 3 * e), rtol=1e-6)
    # edge rule cur_max_disp = min(max_disp, x+1): pixel x=1 sees d in {0,1}
    np.testing.assert_allclose(o[0, 0, 1], (1e-6 + e * 1) / (1e-6 + 2 * e), rtol=1e-6)


def test_max_disp_larger_than_width():
    L, R, rm, tm = _case(5, 1, 4, 2, 7, 1.0, 1.0)
    o, s, m = oracle.spamat_forward(L, R, rm, tm, 20)
    vo, vs, vm = V.spamat(L.double(), R.double(), rm.double(), tm.double(), 20)
    np.testing.assert_allclose(o, vo.numpy(), rtol=1e-5, atol=1e-5)
