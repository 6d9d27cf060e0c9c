"""Parity of the fused conv-chain kernel (tools/experiments/chain2d.hip, decnet_chain2d_forward; EXPERIMENT, run by hand after tools/experiments/build.sh) with the torch CPU ops the
reference calls layer by layer (modules/submodule.py:15-45: conv2d -> batch_norm(eval) -> relu), in float64.  -m gpu.
Tolerance: 1e-5 * max|y| per chain (bf16x3 products on the matrix cores = every product above 2^-24, fp32 sums)."""
import os
import sys

import pytest
import torch

_H = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(_H, "..", ".."), _H]

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import decnet_amd  # noqa: F401
    return torch.device("cuda:0")


def _unit(cin, cout, k=3, dil=1, relu=True, bn=True, seed=0):
    from decnet_amd.model import Unit
    torch.manual_seed(seed)
    u = Unit(cin, cout, k, pad=dil * (k // 2), dil=dil, relu=relu, bn=bn)
    u.conv.weight.data.normal_(0, (2.0 / (k * k * cin)) ** 0.5)
    if bn:
        u.bn.weight.data.uniform_(0.5, 1.5)
        u.bn.bias.data.normal_(0, 0.2)
        u.bn.running_mean.data.normal_(0, 0.2)
        u.bn.running_var.data.uniform_(0.5, 1.5)
    else:
        u.conv.bias.data.normal_(0, 0.2)
    return u.eval()


def _ref(units, x):
    y = x.double()
    with torch.no_grad():
        for u in units:
            y = u.double()(y)
            u.float()
    return y


CHAINS = [
    # (layer specs (cin, cout, k, dil, relu, bn), input channel split, shape (B, H, W))
    ([(8, 8, 3, 1, True, True)], (8,), (2, 37, 100)),
    ([(8, 8, 3, 1, True, True), (8, 8, 3, 1, True, True)], (8,), (2, 41, 131)),
    ([(3, 8, 3, 1, True, True), (8, 8, 3, 1, True, True)], (3,), (1, 54, 243)),
    ([(12, 8, 3, 1, True, True), (8, 8, 3, 1, True, True), (8, 1, 3, 1, False, True)], (8, 1, 1, 1, 1), (2, 60, 108)),
    ([(17, 8, 3, 3, True, True), (8, 8, 3, 1, True, True)], (8, 8, 1), (1, 70, 150)),
    ([(8, 8, 3, 6, True, True), (8, 4, 3, 1, True, True)], (8,), (1, 64, 97)),
    ([(8, 8, 3, 1, True, False), (8, 3, 3, 1, False, True)], (8,), (3, 27, 54)),
    ([(16, 8, 3, 1, True, True), (8, 8, 3, 1, True, True)], (8, 8), (1, 45, 300)),
    ([(24, 8, 3, 1, True, False), (8, 3, 3, 1, False, True)], (24,), (1, 60, 108)),
    ([(28, 8, 3, 1, True, True), (8, 8, 3, 1, True, True), (8, 1, 3, 1, False, True)], (24, 1, 1, 1, 1), (1, 36, 70)),
    ([(76, 8, 3, 1, True, True), (8, 8, 3, 1, True, True), (8, 1, 3, 1, False, True)], (72, 1, 1, 1, 1), (1, 20, 36)),
    ([(8, 8, 1, 1, True, True), (8, 8, 3, 2, True, True)], (8,), (1, 33, 80)),
    ([(4, 4, 3, 9, True, True), (4, 4, 3, 1, True, True), (4, 1, 3, 1, False, False)], (4,), (1, 50, 90)),
]


@pytest.mark.parametrize("idx", range(len(CHAINS)))
@pytest.mark.parametrize("force", [(0, 0), (16, 5), (48, 11)])
def test_chain_vs_torch_cpu(dev, idx, force):
    import chain
    specs, split, (B, H, W) = CHAINS[idx]
    units = [_unit(*s, seed=100 * idx + i) for i, s in enumerate(specs)]
    g = torch.Generator().manual_seed(idx)
    xs = [torch.randn(B, c, H, W, generator=g) for c in split]
    ref = _ref(units, torch.cat(xs, 1))
    for u in units:
        u.to(dev)
    assert chain.units_ok(units)
    from decnet_amd._lib import DecnetHipError, UNSUPPORTED
    with torch.no_grad():
        try:
            got = chain.conv_chain(units, [x.to(dev) for x in xs], force_tw=force[0], force_rows=force[1]).cpu()
        except DecnetHipError as e:
            # a pinned strip width may not fit LDS (76 source channels x 48 + halo columns); the planned one must
            assert force[0] and e.code == UNSUPPORTED
            pytest.skip("pinned strip width does not fit LDS for this chain")
    assert got.shape == ref.shape
    err = float((got.double() - ref).abs().max())
    print("chain %d force %s: max err %.2e, max|y| %.2f" % (idx, force, err, float(ref.abs().max())))
    assert err < 1e-5 * max(1.0, float(ref.abs().max()))
