"""Parity of the fused small-channel Conv2dUnit / Deconv2dUnit kernels (csrc/conv2d_small.hip) with
torch CPU conv2d -> batch_norm(eval) -> relu (the third-party arithmetic the reference calls,
modules/submodule.py:15-87).  -m gpu.  Tolerance: 2e-5 * max|y| (fp32, different summation order)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import decnet_amd  # noqa: F401
    return torch.device("cuda:0")


def _unit(cin, cout, k, dil=1, relu=True, bn=True, transposed=False, seed=0):
    from decnet_amd.model import Unit
    torch.manual_seed(seed)
    u = Unit(cin, cout, k, stride=3 if transposed else 1, pad=0 if transposed else dil * (k // 2), dil=dil, relu=relu,
             bn=bn, transposed=transposed)
    if bn:
        u.bn.weight.data.uniform_(0.5, 1.5)
        u.bn.bias.data.normal_(0, 0.2)
        u.bn.running_mean.data.normal_(0, 0.2)
        u.bn.running_var.data.uniform_(0.5, 1.5)
    return u.eval()


@pytest.mark.parametrize("cin,cout,k,dil,relu,bn", [
    (8, 8, 3, 1, True, True), (17, 8, 3, 3, True, True), (8, 8, 3, 6, True, True), (4, 4, 3, 9, True, True),
    (3, 8, 3, 1, True, True), (8, 3, 3, 1, False, True), (8, 1, 3, 1, False, False), (4, 1, 3, 1, False, False),
    (8, 8, 1, 1, True, True), (16, 8, 3, 1, True, True), (3, 1, 1, 1, False, True),
    (49, 24, 3, 2, True, True), (24, 24, 3, 4, True, True), (12, 12, 3, 6, True, True), (8, 24, 3, 1, True, True)])
def test_conv_unit_vs_torch_cpu(dev, cin, cout, k, dil, relu, bn):
    u = _unit(cin, cout, k, dil, relu, bn, seed=cin * 31 + cout)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, cin, 131, 517, generator=g)          # ragged width, > 64 k pixels
    with torch.no_grad():
        ref = u(x)                                          # CPU: torch ops
        ud = u.to(dev)
        assert ud._hip_kind(x.to(dev)) == ("mfma" if (cin, cout) in ((49, 24), (24, 24)) else "conv")
        got = ud(x.to(dev)).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("epi", [1, 2])
def test_single_output_layer_with_fused_tail(dev, epi):
    """decnet_conv2d_cat_epilogue: epilogue 1 (SoftAttention's sigmoid + the stage loop's dense / sparse fusion,
    SparseDenseNetRefinementMask.py:195-202) and 2 (Refinement's residual, submodule.py:716) against the torch ops;
    concatenated input, negated last channel folded into the weights."""
    u = _unit(12, 1, 3, 1, False, True, seed=77)
    g = torch.Generator().manual_seed(6)
    a, b4 = torch.randn(2, 8, 90, 333, generator=g), torch.randn(2, 4, 90, 333, generator=g)
    ea, eb = torch.randn(2, 90, 333, generator=g) * 30, torch.randn(2, 90, 333, generator=g) * 30
    with torch.no_grad():
        xin = torch.cat((a, b4[:, :3], -b4[:, 3:]), 1)
        v = u(xin).squeeze(1)
        ref = ea * (1 - torch.sigmoid(v)) + torch.sigmoid(v) * eb if epi == 1 else ea + v
        ud = u.to(dev)
        parts = (a.to(dev), b4.to(dev))
        assert ud._hip_kind(parts) == "conv"
        got = ud._forward_hip(parts, "conv", epi=epi, ea=ea.to(dev), eb=eb.to(dev) if epi == 1 else None,
                              neg_last=True).squeeze(1).cpu()
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


# the many-channel layers on the bf16 matrix cores (csrc/conv2d_mfma.hip): every (tile height, channel tile count)
# variant, concatenated inputs, ragged sizes, dilation, 1x1; reference = torch CPU in float64
@pytest.mark.parametrize("cins,cout,k,dil,relu,bn,shape,tm", [
    ((81,), 81, 3, 1, True, True, (2, 61, 107), 6), ((81,), 81, 3, 1, True, True, (1, 60, 108), 5),
    ((73,), 81, 3, 1, True, True, (1, 45, 100), 4), ((24, 24), 24, 3, 1, True, True, (2, 70, 131), 8),
    ((24,), 24, 3, 1, True, True, (1, 33, 64), 2), ((72, 72, 1), 72, 3, 1, True, True, (2, 60, 108), 6),
    ((72,), 36, 3, 1, True, True, (1, 64, 64), 8), ((36,), 36, 3, 1, False, True, (1, 40, 103), 5),
    ((217,), 81, 3, 1, True, True, (1, 60, 108), 2), ((216,), 216, 3, 1, True, True, (1, 20, 36), 4),
    ((24, 24, 1), 24, 3, 2, True, True, (1, 90, 162), 4), ((24,), 40, 3, 4, True, True, (1, 64, 70), 2),
    ((24,), 24, 1, 1, True, True, (2, 64, 100), 8), ((432,), 216, 1, 1, True, True, (1, 20, 36), 5),
    ((48,), 64, 3, 1, True, False, (1, 31, 47), 0), ((16,), 24, 3, 1, False, False, (1, 64, 64), 0)])
def test_mfma_conv_unit_vs_torch_cpu(dev, monkeypatch, cins, cout, k, dil, relu, bn, shape, tm):
    cin = sum(cins)
    u = _unit(cin, cout, k, dil, relu, bn, seed=cin * 7 + cout)
    g = torch.Generator().manual_seed(11)
    B, H, W = shape
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    with torch.no_grad():
        ref = u.double()(torch.cat(xs, 1).double())
        ud = u.float().to(dev)
        if tm:
            monkeypatch.setenv("DECNET_CONV2D_MFMA_TM", str(tm))
        xd = [t.to(dev) for t in xs]
        # >= 49 output channels: both the 4-wave and the 8-wave producer / consumer kernel
        for pc in (("0", "1") if cout > 48 else ("",)):
            if pc:
                monkeypatch.setenv("DECNET_CONV2D_MFMA_PC", pc)
            got = ud._forward_mfma(xd if len(xd) > 1 else xd[0]).cpu()
            assert got.shape == ref.shape
            assert float((got.double() - ref).abs().max()) < 4e-6 * max(1.0, float(ref.abs().max())), pc


@pytest.mark.parametrize("seed", range(10))
def test_mfma_conv_random_shapes(dev, seed):
    """Seeded random layer shapes (channel counts that are not multiples of anything, images smaller than a tile, one
    to four concatenated inputs, every dilation the kernel takes) through the kernel's own dispatch."""
    import random
    rnd = random.Random(1234 + seed)
    nseg = rnd.choice((1, 1, 2, 3, 4))
    cins = tuple(rnd.choice((1, 3, 7, 8, 16, 17, 24, 31, 40, 65)) for _ in range(nseg))
    cout = rnd.choice((1, 5, 16, 17, 24, 33, 49, 64, 81, 97, 130))
    k = rnd.choice((1, 3, 3, 3))
    dil = rnd.choice((1, 1, 2, 3, 4)) if k == 3 else 1
    B, H, W = rnd.choice((1, 2, 3)), rnd.choice((1, 5, 16, 23, 40, 67)), rnd.choice((3, 15, 16, 17, 50, 129))
    cin = sum(cins)
    u = _unit(cin, cout, k, dil, rnd.random() < 0.7, rnd.random() < 0.7, seed=seed)
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
    with torch.no_grad():
        ref = u.double()(torch.cat(xs, 1).double())
        ud = u.float().to(dev)
        xd = [t.to(dev) for t in xs]
        got = ud._forward_mfma(xd if len(xd) > 1 else xd[0]).cpu()
    assert got.shape == ref.shape
    assert float((got.double() - ref).abs().max()) < 4e-6 * max(1.0, float(ref.abs().max())), (cins, cout, k, dil, B, H, W)


def test_mfma_conv_is_what_the_many_channel_units_run(dev):
    u = _unit(72, 72, 3).to(dev)
    x = torch.randn(2, 72, 60, 108, device=dev)
    with torch.no_grad():
        assert u._hip_kind(x) == "mfma"
        assert _unit(144, 72, 3).to(dev)._hip_kind((x, x)) == "mfma"
        assert u._hip_kind(x[:, :, :20, :36]) is None                             # small images stay on the library


@pytest.mark.parametrize("cin,cout,shape", [(24, 72, (2, 180, 324)), (72, 216, (2, 60, 108)), (8, 40, (1, 100, 203))])
def test_stride3_conv_on_the_matrix_cores_vs_torch_cpu(dev, cin, cout, shape):
    """Conv2d k 3, stride 3, padding 1 with more than 24 outputs (FeatExtNetChannelPlus conv2[0], conv3_1,
    submodule.py:270-300): decnet_s2d3_pad1 + the matrix-core kernel as a 1 x 1 convolution over 9 Cin channels."""
    from decnet_amd.model import Unit
    torch.manual_seed(cin + cout)
    u = Unit(cin, cout, 3, stride=3, pad=1).eval()
    u.bn.weight.data.uniform_(0.5, 1.5); u.bn.bias.data.normal_(0, 0.2)
    u.bn.running_mean.data.normal_(0, 0.2); u.bn.running_var.data.uniform_(0.5, 1.5)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape[0], cin, shape[1], shape[2], generator=g)
    with torch.no_grad():
        ref = u.double()(x.double()).float()
        ud = u.float().to(dev)
        assert ud._hip_kind(x.to(dev)) == "mfma_s3"
        got = ud(x.to(dev)).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_few_outputs_from_many_inputs_run_on_the_matrix_cores(dev):
    """Round 3: <= 8 outputs from >= 48 inputs (GenerateSparseMask / SoftAttention at 1/9 resolution) and 9..23 outputs
    from >= 16 inputs (Refinement 24 -> 12 at 1/3 resolution) go to csrc/conv2d_mfma.hip; vs torch CPU float64."""
    g = torch.Generator().manual_seed(4)
    for cins, cout, shape in (((72,), 8, (2, 60, 108)), ((72, 1, 1, 1, 1), 8, (2, 60, 108)), ((24,), 12, (1, 180, 324))):
        u = _unit(sum(cins), cout, 3, seed=sum(cins) + cout)
        xs = [torch.randn(shape[0], c, shape[1], shape[2], generator=g) for c in cins]
        with torch.no_grad():
            ref = u.double()(torch.cat(xs, 1).double()).float()
            ud = u.float().to(dev)
            xd = tuple(t.to(dev) for t in xs) if len(xs) > 1 else xs[0].to(dev)
            assert ud._hip_kind(xd) == "mfma"
            got = ud(xd).cpu()
        assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,relu,bn", [(24, 8, True, True), (9, 8, True, False), (24, 3, False, True)])
def test_deconv_unit_vs_torch_cpu(dev, cin, cout, relu, bn):
    u = _unit(cin, cout, 3, relu=relu, bn=bn, transposed=True, seed=cin + cout)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, cin, 45, 173, generator=g)           # output 135 x 519
    with torch.no_grad():
        ref = u(x)
        ud = u.to(dev)
        xd = x.to(dev)
        kind = ud._hip_kind(xd)
        got = ud(xd).cpu() if kind else None
    if kind is None:                                        # input below the size threshold: call directly
        with torch.no_grad():
            got = ud._forward_hip(xd, "deconv").cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,relu,bn,shape", [(72, 24, True, True, (2, 60, 108)), (216, 72, True, True, (2, 20, 36)),
                                                     (16, 9, False, False, (1, 23, 31)), (40, 30, True, True, (1, 33, 50))])
def test_mfma_deconv_unit_vs_torch_cpu(dev, cin, cout, relu, bn, shape):
    """Deconv2dUnit with > 8 output channels: conv2d_mfma as a 1 x 1 convolution to 9 Cout channels + pixel shuffle."""
    u = _unit(cin, cout, 3, relu=relu, bn=bn, transposed=True, seed=cin + cout)
    B, H, W = shape
    x = torch.randn(B, cin, H, W, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        ref = u.double()(x.double())
        ud = u.float().to(dev)
        xd = x.to(dev)
        assert ud._hip_kind(xd) == "mfma_deconv"
        got = ud(xd).cpu()
    assert got.shape == ref.shape == (B, cout, 3 * H, 3 * W)
    assert float((got.double() - ref).abs().max()) < 4e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout", [(8, 24), (3, 8), (24, 20)])
def test_stride3_conv_unit_vs_torch_cpu(dev, cin, cout):
    from decnet_amd.model import Unit
    torch.manual_seed(cin + cout)
    u = Unit(cin, cout, 3, stride=3, pad=1).eval()
    u.bn.weight.data.uniform_(0.5, 1.5); u.bn.bias.data.normal_(0, 0.2)
    u.bn.running_mean.data.normal_(0, 0.2); u.bn.running_var.data.uniform_(0.5, 1.5)
    x = torch.randn(2, cin, 271, 500, generator=torch.Generator().manual_seed(8))    # sizes not multiples of 3
    with torch.no_grad():
        ref = u(x)
        ud = u.to(dev)
        assert ud._hip_kind(x.to(dev)) == "conv_s3"
        got = ud(x.to(dev)).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_concatenated_input_and_warp(dev):
    """Unit on a tuple == Unit on torch.cat; warp_by_disparity kernel == the torch grid_sample path."""
    from decnet_amd import model as M
    u = _unit(17, 8, 3, dil=3, seed=3)
    g = torch.Generator().manual_seed(12)
    a, b, c = (torch.randn(2, n, 140, 500, generator=g) for n in (8, 8, 1))
    with torch.no_grad():
        ref = u(torch.cat((a, b, c), 1))
        ud = u.to(dev)
        parts = (a.to(dev), b.to(dev), c.to(dev))
        assert ud._hip_kind(parts) == "conv"
        got = ud(parts).cpu()
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    right = torch.randn(2, 8, 140, 500, generator=g)
    disp = torch.rand(2, 140, 500, generator=g) * 60 - 5          # some samples fall outside the image
    with torch.no_grad():
        ref = M.warp_by_disparity(right, disp)                     # CPU: meshgrid + grid_sample
        got = M.warp_by_disparity(right.to(dev), disp.to(dev)).cpu()
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_dynamic_upsampling_tail_kernel(dev):
    """DynamicUpsampling on the GPU (fused tail kernel) == the torch ops on CPU."""
    from decnet_amd.model import DynamicUpsampling
    torch.manual_seed(4)
    m = DynamicUpsampling(8, 3).eval()
    g = torch.Generator().manual_seed(14)
    disp, fea = torch.rand(2, 37, 45, generator=g) * 20, torch.randn(2, 8, 111, 135, generator=g)
    with torch.no_grad():
        ref = m(disp, fea)
        got = m.to(dev)(disp.to(dev), fea.to(dev)).cpu()
    assert got.shape == ref.shape == (2, 111, 135)
    assert float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("c,rates,shape", [(216, [4, 8, 12], (2, 20, 36)), (24, [1, 2, 5], (1, 7, 9))])
def test_aspp_tap_gemm_path(dev, c, rates, shape):
    """ASPP (1x1 + three dilated 3x3 branches, concatenated) through the per-tap GEMM + gather path
    == the four torch convolutions on CPU."""
    from decnet_amd.model import ASPP
    torch.manual_seed(c)
    m = ASPP(c, c, rates).eval()
    for u in m.stages.children():
        u.bn.weight.data.uniform_(0.5, 1.5); u.bn.bias.data.normal_(0, 0.2)
        u.bn.running_mean.data.normal_(0, 0.2); u.bn.running_var.data.uniform_(0.5, 1.5)
    B, H, W = shape
    x = torch.randn(B, c, H, W, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = m(x)
        md = m.to(dev)
        assert md._hip_ok(x.to(dev))
        got = md(x.to(dev)).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))


DIRECT = [  # (segments, Cout, k, dilation, (B, H, W))
    ((8,), 3, 3, 1, (1, 9, 131)), ((8,), 8, 3, 1, (2, 7, 257)), ((12,), 12, 3, 1, (1, 6, 99)),
    ((16,), 8, 3, 1, (1, 11, 65)), ((8, 8, 1), 8, 3, 3, (1, 13, 77)), ((3,), 8, 3, 1, (1, 5, 33)),
    ((8, 4), 24, 3, 2, (1, 9, 51)), ((24,), 24, 1, 1, (1, 4, 45)), ((8,), 4, 1, 1, (1, 3, 1027)),
    ((5, 4), 12, 3, 1, (2, 5, 19)),
]


@pytest.mark.parametrize("segs,cout,k,dil,shape", DIRECT)
def test_conv2d_cat_bn_act_c_entry(dev, segs, cout, k, dil, shape):
    """decnet_conv2d_cat_bn_act itself (ctypes): odd widths, dilation > 1, every output-channel tier (3 / 4, 8, 12, 24),
    concatenated inputs, against float64.  Which kernel runs is the dispatcher's choice -- the *_forced_kernels test
    below repeats this with DECNET_CONV2D_SMALL=mfma / valu and nontemporal stores on every output (DECNET_NT_MB=0)."""
    import ctypes
    from decnet_amd import _lib
    L = _lib.lib()
    B, H, W = shape
    cin = sum(segs)
    g = torch.Generator().manual_seed(cin * 100 + cout)
    xs = [torch.randn(B, c, H, W, generator=g) for c in segs]
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(torch.cat(xs, 1).double(), w.double(), padding=dil * (k // 2), dilation=dil)
    ref = torch.relu(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]).float()
    st = torch.cuda.current_stream().cuda_stream
    xd = [x.to(dev).contiguous() for x in xs]
    wp = torch.empty(L.decnet_conv2d_packed_floats(cin, cout, k, 0), device=dev)
    wd = w.to(dev)
    _lib.check(L.decnet_conv2d_pack_weight(wd.data_ptr(), wp.data_ptr(), cin, cout, k, 0, st), "pack")
    y = torch.full((B, cout, H, W), float("nan"), device=dev)
    ptrs = (ctypes.c_void_p * len(xd))(*[x.data_ptr() for x in xd])
    cins = (ctypes.c_int * len(segs))(*segs)
    sd, hd = scale.to(dev), shift.to(dev)
    _lib.check(L.decnet_conv2d_cat_bn_act(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cins, ctypes.c_void_p),
                                          len(segs), wp.data_ptr(), sd.data_ptr(), hd.data_ptr(), y.data_ptr(), B, cout,
                                          H, W, k, dil, 1, st), "decnet_conv2d_cat_bn_act")
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("force", ["mfma", "valu"])
def test_conv2d_cat_bn_act_forced_kernels(force):
    """The same cases with each small-channel kernel forced (conv2d_f32m on v_mfma_f32_4x4x1 incl. the 12-channel tier /
    the packed-FMA kernel) and every output stored nontemporally -- code paths the graph only reaches at its own shapes."""
    import subprocess
    import sys
    env = dict(os.environ, DECNET_CONV2D_SMALL=force, DECNET_NT_MB="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "test_conv2d_cat_bn_act_c_entry or test_conv_unit_vs_torch_cpu"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_mfma_conv_suite_with_two_accumulator_sets():
    """DECNET_CONV2D_ACC=2 (csrc/conv2d_mfma_acc2.hip: its own packed-weight format and kernels behind the same entry
    points; read once per process): every matrix-core convolution case of this file again."""
    import subprocess
    import sys
    env = dict(os.environ, DECNET_CONV2D_ACC="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "mfma and not two_accumulator"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("split", [0, 1])
def test_tap_gemm_c_entry_points_with_and_without_the_split_copy(dev, split):
    """The C protocol itself (ctypes, no model.py): pack -> [split] -> tap_gemm -> gather.  split = 0 is the protocol of
    a caller that never heard of decnet_tapconv_split_weight: the region behind the fp32 matrices stays unwritten (here:
    poisoned with NaN) and must not be read."""
    import ctypes
    from decnet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    B, Ci, Co, H, W, dil = 1, 216, 216, 20, 36, 4
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=dil, dilation=dil).float()
    xd, wd = x.to(dev), w.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    u = torch.full((L.decnet_tapconv_weight_floats(Ci, 9),), float("nan"), device=dev)
    _lib.check(L.decnet_tapconv_pack_weight(wd.data_ptr(), u.data_ptr(), Co, Ci, 3, 0, st), "pack")
    if split:
        _lib.check(L.decnet_tapconv_split_weight(u.data_ptr(), Ci, 9, st), "split")
    P = B * H * W
    V = torch.empty(L.decnet_tapconv_chunk_floats(B, Ci, H, W), device=dev)
    T = torch.empty(9 * ((Co + 15) // 16) * 16 * P, device=dev)
    y = torch.empty(B, Co, H, W, device=dev)
    one, zero = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    arr = lambda v: (ctypes.c_int * len(v))(*v)
    _lib.check(L.decnet_tapconv_to_chunks(xd.data_ptr(), V.data_ptr(), B, Ci, H, W, st), "chunks")
    _lib.check(L.decnet_tap_gemm(V.data_ptr(), u.data_ptr(), T.data_ptr(), P, Ci, Co, 9, split, st), "tap_gemm")
    _lib.check(L.decnet_tapconv_gather(T.data_ptr(), one.data_ptr(), zero.data_ptr(), y.data_ptr(), B, Co, H, W, 1,
                                       arr([0]), arr([3]), arr([dil]), 0, st), "gather")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert float((y.cpu() - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))


def test_unit_falls_back_when_not_covered(dev):
    u = _unit(216, 216, 3).to(dev)                          # many channels on a 20 x 36 image: library path
    x = torch.randn(1, 216, 20, 36, device=dev)
    with torch.no_grad():
        assert u._hip_kind(x) is None
        got = u(x)
        ref = u.cpu()(x.cpu())
    assert tuple(got.shape) == (1, 216, 20, 36)
    assert float((got.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("H,W,quant", [(131, 517, 0.5), (67, 64, 0.9), (5, 1000, 0.3)])
def test_mask_generator_tail_vs_torch_cpu(dev, H, W, quant):
    """csrc/maskgen.hip: (cur - pre)^2 -> Conv2dUnit(3,3,3x3,BN) -> Conv2dUnit(3,1,1x1,BN) -> sigmoid -> > thold
    (submodule.py:366-372, SparseDenseNetRefinementMask.py:158-170) against the same steps as torch CPU ops."""
    from decnet_amd.model import GenerateSparseMask
    torch.manual_seed(H)
    gen = GenerateSparseMask(8, 3).eval()
    for u in gen.conv:
        u.bn.weight.data.uniform_(0.5, 1.5)
        u.bn.bias.data.normal_(0, 0.3)
        u.bn.running_mean.data.normal_(0, 0.2)
        u.bn.running_var.data.uniform_(0.5, 1.5)
    g = torch.Generator().manual_seed(W)
    cur = torch.randn(2, 8, H, W, generator=g)
    pre = torch.randn(2, 24, (H + 2) // 3, (W + 2) // 3, generator=g)
    if H % 3 or W % 3:                                      # the model only ever sees exact x3 pairs
        H, W = 3 * ((H + 2) // 3), 3 * ((W + 2) // 3)
        cur = torch.randn(2, 8, H, W, generator=g)
    with torch.no_grad():
        logit = gen(cur, pre)                               # CPU: torch ops
        sig = torch.sigmoid(logit)
        thold = float(sig.flatten().quantile(quant))        # a threshold that splits this case's pixels
        ref = (sig > thold).float()
        gen = gen.to(dev)
        got = gen.mask(cur.to(dev), pre.to(dev), thold).cpu()
    assert got.shape == ref.shape and set(got.unique().tolist()) <= {0.0, 1.0}
    sure = (sig - thold).abs() > 1e-4                       # away from the threshold the bits must agree
    assert bool((got == ref)[sure].all()), "mask differs away from the threshold"
    assert float((got != ref).float().mean()) < 1e-3
    assert 0.02 < float(ref.mean()) < 0.98                  # the case really has both values


def test_unfold3_cat_vs_torch(dev):
    """csrc/unfold.hip against torch.cat((disp, F.unfold(fea, 3, stride=3))) (submodule.py:578-580): exact."""
    import torch.nn.functional as F
    from decnet_amd import _lib
    g = torch.Generator().manual_seed(2)
    B, C, h, w = 2, 5, 7, 300
    fea = torch.randn(B, C, 3 * h, 3 * w, generator=g)
    disp = torch.randn(B, h, w, generator=g)
    ref = torch.cat((disp.unsqueeze(1), F.unfold(fea, 3, stride=3).view(B, -1, h, w)), 1)
    f, d = fea.to(dev), disp.to(dev)
    out = torch.empty(B, 9 * C + 1, h, w, device=dev)
    _lib.check(_lib.lib().decnet_unfold3_cat(f.data_ptr(), d.data_ptr(), out.data_ptr(), B, C, h, w,
                                             torch.cuda.current_stream().cuda_stream), "decnet_unfold3_cat")
    assert torch.equal(out.cpu(), ref)
