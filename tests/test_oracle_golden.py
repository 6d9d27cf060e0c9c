"""The oracle against the reference-generated golden vectors (CPU only).

stage0_*.npz were produced by the reference's own GetCostVolume / CostRegNetNoDown /
disparity_regression classes (tests/golden/make_golden.py); oracle/stage0.py must
reproduce them.  net_54x243.npz pins the SpaMat/SpaVar call-site contract.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import stage0 as o0


def _params_from_npz(d):
    out = []
    for i in range(8):
        out.append({"w": torch.from_numpy(d["w%d" % i]),
                    "bn": tuple(torch.from_numpy(d["bn%d_%s" % (i, k)])
                                for k in ("gamma", "beta", "mean", "var"))})
    return out


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_stage0_small_matches_reference(golden_dir):
    d = _load(golden_dir, "stage0_small.npz")
    left, right = torch.from_numpy(d["left"]), torch.from_numpy(d["right"])
    params = _params_from_npz(d)
    pred, reg, cv = o0.stage0_forward(left, right, params, int(d["max_disp"]))
    # same torch ops in the same order as the reference -> essentially bitwise
    np.testing.assert_allclose(cv.numpy(), d["cost_vol"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(reg.numpy(), d["reg"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(pred.numpy(), d["pred"], rtol=0, atol=2e-5)


def test_stage0_c216_matches_reference_and_seeded_params_are_stable(golden_dir):
    d = _load(golden_dir, "stage0_c216.npz")
    params = o0.random_params(216, int(d["param_seed"]))
    # the 35 MB of weights are regenerated from the seed, not stored: detect generator drift
    assert abs(params[0]["w"].double().sum().item() - float(d["w0_checksum"])) < 1e-9
    assert abs(params[7]["w"].double().abs().sum().item() - float(d["w7_checksum"])) < 1e-9
    left, right = torch.from_numpy(d["left"]), torch.from_numpy(d["right"])
    pred, reg, cv = o0.stage0_forward(left, right, params, int(d["max_disp"]))
    np.testing.assert_allclose(cv.numpy(), d["cost_vol"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(reg.numpy(), d["reg"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(pred.numpy(), d["pred"], rtol=0, atol=1e-4)


def full_case_inputs(d):
    """The feature maps of stage0_cfg2_full.npz: regenerated from numpy's frozen RandomState stream, CRC-checked."""
    import zlib
    B, C, H, W = (int(v) for v in d["shape"])
    rs = np.random.RandomState(int(d["input_seed"]))
    left = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    right = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    assert zlib.crc32(left.tobytes() + right.tobytes()) == int(d["input_crc"]), "input stream changed"
    return torch.from_numpy(left), torch.from_numpy(right)


def test_stage0_full_size_golden_pins_the_oracle(golden_dir):
    """BASELINE config 2's whole stage-0 batch (8 x 216 x 20 x 36, D = 8) through the REFERENCE's GetCostVolume /
    CostRegNetNoDown / disparity_regression (tests/golden/make_golden.py --only-stage0-full): oracle/stage0.py on two of
    the eight samples (every op is per sample; the GPU test takes all eight)."""
    d = _load(golden_dir, "stage0_cfg2_full.npz")
    params = o0.random_params(216, int(d["param_seed"]))
    assert abs(params[0]["w"].double().sum().item() - float(d["w0_checksum"])) < 1e-9
    left, right = full_case_inputs(d)
    pred, reg, cv = o0.stage0_forward(left[5:7], right[5:7], params, int(d["max_disp"]))
    np.testing.assert_allclose(reg.numpy(), d["reg"][5:7], rtol=0, atol=1e-4 * max(1.0, float(np.abs(d["reg"]).max())))
    np.testing.assert_allclose(pred.numpy(), d["pred"][5:7], rtol=0, atol=1e-4)


@pytest.mark.parametrize("cf", ["ssd", "cat"])
def test_stage0_other_cost_functions_match_reference(golden_dir, cf):
    """cost_func "ssd" (submodule.py:524-530; demo.py:31's default) and "cat" (:512-516 + CostRegNetNoDown.conv_pre
    :618-619, 651-652) through the REFERENCE's classes (make_golden.py --only-costfunc): small case with stored
    parameters, 216-channel case and two samples of config 2's full batch with seeded ones."""
    d = _load(golden_dir, "stage0_%s_small.npz" % cf)
    left, right = torch.from_numpy(d["left"]), torch.from_numpy(d["right"])
    w_pre = torch.from_numpy(d["w_pre"]) if cf == "cat" else None
    pred, reg, cv = o0.stage0_forward(left, right, _params_from_npz(d), int(d["max_disp"]), cf, w_pre)
    assert cv.shape[1] == (2 if cf == "cat" else 1) * left.shape[1]
    np.testing.assert_allclose(cv.numpy(), d["cost_vol"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(reg.numpy(), d["reg"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(pred.numpy(), d["pred"], rtol=0, atol=2e-5)
    for name, sl in (("stage0_%s_c216.npz" % cf, None), ("stage0_cfg2_%s_full.npz" % cf, slice(2, 4)),
                     ("stage0_cfg3_%s_full.npz" % cf, slice(1, 2)), ("stage0_cfg4_%s_full.npz" % cf, slice(0, 1))):
        d = _load(golden_dir, name)
        seed = int(d["param_seed"])
        params = o0.random_params(216, seed)
        assert abs(params[0]["w"].double().sum().item() - float(d["w0_checksum"])) < 1e-9
        w_pre = o0.random_w_pre(216, seed) if cf == "cat" else None
        if w_pre is not None and "w_pre_checksum" in d.files:
            assert abs(w_pre.double().abs().sum().item() - float(d["w_pre_checksum"])) < 1e-9
        if sl is None:
            left, right = torch.from_numpy(d["left"]), torch.from_numpy(d["right"])
            want_reg, want_pred = d["reg"], d["pred"]
        else:
            left, right = (t[sl] for t in full_case_inputs(d))
            want_reg, want_pred = d["reg"][sl], d["pred"][sl]
        pred, reg, cv = o0.stage0_forward(left, right, params, int(d["max_disp"]), cf, w_pre)
        if sl is None:
            np.testing.assert_allclose(cv.numpy(), d["cost_vol"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(reg.numpy(), want_reg, rtol=0, atol=1e-4 * max(1.0, float(np.abs(want_reg).max())))
        np.testing.assert_allclose(pred.numpy(), want_pred, rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", ["stage0_small.npz", "stage0_c216.npz"])
def test_closed_form_warp_equals_grid_sample(golden_dir, name):
    """the stretched, half-pixel-shifted bilinear warp in closed form."""
    d = _load(golden_dir, name)
    left, right = torch.from_numpy(d["left"]), torch.from_numpy(d["right"])
    D = int(d["max_disp"])
    B, C, H, W = right.shape
    ref = o0.warp_right(right, o0.disp_samples(D, B, H, W))
    cf = o0.warp_right_closed_form(right, D)
    np.testing.assert_allclose(cf.numpy(), ref.numpy(), rtol=0, atol=5e-6)
    # and the golden cost volume = masked left x warp
    keep = (torch.arange(W).view(1, 1, 1, 1, W) >= torch.arange(D).view(1, 1, D, 1, 1)).float()
    cv = left.unsqueeze(2) * keep * cf
    np.testing.assert_allclose(cv.numpy(), d["cost_vol"], rtol=0, atol=2e-5)
    # d=0 is NOT the identity (the reason the quirk must be reproduced)
    assert (cf[:, :, 0] - right).abs().max() > 1e-2


def test_net_callsite_contract(golden_dir):
    d = _load(golden_dir, "net_54x243.npz")
    C = {1: 72, 2: 24, 3: 8}
    for i in (1, 2, 3):
        ref = d["sm%d_ref" % i]
        assert ref.shape[1] == C[i]
        assert int(d["sm%d_max_disp" % i]) == 216 // 3 ** (3 - i)
        assert str(d["sm%d_max_disp_type" % i]) == "int64"          # numpy.int64, S13
        assert bool(d["sv%d_same_inputs_as_sm" % i]) and bool(d["sv%d_disparity_is_sm_out" % i])
        for m in ("rmask", "tmask"):
            assert set(np.unique(d["sm%d_%s" % (i, m)])) <= {0.0, 1.0}
        # the oracle reproduces what the stub produced at generation time (determinism)
        o, s, m = oracle.spamat_forward(ref, d["sm%d_tar" % i], d["sm%d_rmask" % i],
                                        d["sm%d_tmask" % i], d["sm%d_max_disp" % i])
        np.testing.assert_array_equal(o, d["sm%d_out" % i])
        v, _, _ = oracle.spavar_forward(ref, d["sm%d_tar" % i], d["sm%d_rmask" % i],
                                        d["sm%d_tmask" % i], d["sv%d_disparity" % i],
                                        d["sv%d_max_disp" % i])
        np.testing.assert_array_equal(v, d["sv%d_out" % i])
        assert (o[d["sm%d_rmask" % i] == 0] == 0).all()
