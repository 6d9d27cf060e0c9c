"""north_star's parity gate (BASELINE config 1): the bundled InputData pairs through this repo's demo path
(decnet_amd.demo.run_pair: pad x27, /255, normalise, max_disp from calib.txt, the MI355X graph, x256 uint16)
against the REFERENCE graph run on the same pixels with the shipped base_channels=8 network
(tests/golden/inputdata/*.npz, made by tests/golden/make_inputdata_golden.py from the imported reference).  -m gpu.

Gates per pair (8 cases: 4 pairs x {init17, fill} weights), all asserted below:
  (1) distance to the reference graph's FLOAT64 run <= 1.5 x the distance of the reference's own float32 run
      (measured round 4: 0.75 - 1.31; with the many-channel 2-D convolutions on the library, DECNET_CONV2D_MFMA=0,
      0.77 - 1.00, asserted <= 1.1 in a second leg below: the excess is the bf16x3 arithmetic of csrc/conv2d_mfma.hip);
  (2) mean abs difference to the reference's float32 run <= 1e-3 px where float32 itself allows it (the reference's own
      float32-vs-float64 noise below 4e-4 px), <= 1e-3 + 2 x that noise otherwise -- i.e. the 1e-3 px gate IS RELAXED on
      the noisy init17 pairs, by exactly the amount two float32 runs may differ;
  (3) at most 10 % of the pixels set aside as flip neighbourhoods (measured: <= 3.5 %), and the ALL-PIXEL mean abs
      difference, flipped neighbourhoods included, <= 1e-2 px (measured: <= 4.8e-3 px).
Two caveats, both measured rather than assumed:
  * The masks are thresholded sigmoids (SparseDenseNetRefinementMask.py:163-170): a logit
    within float noise of the threshold may flip a bit, which moves that pixel (and, through the refinement
    convolutions, its neighbourhood) by whole pixels.  The test counts flips per stage and view (bounded)
    and gates the mean abs difference on the pixels outside the dilated flip neighbourhoods
    (tests/golden/flipmap.py); the all-pixel mean is printed.
  * With the from-scratch weights ("init17": He-normal convolutions, unit BatchNorm statistics -- what demo.py
    runs without --resume) the untrained refinement stack outputs |disparity| ~ 55-160 px with excursions to
    -250 / +440 and amplifies rounding: the REFERENCE graph's own float32 run differs from its own float64
    run by 0.6-1.6e-3 px mean abs on these pairs (fixture key ref_fp32_noise, measured by the generator).
    No float32 implementation can be closer to the reference's float32 run than ~that, so the gate is
    max(1e-3, 2 x ref_fp32_noise) px.  With normalised weights ("fill") the noise is ~7e-5 px and the plain
    1e-3 px gate applies with a 10x margin.  Stage-level results (stage 0, SpaMat outputs) are gated tightly
    in both cases.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "inputdata")
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

CASES = [("Sceneflow", "0006", "init17"), ("KITTI", "000009_10", "init17"), ("real", "00003", "init17"),
         ("real", "00004", "init17"), ("Sceneflow", "0006", "fill"), ("KITTI", "000009_10", "fill"),
         ("real", "00003", "fill"), ("real", "00004", "fill")]
from flipmap import dirty_map  # noqa: E402
from spy_util import spamat_spy  # noqa: E402


def _unpack(d, key):
    shape = tuple(int(v) for v in d[key + "_shape"])
    return np.unpackbits(d[key], axis=-1)[..., :shape[-1]].astype(bool).reshape(shape)


def _build(variant, thold):
    from make_golden import E2E_KW
    from netparams import fill_state_dict
    from decnet_amd.model import get_model
    kw = dict(E2E_KW)
    kw.update(base_channels=8, thold=thold)
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(17)                                  # demo.py:70
        model = get_model(**kw)
    if variant == "fill":
        model.load_state_dict(fill_state_dict(model.state_dict()))
    return model


@pytest.mark.parametrize("ds_name,name,variant", CASES)
def test_inputdata_pair_matches_reference_graph(ds_name, name, variant):
    assert torch.cuda.is_available()
    from decnet_amd import demo
    import decnet_amd.model as M
    d = np.load(os.path.join(GOLD, "%s_%s_%s.npz" % (ds_name, name, variant)))
    dev = torch.device("cuda:0")
    model = _build(variant, float(d["thold"]))
    wsum = sum(v.double().abs().sum().item() for v in model.state_dict().values())
    assert abs(wsum - float(d["w_checksum"])) <= 1e-9 * float(d["w_checksum"]), "weights differ from the fixture's"
    model.skip_stage_id = int(d["skip_stage_id"])
    model = model.to(dev).eval()

    rec, preds = {}, {}

    def spy(L, R, lm, rm, D, o):
        i = len(rec) + 1
        rec[i] = (lm[0].cpu().numpy() != 0, rm[0].cpu().numpy() != 0, o[0][0].cpu().numpy(), int(D))
    hooks = [model.refinement[i].register_forward_hook(
        lambda m, inp, out, i=i: preds.__setitem__(i + 1, out[0][0].cpu().numpy())) for i in range(3)]
    hooks.append(model.register_forward_hook(lambda m, inp, out: preds.__setitem__("final", out[-1][0].cpu().numpy())))
    reg = model.cost_regularizer
    stage0 = reg.stage0
    reg.stage0 = lambda *a, **k: preds.setdefault(0, stage0(*a, **k))
    try:
        pdir = os.path.join(GOLD, ds_name, name)
        limg, rimg = demo.read_rgb(os.path.join(pdir, "im0.png")), demo.read_rgb(os.path.join(pdir, "im1.png"))
        with spamat_spy(spy):
            png, _ = demo.run_pair(model, limg, rimg, dev, demo.read_ndisp(os.path.join(pdir, "calib.txt")))
    finally:
        del reg.stage0
        for h in hooks:
            h.remove()
    assert model.max_disp == int(d["max_disp"])
    assert png.shape == tuple(int(v) for v in d["ori_hw"]) and png.dtype == np.uint16
    pred = preds["final"]
    assert pred.shape == tuple(int(v) for v in d["shape"])
    H, W = pred.shape
    s3 = (slice(1, None, 3), slice(1, None, 3))

    # stage 0: no masks involved
    p0 = preds[0][0].cpu().numpy()
    e0 = np.abs(p0 - d["pred0"])
    print("\n%s/%s %s: max_disp %d; stage 0 mean %.2e max %.2e px" % (ds_name, name, variant, model.max_disp,
                                                                    e0.mean(), e0.max()))
    assert e0.mean() < 1e-4 and e0.max() < 2e-3

    n_stages = len(rec)
    assert n_stages == min(3, int(d["skip_stage_id"]) - 1)
    flips = {}
    for i in range(1, n_stages + 1):
        lm, rm, sp, D = rec[i]
        assert D == model.max_disp // 3 ** (3 - i)
        lf, rf = lm != _unpack(d, "lmask%d" % i), rm != _unpack(d, "rmask%d" % i)
        print("  stage %d: density %.4f / %.4f, flipped mask bits %d left, %d right of %d"
              % (i, lm.mean(), rm.mean(), lf.sum(), rf.sum(), lf.size))
        assert lf.mean() < 2e-3 and rf.mean() < 2e-3
        flips[i] = (lf, rf)
    dirty, hits = dirty_map(flips, model.max_disp, H, W)
    for i in range(1, n_stages + 1):
        lm, rm, sp, D = rec[i]
        ref_sp = d["sparse1"] if i == 1 else d["sparse%d_s3" % i]
        mine, ok = (sp, ~hits[i] & lm) if i == 1 else (sp[s3], (~hits[i] & lm)[s3])
        if ok.any():
            es = np.abs(mine - ref_sp)[ok]
            print("           SpaMat output on %d unflipped active pixels: mean %.2e max %.2e px" % (ok.sum(), es.mean(),
                                                                                                  es.max()))
            assert es.mean() < 2e-4 and es.max() < 5e-3
    # Three float32 / float64 runs of the same graph on the same pixels: this repo (GPU), the reference in float32 and
    # the reference in float64 (the "truth" both float32 runs scatter around; fixture keys pred64_s3, *mask*_64).
    # Pixels in reach of a mask bit that differs between ANY two of them are set aside (a flipped bit moves its
    # neighbourhood by whole pixels in all three comparisons alike).
    flips64 = {}
    for i in range(1, n_stages + 1):
        shp = tuple(int(v) for v in d["lmask%d_shape" % i])
        l64 = np.unpackbits(d["lmask%d_64" % i], axis=-1)[..., :shp[-1]].astype(bool).reshape(shp)
        r64 = np.unpackbits(d["rmask%d_64" % i], axis=-1)[..., :shp[-1]].astype(bool).reshape(shp)
        flips64[i] = (flips[i][0] | (l64 != _unpack(d, "lmask%d" % i)), flips[i][1] | (r64 != _unpack(d, "rmask%d" % i)))
    dirty64, _ = dirty_map(flips64, model.max_disp, H, W)
    clean = ~dirty64[s3]
    err = np.abs(pred[s3] - d["pred_s3"])                               # vs the reference's float32 run
    e_gpu64 = np.abs(pred[s3].astype(np.float64) - d["pred64_s3"])      # vs the float64 run
    e_ref64 = np.abs(d["pred_s3"].astype(np.float64) - d["pred64_s3"])  # the reference's own float32 run vs float64
    assert clean.mean() > 0.9, "more than 10 % of the pixels in reach of a flipped mask bit"
    assert err.mean() < 1e-2, "all-pixel mean abs difference (flip neighbourhoods included) above 1e-2 px"
    g64, r64m = float(e_gpu64[clean].mean()), float(e_ref64[clean].mean())
    print("  final: |gpu - ref32| mean %.2e px over the %.1f%% pixels outside flip neighbourhoods (max %.2e); all pixels "
          "%.2e; |pred| mean %.1f" % (err[clean].mean(), 100 * clean.mean(), err[clean].max(), err.mean(),
                                       float(d["pred_abs_mean"])))
    print("         distance to the float64 run: gpu %.2e px, the reference's own float32 run %.2e px (ratio %.2f); "
          "all pixels: gpu %.2e, reference %.2e" % (g64, r64m, g64 / max(r64m, 1e-12), e_gpu64.mean(), e_ref64.mean()))
    # (1) as close to the truth as the reference's own float32 arithmetic is.  Measured (round 3):
    #     init17 0.74 / 0.81 / 0.84 / 1.38, fill 1.20 / 1.29 / 1.32 / 1.20 x the reference's own distance; with the
    #     many-channel 2-D convolutions on the library instead of the bf16x3 matrix-core kernels 0.91 - 1.00, i.e. the
    #     excess is the six-product bf16 arithmetic of csrc/conv2d_mfma.hip (1.33 - 1.58 before its terms were rounded
    #     to nearest instead of truncated).  Bound: 1.5 x + 2e-5 px; 1.1 x with those convolutions on the library
    #     (second leg, test_fp32_trunk_is_as_close_to_float64_as_the_reference).
    ratio_bound = float(os.environ.get("DECNET_TEST_RATIO_BOUND", "1.5"))
    assert g64 <= ratio_bound * r64m + 2e-5, "much further from the float64 run than the reference's float32 run is"
    # (2) north_star's plain gate, 1e-3 px mean against the reference's float32 run, wherever float32 itself allows it:
    #     two float32 runs that are each e from the truth differ by up to ~2 e, so the plain gate is asserted when the
    #     reference's own float32 noise is below 4e-4 px (every "fill" pair but real/00004 at max_disp 621) and
    #     replaced by 1e-3 + 2 x that noise otherwise (the untrained init17 network amplifies rounding, see above)
    gate = 1e-3 if r64m < 4e-4 else 1e-3 + 2 * r64m
    print("         gate vs the reference's float32 run: %.2e px (%s)" % (gate, "plain 1e-3" if gate == 1e-3 else
                                                                            "1e-3 + 2 x reference noise"))
    assert err[clean].mean() < gate
    # the written image: x256 uint16 (demo.py:191-197); one count = 1/256 px
    dp = np.abs(png.astype(np.int64)[s3] - d["pred_png_s3"].astype(np.int64))
    oh, ow = png.shape
    cl = dirty64[-oh:, -ow:][s3]
    print("  png: mean |difference| %.3f counts, %.3f%% of the sampled counts differ by more than 1, outside flip "
          "neighbourhoods" % (dp[~cl].mean(), 100 * (dp[~cl] > 1).mean()))
    assert dp[~cl].mean() < 256 * gate + 0.5      # both sides truncate to 1/256 px


def test_fp32_trunk_is_as_close_to_float64_as_the_reference():
    """Second leg: with the many-channel 2-D convolutions on the library's fp32 kernels
    (DECNET_CONV2D_MFMA=0; the switch is read once per process, hence the child) the graph must be as close to the
    float64 run as the reference's own float32 run: ratio <= 1.1 (measured 0.77 - 1.00) on three pairs.  A regression of
    the fp32 kernels (SpaMat / SpaVar, stage 0, the few-channel convolutions) shows here without the bf16x3 noise."""
    import subprocess
    env = dict(os.environ, DECNET_CONV2D_MFMA="0", DECNET_TEST_RATIO_BOUND="1.1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "pair_matches and (Sceneflow-0006-fill or KITTI-000009_10-fill or real-00004-init17)"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_accumulator_trunk_is_as_close_to_float64_as_the_reference():
    """Third leg (round 5): DECNET_CONV2D_ACC=2 keeps the many-channel 2-D convolutions on the bf16 matrix pipe but
    collects the small bf16x3 term groups in a second accumulator set (csrc/conv2d_mfma_acc2.hip) -- the supported switch
    for reference-grade fp32 in the trunk.  Same bound as the library leg: ratio <= 1.1 on the same three pairs."""
    import subprocess
    env = dict(os.environ, DECNET_CONV2D_ACC="2", DECNET_TEST_RATIO_BOUND="1.1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-s", "-k",
                        "pair_matches and (Sceneflow-0006-fill or KITTI-000009_10-fill or real-00004-init17)"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    print("\n".join(ln for ln in r.stdout.splitlines() if "distance to the float64 run" in ln))
