"""decnet_amd.eval / decnet_amd.loader host logic (reference eval.py:114-228, modules/loss.py:427-437,
loader/SceneflowMask.py:115-203, utils/utils.py:281-321).  CPU."""
import os
import pickle

import numpy as np
import pytest
import torch

from decnet_amd import eval as dev_eval
from decnet_amd import loader


def test_test_loss_func_known_answers():
    gt = torch.tensor([[[0.0, 10.0, 100.0, 300.0], [50.0, 20.0, 1.0, 191.0]]])
    pred = torch.tensor([[[5.0, 12.5, 104.9, 0.0], [53.0, 16.9, 9.0, 191.0]]])
    # valid: 0 < gt < 192 -> 10, 100, 50, 20, 1, 191 (6 px); errors 2.5, 4.9, 3.0, 3.1, 8.0, 0
    # good: 2.5 (<3), 4.9 (< 5 % of 100), 3.0 (not < 3, not < 2.5) bad, 3.1 bad, 8 bad, 0 good -> 3 of 6
    epe, l3 = dev_eval.test_loss_func(pred, gt, 192)
    assert abs(float(epe) - (2.5 + 4.9 + 3.0 + 3.1 + 8.0 + 0.0) / 6) < 1e-5
    assert abs(float(l3) - 50.0) < 1e-4
    epe2, l32 = dev_eval.test_loss_func(pred, gt, 400)                 # now 300 is valid too (error 300)
    assert abs(float(epe2) - (21.5 + 300) / 7) < 1e-4 and abs(float(l32) - (100 - 300 / 7)) < 1e-4


def test_pfm_reader(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = tmp_path / "d.pfm"
    with open(p, "wb") as f:
        f.write(b"Pf\n4 3\n-1.0\n")
        f.write(np.flipud(a).astype("<f4").tobytes())
    d, scale = loader.read_pfm(str(p))
    assert scale == 1.0 and np.array_equal(d, a)


def test_npy_pairs_layout(tmp_path):
    rng = np.random.RandomState(0)
    root = tmp_path / "sf"
    (root / "test").mkdir(parents=True)
    (root / "test_mask").mkdir()
    arr = np.concatenate([rng.randint(0, 255, (30, 50, 6)).astype(np.float32),
                          rng.rand(30, 50, 1).astype(np.float32) * 40], -1)
    np.save(root / "test" / "a.npy", arr)
    masks = [np.ones((54, 54)), np.ones((18, 18)), np.zeros((6, 6))] * 2     # fine -> coarse, left then right
    with open(root / "test_mask" / "a", "wb") as f:
        pickle.dump(masks, f)
    ds = loader.get_loader("SceneflowMask")(str(root), split="test", use_detail=True)
    assert len(ds) == 1
    left, right, disp, image, lm1, lm2, lm3, rm1, rm2, rm3, oh, ow, name, nd = ds[0]
    assert left.shape == (3, 54, 54) and disp.shape == (54, 54) and (oh, ow, name, nd) == (30, 50, "a", -1)   # no per-sample range
    assert lm1.shape == (6, 6) and lm3.shape == (54, 54) and float(lm1.sum()) == 0      # coarsest first
    assert float(disp[:24].abs().sum()) == 0 and float(disp[:, :4].abs().sum()) == 0   # top/left padding
    np.testing.assert_allclose(disp[24:, 4:].numpy(), arr[..., 6])
    want = (arr[0, 0, 0:3] / 255 - loader.MEAN) / loader.STD
    np.testing.assert_allclose(left[:, 24, 4].numpy(), want, rtol=1e-5)
    np.testing.assert_allclose(left[:, 0, 0].numpy(), (0 - loader.MEAN) / loader.STD, rtol=1e-5)   # padded pixel


def test_pair_directory_and_batches(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(1)
    for n, nd in (("p0", None), ("p1", 40)):
        d = tmp_path / n
        d.mkdir()
        for f in ("im0.png", "im1.png"):
            Image.fromarray(rng.randint(0, 255, (27, 54, 3)).astype(np.uint8)).save(str(d / f))
        Image.fromarray((rng.rand(27, 54) * 30 * 256).astype(np.uint16)).save(str(d / "disp0.png"))
        if nd:
            (d / "calib.txt").write_text("ndisp=%d\n" % nd)
    ds = loader.get_loader("pairs")(str(tmp_path), use_detail=False)
    assert len(ds) == 2
    s0, s1 = ds[0], ds[1]
    assert s0[-1] <= 0 and s1[-1] == 54 and s0[-2] == "p0"          # no calib.txt: eval keeps --max_disp
    assert s0[4].shape == (3, 6) and s0[6].shape == (27, 54) and set(np.unique(s0[6].numpy())) <= {0.0, 1.0}
    assert float(s0[2].max()) < 30.001 and float(s0[2].max()) > 1
    b = dev_eval.batches_of(ds, 8)
    assert b == [[0, 1]]
    # MiddleburyMask: per-sample ranges and sizes -> batch size 1, decided before any batch is computed
    assert dev_eval.batches_of(ds, 8, "MiddleburyMask") == [[0], [1]]
    assert dev_eval.batches_of(ds, 8, "kitti15mask") == [[0, 1]]
    cols = dev_eval.collate([s0, s1])
    assert cols[0].shape == (2, 3, 27, 54) and cols[12] == ["p0", "p1"]


def test_middlebury_pickle_layout(tmp_path):
    """loader/MiddleburyMask.py:117-131: pickled dicts {ndisp, im0, im1, disparity} under
    <root>/MiddEval3H_processed/trainingH, masks under trainingH_mask; the sample's ndisp travels as n_disp."""
    rng = np.random.RandomState(2)
    d = tmp_path / "MiddEval3H_processed" / "trainingH"
    d.mkdir(parents=True)
    (tmp_path / "MiddEval3H_processed" / "trainingH_mask").mkdir()
    gt = rng.rand(20, 31).astype(np.float32) * 50
    gt[3, 4] = np.inf
    with open(d / "Adirondack.pkl", "wb") as f:
        pickle.dump({"ndisp": 145, "im0": rng.randint(0, 255, (20, 31, 3)).astype(np.uint8),
                     "im1": rng.randint(0, 255, (20, 31, 3)).astype(np.uint8), "disparity": gt.copy()}, f)
    masks = [np.ones((27, 54)), np.ones((9, 18)), np.zeros((3, 6))] * 2
    with open(tmp_path / "MiddEval3H_processed" / "trainingH_mask" / "Adirondack", "wb") as f:
        pickle.dump(masks, f)
    ds = loader.get_loader("MiddleburyMask")(str(tmp_path), split="eval_H", use_detail=True)
    left, right, disp, image, lm1, lm2, lm3, rm1, rm2, rm3, oh, ow, name, nd = ds[0]
    assert (oh, ow, name, nd) == (20, 31, "Adirondack", 145) and left.shape == (3, 27, 54)
    assert float(disp[7 + 3, 23 + 4]) == 0 and np.isfinite(disp.numpy()).all()      # inf -> 0 (MiddleburyMask.py:128)
    np.testing.assert_allclose(disp[7:, 23:].numpy()[5], gt[5])
    assert lm1.shape == (3, 6) and lm3.shape == (27, 54)
    try:
        loader.get_loader("MiddleburyMask")(str(tmp_path), split="nope")
        assert False
    except Exception as e:
        assert "split" in str(e)


def test_parser_defaults_follow_eval_sh():
    a = dev_eval.build_parser().parse_args([])
    assert a.max_disp == 216 and a.base_channels == 8 and a.num_stage == 4 and a.gpus == 1


def test_batch_max_disp_follows_each_callers_rule():
    """eval.py:173-175 (Middlebury: int(n_disp), unrounded -> floor-divided per stage, …Mask.py:124),
    demo.py:149-155 ('pairs': ceil to a multiple of 27), everything else keeps --max_disp."""
    from decnet_amd.eval import batch_max_disp
    assert batch_max_disp("MiddleburyMask", [290], 216) == 290
    assert [290 // 3 ** k for k in (3, 2, 1, 0)] == [10, 32, 96, 290]
    with pytest.raises(ValueError):
        batch_max_disp("MiddleburyMask", [290, 145], 216)
    assert batch_max_disp("pairs", [400, -1], 216) == 405
    assert batch_max_disp("pairs", [-1, 0], 216) == 216
    assert batch_max_disp("KITTI15Mask", [300], 216) == 216
    assert batch_max_disp("SceneflowMask", [-1], 192) == 192
