"""decnet_amd.augment / NpyPairs(is_training=True) -- the reference's training-time sample preparation
(loader/SceneflowMask.py:131-196, loader/KITTI15Mask.py:125-245, 256-366).  The stripe-noise functions and
RandomPhotometric are compared, draw for draw, with outputs recorded from the reference's own functions
(tests/golden/make_loader_train_golden.py); the crop / occlusion / tuple plumbing by its properties.  CPU only."""
import os
import pickle

import numpy as np
import pytest
import torch

from decnet_amd import augment, loader

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_train.npz")


def _images():
    yy, xx, cc = np.meshgrid(np.arange(200), np.arange(340), np.arange(3), indexing="ij")
    return (((7 * yy + 3 * xx + 50 * cc) % 256).astype(np.float32),
            ((5 * yy + 11 * xx + 90 * cc) % 256).astype(np.float32))


@pytest.mark.parametrize("kind,fn", [("grey", augment.stripe_noise_grey), ("colour", augment.stripe_noise_colour)])
@pytest.mark.parametrize("seed", [1, 2])
def test_stripe_noise_equals_the_references(kind, fn, seed):
    z = np.load(GOLD)
    left, right = _images()
    np.random.seed(seed)
    l, r = fn(left, right)
    for side, mine, base in (("l", l, left), ("r", r, right)):
        key = "%s_%s_%d" % (kind, side, seed)
        y0, y1, x0, x1 = (int(v) for v in z[key + "_box"])
        np.testing.assert_array_equal(mine[y0:y1, x0:x1], z[key])            # same draws, same arithmetic: bit equal
        outside = mine.copy()
        outside[y0:y1, x0:x1] = base[y0:y1, x0:x1]
        np.testing.assert_array_equal(outside, base)                          # nothing else touched
    assert l.max() <= 255 and r.max() <= 255 and l is not left and r is not right


@pytest.mark.parametrize("seed", [1, 2])
def test_random_photometric_equals_the_references(seed):
    z = np.load(GOLD)
    np.random.seed(seed)
    out = augment.RandomPhotometric()(torch.from_numpy(z["photo_in"]).clone())
    np.testing.assert_array_equal(out.numpy(), z["photo_%d" % seed])


def test_random_crop_draws_and_crops_the_masks_with_the_image():
    data = np.arange(54 * 81 * 7, dtype=np.float32).reshape(54, 81, 7)
    masks = [np.arange(54 * 81).reshape(54, 81), np.arange(18 * 27).reshape(18, 27), np.arange(6 * 9).reshape(6, 9)] * 2
    np.random.seed(3)
    x1, y1 = np.random.randint(0, 54 - 27 + 1), np.random.randint(0, 81 - 54 + 1)
    np.random.seed(3)
    out, m, corner = augment.random_crop(data, masks, (20, 50))              # -> 27 x 54 (rounded up to x27)
    assert corner == (x1, y1) and out.shape == (27, 54, 7)
    np.testing.assert_array_equal(out, data[x1:x1 + 27, y1:y1 + 54])
    for i, s in enumerate((1, 3, 9, 1, 3, 9)):
        np.testing.assert_array_equal(m[i], masks[i][x1 // s:(x1 + 27) // s, y1 // s:(y1 + 54) // s])
    st = np.random.get_state()[1][:4].copy()
    same, m2, c2 = augment.random_crop(data, masks, (54, 81))                # full size: no draw at all
    assert same is data and c2 == (0, 0) and (np.random.get_state()[1][:4] == st).all()


def test_occlusion_rectangle_is_the_mean_colour():
    np.random.seed(4)
    r = np.random.rand(200, 300, 3).astype(np.float32)
    mean = np.mean(np.mean(r, 0), 0)
    np.random.seed(9)
    sh, sw = int(np.random.uniform(30, 80)), int(np.random.uniform(10, 80))
    ch, cw = int(np.random.uniform(sh, 200 - sh)), int(np.random.uniform(sw, 300 - sw))
    np.random.seed(9)
    o = augment.occlude_right(r.copy())
    np.testing.assert_allclose(o[ch - sh:ch + sh, cw - sw:cw + sw], np.broadcast_to(mean, (2 * sh, 2 * sw, 3)), rtol=1e-6)
    o[ch - sh:ch + sh, cw - sw:cw + sw] = r[ch - sh:ch + sh, cw - sw:cw + sw]
    np.testing.assert_array_equal(o, r)


@pytest.mark.parametrize("name,policy", [("SceneflowMask", "sceneflow"), ("KITTI15Mask", "kitti"),
                                         ("DrivingStereoMask", "drivingstereo")])
def test_training_sample_tuple(tmp_path, name, policy):
    """The 10-tuple of SceneflowMask.py:195-196 at the cropped size, masks cropped with it, deterministic per seed."""
    rs = np.random.RandomState(0)
    os.makedirs(tmp_path / "train")
    os.makedirs(tmp_path / "train_mask")
    arr = rs.uniform(0, 255, (300, 520, 8)).astype(np.float32)
    arr[..., 6] = rs.uniform(0, 100, (300, 520))
    arr[..., 7] = rs.rand(300, 520) < 0.5
    np.save(tmp_path / "train" / "a.npy", arr)
    Hp, Wp = 324, 540
    masks = [(rs.rand(Hp // s, Wp // s) < 0.5).astype(np.float64) for s in (1, 3, 9)] * 2
    with open(tmp_path / "train_mask" / "a", "wb") as f:
        pickle.dump(masks, f)
    assert loader.training_policy(name) == policy
    ds = loader.get_loader(name)(str(tmp_path), split="train", is_training=True, img_size=(260, 500), policy=policy)
    np.random.seed(11)
    a = ds[0]
    np.random.seed(11)
    b = ds[0]
    assert len(a) == 10
    left, right, disp, image, lm1, lm2, lm3, rm1, rm2, rm3 = a
    assert left.shape == right.shape == (3, 270, 513) and disp.shape == (270, 513) and image.shape == (3, 270, 513)
    assert left.dtype == torch.float32 and lm1.shape == (30, 57) and lm2.shape == (90, 171) and rm3.shape == (270, 513)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert float(image.max()) > 1.5                                          # the raw 0..255 left crop (ToTensor on floats)
    np.random.seed(12)
    c = ds[0]
    assert not torch.equal(c[0], left)
    # evaluation mode is untouched by the flag's existence
    ev = loader.get_loader(name)(str(tmp_path), split="train")[0]
    assert len(ev) == 14 and ev[0].shape == (3, 324, 540)
