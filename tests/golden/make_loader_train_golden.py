"""Fixtures for decnet_amd.augment from the REFERENCE's own functions (build container only; the reference never ships).

    python tests/golden/make_loader_train_golden.py        # -> tests/golden/loader_train.npz

Runs, after np.random.seed(s), the reference's
  SceneflowMask.add_paralex_noise      loader/SceneflowMask.py:255-283
  KITTI15Mask.add_paralex_noise        loader/KITTI15Mask.py:256-304   (spells np.int, removed from numpy >= 1.24:
                                                                       aliased to int for the duration of the call)
  RandomPhotometric.__call__           loader/KITTI15Mask.py:340-366
on seeded synthetic images and stores their outputs.  The loaders import cv2 / torchvision at module level (absent
here): empty stand-in modules are registered for the import only -- none of the three functions touches them.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    from loader.SceneflowMask import SceneflowMask
    from loader.KITTI15Mask import KITTI15Mask, RandomPhotometric
    rs = np.random.RandomState(5)
    yy, xx, cc = np.meshgrid(np.arange(200), np.arange(340), np.arange(3), indexing="ij")
    left = ((7 * yy + 3 * xx + 50 * cc) % 256).astype(np.float32)             # (smooth: the fixture compresses)
    right = ((5 * yy + 11 * xx + 90 * cc) % 256).astype(np.float32)
    out = {}
    sf, kt = object.__new__(SceneflowMask), object.__new__(KITTI15Mask)
    for seed in (1, 2):
        np.random.seed(seed)
        l, r = sf.add_paralex_noise(left, right)
        out["grey_l_%d" % seed], out["grey_r_%d" % seed] = l, r
        np.random.seed(seed)
        np.int = int
        try:
            l, r = kt.add_paralex_noise(left, right)
        finally:
            del np.int
        out["colour_l_%d" % seed], out["colour_r_%d" % seed] = l, r
    im = torch.from_numpy(rs.uniform(0, 1, (3, 40, 60)).astype(np.float32))
    out["photo_in"] = im.numpy()
    jit = RandomPhotometric(noise_stddev=0.0, min_contrast=-0.37, max_contrast=0.37, brightness_stddev=0.02, min_color=0.9,
                            max_color=1.1, min_gamma=0.7, max_gamma=1.7)
    for seed in (1, 2):
        np.random.seed(seed)
        out["photo_%d" % seed] = jit(im.clone()).numpy()
    # keep the fixture small: store the inputs by seed, the outputs as differences on the touched region only
    keep = {"photo_in": out["photo_in"], "photo_1": out["photo_1"], "photo_2": out["photo_2"]}
    for k in list(out):
        if k.startswith(("grey", "colour")):
            base = left if "_l_" in k else right
            d = out[k] - base
            ys, xs = np.nonzero(np.abs(d).sum(-1))
            keep[k + "_box"] = np.array([ys.min(), ys.max() + 1, xs.min(), xs.max() + 1])
            keep[k] = out[k][ys.min():ys.max() + 1, xs.min():xs.max() + 1]
            keep[k + "_outside_equal"] = np.array(int(np.count_nonzero(d) == np.count_nonzero(
                d[ys.min():ys.max() + 1, xs.min():xs.max() + 1])))
    np.savez_compressed(os.path.join(HERE, "loader_train.npz"), **keep)
    print(os.path.getsize(os.path.join(HERE, "loader_train.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
