"""Evaluation-time data sources for decnet_amd.eval.

Three layouts, all yielding the tuple the reference's datasets return in test mode
(loader/SceneflowMask.py:198-203): ``left, right, disparity, image, left_mask1..3, right_mask1..3,
ori_h, ori_w, name, n_disp`` with the images padded on the top/left to multiples of 27
(SceneflowMask.py:118-130), scaled to [0,1] and normalised with the ImageNet statistics (:152-153,
:206-210):

* ``NpyPairs``  -- the reference's pre-baked format: ``<root>/<split>/*.npy`` arrays ``[H,W,7]`` (left RGB,
  right RGB, disparity; SceneflowMask.py:115,144-146) and, optionally, the pickled list of six detail masks
  next to them in ``<root>/<split>_mask/<name>`` (left fine->coarse, right fine->coarse; :166-185).
* ``MiddleburyPickles`` -- loader/MiddleburyMask.py:117-131: ``<root>/<MiddEval3?_processed>/<split dir>/*.pkl``, each a
  pickled dict ``{ndisp, im0, im1 [, disparity] [, disparity_right]}`` (inf in the ground truth -> 0), masks pickled in
  ``<split dir>_mask/<name>`` as above; the sample's own ``ndisp`` travels as ``n_disp`` (eval.py:173-174).
* ``PairDirectory`` -- demo.py's layout: ``<root>/<name>/im0.png, im1.png [, calib.txt] [, disp0.pfm |
  disp0.png (uint16, x256)]``.

Without stored masks the three per-view detail masks come from decnet_amd.masks.detail_detection (the
reference bakes the same function's output into the pickles); with ``use_detail`` the network ignores them
(SparseDenseNetRefinementMask.py:148-170) and ones are returned.

Training mode (``NpyPairs(..., is_training=True, img_size=...)``, round 4): the reference's training branch --
random crop (masks cropped with it), reflected-light stripes, KITTI's occlusion rectangle / object mask /
RandomPhotometric -- in decnet_amd.augment, drawn from numpy's global generator in the reference's order; the sample
is then the 10-tuple of SceneflowMask.py:195-196.  Middlebury's training branch (its own crop policy,
MiddleburyMask.py:152-237) is not reproduced.
"""
import math
import os
import pickle
import re

import numpy as np
import torch
from torch.utils import data

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def pad_top_left(arr, multiple=27):
    """Zero pad [H,W,C] on the TOP and LEFT up to the next multiple (demo.py:75-81, SceneflowMask.py:118-128)."""
    h, w = arr.shape[:2]
    rh = int(math.ceil(h / multiple) * multiple) - h
    rw = int(math.ceil(w / multiple) * multiple) - w
    out = np.zeros((h + rh, w + rw) + arr.shape[2:], dtype=np.float32)
    out[rh:, rw:] = arr
    return out


def normalise(img01):
    x = (img01.astype(np.float32) - MEAN) / STD
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).float()


def read_pfm(path):
    """utils/utils.py:281-321: 'Pf' / 'PF' header, dims, scale (negative = little endian), rows bottom-up."""
    with open(path, "rb") as f:
        header = f.readline().rstrip().decode("utf-8")
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = map(int, m.groups())
        scale = float(f.readline().rstrip().decode("utf-8"))
        endian = "<" if scale < 0 else ">"
        d = np.frombuffer(f.read(), endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(d.reshape(shape)).astype(np.float32), abs(scale)


def _masks(img01_padded, stored, use_detail):
    """Three masks of one view, coarsest first (stage 1..3), float [h,w]."""
    if stored is not None:                              # fine -> coarse in the pickle (SceneflowMask.py:177-184)
        return [torch.from_numpy(np.asarray(m)).float() for m in stored[::-1]]
    H, W = img01_padded.shape[:2]
    if use_detail:                                      # placeholders: the network makes its own masks
        return [torch.ones(H // s, W // s) for s in (9, 3, 1)]
    from .masks import detail_detection
    return [torch.from_numpy(m.astype(np.float32)) for m in detail_detection(img01_padded)[::-1]]


class _Base(data.Dataset):
    def __init__(self, use_detail=True, max_disp=192):
        self.use_detail, self.n_disp = use_detail, max_disp

    def _item(self, left_u8, right_u8, disp, name, n_disp, lmasks=None, rmasks=None):
        ori_h, ori_w = left_u8.shape[:2]
        lp, rp = pad_top_left(left_u8.astype(np.float32)) / 255, pad_top_left(right_u8.astype(np.float32)) / 255
        gt = torch.from_numpy(pad_top_left(disp.astype(np.float32)))
        lm, rm = _masks(lp, lmasks, self.use_detail), _masks(rp, rmasks, self.use_detail)
        image = torch.from_numpy(np.ascontiguousarray(pad_top_left(left_u8.astype(np.float32)).transpose(2, 0, 1)))
        return (normalise(lp), normalise(rp), gt, image, lm[0], lm[1], lm[2], rm[0], rm[1], rm[2], ori_h, ori_w,
                name, n_disp)


class NpyPairs(_Base):
    """policy: which dataset's training branch ``is_training`` follows ('sceneflow' | 'kitti' | 'drivingstereo')."""

    def __init__(self, root, split="test", is_training=False, img_size=(540, 960), policy="sceneflow", **kw):
        super().__init__(**kw)
        self.is_training, self.img_size, self.policy = is_training, tuple(img_size), policy
        p = os.path.join(root, split)
        if os.path.isfile(p):
            self.paths = sorted(str(s) for s in np.load(p))
        else:
            self.paths = sorted(os.path.join(p, f) for f in os.listdir(p) if f.endswith(".npy"))
        if not self.paths:
            raise Exception("No files under/in {}/{}".format(root, split))

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        path = self.paths[i]
        arr = np.load(path)
        name = os.path.basename(path).split(".")[0]
        d = os.path.dirname(path)
        mpath = os.path.join(d + "_mask", name)
        lm = rm = None
        if os.path.exists(mpath):
            with open(mpath, "rb") as f:
                m = pickle.load(f)
            lm, rm = m[0:3], m[3:6]
        if self.is_training:
            return self._train_item(arr, None if lm is None else list(lm) + list(rm))
        # no per-sample disparity range in this layout: n_disp <= 0 tells eval to keep --max_disp
        return self._item(arr[..., 0:3], arr[..., 3:6], arr[..., 6], name, -1, lm, rm)

    def _train_item(self, arr, stored):
        """SceneflowMask.py:115-196 / KITTI15Mask.py:105-211 with is_training: pad, crop, augment, normalise; the
        10-tuple left, right, disparity, image, left_mask1..3, right_mask1..3."""
        from . import augment
        data = pad_top_left(arr.astype(np.float32))
        left, right, disp, image, stored = augment.prepare_training_sample(data, stored, self.policy, self.img_size)
        lm = _masks(left, None if stored is None else stored[0:3], self.use_detail)
        rm = _masks(right, None if stored is None else stored[3:6], self.use_detail)
        jitter = augment.photometric(self.policy)
        views = []
        for v in (left, right):                                             # left first, then right (KITTI15Mask.py:240-241)
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32).transpose(2, 0, 1)))
            if jitter is not None:
                t = jitter(t)
            views.append(((t - torch.from_numpy(MEAN)[:, None, None]) / torch.from_numpy(STD)[:, None, None]).float())
        img = torch.from_numpy(np.ascontiguousarray(np.asarray(image, np.float32).transpose(2, 0, 1)))
        gt = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32))
        return (views[0], views[1], gt, img, lm[0], lm[1], lm[2], rm[0], rm[1], rm[2])


class MiddleburyPickles(_Base):
    """loader/MiddleburyMask.py:13-131 in test / eval mode (no augmentation)."""
    _SPLITS = {"train_Q": ("MiddEval3Q_processed", "trainingQ"), "eval_Q": ("MiddEval3Q_processed", "trainingQ"),
               "train_H": ("MiddEval3H_processed", "trainingH"), "eval_H": ("MiddEval3H_processed", "trainingH"),
               "train_F": ("MiddEval3F_processed", "trainingF"), "train_AG": ("", "MiddZip_raw_split_dense"),
               "train_allF": ("", "MiddZip_processed"), "eval_allF": ("", "MiddZip_processed"),
               "train_allF_EL": ("", "MiddZip_processed_EL"), "eval_allF_EL": ("", "MiddZip_processed_EL"),
               "train_merge": ("", "MiddMerged"), "test_Q": ("MiddEval3Q_processed", "testQ"),
               "test_H": ("MiddEval3H_processed", "testH"), "test_F": ("MiddEval3F_processed", "testF")}

    def __init__(self, root, split="eval_H", **kw):
        super().__init__(**kw)
        if split not in self._SPLITS:
            raise Exception("Nu such split: {}".format(split))               # MiddleburyMask.py:78
        sub, self.split = self._SPLITS[split]
        self.datapath = os.path.join(root, sub) if sub else root
        d = os.path.join(self.datapath, self.split)
        self.files = sorted(os.listdir(d)) if os.path.isdir(d) else []
        if not self.files:
            raise Exception("No files for ld=[%s] found in %s" % (self.split, self.datapath))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        with open(os.path.join(self.datapath, self.split, self.files[i]), "rb") as f:
            raw = pickle.load(f)
        left, right = np.asarray(raw["im0"]), np.asarray(raw["im1"])
        disp = raw.get("disparity")
        disp = np.zeros(left.shape[:2], np.float32) if disp is None else np.array(disp, dtype=np.float32)
        disp[~np.isfinite(disp)] = 0                                       # MiddleburyMask.py:128
        name = self.files[i].split(".pkl")[0]
        lm = rm = None
        mpath = os.path.join(self.datapath, self.split + "_mask", name)
        if os.path.exists(mpath):
            with open(mpath, "rb") as f:
                m = pickle.load(f)
            lm, rm = m[0:3], m[3:6]
        return self._item(left, right, disp, self.files[i].split(".")[0], int(raw["ndisp"]), lm, rm)


class PairDirectory(_Base):
    def __init__(self, root, **kw):
        super().__init__(**kw)
        self.root = root
        self.names = sorted(n for n in os.listdir(root) if os.path.isfile(os.path.join(root, n, "im0.png")))
        if not self.names:
            raise Exception("No pairs under {}".format(root))

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        from .demo import read_ndisp, read_rgb
        d = os.path.join(self.root, self.names[i])
        left, right = read_rgb(os.path.join(d, "im0.png")), read_rgb(os.path.join(d, "im1.png"))
        if os.path.exists(os.path.join(d, "disp0.pfm")):
            disp = read_pfm(os.path.join(d, "disp0.pfm"))[0]
            disp = np.where(np.isfinite(disp), disp, 0).astype(np.float32)
        elif os.path.exists(os.path.join(d, "disp0.png")):
            from PIL import Image
            disp = np.asarray(Image.open(os.path.join(d, "disp0.png"))).astype(np.float32) / 256
        else:
            disp = np.zeros(left.shape[:2], np.float32)
        n = read_ndisp(os.path.join(d, "calib.txt"))
        return self._item(left, right, disp, self.names[i], n)            # n <= 0: no calib.txt, eval keeps --max_disp


def get_loader(name):
    """loader/__init__.py:8-21: KITTI15 / Sceneflow / DrivingStereo share the pre-baked .npy layout, Middlebury has its
    pickled-dict layout; + 'pairs' (demo.py's directories)."""
    return {"kitti15mask": NpyPairs, "sceneflowmask": NpyPairs, "drivingstereomask": NpyPairs,
            "middleburymask": MiddleburyPickles, "pairs": PairDirectory}[name.lower()]


def training_policy(name):
    """Which training branch a dataset name selects in NpyPairs(is_training=True, policy=...)."""
    return {"kitti15mask": "kitti", "sceneflowmask": "sceneflow", "drivingstereomask": "drivingstereo"}[name.lower()]
