"""Training-time sample preparation of the reference's datasets, host side, numpy + torch only.

What the reference does between ``np.load`` and the returned tuple when ``is_training`` is set:

* ``random_crop``         loader/SceneflowMask.py:131-141, 170-175 (= KITTI15Mask.py:125-133, 179-184): a crop of
                          ``img_size`` rounded up to a multiple of 27 at a uniformly drawn corner; the six stored detail
                          masks are cropped at the corner divided by their scale.
* ``stripe_noise_grey``   SceneflowMask.add_paralex_noise (:255-283): a vertical Gaussian stripe ("reflected light") added
                          to the right view and, shifted by a drawn disparity, to the left view; p = 0.5 (:148-150).
* ``stripe_noise_colour`` KITTI15Mask.add_paralex_noise (:256-304): the same with per-channel gains 400 / 300 / 500 and a
                          sheared column position; drawn with p = 0.8 and again with p = 0.5 (:140-145).
* ``occlude_right``       KITTI15Mask.py:150-157: a rectangle of the right view replaced by the view's mean colour, p = 0.5.
* ``RandomPhotometric``   KITTI15Mask.py:307-366: contrast / brightness / colour / gamma jitter, drawn per view.

Every function draws from numpy's GLOBAL legacy generator in the reference's own call order, so that after
``np.random.seed(s)`` the outputs equal the reference's for the same seed (tests/test_loader_train_cpu.py checks the two
``add_paralex_noise`` variants and ``RandomPhotometric`` against fixtures recorded from the reference's own functions).
"""
import numpy as np
import torch


def crop_size(img_size, interval=27):
    th, tw = img_size
    return int(np.ceil(th / interval) * interval), int(np.ceil(tw / interval) * interval)


def random_crop(data, masks, img_size, scale=3, interval=27):
    """data [H,W,C] (already padded to multiples of ``interval``), masks: list of six arrays (left fine->coarse, right
    fine->coarse) or None -> (cropped data, cropped masks, (x1, y1)).  Draws: randint(0, h-th+1), randint(0, w-tw+1),
    only when the crop differs from the padded size (SceneflowMask.py:136-141)."""
    h, w = data.shape[:2]
    th, tw = crop_size(img_size, interval)
    if (th, tw) == (h, w):
        return data, masks, (0, 0)
    x1 = np.random.randint(0, h - th + 1)
    y1 = np.random.randint(0, w - tw + 1)
    data = data[x1:x1 + th, y1:y1 + tw, :]
    if masks is not None:
        out = []
        for idx, m in enumerate(masks):
            ds = scale ** (idx % 3)                                                   # :173-174
            out.append(m[x1 // ds:(x1 + th) // ds, y1 // ds:(y1 + tw) // ds])
        masks = out
    return data, masks, (x1, y1)


def _stripe_geometry(h, w):
    sel_h = np.random.randint(100, 180)
    sel_w = np.random.randint(30, 70)
    parallel_d = np.random.randint(60, 200)
    sta_h = int(np.random.uniform(0, h - sel_h))
    sta_w = int(np.random.uniform(0, w - sel_w - parallel_d))
    return sel_h, sel_w, parallel_d, sta_h, sta_w


def _gauss(sel_w, gain):
    x = np.arange(sel_w)
    u, sig = sel_w // 2, 7
    return np.exp(-(x - u) ** 2 / (2 * sig ** 2)) / (np.sqrt(2 * np.pi) * sig) * gain


def stripe_noise_grey(left, right):
    """SceneflowMask.add_paralex_noise: images [H,W,3] in 0..255 -> noisy copies (left, right)."""
    h, w, _ = left.shape
    sel_h, sel_w, pd, sta_h, sta_w = _stripe_geometry(h, w)
    noise = _gauss(sel_w, 400) * np.random.uniform(0.7, 1.2)
    noise = np.repeat(np.repeat(noise[np.newaxis], sel_h, axis=0)[..., np.newaxis], 3, axis=-1)
    r = right.copy()
    r[sta_h:sta_h + sel_h, sta_w:sta_w + sel_w] = r[sta_h:sta_h + sel_h, sta_w:sta_w + sel_w] + noise
    r[r > 255] = 255.
    l = left.copy()
    l[sta_h:sta_h + sel_h, sta_w + pd:sta_w + sel_w + pd] = l[sta_h:sta_h + sel_h, sta_w + pd:sta_w + sel_w + pd] + noise
    l[l > 255] = 255.
    return l, r


def stripe_noise_colour(left, right):
    """KITTI15Mask.add_paralex_noise: per-channel gains, column position sheared by a drawn step per row."""
    h, w, _ = left.shape
    sel_h, sel_w, pd, sta_h, sta_w = _stripe_geometry(h, w)
    noise = np.stack([np.repeat(_gauss(sel_w, g)[np.newaxis], sel_h, axis=0) for g in (400, 300, 500)], axis=-1)
    noise = noise.reshape(-1, 3)
    pos_h = np.repeat(np.arange(sta_h, sta_h + sel_h), sel_w)
    pos_w = np.repeat(np.arange(sta_w, sta_w + sel_w)[..., np.newaxis], sel_h, axis=1).transpose().reshape(-1)
    step = np.random.rand() * 0.3
    shift = ((np.arange(sel_h) - sel_h // 2) * step).astype(int)          # (the reference spells it np.int)
    pos_w = np.clip(pos_w + np.repeat(shift[..., np.newaxis], sel_w, axis=1).reshape(-1), a_min=0, a_max=w - pd - 1)
    r = right.copy()
    r[pos_h, pos_w] = r[pos_h, pos_w] + noise
    r[r > 255] = 255.
    l = left.copy()
    l[pos_h, pos_w + pd] = l[pos_h, pos_w + pd] + noise
    l[l > 255] = 255.
    return l, r


def occlude_right(right01):
    """KITTI15Mask.py:152-157, in place on the [0,1] right view (the caller has drawn binomial(1, 0.5))."""
    sh = int(np.random.uniform(30, 80))
    sw = int(np.random.uniform(10, 80))
    ch = int(np.random.uniform(sh, right01.shape[0] - sh))
    cw = int(np.random.uniform(sw, right01.shape[1] - sw))
    right01[ch - sh:ch + sh, cw - sw:cw + sw] = np.mean(np.mean(right01, 0), 0)[np.newaxis, np.newaxis]
    return right01


class RandomPhotometric:
    """KITTI15Mask.py:307-366 on one [3,H,W] tensor in [0,1]: (im (contrast + 1) + brightness) colour, clamp, ^(1/gamma),
    + noise.  KITTI's training transform uses noise 0, contrast +-0.37, brightness sigma 0.02, colour 0.9-1.1, gamma
    0.7-1.7 (:226-236)."""

    def __init__(self, noise_stddev=0.0, min_contrast=-0.37, max_contrast=0.37, brightness_stddev=0.02, min_color=0.9,
                 max_color=1.1, min_gamma=0.7, max_gamma=1.7):
        self.noise_stddev, self.brightness_stddev = noise_stddev, brightness_stddev
        self.min_contrast, self.max_contrast = min_contrast, max_contrast
        self.min_color, self.max_color = min_color, max_color
        self.min_gamma, self.max_gamma = min_gamma, max_gamma

    def __call__(self, im):
        contrast = np.random.uniform(self.min_contrast, self.max_contrast)
        gamma_inv = 1.0 / np.random.uniform(self.min_gamma, self.max_gamma)
        color = torch.from_numpy(np.random.uniform(self.min_color, self.max_color, (3))).float()
        noise = np.random.normal(scale=self.noise_stddev) if self.noise_stddev > 0.0 else 0
        brightness = np.random.normal(scale=self.brightness_stddev) if self.brightness_stddev > 0.0 else 0
        x = im.permute(1, 2, 0)
        x = (x * (contrast + 1.0) + brightness) * color
        x = torch.pow(torch.clamp(x, min=0.0, max=1.0), gamma_inv)
        x += noise
        return x.permute(2, 0, 1)


def photometric(policy):
    """The per-view jitter between ToTensor and Normalize: KITTI / DrivingStereo only (SceneflowMask.transform has it
    commented out, :211-240)."""
    return RandomPhotometric() if policy != "sceneflow" else None


def prepare_training_sample(data, masks, policy, img_size, scale=3, interval=27):
    """The whole training branch of ``__getitem__`` after the top/left padding, in the reference's draw order.
    data [H,W,7|8] float32 (left RGB, right RGB, disparity [, object mask]); policy 'sceneflow' | 'kitti' |
    'drivingstereo' (DrivingStereoMask.py:115-150: the grey stripe with p = 0.5, no occlusion, the object mask).
    -> left01, right01 [h,w,3] in [0,1], disparity [h,w], image (the un-noised left crop, 0..255), masks."""
    data, masks, _ = random_crop(data, masks, img_size, scale, interval)
    left, right, disparity = data[..., 0:3], data[..., 3:6], data[..., 6]
    if policy in ("sceneflow", "drivingstereo"):
        if np.random.binomial(1, 0.5):                                                  # SceneflowMask.py:148-150
            left, right = stripe_noise_grey(left, right)
    else:
        if np.random.binomial(1, 0.8):                                                  # KITTI15Mask.py:142-145
            left, right = stripe_noise_colour(left, right)
        if np.random.binomial(1, 0.5):
            left, right = stripe_noise_colour(left, right)
    left, right = left / 255, right / 255
    if policy == "kitti" and np.random.binomial(1, 0.5):                                # KITTI15Mask.py:152
        right = occlude_right(right)
    if policy != "sceneflow" and data.shape[-1] == 8 and np.random.rand() < 0.3:       # :160-162
        disparity = disparity * data[..., 7]
    return left, right, disparity, data[..., 0:3], masks
