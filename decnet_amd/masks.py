"""Host-side detail masks for `use_detail=0` (reference: utils/utils.py:483-534 `detailDetection`, with
`GaussianDown` :447-463 and `GaussianUp` :465-479): a Laplacian-pyramid residual per level, summed over
channels, min-max normalised and thresholded.  Not on the GPU hot path (the shipped scripts all run with
--use_detail=1 and never call it); numpy only, no cv2.

The reference's arithmetic lives in OpenCV, which is absent here, so this restatement follows the
published semantics of the three calls it makes and is *parity unpinned* against cv2 itself (the two primitives are
pinned against scipy.ndimage and torch's bilinear interpolate, two independent implementations of those semantics:
tests/test_demo_cpu.py):
  * cv2.GaussianBlur(img, (k,k), 1): separable kernel exp(-x^2/2)/sum, BORDER_REFLECT_101;
  * cv2.resize(img, dsize, cv2.INTER_AREA): the third positional argument of cv2.resize is `dst`, not
    the interpolation flag, so the reference actually resizes with the default INTER_LINEAR --
    half-pixel centres, no antialiasing, clamped at the border.  Down by an exact factor 3 that is the
    plain subsampling src[3i+1]; up by 3 it is a two-tap lerp.
"""
import numpy as np


def _gauss_kernel(k, sigma=1.0):
    x = np.arange(k, dtype=np.float64) - (k - 1) / 2
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (g / g.sum()).astype(np.float32)


def _blur_axis(a, g, axis):
    r = len(g) // 2
    pad = [(0, 0)] * a.ndim
    pad[axis] = (r, r)
    p = np.pad(a, pad, mode="reflect")                   # numpy "reflect" == BORDER_REFLECT_101
    out = np.zeros_like(a)
    n = a.shape[axis]
    for t, w in enumerate(g):
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(t, t + n)
        out += w * p[tuple(sl)]
    return out


def gaussian_blur(img, k):
    g = _gauss_kernel(k)
    return _blur_axis(_blur_axis(img.astype(np.float32), g, 1), g, 0)


def _resize_axis_linear(a, n_out, axis):
    n_in = a.shape[axis]
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    i0 = np.floor(src).astype(np.int64)
    f = (src - i0).astype(np.float32)
    lo = i0 < 0
    i0[lo], f[lo] = 0, 0.0
    hi = i0 >= n_in - 1
    i0[hi], f[hi] = n_in - 1, 0.0
    i1 = np.minimum(i0 + 1, n_in - 1)
    shp = [1] * a.ndim
    shp[axis] = n_out
    f = f.reshape(shp)
    return np.take(a, i0, axis=axis) * (1.0 - f) + np.take(a, i1, axis=axis) * f


def resize_linear(img, h_out, w_out):
    return _resize_axis_linear(_resize_axis_linear(img.astype(np.float32), w_out, 1), h_out, 0).astype(np.float32)


def detail_detection(img, scale=3, downsampling_iteration=3, thold=0.3):
    """img: HxWxC float image in [0,1] (already padded as demo.py:75-81 does).  Returns the list of boolean
    masks [H x W, H/scale x W/scale, ...] of utils/utils.py:483-534 (finest first; demo.py:166-167 feeds
    them to the model in reverse order)."""
    h, w, c = img.shape
    interval = scale ** downsampling_iteration
    rh = (interval - h % interval) % interval
    rw = (interval - w % interval) % interval
    data = np.zeros((h + rh, w + rw, c), dtype=np.float32)        # top/left padding, :487-494
    data[rh:, rw:] = img
    data[:rh, rw:] = img[:1]
    data[rh:, :rw] = img[:, :1]
    masks = []
    for i in range(downsampling_iteration):
        blurred = gaussian_blur(data, 3)                                                  # GaussianDown
        down = resize_linear(blurred, blurred.shape[0] // scale, blurred.shape[1] // scale)
        up = gaussian_blur(resize_linear(down, down.shape[0] * scale, down.shape[1] * scale), 5)   # GaussianUp
        if up.shape != data.shape:
            up = resize_linear(up, data.shape[0], data.shape[1])
        residual = np.abs(data - up).sum(axis=2)
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (residual - residual.min()) / (residual.max() - residual.min())
        m = t >= thold                                                                    # NaN (flat image) -> False
        m[:rh // scale ** i, :] = False
        m[:, :rw // scale ** i] = False
        masks.append(m)
        data = down
    return masks
