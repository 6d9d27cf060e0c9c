"""Host-side pre/post-processing of the demo counterpart (demo.py:75-98, 149-155, 191-197). CPU."""
import os
import sys

import numpy as np
import torch

from decnet_amd import demo

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_padding_is_top_left_to_multiple_of_27():
    img = np.arange(5 * 7 * 3, dtype=np.float32).reshape(5, 7, 3) + 1
    p = demo.padding(img)
    assert p.shape == (27, 27, 3)
    assert (p[-5:, -7:] == img).all() and p[:22].sum() == 0 and p[:, :20].sum() == 0
    assert demo.padding(np.zeros((540, 960, 3), np.float32)).shape == (540, 972, 3)     # S3
    assert demo.padding(np.zeros((375, 1242, 3), np.float32)).shape == (378, 1242, 3)


def test_transform_and_ndisp(tmp_path):
    x = demo.transform(np.full((2, 3, 3), 0.5, np.float32))
    assert x.shape == (1, 3, 2, 3)
    np.testing.assert_allclose(x[0, :, 0, 0].numpy(), (0.5 - demo.MEAN) / demo.STD, rtol=1e-6)
    c = tmp_path / "calib.txt"
    c.write_text("cam0=[1 0 0]\nndisp=400\n")
    assert demo.read_ndisp(str(c)) == 405                                      # ceil(400/27)*27
    assert demo.read_ndisp(str(tmp_path / "missing.txt")) == -1


def test_uint16_output_and_png_roundtrip(tmp_path):
    pred = torch.tensor([[[-1.0, 0.5, 10.0], [300.0, 255.99, 1.0]]])
    out = demo.disparity_to_uint16(pred, 2, 2)
    assert out.dtype == np.uint16 and out.shape == (2, 2)
    assert out.tolist() == [[128, 2560], [65533, 256]]     # crop bottom-right; 255.99*256 = 65533.4
    full = demo.disparity_to_uint16(pred, 2, 3)
    assert full[0, 0] == 0 and full[1, 0] == 65535         # clamp
    path = str(tmp_path / "d.png")
    demo.write_png16(path, full)
    from PIL import Image
    assert (np.asarray(Image.open(path)) == full).all()


def test_host_detail_masks():
    """decnet_amd.masks.detail_detection (utils/utils.py:483-534 without cv2; parity unpinned): shapes,
    the INTER_LINEAR quirk of the reference's cv2.resize calls, where the detail lands, flat images."""
    import numpy as np
    from decnet_amd.masks import detail_detection, resize_linear, gaussian_blur
    img = np.zeros((54, 81, 3), np.float32)
    img[:, 40:] = 1.0                                         # one vertical edge
    ms = detail_detection(img, scale=3, downsampling_iteration=3, thold=0.3)
    assert [m.shape for m in ms] == [(54, 81), (18, 27), (6, 9)] and all(m.dtype == bool for m in ms)
    cols = np.nonzero(ms[0].any(0))[0]
    assert cols.min() >= 36 and cols.max() <= 43 and ms[0][:, 38:42].all()      # only around the edge
    assert not ms[0][:, :30].any() and not ms[0][:, 50:].any()
    # down by 3 = src[3i+1] (half-pixel-centre bilinear with an integer source position)
    a = np.random.default_rng(0).random((12, 15, 2)).astype(np.float32)
    assert np.array_equal(resize_linear(a, 4, 5), a[1::3, 1::3])
    # up by 3 keeps constants and clamps at the border
    up = resize_linear(a[:2, :2], 6, 6)
    assert np.allclose(up[0, 0], a[0, 0]) and np.allclose(up[-1, -1], a[1, 1])
    assert np.allclose(gaussian_blur(np.full((7, 9, 1), 0.25, np.float32), 5), 0.25, atol=1e-6)
    # flat image: 0/0 -> no detail anywhere (the reference's NaN compares false too)
    assert not any(m.any() for m in detail_detection(np.full((27, 27, 3), 0.5, np.float32)))
    # sizes that are not multiples of 27 get the reference's top/left padding, whose mask part is cleared
    ms = detail_detection(np.random.default_rng(1).random((50, 70, 3)).astype(np.float32))
    assert ms[0].shape == (54, 81) and not ms[0][:4].any() and not ms[0][:, :11].any()


def test_checkpoint_loader_is_strict():
    """demo.py:124-133 merges whatever matches; a checkpoint with foreign key names would silently leave the
    net on its random init.  Here that is an error, and DataParallel's prefix is stripped."""
    import pytest
    from make_golden import E2E_KW
    from decnet_amd.model import get_model, load_reference_checkpoint
    torch.manual_seed(0)
    a, b = get_model(**E2E_KW), get_model(**E2E_KW)
    sd = a.state_dict()
    load_reference_checkpoint(b, {"module." + k: v for k, v in sd.items()})
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in sd.items())
    # tolerated: no num_batches_tracked (old PyTorch), the reference's loss-module keys
    ok = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    ok["train_loss_func.dummy"] = torch.zeros(1)
    load_reference_checkpoint(b, ok)
    with pytest.raises(RuntimeError, match="missing"):
        load_reference_checkpoint(b, {"model." + k: v for k, v in sd.items()})      # foreign prefix: nothing matches
    part = dict(sd)
    del part["cost_regularizer.conv0.0.conv.weight"]
    with pytest.raises(RuntimeError, match="cost_regularizer.conv0.0.conv.weight"):
        load_reference_checkpoint(b, part)
    load_reference_checkpoint(b, part, strict=False)                                # the reference's behaviour


def test_host_detail_mask_primitives_against_scipy_and_torch():
    """cv2 is absent here, so decnet_amd.masks cannot be pinned against the library the reference calls
    (utils/utils.py:447-534).  The two primitives are pinned against two OTHER independent implementations of the same
    published semantics instead: cv2.GaussianBlur(k, sigma=1, BORDER_REFLECT_101) == a separable correlation with
    exp(-x^2/2)/sum and scipy's 'mirror' boundary; cv2.resize(..., INTER_LINEAR) == half-pixel-centre bilinear without
    antialiasing, clamped at the border == torch's bilinear interpolate with align_corners=False."""
    import numpy as np
    import scipy.ndimage as nd
    from decnet_amd.masks import gaussian_blur, resize_linear
    rng = np.random.default_rng(5)
    img = rng.random((31, 44, 3)).astype(np.float32)
    for k in (3, 5, 7):
        x = np.arange(k, dtype=np.float64) - (k - 1) / 2
        g = np.exp(-x * x / 2.0)
        g /= g.sum()
        ref = nd.correlate1d(nd.correlate1d(img.astype(np.float64), g, axis=1, mode="mirror"), g, axis=0, mode="mirror")
        np.testing.assert_allclose(gaussian_blur(img, k), ref, rtol=0, atol=2e-6)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    for (ho, wo) in ((10, 14), (93, 132), (31, 44), (17, 50)):
        ref = torch.nn.functional.interpolate(t, size=(ho, wo), mode="bilinear", align_corners=False, antialias=False)
        np.testing.assert_allclose(resize_linear(img, ho, wo), ref[0].permute(1, 2, 0).numpy(), rtol=0, atol=5e-6)
