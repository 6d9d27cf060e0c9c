"""Host-side pre/post-processing of the demo counterpart (demo.py:75-98, 149-155, 191-197). CPU."""
import numpy as np
import torch

from decnet_amd import demo


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
