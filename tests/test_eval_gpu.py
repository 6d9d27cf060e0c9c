"""decnet_amd.eval end to end on the GPU: the 'pairs' layout built from a bundled InputData pair, evaluation
mode (EPE / 3-px of modules/loss.py:427-437 against a ground truth made from the model's own output, so the
expected numbers are known) and submission mode (PNG identical to decnet_amd.demo.run_pair).  -m gpu."""
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PAIR = os.path.join(HERE, "golden", "inputdata", "KITTI", "000009_10")


def test_eval_and_submission_modes(tmp_path):
    from PIL import Image
    from decnet_amd import demo
    from decnet_amd import eval as E
    dev = torch.device("cuda:0")
    root = tmp_path / "data"
    for n in ("a", "b"):
        shutil.copytree(PAIR, str(root / n))
    flags = ["--dataset", "pairs", "--data_path", str(root), "--base_channels", "2", "--thold", "0.5",
             "--batch_size", "2", "--skip_stage_id", "4"]
    args = E.build_parser().parse_args(flags + ["--is_eval", "0", "--save2where", str(tmp_path / "out")])
    torch.manual_seed(5)
    model = E.build_model(args, dev)
    assert E.test(args, model=model) is None
    png = np.asarray(Image.open(str(tmp_path / "out" / "a.png")))
    limg, rimg = demo.read_rgb(os.path.join(PAIR, "im0.png")), demo.read_rgb(os.path.join(PAIR, "im1.png"))
    model.max_disp = 216                      # no calib.txt: --max_disp stays (as demo.py keeps it when ndisp <= 0)
    want, _ = demo.run_pair(model, limg, rimg, dev)
    assert png.shape == (375, 1242) and png.dtype == np.uint16
    # batch of 2 vs batch of 1 may pick other MIOpen algorithms for the library-side convolutions, and the
    # untrained refinement stack amplifies that rounding (tests/test_inputdata_gpu.py): counts of 1/256 px
    dcount = np.abs(png.astype(np.int64) - want.astype(np.int64))
    assert dcount.mean() < 5.0 and np.median(dcount) <= 1
    assert np.array_equal(png, np.asarray(Image.open(str(tmp_path / "out" / "b.png"))))
    # evaluation mode: ground truth = that output + 1 px where it is a valid disparity
    gt = np.where(want > 0, want.astype(np.float32) / 256 + 1.0, 0.0)       # clamped (negative) outputs: invalid
    for n in ("a", "b"):
        Image.fromarray(np.clip(gt * 256, 0, 65535).astype(np.uint16)).save(str(root / n / "disp0.png"))
    args = E.build_parser().parse_args(flags + ["--is_eval", "1"])
    epe, l3 = E.test(args, model=model)
    valid = (gt > 0) & (gt < 216)
    assert valid.mean() > 0.05
    assert abs(epe - 1.0) < 0.02 and l3 < 0.5           # every valid pixel is 1 px off: inside the 3-px band


def test_eval_metrics_through_rccl_in_a_world_of_one(tmp_path):
    """--force-collective: init_process_group("nccl") and the metric all-gather really run on the one GPU; same numbers."""
    from PIL import Image
    from decnet_amd import eval as E
    dev = torch.device("cuda:0")
    root = tmp_path / "data"
    for n in ("a", "b", "c"):
        shutil.copytree(PAIR, str(root / n))
        Image.fromarray((np.random.RandomState(3).rand(375, 1242) * 60 * 256).astype(np.uint16)).save(
            str(root / n / "disp0.png"))
    flags = ["--dataset", "pairs", "--data_path", str(root), "--base_channels", "2", "--thold", "0.5",
             "--batch_size", "1", "--skip_stage_id", "4", "--is_eval", "1"]
    torch.manual_seed(5)
    model = E.build_model(E.build_parser().parse_args(flags), dev)
    plain = E.test(E.build_parser().parse_args(flags), model=model)
    forced = E.test(E.build_parser().parse_args(flags + ["--force-collective"]), model=model)
    assert not torch.distributed.is_initialized()
    assert plain is not None and forced is not None
    assert abs(plain[0] - forced[0]) < 1e-9 + 1e-6 * abs(plain[0]) and abs(plain[1] - forced[1]) < 1e-6
