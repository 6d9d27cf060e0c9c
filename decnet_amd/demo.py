#!/usr/bin/env python
"""Inference on a directory of stereo pairs -- this repo's counterpart of the reference's demo.py.

    python -m decnet_amd.demo --root ./InputData/Sceneflow --save2where ./out [--resume ckpt.pkl]

Same flags, pre- and post-processing as demo.py:22-67, 75-98, 140-198: every sub-directory
``name/`` holding ``im0.png`` / ``im1.png`` (and optionally ``calib.txt`` whose last line ``ndisp=N``
sets max_disp = ceil(N/27)*27) is padded on the top/left to multiples of 27, scaled to [0,1],
normalised with the ImageNet statistics, run through the network, and the disparity is written as
``name.png`` = uint16(disp * 256) cropped back to the original size (bottom-right).
Images are read / written with PIL (the reference uses cv2.imread + BGR2RGB: the same RGB array).
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from .model import get_model, load_reference_checkpoint

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def padding(img, multiple=27):
    """demo.py:75-81: zero pad on the TOP and LEFT up to the next multiple of 27."""
    h, w, c = img.shape
    rh = int(math.ceil(h / multiple) * multiple) - h
    rw = int(math.ceil(w / multiple) * multiple) - w
    out = np.zeros((h + rh, w + rw, c), dtype=np.float32)
    out[rh:, rw:] = img
    return out


def transform(img01):
    """demo.py:83-89: HWC [0,1] -> normalised 1x3xHxW float tensor."""
    x = (img01.astype(np.float32) - MEAN) / STD
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).float().unsqueeze(0)


def read_ndisp(calib_path):
    """demo.py:149-155: last line 'ndisp=N' -> ceil(N/27)*27, or -1."""
    if not os.path.exists(calib_path):
        return -1
    with open(calib_path) as f:
        lines = f.readlines()
    return int(math.ceil(float(lines[-1].strip().split("=")[-1]) / 27) * 27)


def disparity_to_uint16(pred, ori_h, ori_w):
    """demo.py:191-197: x256, clamp to [0, 65535], uint16, crop the bottom-right ori_h x ori_w."""
    out = (pred * 256).clamp(0, 65535)
    return out.detach().cpu().numpy().astype("uint16")[0, -ori_h:, -ori_w:]


def write_png16(path, arr):
    from PIL import Image
    Image.fromarray(arr.astype(np.uint16)).save(path)


def read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def build_parser():
    p = argparse.ArgumentParser(description="DecNet inference on MI355X")
    p.add_argument("--arch", default="sparsedensenetrefinementmask")
    p.add_argument("--max_disp", type=int, default=216)
    p.add_argument("--base_channels", type=int, default=8)
    p.add_argument("--cost_func", default="ssd", help="ssd | cor | cat (demo.py:31; demo.sh and eval.sh pass cor)")
    p.add_argument("--grad_method", default="detach")
    p.add_argument("--num_stage", type=int, default=4)
    p.add_argument("--down_scale", type=int, default=3)
    p.add_argument("--step", default="-1,1,1,1")
    p.add_argument("--samp_num", default="-1,12,10,6")
    p.add_argument("--sample_spa_size_list", default="-1,3,5,7")
    p.add_argument("--down_func_name", default="bicubic")
    p.add_argument("--loss_weights", default="1,1,1,1")
    p.add_argument("--skip_stage_id", type=int, default=4)
    p.add_argument("--use_detail", type=int, default=1)
    p.add_argument("--thold", type=float, default=0.9)
    p.add_argument("--seed", type=int, default=17)
    p.add_argument("--root", default="./InputData/Sceneflow")
    p.add_argument("--resume", default=None)
    p.add_argument("--save2where", default="./Log/FirstTry")
    return p


def build_model(args, device):
    model = get_model(name=args.arch, max_disp=args.max_disp, base_channels=args.base_channels,
                      cost_func=args.cost_func, grad_method=args.grad_method, num_stage=args.num_stage,
                      down_scale=args.down_scale, step=list(map(float, args.step.split(","))),
                      samp_num=list(map(float, args.samp_num.split(","))),
                      sample_spa_size_list=list(map(int, args.sample_spa_size_list.split(","))),
                      down_func_name=args.down_func_name,
                      weights=list(map(float, args.loss_weights.split(","))), if_overmask=False,
                      skip_stage_id=args.skip_stage_id, use_detail=bool(args.use_detail), thold=args.thold)
    if args.resume is not None:
        if not os.path.isfile(args.resume):
            raise Exception("No such model file, please check it: {}".format(args.resume))
        ckpt = torch.load(args.resume, map_location="cpu")
        load_reference_checkpoint(model, ckpt["model_state"])
    else:
        print("From scratch!")
    return model.to(device).eval()


def host_masks(padded_img, device):
    """demo.py:161-167: the three detailDetection masks of one view as [1,H,W] float tensors, coarsest
    first (stage 1 .. 3).  Used only with --use_detail=0."""
    from .masks import detail_detection
    ms = detail_detection(padded_img, scale=3, downsampling_iteration=3, thold=0.3)[::-1]
    return [torch.from_numpy(m.astype(np.float32))[None].to(device) for m in ms]


def run_pair(model, left_img, right_img, device, n_disp=-1):
    """One pair of HxWx3 uint8 RGB arrays -> (uint16 disparity image, seconds)."""
    ori_h, ori_w, _ = left_img.shape
    lp, rp = padding(left_img) / 255, padding(right_img) / 255
    left, right = transform(lp).to(device), transform(rp).to(device)
    lm = rm = None
    if not getattr(model, "use_detail", True):
        lm, rm = host_masks(lp, device), host_masks(rp, device)
    with torch.no_grad():
        if n_disp > 0:
            model.max_disp = int(n_disp)
        torch.cuda.synchronize()
        t0 = time.time()
        pred = model(left, right, None, lm, rm)[-1]
        torch.cuda.synchronize()
        dt = time.time() - t0
    return disparity_to_uint16(pred, ori_h, ori_w), dt


def test(args):
    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("decnet_amd.demo needs an MI355X (no CPU path)")
    device = torch.device("cuda:0")
    os.makedirs(args.save2where, exist_ok=True)
    model = build_model(args, device)
    for name in sorted(os.listdir(args.root)):
        d = os.path.join(args.root, name)
        if not os.path.isdir(d):
            continue
        left_img, right_img = read_rgb(os.path.join(d, "im0.png")), read_rgb(os.path.join(d, "im1.png"))
        img, dt = run_pair(model, left_img, right_img, device, read_ndisp(os.path.join(d, "calib.txt")))
        write_png16(os.path.join(args.save2where, name + ".png"), img)
        print("rebuild version, cost time: {}".format(dt))
    print("The testing is completed: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time()))))


if __name__ == "__main__":
    test(build_parser().parse_args())
