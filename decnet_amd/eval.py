#!/usr/bin/env python
"""Dataset evaluation / submission -- this repo's counterpart of the reference's eval.py.

    python -m decnet_amd.eval --arch sparsedensenetrefinementmask --dataset sceneflowmask --data_path D \\
        --test_split test --batch_size 8 --is_eval 1 --resume ckpt.pkl [--gpus 8]

Same flags and flow as eval.py:34-101, 114-228: build the network, load ``--resume`` (DataParallel prefix
stripped), run every test batch under ``no_grad``; with ``--is_eval`` report the end-point error and the
3-px / 5 % error of ``test_loss_func`` (modules/loss.py:427-437) per batch and their means, otherwise write
``name.png`` = uint16(disp * 256) cropped bottom-right to the original size (eval.py:197-206).

Multi-GPU: the reference wraps the model in one-process ``torch.nn.DataParallel`` (eval.py:145-146), which
scatters every batch and re-broadcasts the 52.7 MB of weights on every forward.  Here ``--gpus N`` starts N
processes (one per GPU, RCCL); rank r evaluates the contiguous shard ``decnet_amd.dist.shard_range`` of the
batches with weights resident, and the per-batch metrics are gathered once at the end.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

from .demo import disparity_to_uint16, write_png16
from .dist import shard_range
from .model import get_model, load_reference_checkpoint


def test_loss_func(pred, gt, max_disp):
    """modules/loss.py:427-437: over pixels with 0 < gt < max_disp -- EPE = mean |pred - gt|;
    loss_3 = 100 - % of pixels with |err| < 3 px or |err| < 5 % of gt."""
    assert tuple(pred.shape[-2:]) == tuple(gt.shape[-2:]), \
        "the size of predcited disparity map is not equal to the size of groundtruth."
    mask = (gt < max_disp) & (gt > 0)
    err = torch.abs(pred[mask] - gt[mask])
    good = ((err < 3) | (err < 0.05 * gt[mask])).to(torch.float32)
    loss_3 = 100 - torch.sum(good) / torch.sum(mask) * 100
    epe = torch.mean(err)
    return epe, loss_3


def build_parser():
    p = argparse.ArgumentParser(description="DecNet evaluation on MI355X")
    p.add_argument("--seed", type=int, default=7)
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
    p.add_argument("--skip_stage_id", type=int, default=100)
    p.add_argument("--use_detail", type=int, default=1)
    p.add_argument("--is_eval", type=int, default=0, help="1: report EPE / 3-px error; 0: write submission PNGs")
    p.add_argument("--thold", type=float, default=0.9)
    p.add_argument("--dataset", default="sceneflowmask",
                   help="kitti15mask | sceneflowmask | drivingstereomask | middleburymask (pre-baked .npy layout) "
                        "| pairs (demo.py directory layout)")
    p.add_argument("--data_path", default=None, help="dataset root (else config.json[dataset]['data_path'])")
    p.add_argument("--test_split", default="test")
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--resume", default=None)
    p.add_argument("--save2where", default="./Log/FirstTry")
    p.add_argument("--gpus", type=int, default=1, help="processes (one per GPU); > 1 without a launcher starts them")
    p.add_argument("--force-collective", action="store_true",
                   help="initialise RCCL and gather the metrics through it even with one rank (the N > 1 path on a "
                        "one-GPU box; same result)")
    return p


def data_path(args):
    if args.data_path:
        return args.data_path
    import json
    with open("config.json") as f:                                     # loader/__init__.py:24-30
        return json.load(f)[args.dataset.lower()]["data_path"]


def build_model(args, device):
    model = get_model(name=args.arch, max_disp=args.max_disp, base_channels=args.base_channels,
                      cost_func=args.cost_func, grad_method=args.grad_method, num_stage=args.num_stage,
                      down_scale=args.down_scale, step=list(map(float, args.step.split(","))),
                      samp_num=list(map(float, args.samp_num.split(","))),
                      sample_spa_size_list=list(map(int, args.sample_spa_size_list.split(","))),
                      down_func_name=args.down_func_name, weights=list(map(float, args.loss_weights.split(","))),
                      if_overmask=False, skip_stage_id=args.skip_stage_id, use_detail=bool(args.use_detail),
                      thold=args.thold)
    if args.resume is not None:
        if not os.path.isfile(args.resume):
            raise Exception("No such model file, please check it: {}".format(args.resume))
        load_reference_checkpoint(model, torch.load(args.resume, map_location="cpu")["model_state"])
    else:
        print("From scratch!")
    return model.to(device).eval()


def batches_of(dataset, batch_size, dataset_name=""):
    """Consecutive batches of equal-shape samples (DataLoader(shuffle=False) order, eval.py:122).
    MiddleburyMask runs at batch size 1 whatever --batch_size says: every sample carries its own disparity range
    (eval.py:173-175 takes ``int(n_disp)`` of a one-element tensor) and image size, so a larger batch could only fail
    in ``batch_max_disp`` / ``collate`` after earlier batches had been computed.  Decided here, before any work."""
    if dataset_name.lower() == "middleburymask" and batch_size != 1:
        print("MiddleburyMask: per-sample disparity ranges -> batch size 1 (asked for %d)" % batch_size)
        batch_size = 1
    idx = list(range(len(dataset)))
    return [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]


def collate(samples):
    cols = list(zip(*samples))
    out = []
    for c in cols:
        out.append(torch.stack(c) if isinstance(c[0], torch.Tensor) else list(c))
    return out


def test(args, model=None):
    from .loader import get_loader
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("decnet_amd.eval needs an MI355X (no CPU path)")
    torch.manual_seed(17)                                               # eval.py:106
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    coll = world > 1 or getattr(args, "force_collective", False)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                             # --force-collective without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.distributed.init_process_group("nccl", device_id=device)
    kind = get_loader(args.dataset)
    kw = dict(use_detail=bool(args.use_detail), max_disp=192)
    dataset = kind(data_path(args), **kw) if args.dataset.lower() == "pairs" else \
        kind(data_path(args), split=args.test_split, **kw)
    if not args.is_eval and rank == 0:
        os.makedirs(args.save2where, exist_ok=True)
    if model is None:
        model = build_model(args, device)
    batches = batches_of(dataset, args.batch_size, args.dataset)
    lo, hi = shard_range(len(batches), rank, world)
    rec = []                                                            # (batch index, epe, loss_3)
    for bi in range(lo, hi):
        (left, right, disparity, _image, lm1, lm2, lm3, rm1, rm2, rm3, ori_h, ori_w, names, n_disp) = collate(
            [dataset[i] for i in batches[bi]])
        with torch.no_grad():
            model.max_disp = batch_max_disp(args.dataset, n_disp, args.max_disp)
            left, right, disparity = left.to(device), right.to(device), disparity.to(device)
            lms, rms = [m.to(device) for m in (lm1, lm2, lm3)], [m.to(device) for m in (rm1, rm2, rm3)]
            torch.cuda.synchronize()
            t0 = time.time()
            pred = model(left, right, disparity, lms, rms, is_check=False, is_eval=bool(args.is_eval))[-1]
            torch.cuda.synchronize()
            dt = time.time() - t0
            if args.is_eval:
                epe, loss_3 = test_loss_func(pred, disparity, model.max_disp)
                rec.append((bi, float(epe.mean().item()), float(loss_3.mean().item())))
                print("[{}/{}]   evaluation cost time: {} - epe: {} - loss3: {}".format(bi, len(batches), dt, rec[-1][1],
                                                                                      rec[-1][2]))
            else:
                for j, name in enumerate(names):
                    write_png16(os.path.join(args.save2where, name + ".png"),
                                disparity_to_uint16(pred[j:j + 1], int(ori_h[j]), int(ori_w[j])))
                print("[{}/{}]   submission cost time: {}".format(bi, len(batches), dt))
    result = None
    if coll:
        # the per-batch metrics of every rank, once, as one all-gather of a fixed-size tensor over RCCL
        nb = len(batches)
        mine = torch.zeros((nb, 3), device=device, dtype=torch.float64)    # (epe, loss_3, owned): a NaN metric stays a NaN
        for bi, epe, l3 in rec:
            mine[bi, 0], mine[bi, 1], mine[bi, 2] = epe, l3, 1.0
        allm = torch.empty((world, nb, 3), device=device, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(allm, mine)
        allm = allm.cpu()
        owner = allm[..., 2].argmax(0)                                      # shard_range: exactly one owner per batch
        have = allm[..., 2].sum(0) > 0
        rec = [(bi, float(allm[owner[bi], bi, 0]), float(allm[owner[bi], bi, 1])) for bi in range(nb) if bool(have[bi])]
    if args.is_eval and rec:
        result = (float(np.mean([r[1] for r in rec])), float(np.mean([r[2] for r in rec])))
        if rank == 0:
            print("epe: {}, loss_3: {}".format(result[0], result[1]))
    if rank == 0:
        print("The testing is completed: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time()))))
    if coll:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return result


def batch_max_disp(dataset, n_disp, default):
    """The disparity range a batch runs with.
    * MiddleburyMask: eval.py:173-175 sets ``model.max_disp = int(n_disp)`` UNROUNDED (the per-stage ranges are then the
      floor divisions of SparseDenseNetRefinementMask.py:124 and the metric mask is gt < n_disp, modules/loss.py:427-437).
      The reference runs that dataset at batch size 1 (``int()`` of a one-element tensor); a batch whose samples disagree
      is refused here instead of silently taking one of them.
    * 'pairs' (demo.py's directory layout): a calib.txt range rounded up to a multiple of 27, demo.py:149-155.
    * every other dataset: --max_disp (eval.py never touches it).
    Ranges above 272 at full resolution (stage-3 bands wider than 18 tiles) leave the matrix-core SpaMat kernels for the
    row-tile fallback (`spamat_rowtile.hip`): same results, several times slower -- Middlebury at full resolution."""
    ds = dataset.lower()
    nds = sorted({int(v) for v in n_disp})
    if ds == "middleburymask":
        if len(nds) != 1:
            raise ValueError("MiddleburyMask batch with mixed n_disp %s: the reference evaluates it at --batch_size 1" % nds)
        return nds[0] if nds[0] > 0 else int(default)
    if ds == "pairs" and nds[-1] > 0:
        return int(math.ceil(nds[-1] / 27.0) * 27)
    return int(default)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU, started before anything here touches HIP (never an exec after GPU init)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
             "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "decnet_amd.eval"] +
            list(sys.argv[1:] if argv is None else argv), env=env))
    test(args)


if __name__ == "__main__":
    main()
