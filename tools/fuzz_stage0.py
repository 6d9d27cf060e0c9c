"""Random stage-0 shapes (B, C, H, W, D) and random CostRegNetNoDown weights: decnet_amd.Stage0 (cost volume + Winograd stack
+ fused last layer / soft-argmax on the MI355X; cost_func cor / ssd / cat by seed) against oracle/stage0.py (the CPU restatement
of submodule.py:479-530, 608-662, 766-777), the tolerances of tests/test_stage0_gpu.py::test_vs_oracle_seeded.
python tools/fuzz_stage0.py [first_seed [n [seconds]]]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_stage0_gpu as t  # noqa: E402
from oracle import stage0 as o0  # noqa: E402
import decnet_amd  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
dev = torch.device("cuda:0")
t0, bad, done, cond = time.time(), 0, 0, 0
for seed in range(first, first + n):
    if time.time() - t0 > budget:
        break
    done += 1
    rng = np.random.RandomState(90000 + seed)
    C = int(rng.choice([8, 20, 24, 54, 72, 216, 216, 216]))
    B = int(rng.randint(1, 4))
    H, W = int(rng.randint(2, 24)), int(rng.randint(2, 40))     # H = 1 / W = 1: the reference's grid is 0 / 0 (submodule.py:498-499);
                                                                # decnet_stage0_forward answers BAD_SHAPE
    D = int(rng.randint(2, 13))
    if C == 216:
        H, W = min(H, 14), min(W, 24)
    cf = ("cor", "ssd", "cat")[seed % 3]                        # every --cost_func of the reference (submodule.py:552-560)
    tag = dict(seed=seed, B=B, C=C, H=H, W=W, D=D, cost_func=cf)
    g = torch.Generator().manual_seed(seed)
    left = torch.relu(torch.randn(B, C, H, W, generator=g))
    right = torch.relu(torch.randn(B, C, H, W, generator=g))
    params = o0.random_params(C, 1000 + seed)
    w_pre = o0.random_w_pre(C, 1000 + seed) if cf == "cat" else None
    with torch.no_grad():
        pred_o, reg_o, _ = o0.stage0_forward(left, right, params, D, cf, w_pre)
        reg = t.load_reg(C, params, dev, cf, w_pre)
        try:
            pred, r = decnet_amd.Stage0(reg)(left.to(dev), right.to(dev), D, return_reg=True)
        except Exception as e:
            bad += 1
            print("RAISED", tag, type(e).__name__, str(e)[:200], flush=True)
            continue
    tol = 2e-4 * max(1.0, float(reg_o.abs().max()))
    e_reg = float((r.cpu() - reg_o).abs().max())
    d_pred = (pred.cpu() - pred_o).abs()
    e_pred, e_mean = float(d_pred.max()), float(d_pred.mean())
    # the soft-argmax is ill conditioned where the regularised volume has two distant near-equal maxima (random weights:
    # |reg| ~ 100): d pred = sum_d p_d (d - pred) d reg_d.  A pixel beyond 1e-3 px passes only if the volume's MEASURED
    # error explains it: |d pred| <= 1e-3 + 2 max|d reg| sum_d p_d |d - pred|   (seed 5864: two maxima 0.037 apart at
    # d = 2 and d = 8, volume off by 1.0e-3 of 178 -> 1.5e-3 px; the torch-CPU oracle is 1.3e-4 px from float64 there)
    p_o = torch.softmax(reg_o, dim=1)
    sens = (p_o * (torch.arange(D, dtype=p_o.dtype).view(1, D, 1, 1) - pred_o.unsqueeze(1)).abs()).sum(1)
    over = d_pred > 1e-3
    pred_ok = bool((d_pred <= 1e-3 + 2.0 * e_reg * sens).all())
    if bool(over.any()) and pred_ok:
        cond += int(over.sum())
    if not (e_reg <= tol and pred_ok and e_mean < 1e-4):
        bad += 1
        print("FAILED", tag, "reg %.3e (tol %.3e) pred max %.3e mean %.3e" % (e_reg, tol, e_pred, e_mean), flush=True)
print("%d stage-0 cases (seeds %d ..), %d failed, %d pixels beyond 1e-3 px explained by the volume's own error at a "
      "two-peaked pixel, %.0f s" % (done, first, bad, cond, time.time() - t0))
sys.exit(1 if bad else 0)
