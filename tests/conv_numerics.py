"""tests/conv_numerics.py -- how far each Conv3d algorithm of the HIP stage-0 path is from the exact
result, next to the fp32 CPU oracle's own rounding error.  A measurement script, not a pytest module; it
lives under tests/ because it uses oracle/ as the checker.

Ground truth = the oracle (oracle/stage0.py) evaluated in float64.  For each `sharpness` s the last
BatchNorm's gamma/beta are multiplied by s, which scales the logits of the soft-argmax the way a
trained, confident network's are (random-init logits are almost flat).

    python tests/conv_numerics.py [--shape 8,20,36] [--c 216] [--sharp 1,10,100]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import stage0 as o0  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="8,20,36")
    ap.add_argument("--c", type=int, default=216)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--sharp", default="1,10,100")
    a = ap.parse_args()
    D, H, W = map(int, a.shape.split(","))
    C, B = a.c, a.batch
    import decnet_amd
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    left = torch.relu(torch.randn(B, C, H, W, generator=g))
    right = torch.relu(torch.randn(B, C, H, W, generator=g))
    print("%-10s %-10s %12s %12s %12s %12s" % ("sharpness", "path", "reg rel max", "pred mean", "pred max",
                                                 "logit range"))
    for s in map(float, a.sharp.split(",")):
        params = o0.random_params(C, 99)
        gm, bt, mu, var = params[7]["bn"]
        params[7]["bn"] = (gm * s, bt * s, mu, var)
        p64 = [{"w": p["w"].double(), "bn": tuple(t.double() for t in p["bn"])} for p in params]
        with torch.no_grad():
            pred64, reg64, _ = o0.stage0_forward(left.double(), right.double(), p64, D)
            pred32, reg32, _ = o0.stage0_forward(left, right, params, D)
        rng = float((reg64.max(1).values - reg64.min(1).values).mean())
        rows = [("oracle32", pred32, reg32)]
        reg = decnet_amd.CostRegNetNoDown(in_channels=C, base_channels=2 * C, cost_func="cor")
        for u, p in zip(reg.units(), params):
            u.conv.weight.data = p["w"].clone()
            u.bn.weight.data, u.bn.bias.data = p["bn"][0].clone(), p["bn"][1].clone()
            u.bn.running_mean.data, u.bn.running_var.data = p["bn"][2].clone(), p["bn"][3].clone()
        reg = reg.to(dev).eval()
        for algo in ("direct", "winograd", "winograd4", "winograd444"):
            os.environ["DECNET_CONV_ALGO"] = algo
            with torch.no_grad():
                pred, r = decnet_amd.Stage0(reg)(left.to(dev), right.to(dev), D, return_reg=True)
            rows.append((algo, pred.cpu(), r.cpu()))
        for name, pred, r in rows:
            print("%-10g %-10s %12.3e %12.3e %12.3e %12.3f" % (
                s, name, float((r.double() - reg64).abs().max() / reg64.abs().max()),
                float((pred.double() - pred64).abs().mean()), float((pred.double() - pred64).abs().max()), rng))


if __name__ == "__main__":
    main()
