"""Random Conv2dUnit / Deconv2dUnit shapes through decnet_amd.model.Unit's HIP kernels (conv2d_small, conv2d_f32m, conv2d_mfma,
the stride-3 and transposed kernels, concatenated inputs) against the same layer on torch's own kernels with float64 weights
and inputs.  python tools/fuzz_conv2d.py [first_seed [n [seconds]]]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from decnet_amd.model import Unit  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
dev = torch.device("cuda:0")
t0, bad, done, hip = time.time(), 0, 0, 0
kinds = {}
for seed in range(first, first + n):
    if time.time() - t0 > budget:
        break
    done += 1
    rng = np.random.RandomState(70000 + seed)
    torch.manual_seed(seed)
    mode = rng.choice(["conv", "conv", "conv", "conv_s3", "deconv"])
    B = int(rng.randint(1, 4))
    H, W = int(rng.choice([1, 2, 5, 17, 36, 60, 64, 90])), int(rng.choice([3, 16, 33, 64, 108, 130, 255, 324]))
    cout = int(rng.choice([1, 3, 4, 8, 8, 12, 24, 36, 72, 81]))
    if mode == "conv":
        k = int(rng.choice([1, 3, 3, 3]))
        dil = int(rng.choice([1, 1, 2, 3, 6, 9])) if k == 3 else 1
        nseg = int(rng.choice([1, 1, 1, 2, 3, 5]))
        cins = [int(rng.choice([1, 3, 4, 8, 8, 16, 24, 49, 72])) for _ in range(nseg)]
        u = Unit(sum(cins), cout, k, pad=dil * (k // 2), dil=dil, relu=bool(rng.randint(2)), bn=bool(rng.randint(2)))
    elif mode == "conv_s3":
        cins = [int(rng.choice([3, 8, 24, 72]))]
        u = Unit(cins[0], cout, 3, stride=3, pad=1, relu=bool(rng.randint(2)), bn=bool(rng.randint(2)))
    else:
        cins = [int(rng.choice([8, 24, 72, 216]))]
        cout = int(rng.choice([3, 8, 8, 24, 72]))
        H, W = min(H, 36), min(W, 108)
        u = Unit(cins[0], cout, 3, stride=3, relu=bool(rng.randint(2)), bn=bool(rng.randint(2)), transposed=True)
    if u.bn is not None:
        with torch.no_grad():
            u.bn.running_mean.normal_(0, 0.2); u.bn.running_var.uniform_(0.5, 2.0)
            u.bn.weight.normal_(1, 0.2); u.bn.bias.normal_(0, 0.2)
    u = u.to(dev).eval()
    xs = [torch.randn(B, c, H, W, device=dev) for c in cins]
    tag = dict(seed=seed, mode=mode, B=B, cins=cins, cout=cout, H=H, W=W, k=u.conv.kernel_size[0], dil=u.conv.dilation[0],
               relu=u.relu, bn=u.bn is not None)
    with torch.no_grad():
        arg = tuple(xs) if len(xs) > 1 else xs[0]
        kind = u._hip_kind(arg)
        kinds[kind] = kinds.get(kind, 0) + 1
        y = u(arg)
        ud = Unit.__new__(Unit)
        x64 = torch.cat(xs, 1).double()
        c = u.conv
        if mode == "deconv":
            r = F.conv_transpose2d(x64, c.weight.double(), None if c.bias is None else c.bias.double(), c.stride, c.padding)
        else:
            r = F.conv2d(x64, c.weight.double(), None if c.bias is None else c.bias.double(), c.stride, c.padding, c.dilation)
        if u.bn is not None:
            bn = u.bn
            r = (r - bn.running_mean.double().view(1, -1, 1, 1)) / torch.sqrt(bn.running_var.double().view(1, -1, 1, 1) + bn.eps) \
                * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1)
        if u.relu:
            r = torch.relu(r)
    err = float((y.double() - r).abs().max())
    sc = max(1.0, float(r.abs().max()))
    if y.shape != r.shape or not err < 3e-5 * sc:
        bad += 1
        print("FAILED", tag, "kind", kind, "max err %.3e scale %.2f" % (err, sc), flush=True)
print("%d layers (seeds %d ..), %d failed, %.0f s; kernels chosen: %s" % (done, first, bad, time.time() - t0, kinds))
sys.exit(1 if bad else 0)
