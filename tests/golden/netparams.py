"""Deterministic synthetic parameters for an e2e golden: every tensor of a state_dict is filled
from its own seeded generator (key order independent), so the generator script and the test
build bit-identical weights without storing 5 MB of them."""
import zlib

import torch


def fill_state_dict(sd, seed=1234):
    out = {}
    for k, v in sd.items():
        g = torch.Generator().manual_seed(seed + (zlib.crc32(k.encode()) & 0x7FFFFFF))
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        elif k.endswith("bn.weight"):
            out[k] = torch.rand(v.shape, generator=g) * 0.4 + 0.8
        elif k.endswith("bias"):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        else:                                   # conv / deconv weights: He-style scale
            fan = v[0].numel() if v.dim() > 1 else v.numel()
            out[k] = torch.randn(v.shape, generator=g) * (2.0 / max(fan, 1)) ** 0.5
    return out
