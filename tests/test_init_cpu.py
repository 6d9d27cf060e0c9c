"""From-scratch initialisation: ``torch.manual_seed(17)`` (demo.py:70) + the constructor must give the
reference's tensors bit for bit (SparseDenseNetRefinementMask.py:239-257 runs inside the ctor; transposed
convolutions keep PyTorch's default init).  tests/golden/init17_fingerprint.npz holds the CRC32 of every
tensor of the reference's state_dict, made by tests/golden/make_inputdata_golden.py from the imported
reference."""
import contextlib
import io
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _kwargs():
    from make_golden import E2E_KW
    kw = dict(E2E_KW)
    kw.update(base_channels=8, thold=0.9)
    return kw


def test_seed17_init_matches_reference_bit_for_bit():
    from decnet_amd.model import get_model
    fp = np.load(os.path.join(HERE, "golden", "init17_fingerprint.npz"))
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(17)
        sd = get_model(**_kwargs()).state_dict()
    assert sorted(sd) == [str(k) for k in fp["keys"]]
    bad = [k for k, c, n in zip(fp["keys"], fp["crc32"], fp["numel"])
           if sd[str(k)].numel() != n or zlib.crc32(sd[str(k)].contiguous().numpy().tobytes()) != c]
    assert not bad, "tensors differing from the reference's seed-17 init: %s" % bad[:8]


def test_initialize_weights_rules():
    """He-normal (fan-out) Conv2d/Conv3d, zero conv biases, unit BatchNorm; ConvTranspose2d untouched."""
    from decnet_amd.model import get_model
    kw = _kwargs()
    kw["base_channels"] = 2
    torch.manual_seed(3)
    m = get_model(**kw)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all())
        if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
            assert bool((mod.bias == 0).all())
    w = m.cost_regularizer.conv0[0].conv.weight
    assert abs(float(w.std()) - (2.0 / (27 * w.shape[0])) ** 0.5) < 0.05 * float(w.std())
    tr = m.detail_detection[0].deconv[0].conv          # ConvTranspose2d with a bias: default (non-zero) init
    assert isinstance(tr, torch.nn.ConvTranspose2d) and float(tr.bias.abs().sum()) > 0
