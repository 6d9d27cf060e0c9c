"""The forward parity suite again with the dispatcher pinned to each kernel variant
(DECNET_SPAMAT_KERNEL is read once per process, hence the subprocess).  -m gpu."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["rowtile", "mfma", "mfma_dense"])
def test_forward_suite_with_pinned_variant(variant):
    env = dict(os.environ, DECNET_SPAMAT_KERNEL=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spamat_gpu.py"),
                        "-m", "gpu", "-q", "-x", "-k", "forward or golden or full_size or backward"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_forward_suite_with_fp32_dense_rows():
    """Dense rows at C <= 8 default to bf16x3 cost tiles on the bf16 matrix cores (dense16_body); the fp32 MFMA chain
    (DECNET_SPAMAT_DENSE=fp32, also what C > 8 uses) must pass the same suite."""
    env = dict(os.environ, DECNET_SPAMAT_DENSE="fp32")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spamat_gpu.py"),
                        "-m", "gpu", "-q", "-x", "-k", "forward or golden or full_size or mixed_row"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_forward_suite_without_the_mask_bit_handover():
    """Round 5: the sparse-row launch hands the activity bits of the rows it rejects to the band launch (in the rows' own
    max_cost entries).  DECNET_SPAMAT_HANDOVER=0 keeps the band launch on the float mask planes: same results."""
    env = dict(os.environ, DECNET_SPAMAT_HANDOVER="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_spamat_gpu.py"),
                        os.path.join(ROOT, "tests", "test_spamat_ref.py"), "-m", "gpu", "-q", "-x", "-k",
                        "forward or golden or full_size or mixed_row or reference_kernels"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_handover_equals_float_masks_bit_for_bit():
    """The two routes of the mask information give the same bits, so the same outputs: mid-density and dense rows, widths
    that are not multiples of 32, one process with the switch off for the reference run."""
    import numpy as np
    import torch
    code = ("import sys, torch, numpy as np; sys.path.insert(0, %r); import decnet_amd; dev = torch.device('cuda:0'); out = {}\n"
            "for i, (C, H, W, D, p) in enumerate(((8, 6, 972, 216, 0.4), (8, 5, 333, 100, 0.9), (7, 4, 1001, 216, 0.6), (8, 3, 70, 40, 1.0))):\n"
            "    g = torch.Generator().manual_seed(40 + i)\n"
            "    L = torch.randn(2, C, H, W, generator=g).to(dev); R = torch.randn(2, C, H, W, generator=g).to(dev)\n"
            "    rm = (torch.rand(2, H, W, generator=g) < p).float().to(dev); tm = (torch.rand(2, H, W, generator=g) < p).float().to(dev)\n"
            "    o = decnet_amd.spamatvar_forward(L, R, rm, tm, D)\n"
            "    out['c%%d' %% i] = torch.stack(o).cpu().numpy()\n"
            "np.savez(sys.argv[1], **out)\n" % ROOT)
    import tempfile
    res = {}
    for sw in ("1", "0"):
        with tempfile.NamedTemporaryFile(suffix=".npz") as f:
            env = dict(os.environ, DECNET_SPAMAT_HANDOVER=sw)
            r = subprocess.run([sys.executable, "-c", code, f.name], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[sw] = {k: v.copy() for k, v in np.load(f.name).items()}
    for k in res["1"]:
        assert np.array_equal(res["1"][k], res["0"][k]), k
