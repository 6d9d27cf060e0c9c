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
