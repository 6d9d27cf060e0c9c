"""More seeds of tests/test_spamat_gpu.py::test_randomized_shapes_forward_and_backward (5 random cases per seed, HIP path
against the CPU oracle, forward + both backward passes):  python tools/fuzz_spamat.py [first_seed [n_seeds [seconds]]]"""
import os
import sys
import time

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_spamat_gpu as t  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
dev = torch.device("cuda:0")
t0 = time.time()
bad = 0
done = 0
for seed in range(first, first + n):
    if time.time() - t0 > budget:
        break
    done += 1
    try:
        t.test_randomized_shapes_forward_and_backward.__wrapped__(dev, seed) if hasattr(
            t.test_randomized_shapes_forward_and_backward, "__wrapped__") else t.test_randomized_shapes_forward_and_backward(dev, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:400], flush=True)
print("%d seeds (%d cases), %d failed, %.0f s" % (done, 5 * done, bad, time.time() - t0))
sys.exit(1 if bad else 0)
