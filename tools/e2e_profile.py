"""Whole-graph forward of decnet_amd.model on B synthetic pairs (the --e2e leg of bench.py) a few
times, for `rocprofv3 --kernel-trace --stats -- python3 tools/e2e_profile.py`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    t0 = time.time()
    r = bench.e2e_bench(B, torch.device("cuda:0"), iters=5)
    print(r, "wall %.1f s" % (time.time() - t0))
