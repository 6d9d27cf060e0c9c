"""Where can a flipped mask bit show up in the final disparity map?  Shared by the fixture generator
(reference fp32 run vs reference fp64 run) and tests/test_inputdata_gpu.py (HIP graph vs reference fp32 run).

The masks are thresholded sigmoids (SparseDenseNetRefinementMask.py:163-170): a logit within float noise of
the threshold may flip, which changes the candidate set of that left pixel (own flip) or of the left pixels
x .. x+D-1 of the row (right-view flip), and from there everything in reach of the attention / refinement /
upsampling convolutions of that and every finer stage."""
import numpy as np
import torch

# full-resolution reach of a stage-s flip: soft attention (3 convs) + refinement (7 convs, dilations
# submodule.py:666-700) of its own stage, then x3 + dynamic upsampling + attention + refinement of each finer one
REACH = {1: 224, 2: 96, 3: 32}


def dilate(mask, r):
    if r <= 0 or not mask.any():
        return mask
    t = torch.from_numpy(mask.astype(np.float32))[None, None]
    t = torch.nn.functional.max_pool2d(t, (1, 2 * r + 1), 1, (0, r))
    t = torch.nn.functional.max_pool2d(t, (2 * r + 1, 1), 1, (r, 0))
    return t[0, 0].numpy() > 0


def hit_map(lflip, rflip, D):
    """Left pixels of one stage whose candidate set changed: own flip, or a right flip at x-d, d in [0, D)."""
    hit = lflip.copy()
    if rflip.any():
        t = torch.from_numpy(rflip.astype(np.float32))[None, None]
        t = torch.nn.functional.max_pool2d(torch.nn.functional.pad(t, (D - 1, 0)), (1, D), 1)
        hit |= t[0, 0].numpy() > 0
    return hit


def dirty_map(flips, max_disp, H, W):
    """flips: {stage: (lflip, rflip)} boolean maps at stage resolution -> full-resolution map of the pixels
    a flip can reach, and {stage: hit map}."""
    dirty = np.zeros((H, W), bool)
    hits = {}
    for s, (lf, rf) in flips.items():
        hit = hit_map(lf, rf, max_disp // 3 ** (3 - s))
        hits[s] = hit
        scale = 3 ** (3 - s)
        full = np.kron(hit, np.ones((scale, scale), bool)) if scale > 1 else hit
        dirty |= dilate(full, REACH[s])
    return dirty, hits
