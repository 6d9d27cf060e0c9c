"""Case list + seeded input generator shared by make_spamat_ref_golden.py (which runs the REFERENCE's
own SpaMat / SpaVar kernels, built unmodified by oracle/ref_build.sh, on an MI355X) and by the tests
that compare the oracle (CPU) and the HIP path (GPU) with the recorded outputs.

Inputs come from numpy's legacy RandomState (its streams are frozen by numpy's compatibility
policy), so only the reference's OUTPUTS are stored; each fixture carries a CRC32 of its inputs.
"""
import zlib

import numpy as np

# name: (B, C, H, W, max_disp, per-row (p_ref, p_tar) cycle, relu, scale, seed)
#   cfgN_sK = the stage-K row shape of BASELINE config N (SURVEY.md §8 shape table), a few rows each.
CASES = {
    # --- config 2 / 5 (972x540 padded, max_disp 216) ---
    "cfg2_s1": (1, 72, 3, 108, 24, [(1.0, 1.0), (0.5, 0.5), (0.1, 0.1)], True, 1.0, 101),
    "cfg2_s2": (1, 24, 3, 324, 72, [(1.0, 1.0), (0.5, 0.5), (0.1, 0.1)], True, 1.0, 102),
    "cfg2_s3": (1, 8, 4, 972, 216, [(1.0, 1.0), (0.5, 0.5), (0.3, 0.3), (0.1, 0.1)], True, 1.0, 103),
    # --- config 3 (KITTI 1242x378) ---
    "cfg3_s1": (1, 72, 2, 138, 24, [(1.0, 1.0), (0.3, 0.6)], True, 1.0, 111),
    "cfg3_s2": (1, 24, 2, 414, 72, [(1.0, 1.0), (0.3, 0.6)], True, 1.0, 112),
    "cfg3_s3": (1, 8, 2, 1242, 216, [(1.0, 1.0), (0.2, 0.2)], True, 1.0, 113),
    # --- config 4 (Middlebury half, 1512x1026, max_disp 270) ---
    "cfg4_s1": (1, 72, 2, 168, 30, [(1.0, 1.0), (0.4, 0.4)], True, 1.0, 121),
    "cfg4_s2": (1, 24, 2, 504, 90, [(1.0, 1.0), (0.4, 0.4)], True, 1.0, 122),
    "cfg4_s3": (1, 8, 2, 1512, 270, [(1.0, 1.0), (0.15, 0.15)], True, 1.0, 123),
    # --- shapes off the shipped grid / edge rules ---
    "batch2": (2, 8, 2, 300, 216, [(0.8, 0.8), (0.5, 1.0)], True, 1.0, 131),     # batch indexing
    "disp_gt_w": (1, 8, 2, 100, 216, [(1.0, 1.0), (0.6, 0.6)], True, 1.0, 132),  # max_disp > W: cur = x + 1
    "tiny_oddc": (1, 3, 2, 7, 5, [(1.0, 0.5)], True, 1.0, 133),
    "oddc5": (1, 5, 2, 9, 12, [(1.0, 1.0)], False, 0.5, 134),
    # --- signed features (gradients of both signs, costs below the 1e-6 floor) ---
    "signed_s3": (1, 8, 2, 300, 216, [(1.0, 1.0), (0.5, 0.5)], False, 0.5, 141),
    "signed_s2": (1, 24, 2, 81, 72, [(0.9, 0.8)], False, 0.5, 142),
    "signed_s1": (1, 72, 2, 27, 24, [(0.5, 0.5)], False, 0.5, 143),
    # --- large costs: exp() underflows for all but the best candidates ---
    "sharp_s3": (1, 8, 2, 300, 216, [(1.0, 1.0), (0.3, 0.3)], True, 3.0, 151),
}

QUIRKS = ("all_tar_off", "all_ref_off", "negative_costs", "single_candidate", "left_edge")


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


def make_inputs(name):
    """-> dict(L, R, rm, tm, g, mu_noise, max_disp) float32 numpy arrays."""
    if name in QUIRKS:
        return _quirk(name)
    B, C, H, W, D, rows, relu, scale, seed = CASES[name]
    rs = np.random.RandomState(seed)
    L = (rs.standard_normal((B, C, H, W)) * scale).astype(np.float32)
    R = (rs.standard_normal((B, C, H, W)) * scale).astype(np.float32)
    if relu:
        L, R = np.maximum(L, 0), np.maximum(R, 0)
    pr = np.array([rows[y % len(rows)][0] for y in range(H)])[None, :, None]
    pt = np.array([rows[y % len(rows)][1] for y in range(H)])[None, :, None]
    rm = (rs.random_sample((B, H, W)) < pr).astype(np.float32)
    tm = (rs.random_sample((B, H, W)) < pt).astype(np.float32)
    g = rs.standard_normal((B, H, W)).astype(np.float32)
    mu_noise = rs.standard_normal((B, H, W)).astype(np.float32)
    return dict(L=L, R=R, rm=rm, tm=tm, g=g, mu_noise=mu_noise, max_disp=D)


def _quirk(name):
    """Known-answer situations of SM_kernel.cu:45,100,123-124."""
    rs = np.random.RandomState(7)
    B, C, H, W, D = 1, 8, 2, 64, 64
    L = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    R = np.maximum(rs.standard_normal((B, C, H, W)), 0).astype(np.float32)
    rm = np.ones((B, H, W), np.float32)
    tm = np.ones((B, H, W), np.float32)
    if name == "all_tar_off":          # ref on, no valid candidate -> 1e-6 / 1e-6 = 1.0
        tm[:] = 0
    elif name == "all_ref_off":        # ref off -> the caller's zero fill survives
        rm[:] = 0
    elif name == "negative_costs":     # every cost < 0 -> max_cost stays at its 1e-6 floor
        R = -R - 0.1
        L = L + 0.1
    elif name == "single_candidate":   # one tar pixel on per row
        tm[:] = 0
        tm[:, :, 10] = 1
    elif name == "left_edge":          # only x < 4 on: cur_max_disp = x + 1
        rm[:] = 0
        rm[:, :, :4] = 1
    g = rs.standard_normal((B, H, W)).astype(np.float32)
    mu_noise = rs.standard_normal((B, H, W)).astype(np.float32)
    return dict(L=L, R=R, rm=rm, tm=tm, g=g, mu_noise=mu_noise, max_disp=D)


def all_names():
    return list(QUIRKS) + list(CASES)


GROUPS = {
    "quirks": list(QUIRKS),
    "cfg2": ["cfg2_s1", "cfg2_s2", "cfg2_s3"],
    "cfg3": ["cfg3_s1", "cfg3_s2", "cfg3_s3"],
    "cfg4": ["cfg4_s1", "cfg4_s2", "cfg4_s3"],
    "misc": ["batch2", "disp_gt_w", "tiny_oddc", "oddc5", "signed_s3", "signed_s2", "signed_s1", "sharp_s3"],
}
