// decnet_amd/csrc/common.h -- shared helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/decnet_hip.h"

#define DECNET_WAVE 64                  // CDNA4 wavefront
#define DECNET_LDS_BYTES (160 * 1024)   // LDS per CU on MI355X
#define DECNET_LDS_BUDGET (64 * 1024)   // per-workgroup target: >= 2 workgroups per CU

static inline int decnet_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DECNET_OK : (int)e;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// The cost of one (left, right) feature pair, GetCostVolume.cost_computation_* (submodule.py:511-530), with the
// reference's fp32 operation sequence (no contraction): CF = DECNET_COST_COR l * r (:521); DECNET_COST_SSD
// (l^2 + r^2) / 2 - ((l + r) / 2)^2 as torch evaluates :527-529 (pow_(2) is x * x, div_(2) is exact);
// DECNET_COST_SUM l + r (the "cat" volume behind conv_pre, see decnet_stage0_forward_cf).
template <int CF>
__device__ __forceinline__ float decnet_cost(float l, float r) {
#pragma clang fp contract(off)
    if (CF == DECNET_COST_COR) return l * r;
    if (CF == DECNET_COST_SSD) {
        const float volume_sum = l + r;
        const float ll = l * l, rr = r * r;
        const float volume_sqr = ll + rr;
        const float half = volume_sum * 0.5f;
        const float hh = half * half;
        return volume_sqr * 0.5f - hh;
    }
    return l + r;
}

// XCD-aware block order for row-stencil kernels.  Workgroups are dealt round-robin to the 8 XCDs, each with its
// own L2: in the natural (x block, row) order the rows y-1, y, y+1 that a 3x3 stencil reads are fetched into three
// different L2s (measured on the 8 -> 8 full-resolution convolution: ~3x the input bytes over the fabric, 0.109 ms
// where the FMA issue floor is 0.045).  Here the launch is a 1-D grid of 8 * ceil(T / 8) blocks, T = gx * nrows, and
// block ids that are equal mod 8 (one XCD) walk a contiguous band of rows.  Returns false for the padding blocks.
__device__ __forceinline__ bool decnet_xcd_rows(int gx, int nrows, int &bx, int &row) {
    const int T = gx * nrows, per = (T + 7) >> 3;
    const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (v >= T) return false;
    row = v / gx;
    bx = v - row * gx;
    return true;
}
#endif
static inline unsigned decnet_xcd_grid(int gx, long nrows) { return (unsigned)(8 * ((gx * nrows + 7) / 8)); }
