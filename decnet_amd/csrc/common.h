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
