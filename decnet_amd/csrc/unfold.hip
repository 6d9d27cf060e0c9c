// decnet_amd/csrc/unfold.hip -- head of DynamicUpsampling.forward (modules/submodule.py:578-580):
//     torch.cat((disp.unsqueeze(1), F.unfold(fea, 3, stride=3).view(B, 9C, h, w)), 1)
// a space-to-depth of the fine-level features (channel c*9 + i*3 + j = fea[c, 3y+i, 3x+j]) behind the coarse
// disparity plane, as one pass (the stock path is im2col + a view + a concatenation copy).
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void unfold3_cat(const float *__restrict__ fea, const float *__restrict__ disp,
                                                   float *__restrict__ out, int C, int h, int w) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int b = blockIdx.z / (C + 1), c = blockIdx.z - b * (C + 1);   // c == C: the disparity plane
    if (x >= w) return;
    const size_t cp = (size_t)h * w;                                    // coarse plane
    float *o = out + ((size_t)b * (9 * C + 1)) * cp + (size_t)y * w + x;
    if (c == C) {
        o[0] = disp[((size_t)b * h + y) * w + x];
        return;
    }
    const float *f = fea + (((size_t)b * C + c) * 3 * h + 3 * y) * (3 * (size_t)w) + 3 * x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v0 = f[(size_t)i * 3 * w], v1 = f[(size_t)i * 3 * w + 1], v2 = f[(size_t)i * 3 * w + 2];
        o[(size_t)(1 + c * 9 + i * 3 + 0) * cp] = v0;
        o[(size_t)(1 + c * 9 + i * 3 + 1) * cp] = v1;
        o[(size_t)(1 + c * 9 + i * 3 + 2) * cp] = v2;
    }
}
}  // namespace

extern "C" int decnet_unfold3_cat(const float *fea, const float *disp, float *out, int B, int C, int h, int w,
                                  void *stream) {
    if (!fea || !disp || !out) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || h < 1 || w < 1 || h > 65535 || (long)B * (C + 1) > 65535) return DECNET_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(unfold3_cat, dim3((unsigned)ceil_div(w, 256), (unsigned)h, (unsigned)(B * (C + 1))), dim3(256), 0,
                       (hipStream_t)stream, fea, disp, out, C, h, w);
    return decnet_launch_status();
}
