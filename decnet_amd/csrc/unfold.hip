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
    float v[9];                                     // nine requests, then nine stores
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[i * 3 + j] = f[(size_t)i * 3 * w + j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) o[(size_t)(1 + c * 9 + t) * cp] = v[t];
}
// The stride-3 convolutions of FeatExtNetChannelPlus (Conv2d k 3, stride 3, padding 1: submodule.py:270-300) as a
// space-to-depth gather in front of a 1 x 1 convolution on the matrix cores (csrc/conv2d_mfma.hip):
//     out[b, c*9 + ky*3 + kx, yo, xo] = x[b, c, 3 yo - 1 + ky, 3 xo - 1 + kx]   (0 outside the image)
// -- every input pixel feeds exactly one output pixel, so this is a permutation with a zero border; the channel order
// is the row-major flattening of the weight [Cout, Cin, 3, 3].
__global__ __launch_bounds__(256) void s2d3_pad1(const float *__restrict__ x, float *__restrict__ out, int C, int H, int W,
                                                 int Ho, int Wo) {
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
    const int b = blockIdx.z / C, c = blockIdx.z - b * C;
    if (xo >= Wo) return;
    const size_t cp = (size_t)Ho * Wo;
    float *o = out + ((size_t)b * 9 * C + 9 * c) * cp + (size_t)yo * Wo + xo;
    const float *f = x + ((size_t)b * C + c) * H * (size_t)W;
    float v[9];                                     // nine requests, then nine stores (a store right behind its load
#pragma unroll                                      // is a memory round trip per tap: round 5, from the ISA)
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = 3 * yo - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xi = 3 * xo - 1 + kx;
            v[ky * 3 + kx] = (yi >= 0 && yi < H && xi >= 0 && xi < W) ? f[(size_t)yi * W + xi] : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t) o[(size_t)t * cp] = v[t];
}
}  // namespace

extern "C" int decnet_s2d3_pad1(const float *x, float *out, int B, int C, int H, int W, void *stream) {
    if (!x || !out) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    const int Ho = (H - 1) / 3 + 1, Wo = (W - 1) / 3 + 1;            // floor((H + 2 - 3) / 3) + 1
    if (Ho > 65535 || (long)B * C > 65535) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(s2d3_pad1, dim3((unsigned)ceil_div(Wo, 256), (unsigned)Ho, (unsigned)(B * C)), dim3(256), 0,
                       (hipStream_t)stream, x, out, C, H, W, Ho, Wo);
    return decnet_launch_status();
}

extern "C" int decnet_unfold3_cat(const float *fea, const float *disp, float *out, int B, int C, int h, int w,
                                  void *stream) {
    if (!fea || !disp || !out) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || h < 1 || w < 1 || h > 65535 || (long)B * (C + 1) > 65535) return DECNET_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(unfold3_cat, dim3((unsigned)ceil_div(w, 256), (unsigned)h, (unsigned)(B * (C + 1))), dim3(256), 0,
                       (hipStream_t)stream, fea, disp, out, C, h, w);
    return decnet_launch_status();
}
