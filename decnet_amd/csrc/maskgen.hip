// decnet_amd/csrc/maskgen.hip -- the tail of GenerateSparseMask + the thresholding of the model loop in
// one pass.
//
// Reference: GenerateSparseMask.forward (modules/submodule.py:366-372) ends with
//     res_info = (cur_fea - pre_fea)^2                       [B,3,H,W] each
//     detail   = conv( res_info )   conv = Conv2dUnit(3,3,3x3,BN,no ReLU) -> Conv2dUnit(3,1,1x1,BN,no ReLU)
// and SparseDenseNetRefinementMask.forward :158-170 turns it into the binary masks SpaMat reads:
//     m = sigmoid(detail);  mask = (m > thold) ? 1 : 0       (float 0/1 plane, the reference's contract)
// As separate kernels that is a subtraction, a square, two convolutions, a sigmoid, two compares-and-fills
// and a cast: eight passes over full-resolution planes per view and stage.  Here: one kernel, a thread per
// pixel; the 27 taps of both inputs come through L1 (neighbouring pixels share them), the 81 + 3 weights and
// the folded BatchNorm constants are wave-uniform scalar loads; outputs are the float mask and, optionally,
// the logits (for callers that want `detail`) and a bit-packed copy (one 64-bit word per 64 pixels of a
// row, bit i = pixel 64 w + i) for consumers that do not need 4 bytes per pixel.
#include "common.h"

namespace {

struct MaskGenParams {
    float w3[3][3][3][3];      // [co][ci][ky][kx] of the 3x3 unit
    float scale3[3], shift3[3];
    float w1[3];               // 1x1 unit (3 -> 1)
    float scale1, shift1, thold;
};

__global__ __launch_bounds__(256) void detail_mask(const float *__restrict__ cur, const float *__restrict__ pre,
                                                   MaskGenParams P, float *__restrict__ mask,
                                                   float *__restrict__ logits,
                                                   unsigned long long *__restrict__ bits, int H, int W,
                                                   int words_per_row, int nrows) {
    int bx, row;
    if (!decnet_xcd_rows((W + 255) >> 8, nrows, bx, row)) return;        // a 3-row stencil: rows of one XCD are neighbours
    const int x = bx * 256 + threadIdx.x, b = row / H, y = row - b * H;
    const size_t plane = (size_t)H * W;
    const float *c0 = cur + (size_t)b * 3 * plane, *p0 = pre + (size_t)b * 3 * plane;
    float t[3] = {0.f, 0.f, 0.f};
    if (x < W) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if ((unsigned)yy >= (unsigned)H) continue;                     // block-uniform: zero padding
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                const bool ok = (unsigned)xx < (unsigned)W;
                const size_t o = (size_t)yy * W + (ok ? xx : x);
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float df = c0[ci * plane + o] - p0[ci * plane + o];
                    const float d2 = ok ? df * df : 0.f;                   // torch.pow(cur - pre, 2), zero padded
#pragma unroll
                    for (int co = 0; co < 3; ++co) t[co] = fmaf(P.w3[co][ci][ky][kx], d2, t[co]);
                }
            }
        }
    }
    float z = 0.f;
#pragma unroll
    for (int co = 0; co < 3; ++co) z = fmaf(P.w1[co], fmaf(t[co], P.scale3[co], P.shift3[co]), z);
    z = fmaf(z, P.scale1, P.shift1);
    const float s = 1.f / (1.f + expf(-z));                                // torch.sigmoid
    const bool on = x < W && s > P.thold;                                  // mask[m > thold] = 1, else 0
    if (x < W) {
        const size_t pix = ((size_t)b * H + y) * W + x;
        mask[pix] = on ? 1.f : 0.f;
        if (logits) logits[pix] = z;
    }
    if (bits) {
        const unsigned long long word = __ballot(on);                      // lane i = pixel 64 w + i
        if ((threadIdx.x & 63) == 0 && (x >> 6) < words_per_row)
            bits[((size_t)b * H + y) * words_per_row + (x >> 6)] = word;
    }
}

}  // namespace

extern "C" int decnet_detail_mask(const float *cur3, const float *pre3, const float *w3x3, const float *scale3,
                                  const float *shift3, const float *w1x1, float scale1, float shift1,
                                  float thold, float *mask, float *logits, unsigned long long *bits, int B,
                                  int H, int W, void *stream) {
    if (!cur3 || !pre3 || !w3x3 || !scale3 || !shift3 || !w1x1 || !mask) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || H < 1 || W < 1 || B > 65535 || H > 65535) return DECNET_ERR_BAD_SHAPE;
    if ((double)B * H * ceil_div(W, 256) >= 2.0e9) return DECNET_ERR_UNSUPPORTED;
    MaskGenParams P;                       // host arrays: 90 floats, passed by value as a kernel argument
    for (int i = 0; i < 81; ++i) (&P.w3[0][0][0][0])[i] = w3x3[i];
    for (int i = 0; i < 3; ++i) { P.scale3[i] = scale3[i]; P.shift3[i] = shift3[i]; P.w1[i] = w1x1[i]; }
    P.scale1 = scale1; P.shift1 = shift1; P.thold = thold;
    const int wpr = (W + 63) / 64;
    hipLaunchKernelGGL(detail_mask, dim3(decnet_xcd_grid(ceil_div(W, 256), (long)H * B)), dim3(256), 0,
                       (hipStream_t)stream, cur3, pre3, P, mask, logits, bits, H, W, wpr, H * B);
    return decnet_launch_status();
}
