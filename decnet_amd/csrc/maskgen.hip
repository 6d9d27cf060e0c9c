// decnet_amd/csrc/maskgen.hip -- the tail of GenerateSparseMask + the thresholding of the model loop in
// one pass.
//
// Reference: GenerateSparseMask.forward (modules/submodule.py:366-372) ends with
//     res_info = (cur_fea - pre_fea)^2                       [B,3,H,W] each
//     detail   = conv( res_info )   conv = Conv2dUnit(3,3,3x3,BN,no ReLU) -> Conv2dUnit(3,1,1x1,BN,no ReLU)
// and SparseDenseNetRefinementMask.forward :158-170 turns it into the binary masks SpaMat reads:
//     m = sigmoid(detail);  mask = (m > thold) ? 1 : 0       (float 0/1 plane, the reference's contract)
// As separate kernels that is a subtraction, a square, two convolutions, a sigmoid, two compares-and-fills
// and a cast: eight passes over full-resolution planes per view and stage.  Here: one kernel, four pixels per
// thread; the 81 + 3 weights and the folded BatchNorm constants are wave-uniform scalar loads; outputs are the float mask and, optionally,
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

typedef int i32x4_g __attribute__((ext_vector_type(4)));

// bit i of a 16-bit value -> bit 4 i of a 64-bit word
__device__ __forceinline__ unsigned long long spread4(unsigned long long x) {
    x &= 0xffffull;
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}

// Round 4: FOUR consecutive pixels per thread.  The first version (one pixel per thread) issued 54 dword loads per
// pixel and ran at 0.11 ms for the 118 MB of a full-resolution launch -- bound by vector-memory INSTRUCTIONS, not
// bytes.  Here a row of an input plane is one bounds-checked 16-byte load + the two neighbour dwords (rows / columns
// outside the image read as zero through the buffer descriptor: (0 - 0)^2 is the zero padding of res_info), the squared
// difference is formed once per loaded value instead of once per tap, and the mask leaves as one 16-byte store; the
// bit-packed copy is assembled from four ballots (lane l holds pixels 4 l .. 4 l + 3: word w of a 256-pixel block =
// lanes 16 w .. 16 w + 15).
__global__ __launch_bounds__(256) void detail_mask(const float *__restrict__ cur, const float *__restrict__ pre,
                                                   MaskGenParams P, float *__restrict__ mask,
                                                   float *__restrict__ logits,
                                                   unsigned long long *__restrict__ bits, int H, int W,
                                                   int words_per_row, int nrows) {
    int bx, row;
    if (!decnet_xcd_rows((W + 1023) >> 10, nrows, bx, row)) return;      // a 3-row stencil: rows of one XCD are neighbours
    const int x0 = (bx * 256 + threadIdx.x) * 4, b = row / H, y = row - b * H;
    const size_t plane = (size_t)H * W;
    const float *c0 = cur + (size_t)b * 3 * plane, *p0 = pre + (size_t)b * 3 * plane;
    float t[3][4];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[co][e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool rok = (unsigned)yy < (unsigned)H;                       // block-uniform: zero padding
        const size_t ro = rok ? (size_t)yy * W : 0;
        const int rb = rok ? W * 4 : 0;
        float d2[3][6];                                                    // torch.pow(cur - pre, 2) at x0 - 1 .. x0 + 4
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(c0 + ci * plane + ro), 0, rb, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(p0 + ci * plane + ro), 0, rb, 0x00020000);
            const i32x4_g cq = __builtin_amdgcn_raw_buffer_load_b128(rc, x0 * 4, 0, 0);
            const i32x4_g pq = __builtin_amdgcn_raw_buffer_load_b128(rp, x0 * 4, 0, 0);
            const float cl = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, (x0 - 1) * 4, 0, 0));
            const float pl = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, (x0 - 1) * 4, 0, 0));
            const float cr = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, (x0 + 4) * 4, 0, 0));
            const float pr = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, (x0 + 4) * 4, 0, 0));
            const float dl = cl - pl, dr = cr - pr;
            const float d0 = __int_as_float(cq.x) - __int_as_float(pq.x), d1 = __int_as_float(cq.y) - __int_as_float(pq.y);
            const float d2_ = __int_as_float(cq.z) - __int_as_float(pq.z), d3 = __int_as_float(cq.w) - __int_as_float(pq.w);
            d2[ci][0] = dl * dl; d2[ci][1] = d0 * d0; d2[ci][2] = d1 * d1; d2[ci][3] = d2_ * d2_; d2[ci][4] = d3 * d3;
            d2[ci][5] = dr * dr;
        }
        // the accumulation order of the one-pixel version (ky, kx, ci): the same fma chain per output, bit for bit
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int co = 0; co < 3; ++co)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[co][e] = fmaf(P.w3[co][ci][ky][kx], d2[ci][e + kx], t[co][e]);
    }
    float zv[4], mv[4];
    bool on[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float z = 0.f;
#pragma unroll
        for (int co = 0; co < 3; ++co) z = fmaf(P.w1[co], fmaf(t[co][e], P.scale3[co], P.shift3[co]), z);
        z = fmaf(z, P.scale1, P.shift1);
        const float s = 1.f / (1.f + expf(-z));                            // torch.sigmoid
        on[e] = x0 + e < W && s > P.thold;                                 // mask[m > thold] = 1, else 0
        zv[e] = z;
        mv[e] = on[e] ? 1.f : 0.f;
    }
    if (x0 < W) {
        const size_t pix = ((size_t)b * H + y) * W + x0;
        const bool vec = (W & 3) == 0 && (((uintptr_t)mask) & 15) == 0 && (!logits || (((uintptr_t)logits) & 15) == 0);
        if (vec) {
            *reinterpret_cast<float4 *>(mask + pix) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            if (logits) *reinterpret_cast<float4 *>(logits + pix) = make_float4(zv[0], zv[1], zv[2], zv[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < W) {
                    mask[pix + e] = mv[e];
                    if (logits) logits[pix + e] = zv[e];
                }
        }
    }
    if (bits) {
        const unsigned long long b0 = __ballot(on[0]), b1 = __ballot(on[1]), b2 = __ballot(on[2]), b3 = __ballot(on[3]);
        const int lane = threadIdx.x & 63;
        if ((lane & 15) == 0) {                                            // lanes 0, 16, 32, 48: one word each
            const int sh = lane;                                           // 16 w
            const unsigned long long word = spread4(b0 >> sh) | (spread4(b1 >> sh) << 1) | (spread4(b2 >> sh) << 2) |
                                            (spread4(b3 >> sh) << 3);
            const int wi = x0 >> 6;                                        // word of pixel x0 = 64-aligned block
            if (wi < words_per_row) bits[((size_t)b * H + y) * words_per_row + wi] = word;
        }
    }
}

}  // namespace

extern "C" int decnet_detail_mask(const float *cur3, const float *pre3, const float *w3x3, const float *scale3,
                                  const float *shift3, const float *w1x1, float scale1, float shift1,
                                  float thold, float *mask, float *logits, unsigned long long *bits, int B,
                                  int H, int W, void *stream) {
    if (!cur3 || !pre3 || !w3x3 || !scale3 || !shift3 || !w1x1 || !mask) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || H < 1 || W < 1 || B > 65535 || H > 65535) return DECNET_ERR_BAD_SHAPE;
    if ((double)B * H * ceil_div(W, 1024) >= 2.0e9 || W > (1 << 28)) return DECNET_ERR_UNSUPPORTED;
    MaskGenParams P;                       // host arrays: 90 floats, passed by value as a kernel argument
    for (int i = 0; i < 81; ++i) (&P.w3[0][0][0][0])[i] = w3x3[i];
    for (int i = 0; i < 3; ++i) { P.scale3[i] = scale3[i]; P.shift3[i] = shift3[i]; P.w1[i] = w1x1[i]; }
    P.scale1 = scale1; P.shift1 = shift1; P.thold = thold;
    const int wpr = (W + 63) / 64;
    hipLaunchKernelGGL(detail_mask, dim3(decnet_xcd_grid(ceil_div(W, 1024), (long)H * B)), dim3(256), 0,
                       (hipStream_t)stream, cur3, pre3, P, mask, logits, bits, H, W, wpr, H * B);
    return decnet_launch_status();
}
