// decnet_amd/csrc/conv2d_small.hip -- the full-resolution, few-channel 2-D convolutions of the trunk
// (SURVEY.md 8f-2): Conv2dUnit / Deconv2dUnit in eval mode (modules/submodule.py:15-87) =
// conv -> BatchNorm2d(running stats) -> ReLU, fused.
//
// At 540x972 with 3..17 channels these layers are bandwidth-shaped (a [8,8,540,972] tensor is
// 134 MB; 8 -> 8 channels is 576 MACs per output pixel), and the stock path runs each as three
// kernels (MIOpen Winograd/implicit-GEMM conv, BatchNorm, ReLU: 0.35-1.0 ms per layer).  Here a
// thread owns one output pixel and all (<= 8) output channels: the input taps come through L1
// (neighbouring pixels share them), the weights are wave-uniform (scalar loads), BN is a folded
// per-channel scale/shift, and every tensor is read / written exactly once.
//   conv2d_small<CO,K>     k = 1 or 3, stride 1, dilation d, padding d*(k/2)       (Conv2dUnit)
//   deconv2d_k3s3<CO>      ConvTranspose2d k = 3, stride 3, padding 0: every output pixel has exactly
//                          one tap, in[y/3][x/3] * w[ci][co][y%3][x%3]              (Deconv2dUnit)
//   conv2d_k3s3<CO>        k = 3, stride 3, padding 1 (the down-sampling convs of FeatExtNet)
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef int i32x4_c __attribute__((ext_vector_type(4)));
typedef float f32x3_u __attribute__((ext_vector_type(3), aligned(4)));   // dword-aligned 12-byte vector (global_store_dwordx3)

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
namespace {

// A thread owns 4 consecutive output pixels x CO output channels.  Input taps are bounds-checked
// 16-byte buffer loads against a descriptor of ONE input row (base = row start, size = W floats):
// dwords past the end of the row read as zero one by one, and a load that STARTS left of the row
// (negative = huge unsigned offset) reads as zero entirely (tools/ubench/buf_oob.hip) -- so only
// the one thread per row whose left tap straddles x = 0 patches up to three dwords; no other
// branch or select in the loop.  Weights [Cin][K][K][CO] are wave-uniform scalar loads.
// The input may be the channel concatenation of up to 6 tensors [B,c_i,H,W] (the torch.cat in front
// of Deconv2dBlock / Refinement / SoftAttention convolutions is never materialised).
constexpr int MAXSEG = 6;
struct Segs {
    const float *p[MAXSEG];
    int c[MAXSEG];
    int n;
};

template <int CO, int K>
__global__ __launch_bounds__(256) void conv2d_small(Segs in, const float *__restrict__ w,
                                                    const float *__restrict__ scale,
                                                    const float *__restrict__ shift, float *__restrict__ y,
                                                    int Cout, int H, int W, int dil, int relu, int nrows, int epi,
                                                    const float *__restrict__ ea, const float *__restrict__ eb) {
    int bx, row;
    if (!decnet_xcd_rows((W + 1023) >> 10, nrows, bx, row)) return;      // rows of one XCD's blocks are neighbours
    const bool nt = relu & 2;                                            // streaming stores (store_policy)
    relu &= 1;
    const int x0 = (bx * 256 + threadIdx.x) * 4, b = row / H, yy = row - b * H;
    if (x0 >= W) return;
    const size_t plane = (size_t)H * W;
    const int xl = x0 - (K / 2) * dil;                                   // first element of the left tap
    const bool straddle = xl < 0 && xl > -4;
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[co][e] = 0.f;
    int ci = 0;                                                          // channel of the concatenation
    for (int sg = 0; sg < in.n; ++sg)
    for (int cs = 0; cs < in.c[sg]; ++cs, ++ci) {
        const float *xp = in.p[sg] + ((size_t)b * in.c[sg] + cs) * plane;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yi = yy + (ky - K / 2) * dil;
            if ((unsigned)yi >= (unsigned)H) continue;                   // block-uniform
            const __amdgpu_buffer_rsrc_t rr =
                __builtin_amdgcn_make_buffer_rsrc((void *)(xp + (size_t)yi * W), 0, W * 4, 0x00020000);
            float v[K][4];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const i32x4_c t = __builtin_amdgcn_raw_buffer_load_b128(rr, (x0 + (kx - K / 2) * dil) * 4, 0, 0);
                v[kx][0] = __int_as_float(t.x); v[kx][1] = __int_as_float(t.y);
                v[kx][2] = __int_as_float(t.z); v[kx][3] = __int_as_float(t.w);
            }
            if (K > 1 && straddle) {
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (xl + e >= 0) v[0][e] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (xl + e) * 4, 0, 0));
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int co = 0; co < CO; ++co) {                  // co >= Cout: zero weights (packed so)
                    const float wv = w[((ci * K + ky) * K + kx) * CO + co];     // wave-uniform, co contiguous
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(v[kx][e], wv, acc[co][e]);
                }
        }
    }
    const bool vec = (W & 3) == 0 && ((uintptr_t)y & 15) == 0;
#pragma unroll
    for (int co = 0; co < CO; ++co)
        if (co < Cout) {
            const float sc = scale[co], sh = shift[co];
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = fmaf(acc[co][e], sc, sh);
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
            if (CO == 1 && epi) {
#pragma clang fp contract(off)
                // single-output layers with the elementwise tail of their caller fused (ea, eb: [B,H,W] planes):
                //   1: SoftAttention + fusion (SparseDenseNetRefinementMask.py:195-202): s = sigmoid(o); ea (1 - s) + s eb
                //   2: Refinement's residual (submodule.py:716): ea + o
                const size_t pix = (size_t)b * plane + (size_t)yy * W + x0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (x0 + e >= W) continue;
                    if (epi == 1) {
                        const float sft = 1.f / (1.f + expf(-o[e]));
                        const float t1 = ea[pix + e] * (1.f - sft), t2 = sft * eb[pix + e];
                        o[e] = t1 + t2;
                    } else {
                        o[e] = ea[pix + e] + o[e];
                    }
                }
            }
            float *yp = y + ((size_t)b * Cout + co) * plane + (size_t)yy * W + x0;
            if (vec) {
                if (nt) __builtin_nontemporal_store(f32x4_nt{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4_nt *>(yp));
                else *reinterpret_cast<float4 *>(yp) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x0 + e < W) yp[e] = o[e];
            }
        }
}

// ---- the same layers on the fp32 matrix pipe (round 4) ------------------------------------------------------------
// conv2d_small above runs its multiply-adds as v_pk_fma_f32 with an SGPR weight pair: 60 - 70 TFLOP/s whatever the
// occupancy (8 -> 8 is HBM-shaped at that rate, but 12 / 16 / 17 -> 8 -- the concatenated inputs of Deconv2dBlock,
// SoftAttention and Refinement -- take 1.5 - 2 x their HBM time).  v_mfma_f32_4x4x1_16B_f32 does 16 independent
// 4 x 4 outer products per wave-instruction (block b = lane / 4: D[i][j] += A[4b + i] * B[4b + j]) at the full fp32
// matrix rate (tools/ubench/mfma4x4: 125 TFLOP/s from two accumulator chains) and, unlike the 16 x 16 / 32 x 32
// shapes, wastes nothing at 4, 8, 12 or 24 output channels:
//     A = the input value of the lane's OWN pixel (lane <-> pixel x0 + lane, one row of 64 pixels per wave),
//     B = the weight of output channel 4 cog + (lane & 3) for this (input channel, tap), read from LDS,
//     D: lane 4b + j, register i = pixel x0 + 4b + i, channel 4 cog + j  -> one 16-byte store per accumulator.
// A wave owns 64 pixels x R output rows spaced by the dilation (so that the R + 2 input rows serve all of them), the
// taps x - d, x, x + d are three bounds-checked dword loads per input row (out of range = 0, no halo logic), the
// result is the same k-ordered fp32 fma chain per output as before (input channels outermost, then ky, kx).
template <int COQ, int K, int R>
__global__ __launch_bounds__(256) void conv2d_f32m(Segs in, const float *__restrict__ w,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   float *__restrict__ y, int Cout, int H, int W, int dil, int relu,
                                                   int ntasks, int tasks_per_img, int cin) {
    constexpr int CO = 4 * COQ, KK = K * K, NR = K == 1 ? R : R + 2;
    extern __shared__ float wl[];                        // [cin][KK][4 (j)][COQ]: a lane's COQ weights are adjacent
    for (int i = threadIdx.x; i < cin * KK * CO; i += 256) {
        const int e = i % CO, rest = i / CO, j = e / COQ, cog = e - j * COQ;
        wl[i] = w[rest * CO + 4 * cog + j];
    }
    __syncthreads();
    int bx, task;
    if (!decnet_xcd_rows((W + 255) >> 8, ntasks, bx, task)) return;
    const bool nt = relu & 2;                            // streaming stores (store_policy)
    relu &= 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = (bx * 4 + wave) * 64;
    if (x0 >= W) return;
    const int x = x0 + lane;
    const int b = task / tasks_per_img, t = task - b * tasks_per_img;
    const int yb = (t / dil) * R * dil + (t % dil);      // output rows yb + r * dil
    const size_t plane = (size_t)H * W;
    size_t roff[NR];
    int rbytes[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {                    // wave-uniform
        const int yi = yb + (rr - K / 2) * dil;
        const bool ok = (unsigned)yi < (unsigned)H;
        roff[rr] = ok ? (size_t)yi * W : 0;
        rbytes[rr] = ok ? W * 4 : 0;                     // a row outside the image reads as zeros
    }
    typedef float f32x4_m __attribute__((ext_vector_type(4)));
    f32x4_m acc[R][COQ];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < COQ; ++c) acc[r][c] = f32x4_m{0.f, 0.f, 0.f, 0.f};
    const float *wlane = wl + (lane & 3) * COQ;
    // CHB channels per step: all their taps are requested before the first MFMA (no loop-carried tap registers)
    constexpr int CHB = 1;            // input channels whose taps are in flight together (2: measured slower)
    int sg = 0, cs = 0;
    auto plane_ptr = [&]() { return in.p[sg] + ((size_t)b * in.c[sg] + cs) * plane; };
    auto advance = [&]() { if (++cs == in.c[sg]) { cs = 0; ++sg; } };
    for (int ci = 0; ci < cin; ci += CHB) {
        float v[CHB][NR][K];
        const float *base = plane_ptr();
#pragma unroll
        for (int u = 0; u < CHB; ++u) {
            const bool live = ci + u < cin;
            const float *xp = live ? plane_ptr() : base;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc((void *)(xp + roff[rr]), 0, live ? rbytes[rr] : 0, 0x00020000);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)          // x - d < 0: a huge unsigned offset; x + d >= W: past the row
                    v[u][rr][kx] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (x + (kx - K / 2) * dil) * 4, 0, 0));
            }
            if (live) advance();
        }
#pragma unroll
        for (int u = 0; u < CHB; ++u) {
            if (ci + u >= cin) break;
            const float *wc = wlane + (ci + u) * KK * CO;
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    float wv[COQ];
#pragma unroll
                    for (int c = 0; c < COQ; ++c) wv[c] = wc[(ky * K + kx) * CO + c];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int c = 0; c < COQ; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u][r + ky][kx], wv[c], acc[r][c], 0, 0, 0);
                }
        }
    }
    const int j = lane & 3, xq = x0 + (lane & ~3);       // this lane's four pixels xq .. xq + 3
    const bool vec = (W & 3) == 0 && ((uintptr_t)y & 15) == 0;
#pragma unroll
    for (int c = 0; c < COQ; ++c) {
        const int co = 4 * c + j;
        if (co >= Cout) continue;
        const float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = yb + r * dil;
            if (row >= H) break;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = fmaf(acc[r][c][e], sc, sh);
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
            float *yp = y + ((size_t)b * Cout + co) * plane + (size_t)row * W + xq;
            if (vec && xq + 3 < W) {
                if (nt) __builtin_nontemporal_store(f32x4_nt{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4_nt *>(yp));
                else *reinterpret_cast<float4 *>(yp) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (xq + e < W) yp[e] = o[e];
            }
        }
    }
}

// weights packed [Cin][3][3][CO] (decnet_conv2d_pack_weight, transposed = 1); output (3H) x (3W).
// A thread owns one INPUT pixel: its Cin values are read once and produce the 3 x 3 output pixels
// x CO channels that depend on it (and on nothing else); a wave's three stores per (co, row) fill
// 768 contiguous bytes.
template <int CO>
__global__ __launch_bounds__(256) void deconv2d_k3s3(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ shift, float *__restrict__ y,
                                                     int Cin, int Cout, int H, int W, int relu) {
    const int Wo = 3 * W, Ho = 3 * H;
    // input pixels of an image numbered row-major across rows: a 324-pixel row does not leave 60 lanes of its sixth wave idle
    const int pi = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    if (pi >= H * W) return;
    const int yi = pi / W, xi = pi - yi * W;
    const size_t plane = (size_t)H * W;
    const float *xp = x + (size_t)b * Cin * plane + (size_t)yi * W + xi;
    float acc[3][3][CO];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[ky][kx][co] = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float v = xp[(size_t)ci * plane];
        const float *wp = w + ci * 9 * CO;                              // [ky][kx][co], wave-uniform
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[ky][kx][co] = fmaf(v, wp[(ky * 3 + kx) * CO + co], acc[ky][kx][co]);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
        if (co < Cout) {
            const float sc = scale[co], sh = shift[co];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float *yp = y + (((size_t)b * Cout + co) * Ho + 3 * yi + ky) * Wo + 3 * xi;
                f32x3_u v3;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float v = fmaf(acc[ky][kx][co], sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    v3[kx] = v;
                }
                *reinterpret_cast<f32x3_u *>(yp) = v3;       // one 12-byte store: a wave fills 768 contiguous bytes
            }
        }
}

// Conv2d k = 3, stride 3, padding 1 (the down-sampling convs of FeatExtNetChannelPlus): every input
// pixel feeds exactly one output pixel.  A thread owns one output pixel and all CO (<= 24) output
// channels; weights packed [Cin][3][3][CO].
template <int CO>
__global__ __launch_bounds__(256) void conv2d_k3s3(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ scale,
                                                   const float *__restrict__ shift, float *__restrict__ y,
                                                   int Cin, int Cout, int H, int W, int Ho, int Wo, int relu) {
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y, b = blockIdx.z;
    if (xo >= Wo) return;
    const size_t plane = (size_t)H * W;
    const float *xb = x + (size_t)b * Cin * plane;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yi = 3 * yo - 1 + ky;
            if ((unsigned)yi >= (unsigned)H) continue;                   // block-uniform
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xb + (size_t)ci * plane + (size_t)yi * W), 0, W * 4, 0x00020000);
            float v[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)               // x = -1 has a huge unsigned offset: reads as zero
                v[kx] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (3 * xo - 1 + kx) * 4, 0, 0));
            const float *wp = w + ((ci * 3 + ky) * 3) * CO;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(v[kx], wp[kx * CO + co], acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
        if (co < Cout) {
            float v = fmaf(acc[co], scale[co], shift[co]);
            if (relu) v = fmaxf(v, 0.f);
            y[(((size_t)b * Cout + co) * Ho + yo) * Wo + xo] = v;
        }
}

// [Cout][Cin][k][k] (Conv2d) or [Cin][Cout][k][k] (ConvTranspose2d) -> [Cin][k][k][CO], zero padded in co
__global__ void pack_weight2d(const float *__restrict__ w, float *__restrict__ wp, int Cin, int Cout, int kk,
                              int CO, int transposed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * kk * CO) return;
    const int co = i % CO, t = (i / CO) % kk, ci = i / (CO * kk);
    float v = 0.f;
    if (co < Cout) v = transposed ? w[((size_t)ci * Cout + co) * kk + t] : w[((size_t)co * Cin + ci) * kk + t];
    wp[i] = v;
}

__host__ __device__ constexpr int co_pad(int Cout) { return Cout <= 1 ? 1 : Cout <= 4 ? 4 : Cout <= 8 ? 8 : Cout <= 12 ? 12 : 24; }

// y[b][c][:] = act(y[b][c][:] + shift[c]) in place: the folded-BatchNorm bias and the ReLU of the layers
// that stay on MIOpen, one pass instead of a bias-add kernel and a clamp kernel.
__global__ __launch_bounds__(256) void bias_act_inplace(float *__restrict__ y, const float *__restrict__ shift,
                                                        int C, int HW, int relu) {
    const int bc = blockIdx.y;                                          // b * C + c
    const float sh = shift[bc % C];
    float *p = y + (size_t)bc * HW;
    const int i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= HW) return;
    if (i4 + 3 < HW && (((uintptr_t)(p + i4)) & 15) == 0) {
        float4 v = *reinterpret_cast<float4 *>(p + i4);
        v.x += sh; v.y += sh; v.z += sh; v.w += sh;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4 *>(p + i4) = v;
    } else {
        for (int e = 0; e < 4 && i4 + e < HW; ++e) {
            float v = p[i4 + e] + sh;
            p[i4 + e] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

template <int CO>
int launch_conv(const Segs &in, const float *w, const float *scale, const float *shift, float *y, int B,
                int Cout, int H, int W, int k, int dil, int relu, hipStream_t s, int epi = 0, const float *ea = nullptr,
                const float *eb = nullptr) {
    const dim3 grid(decnet_xcd_grid(ceil_div(W, 1024), (long)H * B));
    if (k == 3)
        hipLaunchKernelGGL((conv2d_small<CO, 3>), grid, dim3(256), 0, s, in, w, scale, shift, y, Cout, H, W, dil,
                           relu, H * B, epi, ea, eb);
    else
        hipLaunchKernelGGL((conv2d_small<CO, 1>), grid, dim3(256), 0, s, in, w, scale, shift, y, Cout, H, W, dil,
                           relu, H * B, epi, ea, eb);
    return decnet_launch_status();
}

// grid_sample(right, (x - disp) stretched as submodule.py:719-745) -- the warp of Refinement: per-pixel
// disparity, align_corners=False sampling of align_corners=True-normalised coordinates,
// bilinear, zero padding; the arithmetic follows torch's grid_sampler (weights as products of
// differences, taps added in the order nw, ne, sw, se).
__global__ __launch_bounds__(256) void warp_disparity(const float *__restrict__ right,
                                                      const float *__restrict__ disp, float *__restrict__ out,
                                                      int C, int H, int W, int nrows) {
#pragma clang fp contract(off)
    int bx, row;
    if (!decnet_xcd_rows((W + 255) >> 8, nrows, bx, row)) return;        // (two input rows per output row)
    const int x = bx * 256 + threadIdx.x, b = row / H, y = row - b * H;
    if (x >= W) return;
    const size_t plane = (size_t)H * W;
    const float d = disp[(size_t)b * plane + (size_t)y * W + x];
    const float cx = ((float)x - d) / ((float)(W - 1.0) / 2.0f) - 1.0f;
    const float cy = (float)y / ((float)(H - 1.0) / 2.0f) - 1.0f;
    const float ix = ((cx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((cy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = (fx + 1.0f - ix) * (fy + 1.0f - iy), ne = (ix - fx) * (fy + 1.0f - iy);
    const float sw = (fx + 1.0f - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    const float *rb = right + (size_t)b * C * plane;
    float *ob = out + (size_t)b * C * plane + (size_t)y * W + x;
    // clamped tap offsets: every load is issued (8 channels x 4 taps in flight per pass), invalid taps are dropped by
    // the selects -- the guarded form waited for each channel's loads before issuing the next one's (152 us for the
    // [8,8,540,972] warp of stage 3: 1.7 TB/s)
    const size_t o00 = (size_t)(vy0 ? y0 : 0) * W + (vx0 ? x0 : 0), o01 = (size_t)(vy0 ? y0 : 0) * W + (vx1 ? x1 : 0);
    const size_t o10 = (size_t)(vy1 ? y1 : 0) * W + (vx0 ? x0 : 0), o11 = (size_t)(vy1 ? y1 : 0) * W + (vx1 ? x1 : 0);
    for (int c0 = 0; c0 < C; c0 += 8) {
        float t[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float *rp = rb + (size_t)(c0 + e < C ? c0 + e : c0) * plane;
            t[e][0] = rp[o00]; t[e][1] = rp[o01]; t[e][2] = rp[o10]; t[e][3] = rp[o11];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c0 + e >= C) break;
            float v = 0.f;
            if (vy0 && vx0) v += t[e][0] * nw;
            if (vy0 && vx1) v += t[e][1] * ne;
            if (vy1 && vx0) v += t[e][2] * sw;
            if (vy1 && vx1) v += t[e][3] * se;
            ob[(size_t)(c0 + e) * plane] = v;
        }
    }
}

}  // namespace

extern "C" {

size_t decnet_conv2d_packed_floats(int Cin, int Cout, int k, int transposed) {
    if (Cin < 1 || Cout < 1 || Cout > 24 || (k != 1 && k != 3) || (transposed && (k != 3 || Cout > 8))) return 0;
    return (size_t)Cin * k * k * (transposed ? 8 : co_pad(Cout));
}

int decnet_conv2d_pack_weight(const float *w, float *w_packed, int Cin, int Cout, int k, int transposed,
                              void *stream) {
    if (!w || !w_packed) return DECNET_ERR_NULL_POINTER;
    const int n = (int)decnet_conv2d_packed_floats(Cin, Cout, k, transposed);
    if (n == 0) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_weight2d, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, w_packed, Cin,
                       Cout, k * k, transposed ? 8 : co_pad(Cout), transposed);
    return decnet_launch_status();
}

// Tail of DynamicUpsampling.forward (submodule.py:566-589) for down_scale 3: per coarse pixel, nine
// softmaxes over the 3x3 neighbourhood (logits [B,81,h,w]: channel = 9 * sub-position + neighbour) weight
// the replicate-padded neighbours of the coarse disparity; pixel_shuffle and the x3 are the store
// pattern.  One pass over the logits instead of softmax / multiply / sum / pixel_shuffle / scale kernels.
__global__ __launch_bounds__(256) void dynamic_upsample3(const float *__restrict__ logits,
                                                         const float *__restrict__ disp, float *__restrict__ out,
                                                         int h, int w) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const size_t plane = (size_t)h * w;
    const float *dp = disp + (size_t)b * plane;
    float nb[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = min(max(y + ky - 1, 0), h - 1), xx = min(max(x + kx - 1, 0), w - 1);   // ReplicationPad2d(1)
            nb[ky * 3 + kx] = dp[(size_t)yy * w + xx];
        }
    const float *lp = logits + (size_t)b * 81 * plane + (size_t)y * w + x;
    float *op = out + ((size_t)b * 3 * h + 3 * y) * (3 * (size_t)w) + 3 * x;
#pragma unroll
    for (int sy = 0; sy < 3; ++sy)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            float v[9], m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                v[k] = lp[(size_t)((sy * 3 + sx) * 9 + k) * plane];
                m = fmaxf(m, v[k]);
            }
            float sum = 0.f, acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float e = expf(v[k] - m);
                sum += e;
                acc = fmaf(e, nb[k], acc);
            }
            op[(size_t)sy * 3 * w + sx] = acc / sum * 3.0f;
        }
}

// Outputs that the last-level cache cannot keep for their consumer anyway (>= DECNET_NT_MB, default 192 MB against the
// 256 MB Infinity Cache) leave with nontemporal stores: bit 1 of the kernels' relu argument.  Measured at
// [16,8,540,972] (268 MB out): 8 -> 8 3 x 3 0.171 -> 0.141 ms, 1 x 1 0.103 -> 0.082.  DECNET_NT_MB=0: every output, < 0: none.
static int store_policy(double out_bytes) {
    static const double mb = [] { const char *e = getenv("DECNET_NT_MB"); return e ? atof(e) : 192.0; }();
    return mb >= 0 && out_bytes >= mb * 1048576.0 ? 2 : 0;
}

static int conv2d_segs(const Segs &in, const float *w, const float *scale, const float *shift, float *y, int B,
                       int Cout, int H, int W, int k, int dilation, int relu, void *stream, int epi = 0,
                       const float *ea = nullptr, const float *eb = nullptr) {
    if (!w || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Cout < 1 || H < 1 || W < 1 || dilation < 1) return DECNET_ERR_BAD_SHAPE;
    if ((k != 1 && k != 3) || Cout > 24 || H > 65535 || B > 65535 || W > (1 << 28) ||
        (double)B * H * ceil_div(W, 1024) >= 2.0e9) return DECNET_ERR_UNSUPPORTED;
    int cin = 0;
    for (int i = 0; i < in.n; ++i) {
        if (!in.p[i]) return DECNET_ERR_NULL_POINTER;
        if (in.c[i] < 1) return DECNET_ERR_BAD_SHAPE;
        cin += in.c[i];
    }
    if ((double)B * (cin > Cout ? cin : Cout) * H * W >= 9.0e18) return DECNET_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (epi && (Cout != 1 || !ea || (epi == 1 && !eb) || epi < 0 || epi > 2)) return DECNET_ERR_UNSUPPORTED;
    relu = (relu ? 1 : 0) | store_policy(4.0 * B * Cout * H * W);
    // 3 x 3 layers with <= 4 outputs or more than 8 inputs: the fp32 matrix-pipe kernel.  Measured against the packed-FMA
    // kernel at [8,*,540,972] (tools/bench_conv2d_shapes.sh): 16 -> 8 0.149 vs 0.160 ms, 17 -> 8 dilation 3 0.167 vs
    // 0.175, 12 -> 8 0.121 vs 0.129, 8 -> 4 0.045 vs 0.054, 8 -> 3 0.043 vs 0.053; the 8 -> 8 layers (HBM-shaped either
    // way: 0.077 vs 0.072), 3 -> 8 and the 1 x 1 layers stay on the packed-FMA kernel with its eight waves per SIMD.
    // DECNET_CONV2D_SMALL=valu | mfma forces one of them.
    static const int force = [] { const char *e = getenv("DECNET_CONV2D_SMALL"); return !e ? 0 : !strcmp(e, "valu") ? 1 : !strcmp(e, "mfma") ? 2 : 0; }();
    const int cop = co_pad(Cout);
    const bool want_mfma = force == 2 || (force == 0 && k == 3 && (cop <= 4 || cin > 8));
    if (want_mfma && Cout > 1 && (size_t)cin * k * k * cop * 4 <= 64 * 1024 && H <= 65535) {
        const size_t lds = (size_t)cin * k * k * cop * 4;
        const int R = cop <= 12 ? 4 : 2;
        const int tasks_per_img = ceil_div(H, R * dilation) * dilation;
        const long ntasks = (long)tasks_per_img * B;
        if ((double)ntasks * ceil_div(W, 256) < 2.0e9) {
            const dim3 grid(decnet_xcd_grid(ceil_div(W, 256), ntasks));
#define GOM(Q, KK_, R_)                                                                                             \
    hipLaunchKernelGGL((conv2d_f32m<Q, KK_, R_>), grid, dim3(256), lds, s, in, w, scale, shift, y, Cout, H, W, dilation, \
                       relu, (int)ntasks, tasks_per_img, cin)
            if (cop == 4) { if (k == 3) GOM(1, 3, 4); else GOM(1, 1, 4); }
            else if (cop == 8) { if (k == 3) GOM(2, 3, 4); else GOM(2, 1, 4); }
            else if (cop == 12) { if (k == 3) GOM(3, 3, 4); else GOM(3, 1, 4); }   // Refinement's 12-channel layers
            else { if (k == 3) GOM(6, 3, 2); else GOM(6, 1, 2); }
#undef GOM
            return decnet_launch_status();
        }
    }
    if (Cout <= 1) return launch_conv<1>(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, s, epi, ea, eb);
    if (Cout <= 4) return launch_conv<4>(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, s);
    if (Cout <= 8) return launch_conv<8>(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, s);
    if (Cout <= 12) return launch_conv<12>(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, s);
    return launch_conv<24>(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, s);   // co_pad(Cout) = 24
}

int decnet_conv2d_bn_act(const float *x, const float *w, const float *scale, const float *shift, float *y,
                         int B, int Cin, int Cout, int H, int W, int k, int dilation, int relu,
                         void *stream) {
    Segs in{};
    in.p[0] = x; in.c[0] = Cin; in.n = 1;
    return conv2d_segs(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, stream);
}

int decnet_conv2d_cat_bn_act(const float *const *xs, const int *cins, int nseg, const float *w,
                             const float *scale, const float *shift, float *y, int B, int Cout, int H, int W,
                             int k, int dilation, int relu, void *stream) {
    if (!xs || !cins) return DECNET_ERR_NULL_POINTER;
    if (nseg < 1 || nseg > MAXSEG) return DECNET_ERR_UNSUPPORTED;
    Segs in{};
    for (int i = 0; i < nseg; ++i) { in.p[i] = xs[i]; in.c[i] = cins[i]; }
    in.n = nseg;
    return conv2d_segs(in, w, scale, shift, y, B, Cout, H, W, k, dilation, relu, stream);
}

int decnet_conv2d_cat_epilogue(const float *const *xs, const int *cins, int nseg, const float *w, const float *scale,
                               const float *shift, float *y, int B, int H, int W, int k, int dilation, int relu,
                               int epilogue, const float *ea, const float *eb, void *stream) {
    if (!xs || !cins) return DECNET_ERR_NULL_POINTER;
    if (nseg < 1 || nseg > MAXSEG) return DECNET_ERR_UNSUPPORTED;
    Segs in{};
    for (int i = 0; i < nseg; ++i) { in.p[i] = xs[i]; in.c[i] = cins[i]; }
    in.n = nseg;
    return conv2d_segs(in, w, scale, shift, y, B, 1, H, W, k, dilation, relu, stream, epilogue, ea, eb);
}

int decnet_dynamic_upsample3(const float *logits, const float *disp, float *out, int B, int h, int w,
                             void *stream) {
    if (!logits || !disp || !out) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || h < 1 || w < 1) return DECNET_ERR_BAD_SHAPE;
    if (h > 65535 || B > 65535) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dynamic_upsample3, dim3((unsigned)ceil_div(w, 256), (unsigned)h, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, logits, disp, out, h, w);
    return decnet_launch_status();
}

int decnet_warp_disparity(const float *right, const float *disp, float *out, int B, int C, int H, int W,
                          void *stream) {
    if (!right || !disp || !out) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 2 || W < 2) return DECNET_ERR_BAD_SHAPE;
    if (H > 65535 || B > 65535 || (double)B * H * ceil_div(W, 256) >= 2.0e9) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(warp_disparity, dim3(decnet_xcd_grid(ceil_div(W, 256), (long)H * B)), dim3(256), 0,
                       (hipStream_t)stream, right, disp, out, C, H, W, H * B);
    return decnet_launch_status();
}

int decnet_bias_act_inplace(float *y, const float *shift, int B, int C, int H, int W, int relu, void *stream) {
    if (!y || !shift) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 1 || W < 1 || (double)H * W >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    if ((double)B * C > 65535.0) return DECNET_ERR_UNSUPPORTED;
    const int HW = H * W;
    hipLaunchKernelGGL(bias_act_inplace, dim3((unsigned)ceil_div(HW, 1024), (unsigned)(B * C)), dim3(256), 0,
                       (hipStream_t)stream, y, shift, C, HW, relu);
    return decnet_launch_status();
}

int decnet_deconv2d_k3s3_bn_act(const float *x, const float *w, const float *scale, const float *shift,
                                float *y, int B, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    if (Cout > 8 || B > 65535 || 9.0 * H * W >= 2147483648.0) return DECNET_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)ceil_div(H * W, 256), 1u, (unsigned)B);
    hipLaunchKernelGGL((deconv2d_k3s3<8>), grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, Cin,
                       Cout, H, W, relu);
    return decnet_launch_status();
}

int decnet_conv2d_k3s3_bn_act(const float *x, const float *w, const float *scale, const float *shift, float *y,
                              int B, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    if (Cout > 24 || H > 65535 * 3 || B > 65535 || W > (1 << 28)) return DECNET_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 3 + 1, Wo = (W - 1) / 3 + 1;            // floor((H + 2 - 3) / 3) + 1
    const dim3 grid((unsigned)ceil_div(Wo, 256), (unsigned)Ho, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
#define GO(N) hipLaunchKernelGGL((conv2d_k3s3<N>), grid, dim3(256), 0, s, x, w, scale, shift, y, Cin, Cout, H, W, Ho, Wo, relu)
    switch (co_pad(Cout)) {                             // = the co pitch of the packed weights
        case 1: GO(1); break;
        case 4: GO(4); break;
        case 8: GO(8); break;
        case 12: GO(12); break;
        default: GO(24); break;
    }
#undef GO
    return decnet_launch_status();
}

}  // extern "C"
