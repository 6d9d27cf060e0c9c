// decnet_amd/csrc/stage0.hip -- stage-0 dense path on gfx950: cost volume -> Conv3d
// aggregation (f32 MFMA implicit GEMM) -> soft-argmax.
//
// Reference (modules/submodule.py): get_disp_samples :389-390, GetCostVolume :479-522,
// Conv3dUnit :115-123, CostRegNetNoDown :650-662, disparity_regression :766-777.
// The reference materialises [B,C,D,H,W] volumes in NCDHW and runs 8 cuDNN Conv3d +
// 8 BatchNorm + 7 ReLU + softmax/mul/sum kernels.  Here activations are channels-last
// [B,D,H,W,C] so that the implicit-GEMM K dimension (27 taps x Ci) is contiguous, BN and
// ReLU (and the residual add) are the GEMM epilogue, and the 216->1 layer is fused with
// the softmax regression.
//
// Conv3d as implicit GEMM:  Y[M, Co] = sum_{tap, ci} X[shift_tap(m), ci] * Wp[tap, ci, co],
// M = B*D*H*W output positions.  fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, no reduced precision anywhere).
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr int CONV_BN = 224;   // Co tile, 14 MFMA columns-of-16 (216 -> 224, 3.6 % padding)
constexpr int B_PITCH = 240;   // == 16 (mod 32): rows kq, kq+1 land on opposite bank halves
// A tile pitch is BK + 2 (== 2 mod 4): the 32 lanes (row i, k-quad kq < 2) hit 32 distinct banks

// ----------------------------------- cost volume ---------------------------------------
// cost[b,d,y,x,c] = (x >= d ? L[b,c,y,x] : 0) * bilinear(R[b,c]; ix, iy), zero padding.
// Coordinates are formed with the same fp32 operation sequence as the reference + torch:
//   cx = (x-d) / ((W-1)/2) - 1      (submodule.py:497-498)
//   ix = ((cx + 1) * W - 1) / 2     (grid_sample, align_corners=False)
// One workgroup per (b, y): the left row and the two right rows the bilinear taps touch are read
// once, x-contiguous (NCHW), into LDS; every (d, x, c) product is then formed from LDS and
// written c-contiguous (channels-last), so both HBM sides are coalesced.
// CF: the cost function (common.h:decnet_cost); DECNET_COST_CAT writes the masked left value at channel c and the warped
// right value at channel C + c of a 2 C channel volume (submodule.py:514).
template <int CF>
__global__ __launch_bounds__(512) void costvol_ndhwc(const float *__restrict__ left,
                                                     const float *__restrict__ right,
                                                     float *__restrict__ cost, int C, int H,
                                                     int W, int D, int dchunk, int cgn) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int WP = W | 1;                      // odd pitch: column reads across c are conflict-free
    // blockIdx.z picks a group of cgn channels (the "cor" volume is elementwise in c): a workgroup per
    // (row, channel group) stages 3 x cgn x W floats instead of 3 x C x W -- more, smaller workgroups
    const int c_lo = blockIdx.z * cgn, CN = min(cgn, C - c_lo);
    float *Ls = smem;                          // [CN][WP]
    float *R0 = Ls + CN * WP;                  // [CN][WP]  row y0 (zeros if outside)
    float *R1 = R0 + CN * WP;                  // [CN][WP]  row y0 + 1
    const int b = blockIdx.x / H, y = blockIdx.x - b * H;
    const float cy = (float)y / ((float)(H - 1.0) / 2.0f) - 1.0f;
    const float iy = ((cy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fy = floorf(iy);
    const int y0 = (int)fy, y1 = y0 + 1;
    const float wy1 = iy - fy, wy0 = 1.0f - wy1;
    const size_t plane = (size_t)H * W;
    const float *Lb = left + ((size_t)b * C + c_lo) * plane, *Rb = right + ((size_t)b * C + c_lo) * plane;
    // no per-element integer division anywhere: a wave stages one channel row at a time, and later
    // owns one (d, x) output column at a time with its lanes across the channels
    const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    constexpr int U = 9;                       // channel rows in flight per wave (latency, not bandwidth, bounds this)
    for (int x = lane; x < W; x += 64)
        for (int c0 = wave; c0 < CN; c0 += nwaves * U) {
            float l[U], r0[U], r1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * nwaves;
                const bool ok = c < CN;
                l[u] = ok ? Lb[c * plane + (size_t)y * W + x] : 0.f;
                r0[u] = ok && vy0 ? Rb[c * plane + (size_t)y0 * W + x] : 0.f;
                r1[u] = ok && vy1 ? Rb[c * plane + (size_t)y1 * W + x] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * nwaves;
                if (c < CN) {
                    Ls[c * WP + x] = l[u];
                    R0[c * WP + x] = r0[u];
                    R1[c * WP + x] = r1[u];
                }
            }
        }
    __syncthreads();
    // blockIdx.y picks a chunk of disparities (more workgroups than the B*H rows alone)
    const int d_lo = blockIdx.y * dchunk, d_hi = min(D, d_lo + dchunk);
    const int CO = CF == DECNET_COST_CAT ? 2 * C : C;                              // channels of the volume
    float *out = cost + (size_t)b * D * plane * CO + (size_t)y * W * CO + c_lo;   // + d*plane*CO + x*CO + c
    for (int t = d_lo * W + wave; t < d_hi * W; t += nwaves) {
        const int d = t / W, x = t - d * W;                       // wave-uniform
        float cx = (float)(x - d) / ((float)(W - 1.0) / 2.0f) - 1.0f;
        float ix = ((cx + 1.0f) * (float)W - 1.0f) / 2.0f;
        float fx = floorf(ix);
        int x0 = (int)fx, x1 = x0 + 1;
        float wx1 = ix - fx, wx0 = 1.0f - wx1;
        bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        float *o = out + (size_t)d * plane * CO + (size_t)x * CO;
        for (int c = lane; c < CN; c += 64) {
            float l = x >= d ? Ls[c * WP + x] : 0.f;              // submodule.py:506-508
            float r = 0.f;                                        // same tap order as grid_sample
            if (vy0 && vx0) r += R0[c * WP + x0] * w00;
            if (vy0 && vx1) r += R0[c * WP + x1] * w01;
            if (vy1 && vx0) r += R1[c * WP + x0] * w10;
            if (vy1 && vx1) r += R1[c * WP + x1] * w11;
            if (CF == DECNET_COST_CAT) {
                o[c] = l;
                o[C + c] = r;
            } else {
                o[c] = decnet_cost<CF>(l, r);                     // submodule.py:511-530
            }
        }
    }
}

// ----------------------------- Conv3d kernel 1 (conv_pre) ------------------------------
// y[b, co, p] = sum_ci w[co][ci] x[b, ci, p]   (CL = false: x [B][Ci][P], y [B][Co][P]; CL = true: x [B][P][Ci],
// y [B][P][Co]): CostRegNetNoDown.conv_pre of cost_func "cat" (submodule.py:618-619: Conv3d(2 C, C, 1), no bias).  A
// workgroup owns PW_TP = 32 positions: their Ci inputs in LDS, the weights through LDS in chunks of 32 input channels
// (transposed: [k][co]).  Wave = 8 of the positions, lane = 4 consecutive output channels (256 per pass): per input
// channel one 16-byte LDS read of the lane's four weights and two wave-uniform 16-byte reads of the eight inputs feed 32
// fmas (thread = one output channel x 32 positions needed nine reads per 32 fmas and was LDS bound: 48 us against 26 at
// B = 8, C = 216, 20 x 36).  The sum over ci is one fp32 fma chain in channel order.
constexpr int PW_TP = 32, PW_CK = 32, PW_THREADS = 256, PW_WP = PW_THREADS + 4;   // pitch: 16-byte rows, == 4 (mod 32)
// blockIdx.z = 1: the second problem of the same shape (x2, w2, y2): the two halves of conv_pre on the two feature maps
// as one launch (each alone is 184 workgroups on 256 CUs at config 2).
template <bool CL>
__global__ __launch_bounds__(PW_THREADS) void pointwise_conv(const float *__restrict__ x, const float *__restrict__ w,
                                                            float *__restrict__ y, const float *__restrict__ x2,
                                                            const float *__restrict__ w2, float *__restrict__ y2, int Ci,
                                                            int Co, int P, int ldw) {
    if (blockIdx.z) {
        x = x2;
        w = w2;
        y = y2;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                              // [Ci][PW_TP]
    float *Ws = smem + (size_t)Ci * PW_TP;          // [PW_CK][PW_WP]
    const int b = blockIdx.y, p0 = blockIdx.x * PW_TP, np = min(PW_TP, P - p0);
    const float *xb = x + (size_t)b * Ci * P;
    float *yb = y + (size_t)b * Co * P;
    // (loads in batches of eight / thirty-two per thread, all in flight together: with one workgroup per CU a load per
    // loop iteration is a chain of round trips -- 27 for the inputs, 27 per weight chunk -- and was 0.1 ms of a 0.11 ms launch)
    for (int i0 = threadIdx.x; i0 < Ci * PW_TP; i0 += 8 * PW_THREADS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * PW_THREADS;
            if (CL) {
                const int pp = i / Ci, ci = i - pp * Ci;                 // consecutive lanes: consecutive channels
                v[u] = i < Ci * PW_TP && pp < np ? xb[(size_t)(p0 + pp) * Ci + ci] : 0.f;
            } else {
                const int ci = i / PW_TP, pp = i - ci * PW_TP;           // consecutive lanes: consecutive positions
                v[u] = i < Ci * PW_TP && pp < np ? xb[(size_t)ci * P + p0 + pp] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * PW_THREADS;
            if (i < Ci * PW_TP) {
                if (CL) {
                    const int pp = i / Ci, ci = i - pp * Ci;
                    Xs[ci * PW_TP + pp] = v[u];
                } else {
                    Xs[i] = v[u];
                }
            }
        }
    }
    constexpr int WU = PW_CK;                                             // weight elements per thread and chunk
    const int wk = threadIdx.x & (PW_CK - 1), wr0 = threadIdx.x / PW_CK;  // 32 consecutive input channels of one row
    const int cg = threadIdx.x & 63, pg = threadIdx.x >> 6;               // output channels 4 cg .. 4 cg + 3, positions 8 pg ..
    for (int co0 = 0; co0 < Co; co0 += PW_THREADS) {
        const int rows = min(PW_THREADS, Co - co0);
        float acc[4][8], wreg[WU];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
        auto load_w = [&](int c0) {
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int r = wr0 + u * (PW_THREADS / PW_CK);
                wreg[u] = r < rows && c0 + wk < Ci ? w[(size_t)(co0 + r) * ldw + c0 + wk] : 0.f;
            }
        };
        load_w(0);
        for (int c0 = 0; c0 < Ci; c0 += PW_CK) {
            __syncthreads();                                             // Xs written / the previous chunk's Ws read
#pragma unroll
            for (int u = 0; u < WU; ++u) Ws[wk * PW_WP + wr0 + u * (PW_THREADS / PW_CK)] = wreg[u];   // zeros beyond `rows`
            __syncthreads();
            if (c0 + PW_CK < Ci) load_w(c0 + PW_CK);                     // in flight during this chunk's arithmetic
            const int kn = min(PW_CK, Ci - c0);
#pragma unroll 4
            for (int k = 0; k < kn; ++k) {
                const float4 wv = *reinterpret_cast<const float4 *>(Ws + k * PW_WP + 4 * cg);
                const float4 *xr = reinterpret_cast<const float4 *>(Xs + (size_t)(c0 + k) * PW_TP + 8 * pg);
                const float4 xa = xr[0], xc = xr[1];                     // the same address in every lane: broadcast
                const float ws[4] = {wv.x, wv.y, wv.z, wv.w}, xs[8] = {xa.x, xa.y, xa.z, xa.w, xc.x, xc.y, xc.z, xc.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[c][j] = fmaf(ws[c], xs[j], acc[c][j]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int co = co0 + 4 * cg + c;
            if (co < Co) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (8 * pg + j < np) {
                        if (CL) yb[(size_t)(p0 + 8 * pg + j) * Co + co] = acc[c][j];
                        else yb[(size_t)co * P + p0 + 8 * pg + j] = acc[c][j];
                    }
            }
        }
    }
}

// ----------------------------------- weight repack -------------------------------------
// torch [Co][Ci][27] -> [27][Ci][CoP], zero padded in co.
__global__ void pack_weight(const float *__restrict__ w, float *__restrict__ wp, int Co, int Ci,
                            int CoP, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int co = (int)(idx % CoP);
    size_t p = idx / CoP;
    int ci = (int)(p % Ci);
    int tap = (int)(p / Ci);
    wp[idx] = co < Co ? w[((size_t)co * Ci + ci) * 27 + tap] : 0.f;
}

// ------------------------------ Conv3d k3 s1 p1 implicit GEMM --------------------------
// Workgroup: WM x 2 waves (WM = 4: 512 threads, tile 192 x 224; WM = 2: 256 threads, 96 x 224);
// every wave owns a 48 x 112 sub-tile = 3 x 7 MFMA tiles of 16x16 (84 accumulator VGPRs), so two
// waves share each SIMD: while one wave computes addresses, stores its staged operands or sits at
// the barrier, the other keeps the matrix pipe busy.
// K loop: 27 taps x ceil(Ci/BK) chunks.  LDS is double buffered; the next chunk is fetched
// HBM/L2 -> registers with bounds-checked buffer loads (padding taps and the M tail read as 0 by
// pointing their offset out of range, no branches) while the current chunk is on the MFMAs.
// LDS: As[2][BM][BK+2] (k contiguous, as in HBM) | Bs[2][BK][240] (co contiguous).
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z),
                       __int_as_float(v.w));
}

template <int WM, int BK>
__global__ __launch_bounds__(WM * 128) void conv3d_k3_igemm(
    const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ residual, float *__restrict__ y,
    int D, int H, int W, int Ci, int Co, int relu, int M, int x_bytes, int w_bytes) {
    constexpr int THREADS = WM * 128, BM = WM * 48, TM = 3, TN = 7;
    constexpr int A_PITCH = BK + 2;
    constexpr int A_F4 = BM * (BK / 4);
    constexpr int A_PER_T = (A_F4 + THREADS - 1) / THREADS;
    constexpr int B_F4 = BK * (CONV_BN / 4);
    constexpr int B_PER_T = (B_F4 + THREADS - 1) / THREADS;
    constexpr int A_TILE = BM * A_PITCH, B_TILE = BK * B_PITCH;
    constexpr int OOB = 0x7fffffff;     // >= num_records: the buffer load returns zeros

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                 // 2 * A_TILE
    float *Bs = smem + 2 * A_TILE;    // 2 * B_TILE

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kq = lane >> 4;
    const int m_block = blockIdx.x * BM;

    // rows of the A tile this thread stages (the same rows every K step)
    int a_lds[A_PER_T];     // LDS offset (floats) or -1
    int a_off[A_PER_T];     // byte offset of x[pos m][4q]
    int a_tap[A_PER_T];     // bit t set <=> tap t of this output position is inside the volume
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
        int idx = tid + i * THREADS;
        int ml = idx / (BK / 4), q = idx - ml * (BK / 4);
        int m = m_block + ml;
        bool ok = idx < A_F4 && m < M;
        a_lds[i] = idx < A_F4 ? ml * A_PITCH + 4 * q : -1;
        int mm = ok ? m : 0;
        int xx = mm % W; int t = mm / W;
        int yy = t % H; t /= H;
        int dd = t % D;
        a_off[i] = (mm * Ci + 4 * q) * 4;
        int bits = 0;
        if (ok) {
            for (int tap = 0; tap < 27; ++tap) {
                int zd = dd + tap / 9 - 1, zy = yy + (tap / 3) % 3 - 1, zx = xx + tap % 3 - 1;
                if ((unsigned)zd < (unsigned)D && (unsigned)zy < (unsigned)H && (unsigned)zx < (unsigned)W)
                    bits |= 1 << tap;
            }
        }
        a_tap[i] = bits;
    }
    int b_lds[B_PER_T], b_off[B_PER_T];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
        int idx = tid + i * THREADS;
        int kk = idx / (CONV_BN / 4), q = idx - kk * (CONV_BN / 4);
        b_lds[i] = idx < B_F4 ? kk * B_PITCH + 4 * q : -1;
        b_off[i] = idx < B_F4 ? (kk * CONV_BN + 4 * q) * 4 : OOB;
    }

    const int nchunk = (Ci + BK - 1) / BK;
    const int nstep = 27 * nchunk;
    const bool ragged = (Ci % BK) != 0;            // last chunk of a tap is partial

    float4 ra[A_PER_T], rb[B_PER_T];
    auto prefetch = [&](int s) {
        int tap = s / nchunk, ci0 = (s - tap * nchunk) * BK;
        int kd = tap / 9 - 1, kh = (tap / 3) % 3 - 1, kw = tap % 3 - 1;
        int a_step = (((kd * H + kh) * W + kw) * Ci + ci0) * 4;
        int b_step = (tap * Ci + ci0) * CONV_BN * 4;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            bool ok = (a_tap[i] >> tap) & 1;
            if (ragged) ok = ok && (ci0 + (a_lds[i] % A_PITCH) < Ci);
            ra[i] = buf_load4(xr, ok ? a_off[i] + a_step : OOB);
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            bool ok = b_lds[i] >= 0;
            if (ragged) ok = ok && (ci0 + b_lds[i] / B_PITCH < Ci);
            rb[i] = buf_load4(wr, ok ? b_off[i] + b_step : OOB);
        }
    };
    auto stage = [&](int buf) {
        float *a = As + buf * A_TILE, *b = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i)
            if (a_lds[i] >= 0) {      // rows are 8-byte aligned (pitch BK+2), not 16
                *reinterpret_cast<float2 *>(a + a_lds[i]) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2 *>(a + a_lds[i] + 2) = make_float2(ra[i].z, ra[i].w);
            }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<float4 *>(b + b_lds[i]) = rb[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    prefetch(0);
    stage(0);
    __syncthreads();
    const int a_row0 = (wm * 48 + i16) * A_PITCH + kq;
    const int b_col0 = kq * B_PITCH + wn * (CONV_BN / 2) + i16;
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstep) prefetch(s + 1);
        const float *a = As + buf * A_TILE + a_row0;
        const float *b = Bs + buf * B_TILE + b_col0;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[i * 16 * A_PITCH + kk * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[kk * 4 * B_PITCH + j * 16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nstep) stage(buf ^ 1);
        __syncthreads();
    }

    // epilogue: BN (folded scale/shift) -> ReLU -> + residual.  C/D layout of 16x16x4:
    // col = lane & 15, row = 4 * (lane >> 4) + r.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int co = wn * (CONV_BN / 2) + j * 16 + i16;
        if (co >= Co) continue;
        float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m_block + wm * 48 + i * 16 + kq * 4 + r;
                if (m >= M) continue;
                float v = fmaf(acc[i][j][r], sc, sh);
                if (relu) v = fmaxf(v, 0.f);
                size_t o = (size_t)m * Co + co;
                if (residual) v += residual[o];
                y[o] = v;
            }
    }
}

// --------------------- last layer (Ci -> 1) + BN + softmax regression -------------------
// One wave per pixel (b,y,x); lanes stride the channel axis with float4 loads; the D
// regularised costs never leave registers before the soft-argmax.
__global__ __launch_bounds__(256) void conv3d_cout1_softargmax(
    const float *__restrict__ x, const float *__restrict__ w, float scale, float shift,
    float *__restrict__ reg, float *__restrict__ pred, int B, int D, int H, int W, int Ci) {
    extern __shared__ __attribute__((aligned(16))) float ws[];      // [27][Ci]
    for (int i = threadIdx.x; i < 27 * Ci; i += blockDim.x) {
        int tap = i / Ci, ci = i - tap * Ci;
        ws[i] = w[(size_t)ci * 27 + tap];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npix = B * H * W;
    for (int pix = blockIdx.x * 4 + wave; pix < npix; pix += gridDim.x * 4) {
        int xx = pix % W, t = pix / W;
        int yy = t % H, b = t / H;
        float m = -INFINITY, S = 0.f, T = 0.f;
        for (int d = 0; d < D; ++d) {
            float acc = 0.f;
            for (int tap = 0; tap < 27; ++tap) {
                int zd = d + tap / 9 - 1, zy = yy + (tap / 3) % 3 - 1, zx = xx + tap % 3 - 1;
                if ((unsigned)zd >= (unsigned)D || (unsigned)zy >= (unsigned)H ||
                    (unsigned)zx >= (unsigned)W)
                    continue;                                        // wave-uniform
                const float *row = x + ((((size_t)b * D + zd) * H + zy) * W + zx) * Ci;
                const float *wr = ws + tap * Ci;
                for (int c = lane * 4; c < Ci; c += 256) {
                    float4 xv = *reinterpret_cast<const float4 *>(row + c);
                    float4 wv = *reinterpret_cast<const float4 *>(wr + c);
                    acc = fmaf(xv.x, wv.x, acc);
                    acc = fmaf(xv.y, wv.y, acc);
                    acc = fmaf(xv.z, wv.z, acc);
                    acc = fmaf(xv.w, wv.w, acc);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            float cost = fmaf(acc, scale, shift);
            if (reg && lane == 0) reg[(((size_t)b * D + d) * H + yy) * W + xx] = cost;
            float mn = fmaxf(m, cost);
            float r = expf(m - mn), e = expf(cost - mn);             // m = -inf -> r = 0
            S = fmaf(S, r, e);
            T = fmaf(T, r, e * (float)d);
            m = mn;
        }
        if (lane == 0) pred[pix] = T / S;
    }
}

// disparity_regression for arbitrary samples: thread per pixel, coalesced over x.
__global__ void disparity_regression_kernel(const float *__restrict__ cost,
                                            const float *__restrict__ samples,
                                            float *__restrict__ pred, int B, int S, int HW) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    int b = idx / HW, p = idx - b * HW;
    const float *c = cost + (size_t)b * S * HW + p;
    const float *d = samples + (size_t)b * S * HW + p;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, c[(size_t)s * HW]);
    float sum = 0.f, acc = 0.f;
    for (int s = 0; s < S; ++s) {
        float e = expf(c[(size_t)s * HW] - m);
        sum += e;
        acc = fmaf(e, d[(size_t)s * HW], acc);
    }
    pred[idx] = acc / sum;
}

// [B][R][Cn] <-> [B][Cn][R] tiled transpose (32x32 through LDS, both sides coalesced).
__global__ void transpose_inner(const float *__restrict__ src, float *__restrict__ dst, int R,
                                int Cn) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * Cn;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        if (r < R && c < Cn) tile[i][threadIdx.x] = src[base + (size_t)r * Cn + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < Cn) dst[base + (size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

template <int WM, int BK>
int launch_conv(const float *x, const float *wp, const float *scale, const float *shift,
                const float *residual, float *y, int D, int H, int W, int Ci, int Co, int relu,
                int M, hipStream_t stream) {
    constexpr int BM = WM * 48;
    size_t lds = 4 * (size_t)(2 * BM * (BK + 2) + 2 * BK * B_PITCH);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)conv3d_k3_igemm<WM, BK>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    int x_bytes = (int)((size_t)M * Ci * 4), w_bytes = 27 * Ci * CONV_BN * 4;
    hipLaunchKernelGGL((conv3d_k3_igemm<WM, BK>), dim3(ceil_div(M, BM)), dim3(WM * 128), lds, stream,
                       x, wp, scale, shift, residual, y, D, H, W, Ci, Co, relu, M, x_bytes, w_bytes);
    return decnet_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Last Conv3dUnit (Ci -> 1) in two passes that read the activations ONCE (the one-kernel version
// above gathers every input row 27 times):
//   1. cout1_tap_gemm: T[p][tap] = sum_c x[p][c] * w[tap][c] for every input position p -- a
//      [M x Ci] x [Ci x 27] GEMM on the matrix cores (taps padded to 32 = two 16-row MFMA tiles),
//      operands straight from memory to registers with the K-permuted 16-byte-per-lane loads of
//      conv3d_winograd.hip:wino_gemm; the 27 x Ci weights stay in registers.
//   2. cout1_gather_softargmax: cost[d,y,x] = sum_tap T[p + offset(tap)][tap] over the taps inside
//      the volume (T is 32 floats per position: L2 resident), BN scale/shift, then the running
//      softmax expectation over d exactly as conv3d_cout1_softargmax.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <int KC>      // chunks of 16 channels, Ci <= 16*KC
__global__ __launch_bounds__(256) void cout1_tap_gemm(const float *__restrict__ x,
                                                      const float *__restrict__ w,
                                                      float *__restrict__ T, int M, int Ci,
                                                      int x_bytes) {
    constexpr int OOB = 0x7fffffff;
    const int lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // A operand: weights, row = tap (two tiles of 16), lane's k = 16c + 4kq + {0..3}
    f32x4_t wv[2][KC];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tap = t * 16 + i16, k = c * 16 + kq * 4 + e;
                wv[t][c][e] = tap < 27 && k < Ci ? w[(size_t)k * 27 + tap] : 0.f;
            }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    for (int g = wave; g * 16 < M; g += nwaves) {
        const int p = g * 16 + i16;
        f32x4_t xv[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * 16 + kq * 4;
            const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, p < M && k < Ci ? (p * Ci + k) * 4 : OOB, 0, 0);
            xv[c] = f32x4_t{__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)};
        }
        f32x4_t acc[2][2] = {};                         // [tap tile][k parity]: four independent chains
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][c][e], xv[c][e], acc[t][e & 1], 0, 0, 0);
        if (p < M) {
#pragma unroll
            for (int t = 0; t < 2; ++t)                 // rows 4kq + r of tile t = taps, column = position
                *reinterpret_cast<f32x4_t *>(T + (size_t)p * 32 + t * 16 + kq * 4) = acc[t][0] + acc[t][1];
        }
    }
}

__global__ __launch_bounds__(256) void cout1_gather_softargmax(const float *__restrict__ T, float scale,
                                                               float shift, float *__restrict__ reg,
                                                               float *__restrict__ pred, int B, int D,
                                                               int H, int W, int PB) {
    extern __shared__ float costs[];                   // [PB][D]
    const int npix = B * H * W;
    const int pl = threadIdx.x / D, d = threadIdx.x - pl * D;
    const int pix = blockIdx.x * PB + pl;
    if (pl < PB && pix < npix) {
        const int xx = pix % W, t = pix / W;
        const int yy = t % H, b = t / H;
        // all 27 requests before the first use (round 5, from the ISA: `if (inside) acc += T[..]` was 27 serial round
        // trips: a load under a per-lane branch, waited for at once); taps outside the volume read the centre's entry and
        // add 0 -- the same sum in the same order
        float tv[27];
        const size_t centre = ((((size_t)b * D + d) * H + yy) * W + xx) * 32;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int zd = d + tap / 9 - 1, zy = yy + (tap / 3) % 3 - 1, zx = xx + tap % 3 - 1;
            const bool ok = (unsigned)zd < (unsigned)D && (unsigned)zy < (unsigned)H && (unsigned)zx < (unsigned)W;
            tv[tap] = T[ok ? ((((size_t)b * D + zd) * H + zy) * W + zx) * 32 + tap : centre + tap];
        }
        __builtin_amdgcn_sched_barrier(0);
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int zd = d + tap / 9 - 1, zy = yy + (tap / 3) % 3 - 1, zx = xx + tap % 3 - 1;
            const bool ok = (unsigned)zd < (unsigned)D && (unsigned)zy < (unsigned)H && (unsigned)zx < (unsigned)W;
            acc += ok ? tv[tap] : 0.f;
        }
        const float cost = fmaf(acc, scale, shift);
        costs[pl * D + d] = cost;
        if (reg) reg[(((size_t)b * D + d) * H + yy) * W + xx] = cost;
    }
    __syncthreads();
    if ((int)threadIdx.x < PB && (int)(blockIdx.x * PB + threadIdx.x) < npix) {
        float m = -INFINITY, S = 0.f, Tt = 0.f;
        for (int dd = 0; dd < D; ++dd) {
            const float cost = costs[threadIdx.x * D + dd];
            const float mn = fmaxf(m, cost);
            const float r = expf(m - mn), e = expf(cost - mn);       // m = -inf -> r = 0
            S = fmaf(S, r, e);
            Tt = fmaf(Tt, r, e * (float)dd);
            m = mn;
        }
        pred[blockIdx.x * PB + threadIdx.x] = Tt / S;
    }
}

}  // namespace

extern "C" {

int decnet_costvol_forward(const float *left, const float *right, float *cost, int B, int C, int H,
                           int W, int D, void *stream) {
    return decnet_costvol_forward_cf(left, right, cost, B, C, H, W, D, DECNET_COST_COR, stream);
}

int decnet_costvol_forward_cf(const float *left, const float *right, float *cost, int B, int C, int H,
                              int W, int D, int cost_func, void *stream) {
    if (!left || !right || !cost) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 2 || W < 2 || D < 1 || cost_func < DECNET_COST_COR || cost_func > DECNET_COST_SUM)
        return DECNET_ERR_BAD_SHAPE;
    const int CO = cost_func == DECNET_COST_CAT ? 2 * C : C;
    if ((double)B * D * H * W * CO >= 1099511627776.0 || (double)D * W * CO >= 2147483648.0 ||
        (double)B * H >= 2147483648.0)
        return DECNET_ERR_BAD_SHAPE;
    // channel groups of <= 128, equal sizes: 216 -> 2 x 108 with 4 disparities per workgroup measured best
    // (0.031 ms; one workgroup per row with all 216 channels and 8 disparities: 0.046 ms; 4 x 54: 0.044 ms)
    const int groups = ceil_div(C, 128), cgn = ceil_div(C, groups);
    size_t lds = (size_t)3 * cgn * (W | 1) * 4;
    if (lds > DECNET_LDS_BYTES - 1024) return DECNET_ERR_UNSUPPORTED;
    const void *fn = cost_func == DECNET_COST_COR   ? (const void *)costvol_ndhwc<DECNET_COST_COR>
                     : cost_func == DECNET_COST_SSD ? (const void *)costvol_ndhwc<DECNET_COST_SSD>
                     : cost_func == DECNET_COST_CAT ? (const void *)costvol_ndhwc<DECNET_COST_CAT>
                                                    : (const void *)costvol_ndhwc<DECNET_COST_SUM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // disparities per workgroup: the row staging (3 x C x W floats, three dependent load batches) is the
    // expensive part, so a workgroup keeps its rows for as many d as still leaves >= ~1 workgroup per CU
    int dchunk = 1;
    while (dchunk < D && (long)B * H * ceil_div(C, cgn) * ceil_div(D, 2 * dchunk) >= 512) dchunk *= 2;
    const dim3 grid((unsigned)(B * H), (unsigned)ceil_div(D, dchunk), (unsigned)ceil_div(C, cgn));
#define GO(CF)                                                                                                    \
    hipLaunchKernelGGL(costvol_ndhwc<CF>, grid, dim3(512), lds, (hipStream_t)stream, left, right, cost, C, H, W, D, \
                       dchunk, cgn)
    if (cost_func == DECNET_COST_COR) GO(DECNET_COST_COR);
    else if (cost_func == DECNET_COST_SSD) GO(DECNET_COST_SSD);
    else if (cost_func == DECNET_COST_CAT) GO(DECNET_COST_CAT);
    else GO(DECNET_COST_SUM);
#undef GO
    return decnet_launch_status();
}

static int pointwise_launch(const float *x, const float *w, float *y, const float *x2, const float *w2, float *y2, int B,
                            int Ci, int Co, int P, int ldw, int channels_last, void *stream) {
    if (!x || !w || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Ci < 1 || Co < 1 || P < 1 || ldw < Ci) return DECNET_ERR_BAD_SHAPE;
    if (B > 65535 || (double)P * (Ci > Co ? Ci : Co) >= 2147483648.0 * 2) return DECNET_ERR_BAD_SHAPE;
    const size_t lds = ((size_t)Ci * PW_TP + (size_t)PW_CK * PW_WP) * 4;
    if (lds > DECNET_LDS_BYTES - 1024) return DECNET_ERR_UNSUPPORTED;
    const void *fn = channels_last ? (const void *)pointwise_conv<true> : (const void *)pointwise_conv<false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((unsigned)((P + PW_TP - 1) / PW_TP), (unsigned)B, x2 ? 2u : 1u);
    if (channels_last)
        hipLaunchKernelGGL(pointwise_conv<true>, grid, dim3(PW_THREADS), lds, (hipStream_t)stream, x, w, y, x2, w2, y2, Ci, Co,
                           P, ldw);
    else
        hipLaunchKernelGGL(pointwise_conv<false>, grid, dim3(PW_THREADS), lds, (hipStream_t)stream, x, w, y, x2, w2, y2, Ci, Co,
                           P, ldw);
    return decnet_launch_status();
}

int decnet_conv3d_pointwise(const float *x, const float *w, float *y, int B, int Ci, int Co, int P, int ldw,
                            int channels_last, void *stream) {
    return pointwise_launch(x, w, y, nullptr, nullptr, nullptr, B, Ci, Co, P, ldw, channels_last, stream);
}

// two products of one shape in one launch (not in the public header: decnet_stage0_forward_cf's conv_pre halves)
int decnet_conv3d_pointwise_pair(const float *x, const float *w, float *y, const float *x2, const float *w2, float *y2,
                                 int B, int Ci, int Co, int P, int ldw, void *stream) {
    if (!x2 || !w2 || !y2) return DECNET_ERR_NULL_POINTER;
    return pointwise_launch(x, w, y, x2, w2, y2, B, Ci, Co, P, ldw, 0, stream);
}

int decnet_conv3d_packed_cout(int Co) { return Co >= 1 && Co <= CONV_BN ? CONV_BN : -1; }

int decnet_conv3d_pack_weight(const float *w, float *wp, int Co, int Ci, void *stream) {
    if (!w || !wp) return DECNET_ERR_NULL_POINTER;
    if (Co < 1 || Ci < 1) return DECNET_ERR_BAD_SHAPE;
    if (Co > CONV_BN) return DECNET_ERR_UNSUPPORTED;
    size_t total = (size_t)27 * Ci * CONV_BN;
    hipLaunchKernelGGL(pack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, wp, Co, Ci, CONV_BN, total);
    return decnet_launch_status();
}

int decnet_conv3d_bn_act(const float *x, const float *wp, const float *scale, const float *shift,
                         const float *residual, float *y, int B, int D, int H, int W, int Ci,
                         int Co, int relu, void *stream) {
    if (!x || !wp || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1) return DECNET_ERR_BAD_SHAPE;
    if (D > 1023 || H > 1023 || W > 1023) return DECNET_ERR_UNSUPPORTED;
    if (Ci % 4 != 0 || Co > CONV_BN) return DECNET_ERR_UNSUPPORTED;
    double Md = (double)B * D * H * W;
    if (Md * (Ci > Co ? Ci : Co) >= 2147483648.0 * 4) return DECNET_ERR_BAD_SHAPE;
    if (Md >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    int M = (int)Md;
    if (Md * Ci * 4 >= 2147483647.0 || 27.0 * Ci * CONV_BN * 4 >= 2147483647.0)
        return DECNET_ERR_UNSUPPORTED;                 // 32-bit buffer offsets
    // tile height: fewest rounds over the 256 CUs, ties -> the taller tile (more weight reuse,
    // two waves per SIMD)
    long r192 = (ceil_div(M, 192) + 255) / 256 * 192, r96 = (ceil_div(M, 96) + 255) / 256 * 96;
    hipStream_t s = (hipStream_t)stream;
    if (r192 <= r96) {
        if (Ci % 36 == 0)
            return launch_conv<4, 36>(x, wp, scale, shift, residual, y, D, H, W, Ci, Co, relu, M, s);
        return launch_conv<4, 24>(x, wp, scale, shift, residual, y, D, H, W, Ci, Co, relu, M, s);
    }
    return launch_conv<2, 24>(x, wp, scale, shift, residual, y, D, H, W, Ci, Co, relu, M, s);
}

int decnet_conv3d_cout1_softargmax(const float *x, const float *w, float scale, float shift,
                                   float *reg, float *pred, int B, int D, int H, int W, int Ci,
                                   void *stream) {
    if (!x || !w || !pred) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1) return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0) return DECNET_ERR_UNSUPPORTED;
    size_t lds = (size_t)27 * Ci * 4;
    if (lds > DECNET_LDS_BUDGET) return DECNET_ERR_UNSUPPORTED;
    double npix = (double)B * H * W;
    if (npix * D >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    int blocks = (int)((npix + 3) / 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(conv3d_cout1_softargmax, dim3(blocks), dim3(256), lds, (hipStream_t)stream,
                       x, w, scale, shift, reg, pred, B, D, H, W, Ci);
    return decnet_launch_status();
}

/* The same operator in two passes over a workspace of decnet_conv3d_cout1_workspace_floats(B,D,H,W)
 * floats: the activations are read once instead of 27 times. */
size_t decnet_conv3d_cout1_workspace_floats(int B, int D, int H, int W) {
    if (B < 1 || D < 1 || H < 1 || W < 1) return 0;
    return (size_t)B * D * H * W * 32;
}

int decnet_conv3d_cout1_softargmax_ws(const float *x, const float *w, float scale, float shift,
                                      float *reg, float *pred, float *workspace, int B, int D, int H,
                                      int W, int Ci, void *stream) {
    if (!x || !w || !pred || !workspace) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1) return DECNET_ERR_BAD_SHAPE;
    const double Md = (double)B * D * H * W;
    if (Ci % 4 != 0 || Ci > 256 || D > 256 || Md * Ci * 4 >= 2147483647.0) return DECNET_ERR_UNSUPPORTED;
    const int M = (int)Md, x_bytes = (int)(Md * Ci * 4);
    hipStream_t s = (hipStream_t)stream;
    int blocks = ceil_div(ceil_div(M, 16), 4);
    if (blocks > 512) blocks = 512;
    if (Ci <= 224 && Ci > 128)
        hipLaunchKernelGGL((cout1_tap_gemm<14>), dim3(blocks), dim3(256), 0, s, x, w, workspace, M, Ci, x_bytes);
    else if (Ci <= 128)
        hipLaunchKernelGGL((cout1_tap_gemm<8>), dim3(blocks), dim3(256), 0, s, x, w, workspace, M, Ci, x_bytes);
    else
        hipLaunchKernelGGL((cout1_tap_gemm<16>), dim3(blocks), dim3(256), 0, s, x, w, workspace, M, Ci, x_bytes);
    int rc = decnet_launch_status();
    if (rc) return rc;
    const int PB = 256 / D, npix = B * H * W;
    hipLaunchKernelGGL(cout1_gather_softargmax, dim3(ceil_div(npix, PB)), dim3(256), (size_t)PB * D * 4, s,
                       workspace, scale, shift, reg, pred, B, D, H, W, PB);
    return decnet_launch_status();
}

int decnet_disparity_regression(const float *cost, const float *samples, float *pred, int B, int S,
                                int H, int W, void *stream) {
    if (!cost || !samples || !pred) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || S < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    if ((double)B * S * H * W >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    int n = B * H * W;
    hipLaunchKernelGGL(disparity_regression_kernel, dim3(ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, cost, samples, pred, B, S, H * W);
    return decnet_launch_status();
}

static int transpose(const float *src, float *dst, int B, int R, int Cn, void *stream) {
    if (!src || !dst) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || R < 1 || Cn < 1 || B > 65535) return DECNET_ERR_BAD_SHAPE;
    if (ceil_div(R, 32) > 65535) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(transpose_inner, dim3(ceil_div(Cn, 32), ceil_div(R, 32), B), dim3(32, 8), 0,
                       (hipStream_t)stream, src, dst, R, Cn);
    return decnet_launch_status();
}

int decnet_ncdhw_to_ndhwc(const float *src, float *dst, int B, int C, int D, int H, int W,
                          void *stream) {
    if (C < 1 || D < 1 || H < 1 || W < 1 || (double)D * H * W >= 2147483648.0)
        return DECNET_ERR_BAD_SHAPE;
    return transpose(src, dst, B, C, D * H * W, stream);       // [B][C][S] -> [B][S][C]
}

int decnet_ndhwc_to_ncdhw(const float *src, float *dst, int B, int C, int D, int H, int W,
                          void *stream) {
    if (C < 1 || D < 1 || H < 1 || W < 1 || (double)D * H * W >= 2147483648.0)
        return DECNET_ERR_BAD_SHAPE;
    return transpose(src, dst, B, D * H * W, C, stream);       // [B][S][C] -> [B][C][S]
}

}  // extern "C"
