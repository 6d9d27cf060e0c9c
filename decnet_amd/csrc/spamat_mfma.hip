// decnet_amd/csrc/spamat_mfma.hip -- SpaMat / SpaVar forward, banded cost tiles on the
// matrix cores (gfx950).  Replaces get_max_cost + sparse_matching_forward
// (SM_kernel.cu:22-125) and get_max_cost + sparse_var_forward (SV_kernel.cu:22-124), and
// fuses the two the way the model uses them (SparseDenseNetRefinementMask.py:183-192).
//
// Why the matrix cores for an HBM-shaped op: at stage 3 (C=8, D=216) every byte of L/R
// feeds 36 fp32 MACs + 4.5 exp, above the chip's fp32 ridge, so the pass is VALU-bound
// unless the channel dot products leave the VALU.  cost[x'][x] = sum_c R[c][x'] L[c][x]
// over a 16x16 tile of (right pixel, left pixel) IS a K=C matrix product, and
// v_mfma_f32_16x16x4_f32 evaluates it as the same c-ordered fp32 fma chain the reference
// binary runs (exact fp32, no reduced precision), on a pipe that runs beside the VALU.
// The VALU is left with the softmax: max, exp, and the moment sums.
//
// Work decomposition
//   workgroup (8 waves)  one segment of XT 16-pixel tiles of ONE image row (normally the
//                        whole row: no halo is read twice).  R[C][HALO+SW] and the right
//                        mask (as an additive 0 / -1e30 bias) are staged once in LDS with
//                        16-byte row loads, all issued before the first is waited for.
//                        L is NOT staged: every left element is used by exactly one tile,
//                        so it goes HBM -> registers (prefetched one tile ahead).
//   wave                 one 16-pixel left tile at a time: NT = ceil((D-1)/16)+1 cost tiles
//                        of the disparity band, 4*NT costs per lane kept in registers
//                        (accumulator layout: lane&15 = left pixel, 4*(lane>>4)+reg = right
//                        pixel), then max / exp-sum / variance passes over registers and a
//                        4-lane exchange.  <= 128 VGPRs so that 4 waves share a SIMD: one
//                        wave alone issues a VALU op only every 4 cycles, and the matrix
//                        pipe of one wave runs under the VALU passes of the others.
//   right mask           one extra K-step of the MFMA chain adds 0 or -1e30 (SM_kernel.cu:48).
//   disparity range      0 <= d < D only needs checking on the diagonal tile and the last
//                        one or two tiles of the band (SM_kernel.cu:42).
//
// This file is compiled with -fno-honor-nans (see build.py): without it every fmaxf on an
// MFMA result costs an extra canonicalising v_max; NaN inputs give NaN/garbage rows either way.
#include "common.h"

#ifndef DECNET_ABLATE
#define DECNET_ABLATE 0   // 1: skip MFMAs, 2: skip softmax passes, 3: both (diagnostic builds only,
#endif                    // tools/ablate_spamat.sh; results are wrong by construction)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

enum { MODE_MAT = 0, MODE_VAR = 1, MODE_FUSED = 2 };

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int THREADS = 512;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 4 consecutive floats row[x .. x+3], zeros outside [0, W); one 16-byte load when possible.
__device__ __forceinline__ float4 load4(const float *__restrict__ row, int x, int W, bool aligned) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aligned && x >= 0 && x + 3 < W) {
        v = *reinterpret_cast<const float4 *>(row + x);
    } else {
        if (x >= 0 && x < W) v.x = row[x];
        if (x + 1 >= 0 && x + 1 < W) v.y = row[x + 1];
        if (x + 2 >= 0 && x + 2 < W) v.z = row[x + 2];
        if (x + 3 >= 0 && x + 3 < W) v.w = row[x + 3];
    }
    return v;
}

// LDS layout (floats).  The channel pitch is == 16 (mod 32) so that the four channel rows an
// MFMA operand fetch touches (lanes 0-15 / 16-31 / 32-47 / 48-63) never share a bank.
struct Layout {
    int SW, HALO, RP, Cq;      // Cq = channels rounded up to a multiple of 4
    int offR, offB, total;
};
__host__ __device__ inline Layout make_layout(int C, int NT, int XT) {
    Layout l;
    l.SW = XT * 16;
    l.HALO = (NT - 1) * 16;
    l.Cq = (C + 3) & ~3;
    l.RP = ((l.HALO + l.SW + 31) & ~31) + 16;
    l.offR = 0;
    l.offB = l.offR + l.Cq * l.RP;
    l.total = l.offB + l.RP;
    return l;
}

// KQ = number of K=4 channel steps when known at compile time (C <= 4*KQ), 0 = runtime loop.
template <int NT, int MODE, int KQ>
__global__ __launch_bounds__(THREADS, 4) void spamat_fwd_mfma(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int segs_per_row, int XT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Layout lo = make_layout(C, NT, XT);
    float *Rs = smem + lo.offR;    // [Cq][RP]   Rs[c][j] = R[c][xs - HALO + j]
    float *Bi = smem + lo.offB;    // [RP]       0 where the right mask is on, else -1e30
    const int SW = lo.SW, HALO = lo.HALO, RP = lo.RP;
    const int kq_n = KQ ? KQ : lo.Cq / 4;

    const int seg = blockIdx.x % segs_per_row, row = blockIdx.x / segs_per_row;
    const int b = row / H, y = row - b * H;
    const int xs = seg * SW;
    const size_t plane = (size_t)H * W;
    const float *lrow = ref + ((size_t)b * C * H + y) * W;
    {
        // stage R and the mask bias: every load of an 8-channel group is issued before the
        // first LDS store (one HBM round trip per 8 channels instead of one per row)
        const float *rrow = tar + ((size_t)b * C * H + y) * W;
        const float *trow = tmask + (size_t)row * W;
        const int nR = HALO + SW;
        const bool al = ((((uintptr_t)rrow) | ((uintptr_t)(plane * 4))) & 15) == 0;
        const bool alm = (((uintptr_t)trow) & 15) == 0;
        for (int j = threadIdx.x * 4; j < nR; j += THREADS * 4) {
            const int x = xs - HALO + j;
            float4 tv = load4(trow, x, W, alm);
            for (int c0 = 0; c0 < lo.Cq; c0 += 8) {       // Cq % 4 == 0
                float4 v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[c] = c0 + c < C ? load4(rrow + (size_t)(c0 + c) * plane, x, W, al)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c0 + c < lo.Cq) *reinterpret_cast<float4 *>(Rs + (c0 + c) * RP + j) = v[c];
            }
            float4 bv4;
            bv4.x = (x >= 0 && x < W && tv.x != 0.f) ? 0.f : NEG_BIG;
            bv4.y = (x + 1 >= 0 && x + 1 < W && tv.y != 0.f) ? 0.f : NEG_BIG;
            bv4.z = (x + 2 >= 0 && x + 2 < W && tv.z != 0.f) ? 0.f : NEG_BIG;
            bv4.w = (x + 3 >= 0 && x + 3 < W && tv.w != 0.f) ? 0.f : NEG_BIG;
            *reinterpret_cast<float4 *>(Bi + j) = bv4;
        }
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int dl = j - 4 * q;                       // d = 16*m + dl - r
    const float dlf = (float)dl;
    const float bb = q == 0 ? 1.f : 0.f;            // B operand of the mask K-step

    // left operand + left mask of this wave's first tile (HBM -> registers)
    float bv[KQ ? KQ : 1];
    float rm = 0.f;
    auto fetch_left = [&](int xt, float (&dst)[KQ ? KQ : 1], float &m) {
        const int x = xs + xt * 16 + j;
        const bool ok = xt < XT && x < W;
        m = ok ? rmask[(size_t)row * W + x] : 0.f;
        if (KQ) {
#pragma unroll
            for (int s = 0; s < KQ; ++s)
                dst[s] = (ok && 4 * s + q < C) ? lrow[(size_t)(4 * s + q) * plane + x] : 0.f;
        }
    };
    fetch_left(wave, bv, rm);
    __syncthreads();

    for (int xt = wave; xt < XT; xt += THREADS / 64) {
        const int x0 = xs + xt * 16;
        if (x0 >= W) break;
        const int x = x0 + j;
        const size_t pix = (size_t)row * W + x;
        const bool inside = x < W;
        const float rm_cur = rm;
        float bcur[KQ ? KQ : 1];
#pragma unroll
        for (int s = 0; s < (KQ ? KQ : 1); ++s) bcur[s] = bv[s];
        fetch_left(xt + THREADS / 64, bv, rm);      // prefetch the next tile's left operand
        if (__ballot(rm_cur != 0.f) == 0ull) {      // no active left pixel in this tile
            if (inside && q == 0) {
                if (MODE != MODE_VAR) out[pix] = 0.f;
                if (MODE != MODE_MAT) var_out[pix] = 0.f;
                sum_sim[pix] = 0.f;
                max_cost[pix] = 0.f;
            }
            continue;
        }

        // ---- banded costs on the matrix cores: the c-ordered fp32 fma chain of
        //      SM_kernel.cu:52-55, then + (0 | -1e30) for the right mask (:48).  Tiles that lie
        //      left of the image read the zero / -1e30 padding and come out as -1e30. -------
        f32x4 acc[NT];
        const float *ap = Rs + q * RP + (HALO + xt * 16) + j;
        const float *bi = Bi + (HALO + xt * 16) + j;
#if DECNET_ABLATE == 1 || DECNET_ABLATE == 3      // timing-only build: no matrix-core work
        if (KQ) {
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const float ab = q == 0 ? bi[-16 * m] : 0.f;
                acc[m] = f32x4{ap[-16 * m], ap[4 * RP - 16 * m], bcur[0] + ab, bcur[KQ - 1]};
            }
        } else
#endif
        if (KQ) {
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                f32x4 a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[-16 * m], bcur[0],
                                                                 f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < KQ; ++s)
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP - 16 * m], bcur[s], a4, 0, 0, 0);
                const float ab = q == 0 ? bi[-16 * m] : 0.f;
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab, bb, a4, 0, 0, 0);
            }
        } else {
            const float *bp = lrow + (size_t)q * plane + x;     // generic C: left operand from L2/HBM
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < kq_n; ++s) {
                    const float lv = (inside && 4 * s + q < C) ? bp[(size_t)4 * s * plane] : 0.f;
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP - 16 * m], lv, a4, 0, 0, 0);
                }
                const float ab = q == 0 ? bi[-16 * m] : 0.f;
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab, bb, a4, 0, 0, 0);
            }
        }

#if DECNET_ABLATE == 2 || DECNET_ABLATE == 3      // timing-only build: no softmax passes
        {
            float keep = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) asm volatile("" :: "v"(acc[m]));
            if (inside && q == 0) { out[pix] = keep; var_out[pix] = keep; sum_sim[pix] = keep; max_cost[pix] = rm_cur; }
            continue;
        }
#endif
        // ---- pass 1: disparity range 0 <= d < min(D, x+1) (SM_kernel.cu:42,46; only the
        //      diagonal and the last tiles can violate it; x' >= 0 is covered by the padding),
        //      then max_cost = max(1e-6, max_d cost_d)  (SM_kernel.cu:45-59) -----------------
        int dlv = dl;
        asm volatile("" : "+v"(dlv));   // opaque per tile: otherwise LICM hoists every range
                                        // compare out of the tile loop and spills their masks
        float mx0 = 0.000001f, mx1 = 0.000001f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            if (m == 0 || 16 * m + 15 >= D) {
                asm volatile("" ::: "memory");      // keep a real scalar branch (no if-conversion)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * m + dlv - r;
                    acc[m][r] = (unsigned)d >= (unsigned)D ? NEG_BIG : acc[m][r];
                }
            }
            mx0 = fmaxf(fmaxf(mx0, acc[m][0]), acc[m][1]);
            mx1 = fmaxf(fmaxf(mx1, acc[m][2]), acc[m][3]);
        }
        float mx = fmaxf(mx0, mx1);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));

        // ---- pass 2: e = exp(cost - max); S = sum e; T = sum e*d  (SM_kernel.cu:100-122) ---
        float S0 = 0.f, S1 = 0.f, T0 = 0.f, T1 = 0.f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                float e0 = fast_exp2((acc[m][r] - mx) * LOG2E);
                float e1 = fast_exp2((acc[m][r + 1] - mx) * LOG2E);
                acc[m][r] = e0;
                acc[m][r + 1] = e1;
                S0 += e0;
                S1 += e1;
                if (MODE != MODE_VAR) {
                    T0 = fmaf(e0, (float)(16 * m - r), T0);
                    T1 = fmaf(e1, (float)(16 * m - r - 1), T1);
                }
            }
        }
        float Sl = S0 + S1;
        float Tl = fmaf(dlf, Sl, T0 + T1);           // d = (16m - r) + dl
        Sl += __shfl_xor(Sl, 16);
        Sl += __shfl_xor(Sl, 32);
        const float S = Sl + 0.000001f;
        float mu;
        if (MODE == MODE_VAR) {
            mu = inside ? disparity[pix] : 0.f;
        } else {
            Tl += __shfl_xor(Tl, 16);
            Tl += __shfl_xor(Tl, 32);
            mu = (Tl + 0.000001f) / S;
        }

        // ---- pass 3: V = sum e*(d-mu)^2  (SV_kernel.cu:100-121) ---------------------------
        float var = 0.f;
        if (MODE != MODE_MAT) {
            const float c0 = dlf - mu;
            float V0 = 0.f, V1 = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float d0 = (float)(16 * m - r) + c0, d1 = (float)(16 * m - r - 1) + c0;
                    V0 = fmaf(acc[m][r] * d0, d0, V0);
                    V1 = fmaf(acc[m][r + 1] * d1, d1, V1);
                }
            }
            float Vl = V0 + V1;
            Vl += __shfl_xor(Vl, 16);
            Vl += __shfl_xor(Vl, 32);
            var = (Vl + 0.000001f) / S;
        }
        if (inside && q == 0) {
            const bool on = rm_cur != 0.f;
            if (MODE != MODE_VAR) out[pix] = on ? mu : 0.f;
            if (MODE != MODE_MAT) var_out[pix] = on ? var : 0.f;
            sum_sim[pix] = on ? S : 0.f;
            max_cost[pix] = on ? mx : 0.f;
        }
    }
}

template <int NT, int KQ>
int launch_nt(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
              const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost,
              int B, int C, int H, int W, int D, hipStream_t stream) {
    const int xt_row = ceil_div(W, 16);
    auto bytes = [&](int xt) { return (size_t)4 * make_layout(C, NT, xt).total; };
    // whole row per workgroup when two workgroups (16 waves) still fit a CU's LDS; otherwise
    // equal segments that do; otherwise whatever fits once.
    const size_t budget2 = (DECNET_LDS_BYTES - 2048) / 2, budget1 = DECNET_LDS_BYTES - 1024;
    int XT = xt_row;
    if (bytes(XT) > budget2) {
        int segs = 2;
        while (segs < xt_row && bytes(ceil_div(xt_row, segs)) > budget2) ++segs;
        XT = ceil_div(xt_row, segs);
        if (bytes(XT) > budget2) {
            XT = xt_row;
            while (XT > 1 && bytes(XT) > budget1) --XT;
            if (bytes(XT) > budget1) return DECNET_ERR_UNSUPPORTED;
        }
    }
    XT = (XT + 1) & ~1;                               // segment starts stay 32-float aligned
    const int segs = ceil_div(xt_row, XT);
    const size_t lds = bytes(XT);
    if (lds > budget1) return DECNET_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((size_t)B * H * segs)), block(THREADS);
#define LAUNCH(M)                                                                                  \
    do {                                                                                           \
        if (lds > 64 * 1024) {                                                                     \
            hipError_t e = hipFuncSetAttribute((const void *)spamat_fwd_mfma<NT, M, KQ>,           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                    \
        }                                                                                          \
        hipLaunchKernelGGL((spamat_fwd_mfma<NT, M, KQ>), grid, block, lds, stream, ref, tar, rmask, \
                           tmask, disparity, out, var_out, sum_sim, max_cost, C, H, W, D, segs, XT); \
    } while (0)
    if (mode == MODE_MAT) LAUNCH(MODE_MAT);
    else if (mode == MODE_VAR) LAUNCH(MODE_VAR);
    else LAUNCH(MODE_FUSED);
#undef LAUNCH
    return decnet_launch_status();
}

template <int NT>
int launch_kq(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
              const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost,
              int B, int C, int H, int W, int D, hipStream_t stream) {
#define GO(K)                                                                                      \
    return launch_nt<NT, K>(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim,        \
                            max_cost, B, C, H, W, D, stream)
    if (C <= 8 && C > 4) GO(2);        // stage 3 of the shipped network (C = 8)
    if (C <= 24 && C > 20) GO(6);      // stage 2 (C = 24)
    GO(0);                             // anything else, incl. stage 1 (C = 72): runtime K loop,
                                       // left operand read straight from L2/HBM per K-step
#undef GO
}

}  // namespace

// mode: 0 SpaMat, 1 SpaVar, 2 fused.  Returns DECNET_ERR_UNSUPPORTED when the band needs more
// than 20 tiles (max_disp > 305) or a tile does not fit LDS; the dispatcher in capi.hip then
// uses the row-tile kernel.
int decnet_mfma_forward(int mode, const float *ref, const float *tar, const float *rmask,
                        const float *tmask, const float *disparity, float *out, float *var_out,
                        float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                        hipStream_t stream) {
    const int D = max_disp;
    const int need = D <= 1 ? 1 : (D - 1 + 15) / 16 + 1;
#define GO(N)                                                                                     \
    return launch_kq<N>(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim, max_cost, \
                        B, C, H, W, D, stream)
    if (need <= 3) GO(3);       // D <= 32   (stage 1: 24, 30)
    if (need <= 6) GO(6);       // D <= 80   (stage 2: 72)
    if (need <= 8) GO(8);       // D <= 112  (stage 2 at max_disp 270: 90)
    if (need <= 11) GO(11);     // D <= 160
    if (need <= 15) GO(15);     // D <= 224  (stage 3: 216)
    if (need <= 18) GO(18);     // D <= 272  (stage 3 at max_disp 270)
    if (need <= 20) GO(20);     // D <= 304
#undef GO
    return DECNET_ERR_UNSUPPORTED;
}
