// decnet_amd/csrc/spamat_mfma.hip -- SpaMat / SpaVar forward, banded cost tiles on the
// matrix cores (gfx950).  Replaces get_max_cost + sparse_matching_forward
// (SM_kernel.cu:22-125) and get_max_cost + sparse_var_forward (SV_kernel.cu:22-124), and
// fuses the two the way the model uses them (SparseDenseNetRefinementMask.py:183-192).
//
// Why the matrix cores for an HBM-shaped op: at stage 3 (C=8, D=216) every byte of L/R
// feeds 36 fp32 MACs + 4.5 exp, above the chip's fp32 ridge, so the pass is VALU-bound
// unless the channel dot products leave the VALU.  cost[x'][x] = sum_c R[c][x'] L[c][x]
// over a 32x32 tile of (right pixel, left pixel) IS a K=C matrix product, and
// v_mfma_f32_32x32x2_f32 evaluates it as the same c-ordered fp32 fma chain the reference
// binary runs (exact fp32, no reduced precision), on a pipe that runs beside the VALU.
// The VALU is left with the softmax: max, exp, and the three moment sums.
//
// Work decomposition
//   workgroup (4 waves)  one segment of XT 32-pixel tiles of ONE image row (normally the
//                        whole row: no halo is read twice); L[C][SW], R[C][HALO+SW] and the
//                        right mask (as an additive 0 / -1e30 bias) are staged once in LDS.
//   wave                 one 32-pixel left tile at a time: NT = ceil((D-1)/32)+1 cost tiles
//                        of the disparity band, 16*NT costs per lane kept in registers
//                        (accumulator layout: lane&31 = left pixel, register = right pixel),
//                        then max / exp-sum / variance passes over registers and one
//                        partner-lane exchange.
//   right-mask           applied by one extra K-step of the MFMA chain (adds 0 or -1e30).
//   disparity range      0 <= d < D only needs checking on the diagonal tile and the last
//                        one or two tiles of the band.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

enum { MODE_MAT = 0, MODE_VAR = 1, MODE_FUSED = 2 };

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Cooperative staging of `n` floats of one feature row: dst[j] = src_row[xs + j], 0 outside
// [0, W).  float4 when the row base and xs are 16-byte aligned.
__device__ __forceinline__ void stage_row(float *dst, const float *__restrict__ row, int xs, int n,
                                          int W) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool vec = ((((uintptr_t)row) & 15) == 0) && ((xs & 3) == 0);
    if (vec) {
        for (int j = tid * 4; j < n; j += nt * 4) {          // n is a multiple of 32
            int x = xs + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x >= 0 && x + 3 < W) {
                v = *reinterpret_cast<const float4 *>(row + x);
            } else {
                if (x >= 0 && x < W) v.x = row[x];
                if (x + 1 >= 0 && x + 1 < W) v.y = row[x + 1];
                if (x + 2 >= 0 && x + 2 < W) v.z = row[x + 2];
                if (x + 3 >= 0 && x + 3 < W) v.w = row[x + 3];
            }
            *reinterpret_cast<float4 *>(dst + j) = v;
        }
    } else {
        for (int j = tid; j < n; j += nt) {
            int x = xs + j;
            dst[j] = (x >= 0 && x < W) ? row[x] : 0.f;
        }
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void spamat_fwd_mfma(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int segs_per_row, int XT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Cp = (C + 1) & ~1;
    const int SW = XT * 32, HALO = (NT - 1) * 32, RP = HALO + SW;
    float *Ls = smem;             // [Cp][SW]   Ls[c][j]  = L[c][xs + j]
    float *Rs = Ls + Cp * SW;     // [Cp][RP]   Rs[c][j]  = R[c][xs - HALO + j]
    float *Bi = Rs + Cp * RP;     // [RP]       0 where the right mask is on, else -1e30

    const int seg = blockIdx.x % segs_per_row, row = blockIdx.x / segs_per_row;
    const int b = row / H, y = row - b * H;
    const int xs = seg * SW;
    const size_t plane = (size_t)H * W;
    {
        const float *lrow = ref + ((size_t)b * C * H + y) * W;
        const float *rrow = tar + ((size_t)b * C * H + y) * W;
        for (int c = 0; c < Cp; ++c) {
            if (c < C) {
                stage_row(Ls + c * SW, lrow + c * plane, xs, SW, W);
                stage_row(Rs + c * RP, rrow + c * plane, xs - HALO, RP, W);
            } else {                                   // odd C: zero pad channel
                for (int j = threadIdx.x; j < SW; j += blockDim.x) Ls[c * SW + j] = 0.f;
                for (int j = threadIdx.x; j < RP; j += blockDim.x) Rs[c * RP + j] = 0.f;
            }
        }
        const float *trow = tmask + (size_t)row * W;
        for (int j = threadIdx.x; j < RP; j += blockDim.x) {
            int x = xs - HALO + j;
            Bi[j] = (x >= 0 && x < W && trow[x] != 0.f) ? 0.f : NEG_BIG;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int dl = j - 4 * h;                       // d = 32*m + dl - (r&3) - 8*(r>>2)
    const float dlf = (float)dl;

    for (int xt = wave; xt < XT; xt += 4) {
        const int x0 = xs + xt * 32;
        if (x0 >= W) break;
        const int x = x0 + j;
        const size_t pix = (size_t)row * W + x;
        const bool inside = x < W;
        const float rm = inside ? rmask[pix] : 0.f;
        if (__ballot(rm != 0.f) == 0ull) {          // no active left pixel in this tile
            if (inside && h == 0) {
                if (MODE != MODE_VAR) out[pix] = 0.f;
                if (MODE != MODE_MAT) var_out[pix] = 0.f;
                sum_sim[pix] = 0.f;
                max_cost[pix] = 0.f;
            }
            continue;
        }
        const int nact = min(NT, x0 / 32 + 1);      // tiles reaching x' >= 0 (x0 % 32 == 0)

        f32x16 acc[NT];
        // ---- banded costs on the matrix cores: the c-ordered fp32 fma chain of
        //      SM_kernel.cu:52-55, then + (0 | -1e30) for the right mask (:48) --------------
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            f32x16 a16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (m < nact) {
                const float *ap = Rs + h * RP + (HALO + xt * 32 - 32 * m) + j;
                const float *bp = Ls + h * SW + xt * 32 + j;
                for (int s = 0; s < Cp; s += 2)
                    a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * RP], bp[s * SW], a16, 0, 0, 0);
                const float ab = h == 0 ? Bi[HALO + xt * 32 - 32 * m + j] : 0.f;
                const float bb = h == 0 ? 1.f : 0.f;
                a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(ab, bb, a16, 0, 0, 0);
            } else {                                  // tile entirely left of the image
#pragma unroll
                for (int r = 0; r < 16; ++r) a16[r] = NEG_BIG;
            }
            acc[m] = a16;
        }
        // ---- pass 1: disparity range 0 <= d < min(D, x+1) (SM_kernel.cu:42,46; only the
        //      diagonal and the last tiles can violate it), then
        //      max_cost = max(1e-6, max_d cost_d)  (SM_kernel.cu:45-59) ----------------------
        float mx0 = 0.000001f, mx1 = 0.000001f;
        int dlv = dl;
        asm volatile("" : "+v"(dlv));   // opaque per tile: otherwise LICM hoists all 16*NT range
                                        // compares out of the tile loop and spills 2*16*NT SGPRs
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            if (m == 0 || 32 * m + 31 >= D) {
                asm volatile("" ::: "memory");      // keep this a real (scalar) branch: without
                                                    // it the 16 compares are if-converted into
                                                    // EVERY tile (+2 VALU ops per cost)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = 32 * m + dlv - ((r & 3) + 8 * (r >> 2));
                    acc[m][r] = (unsigned)d >= (unsigned)D ? NEG_BIG : acc[m][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                mx0 = fmaxf(mx0, fmaxf(acc[m][r], acc[m][r + 1]));
                mx1 = fmaxf(mx1, fmaxf(acc[m][r + 2], acc[m][r + 3]));
            }
        }
        float mx = fmaxf(mx0, mx1);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // ---- pass 2: e = exp(cost - max); S = sum e; T = sum e*d  (SM_kernel.cu:100-122) ---
        float S0 = 0.f, S1 = 0.f, T0 = 0.f, T1 = 0.f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float k0 = (float)(32 * m - ((r & 3) + 8 * (r >> 2)));
                const float k1 = (float)(32 * m - (((r + 1) & 3) + 8 * ((r + 1) >> 2)));
                float e0 = fast_exp2((acc[m][r] - mx) * LOG2E);
                float e1 = fast_exp2((acc[m][r + 1] - mx) * LOG2E);
                acc[m][r] = e0;
                acc[m][r + 1] = e1;
                S0 += e0;
                S1 += e1;
                if (MODE != MODE_VAR) {
                    T0 = fmaf(e0, k0, T0);
                    T1 = fmaf(e1, k1, T1);
                }
            }
        }
        float Sl = S0 + S1;
        float Tl = fmaf(dlf, Sl, T0 + T1);           // d = k + dl
        float S = Sl + __shfl_xor(Sl, 32) + 0.000001f;
        float mu;
        if (MODE == MODE_VAR) {
            mu = inside ? disparity[pix] : 0.f;
        } else {
            float T = Tl + __shfl_xor(Tl, 32) + 0.000001f;
            mu = T / S;
        }
        // ---- pass 3: V = sum e*(d-mu)^2  (SV_kernel.cu:100-121) ---------------------------
        float var = 0.f;
        if (MODE != MODE_MAT) {
            const float c0 = dlf - mu;
            float V0 = 0.f, V1 = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float k0 = (float)(32 * m - ((r & 3) + 8 * (r >> 2)));
                    const float k1 = (float)(32 * m - (((r + 1) & 3) + 8 * ((r + 1) >> 2)));
                    float d0 = k0 + c0, d1 = k1 + c0;
                    V0 = fmaf(acc[m][r] * d0, d0, V0);
                    V1 = fmaf(acc[m][r + 1] * d1, d1, V1);
                }
            }
            float Vl = V0 + V1;
            var = (Vl + __shfl_xor(Vl, 32) + 0.000001f) / S;
        }
        if (inside && h == 0) {
            const bool on = rm != 0.f;
            if (MODE != MODE_VAR) out[pix] = on ? mu : 0.f;
            if (MODE != MODE_MAT) var_out[pix] = on ? var : 0.f;
            sum_sim[pix] = on ? S : 0.f;
            max_cost[pix] = on ? mx : 0.f;
        }
    }
}

template <int NT>
int launch_nt(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
              const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost,
              int B, int C, int H, int W, int D, hipStream_t stream) {
    const int Cp = (C + 1) & ~1, HALO = (NT - 1) * 32;
    const int xt_row = ceil_div(W, 32);
    // LDS floats: Cp*(2*SW + HALO) + HALO + SW ; aim for two workgroups per CU
    auto bytes = [&](int xt) { return 4 * ((size_t)Cp * (2 * 32 * xt + HALO) + HALO + 32 * xt); };
    int XT = xt_row;
    const size_t budget2 = (DECNET_LDS_BYTES - 2048) / 2;
    if (bytes(XT) > budget2) {
        // split the row into equal segments that fit two-per-CU, else one-per-CU
        int segs = 2;
        while (segs < xt_row && bytes(ceil_div(xt_row, segs)) > budget2) ++segs;
        XT = ceil_div(xt_row, segs);
        if (bytes(XT) > budget2) {
            XT = xt_row;
            while (XT > 1 && bytes(XT) > DECNET_LDS_BYTES - 1024) --XT;
            if (bytes(XT) > DECNET_LDS_BYTES - 1024) return DECNET_ERR_UNSUPPORTED;
        }
    }
    const int segs = ceil_div(xt_row, XT);
    const size_t lds = bytes(XT);
    dim3 grid((unsigned)((size_t)B * H * segs)), block(256);
#define LAUNCH(M)                                                                                  \
    do {                                                                                           \
        if (lds > 64 * 1024) {                                                                     \
            hipError_t e = hipFuncSetAttribute((const void *)spamat_fwd_mfma<NT, M>,               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                    \
        }                                                                                          \
        hipLaunchKernelGGL((spamat_fwd_mfma<NT, M>), grid, block, lds, stream, ref, tar, rmask,    \
                           tmask, disparity, out, var_out, sum_sim, max_cost, C, H, W, D, segs, XT); \
    } while (0)
    if (mode == MODE_MAT) LAUNCH(MODE_MAT);
    else if (mode == MODE_VAR) LAUNCH(MODE_VAR);
    else LAUNCH(MODE_FUSED);
#undef LAUNCH
    return decnet_launch_status();
}

}  // namespace

// mode: 0 SpaMat, 1 SpaVar, 2 fused.  Returns DECNET_ERR_UNSUPPORTED when the band needs more
// than 12 tiles (max_disp > 353) or the tile does not fit LDS; the caller then uses the
// row-tile kernel.
int decnet_mfma_forward(int mode, const float *ref, const float *tar, const float *rmask,
                        const float *tmask, const float *disparity, float *out, float *var_out,
                        float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                        hipStream_t stream) {
    const int D = max_disp;
    const int need = D <= 1 ? 1 : (D - 1 + 31) / 32 + 1;
#define GO(N)                                                                                     \
    return launch_nt<N>(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim, max_cost, \
                        B, C, H, W, D, stream)
    if (need <= 2) GO(2);
    if (need <= 3) GO(3);
    if (need <= 4) GO(4);
    if (need <= 6) GO(6);
    if (need <= 8) GO(8);
    if (need <= 10) GO(10);
    if (need <= 12) GO(12);
#undef GO
    return DECNET_ERR_UNSUPPORTED;
}
