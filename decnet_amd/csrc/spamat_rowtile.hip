// decnet_amd/csrc/spamat_rowtile.hip -- SpaMat / SpaVar, LDS row-tile kernels (gfx950).
//
// What the reference does (modules/SparseMatching/src/SM_kernel.cu, SparseVar/src/SV_kernel.cu):
// one thread per pixel, every operand fetched from global memory for every disparity,
// the channel dot product recomputed in each of 2 (forward) or 2*C (backward) passes.
//
// What these kernels do instead: one workgroup owns TW consecutive pixels of ONE image
// row (b, y).  The left tile L[C][TW], the right tile with its disparity halo
// R[C][TW + D-1] and the right mask row are staged ONCE through LDS with coalesced row
// loads; every candidate (x, d) then reads LDS only.  HBM traffic per launch is the
// compulsory 4*B*H*W*(2C + planes) bytes plus the halo re-reads (which hit L2: the
// neighbouring tile of the same row is resident at the same time).
//
// The backward kernels compute each candidate's softmax weight ONCE into a per-thread
// LDS column Wt[d][tid] and then reduce over d for every channel -- O(C) work per
// candidate instead of the reference's O(C^2).
//
// Arithmetic follows the reference statement by statement (floors, 1e-6 seeds, d order,
// fmaf chain over channels as nvcc contracts it) so results agree with oracle/ to
// rounding; see tests/test_spamat_gpu.py for the stated tolerances.
#include "common.h"

namespace {

enum { MODE_MAT = 0, MODE_VAR = 1, MODE_FUSED = 2 };

struct RowTile {
    int b, y, x0, row;
};

__device__ __forceinline__ RowTile row_tile(int tiles_per_row, int H, int TW) {
    RowTile t;
    int tile = blockIdx.x % tiles_per_row;
    t.row = blockIdx.x / tiles_per_row;
    t.b = t.row / H;
    t.y = t.row - t.b * H;
    t.x0 = tile * TW;
    return t;
}

// Stage a [C][n] strip of one feature row into LDS: dst[c*n + j] = src[b,c,y,xs+j] (0 outside).
__device__ __forceinline__ void stage_rows(float *dst, const float *__restrict__ src, int b, int y,
                                           int xs, int n, int C, int H, int W) {
    const int TW = blockDim.x;
    const size_t plane = (size_t)H * W;
    const float *base = src + ((size_t)b * C * H + y) * W;
    for (int c = 0; c < C; ++c) {
        const float *p = base + c * plane;
        for (int j = threadIdx.x; j < n; j += TW) {
            int x = xs + j;
            dst[c * n + j] = (x >= 0 && x < W) ? p[x] : 0.f;
        }
    }
}

__device__ __forceinline__ void stage_plane(float *dst, const float *__restrict__ src, int row,
                                            int xs, int n, int W) {
    const float *p = src + (size_t)row * W;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        int x = xs + j;
        dst[j] = (x >= 0 && x < W) ? p[x] : 0.f;
    }
}

__device__ __forceinline__ float dot_lds(const float *l, int ls, const float *r, int rs, int C) {
    float cost = 0.f;
    for (int c = 0; c < C; ++c) cost = fmaf(l[c * ls], r[c * rs], cost);   // SM_kernel.cu:52-55
    return cost;
}

// ---------------------------------------------------------------------------------------
// Forward.  MODE_MAT: get_max_cost + sparse_matching_forward (SM_kernel.cu:22-125).
//           MODE_VAR: get_max_cost + sparse_var_forward      (SV_kernel.cu:22-124).
//           MODE_FUSED: both, disparity = this pixel's SpaMat output (…Mask.py:183-192).
// LDS: Ls[C][TW] | Rs[C][RW] | Ts[RW],  RW = TW + halo, halo = D-1,  Rs[.][j] <-> x0-halo+j.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ void spamat_fwd_rowtile(const float *__restrict__ ref, const float *__restrict__ tar,
                                   const float *__restrict__ rmask,
                                   const float *__restrict__ tmask,
                                   const float *__restrict__ disparity, float *__restrict__ out,
                                   float *__restrict__ var_out, float *__restrict__ sum_sim,
                                   float *__restrict__ max_cost, int C, int H, int W, int D,
                                   int tiles_per_row) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TW = blockDim.x, tid = threadIdx.x;
    const int halo = D - 1, RW = TW + halo;
    const RowTile t = row_tile(tiles_per_row, H, TW);
    float *Ls = smem, *Rs = Ls + C * TW, *Ts = Rs + C * RW;

    stage_rows(Ls, ref, t.b, t.y, t.x0, TW, C, H, W);
    stage_rows(Rs, tar, t.b, t.y, t.x0 - halo, RW, C, H, W);
    stage_plane(Ts, tmask, t.row, t.x0 - halo, RW, W);
    __syncthreads();

    const int x = t.x0 + tid;
    if (x >= W) return;
    const size_t pix = (size_t)t.row * W + x;
    if (rmask[pix] == 0.f) {                       // caller's zero fill, functions/SpaMat.py:25-27
        if (MODE != MODE_VAR) out[pix] = 0.f;
        if (MODE != MODE_MAT) var_out[pix] = 0.f;
        sum_sim[pix] = 0.f;
        max_cost[pix] = 0.f;
        return;
    }
    const int cur = x - D + 1 >= 0 ? D : x + 1;    // SM_kernel.cu:42
    const float *l = Ls + tid;
    const float *r0 = Rs + tid + halo;             // r0[-d] = R[., x-d]
    const float *t0 = Ts + tid + halo;

    float m = 0.000001f;                           // SM_kernel.cu:45
    for (int d = 0; d < cur; ++d) {
        if (t0[-d] == 0.f) continue;
        float cost = dot_lds(l, TW, r0 - d, RW, C);
        if (m < cost) m = cost;
    }
    float S = 0.000001f, sd = 0.000001f;           // SM_kernel.cu:100
    float mu = 0.f;
    if (MODE == MODE_VAR) mu = disparity[pix];
    if (MODE != MODE_VAR) {
        for (int d = 0; d < cur; ++d) {
            if (t0[-d] == 0.f) continue;
            float cost = dot_lds(l, TW, r0 - d, RW, C);
            float e = expf(cost - m);
            sd = fmaf(e, (float)d, sd);
            S += e;
        }
        mu = sd / S;
        out[pix] = mu;
    }
    if (MODE != MODE_MAT) {
        float S2 = 0.000001f, sv = 0.000001f;      // SV_kernel.cu:100
        for (int d = 0; d < cur; ++d) {
            if (t0[-d] == 0.f) continue;
            float cost = dot_lds(l, TW, r0 - d, RW, C);
            float e = expf(cost - m);
            float dd = (float)d - mu;
            sv = fmaf(e * dd, dd, sv);             // SV_kernel.cu:120
            S2 += e;
        }
        var_out[pix] = sv / S2;
        S = S2;
    }
    sum_sim[pix] = S;
    max_cost[pix] = m;
}

// ---------------------------------------------------------------------------------------
// Backward w.r.t. the left features (+ the disparity input for SpaVar).
// sparse_matching_ref_backward SM_kernel.cu:143-195 / sparse_var_ref_backward
// SV_kernel.cu:142-195 / sparse_var_dis_backward SV_kernel.cu:275-325.
// LDS: Ls[C][TW] | Rs[C][RW] | Ts[RW] | Wt[D][TW]   (Wt column tid is private to thread tid)
// ---------------------------------------------------------------------------------------
template <bool VAR>
__global__ void spamat_bwd_ref_rowtile(const float *__restrict__ ref, const float *__restrict__ tar,
                                       const float *__restrict__ rmask,
                                       const float *__restrict__ tmask,
                                       const float *__restrict__ disparity,
                                       const float *__restrict__ out,
                                       const float *__restrict__ sum_sim,
                                       const float *__restrict__ max_cost,
                                       const float *__restrict__ grad_out,
                                       float *__restrict__ grad_ref,
                                       float *__restrict__ grad_disp, int C, int H, int W, int D,
                                       int tiles_per_row) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TW = blockDim.x, tid = threadIdx.x;
    const int halo = D - 1, RW = TW + halo;
    const RowTile t = row_tile(tiles_per_row, H, TW);
    float *Ls = smem, *Rs = Ls + C * TW, *Ts = Rs + C * RW, *Wt = Ts + RW;

    stage_rows(Ls, ref, t.b, t.y, t.x0, TW, C, H, W);
    stage_rows(Rs, tar, t.b, t.y, t.x0 - halo, RW, C, H, W);
    stage_plane(Ts, tmask, t.row, t.x0 - halo, RW, W);
    __syncthreads();

    const int x = t.x0 + tid;
    if (x >= W) return;
    const size_t pix = (size_t)t.row * W + x;
    const size_t plane = (size_t)H * W;
    float *g_ref = grad_ref + ((size_t)t.b * C * H + t.y) * W + x;
    if (rmask[pix] == 0.f) {                       // caller's zero fill, functions/SpaMat.py:42
        for (int c = 0; c < C; ++c) g_ref[c * plane] = 0.f;
        if (VAR) grad_disp[pix] = 0.f;
        return;
    }
    const int cur = x - D + 1 >= 0 ? D : x + 1;
    const float *l = Ls + tid;
    const float *r0 = Rs + tid + halo;
    const float *t0 = Ts + tid + halo;
    const float m = max_cost[pix], o = out[pix];
    const float mu = VAR ? disparity[pix] : 0.f;
    float gdis = 0.f;
    for (int d = 0; d < cur; ++d) {
        float w = 0.f;
        if (t0[-d] != 0.f) {
            float cost = dot_lds(l, TW, r0 - d, RW, C);
            float e = expf(cost - m);
            if (VAR) {
                float dd = (float)d - mu;
                w = e * fmaf(dd, dd, -o);          // SV_kernel.cu:191
                gdis = fmaf(e, dd, gdis);          // SV_kernel.cu:321
            } else {
                w = e * ((float)d - o);            // SM_kernel.cu:191
            }
        }
        Wt[d * TW + tid] = w;
    }
    const float g = grad_out[pix], S = sum_sim[pix];
    for (int c = 0; c < C; ++c) {
        const float *r = r0 + c * RW;
        float acc = 0.f;
        for (int d = 0; d < cur; ++d) acc = fmaf(Wt[d * TW + tid], r[-d], acc);
        g_ref[c * plane] = g * acc / S;            // SM_kernel.cu:193
    }
    if (VAR) grad_disp[pix] = -2.f * g * gdis / S; // SV_kernel.cu:323
}

// ---------------------------------------------------------------------------------------
// Backward w.r.t. the right features: sparse_matching_tar_backward SM_kernel.cu:300-355 /
// sparse_var_tar_backward SV_kernel.cu:215-271.  The tile owns TW RIGHT pixels x'; the
// left-side operands are needed on [x0, x0+TW+halo).
// LDS: Rs[C][TW] | Ls[C][RW] | Ms,Os,Qs,Xs,(Us)[RW] | Wt[D][TW],   index j <-> x0 + j.
//      Ms ref mask, Os output, Qs = grad_out/sum_sim, Xs max_cost, Us disparity (VAR).
// ---------------------------------------------------------------------------------------
template <bool VAR>
__global__ void spamat_bwd_tar_rowtile(const float *__restrict__ ref, const float *__restrict__ tar,
                                       const float *__restrict__ rmask,
                                       const float *__restrict__ tmask,
                                       const float *__restrict__ disparity,
                                       const float *__restrict__ out,
                                       const float *__restrict__ sum_sim,
                                       const float *__restrict__ max_cost,
                                       const float *__restrict__ grad_out,
                                       float *__restrict__ grad_tar, int C, int H, int W, int D,
                                       int tiles_per_row) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TW = blockDim.x, tid = threadIdx.x;
    const int halo = D - 1, RW = TW + halo;
    const RowTile t = row_tile(tiles_per_row, H, TW);
    float *Rs = smem, *Ls = Rs + C * TW, *Ms = Ls + C * RW, *Os = Ms + RW, *Qs = Os + RW,
          *Xs = Qs + RW, *Us = Xs + RW, *Wt = Us + (VAR ? RW : 0);

    stage_rows(Rs, tar, t.b, t.y, t.x0, TW, C, H, W);
    stage_rows(Ls, ref, t.b, t.y, t.x0, RW, C, H, W);
    stage_plane(Ms, rmask, t.row, t.x0, RW, W);
    stage_plane(Os, out, t.row, t.x0, RW, W);
    stage_plane(Xs, max_cost, t.row, t.x0, RW, W);
    if (VAR) stage_plane(Us, disparity, t.row, t.x0, RW, W);
    {
        const float *gp = grad_out + (size_t)t.row * W, *sp = sum_sim + (size_t)t.row * W;
        for (int j = tid; j < RW; j += TW) {
            int x = t.x0 + j;
            // masked-off left pixels have sum_sim = 0: never used (Ms gate), keep them finite
            Qs[j] = (x < W && Ms[j] != 0.f) ? gp[x] / sp[x] : 0.f;
        }
    }
    __syncthreads();

    const int x = t.x0 + tid;
    if (x >= W) return;
    const size_t pix = (size_t)t.row * W + x;
    const size_t plane = (size_t)H * W;
    float *g_tar = grad_tar + ((size_t)t.b * C * H + t.y) * W + x;
    if (tmask[pix] == 0.f) {
        for (int c = 0; c < C; ++c) g_tar[c * plane] = 0.f;
        return;
    }
    const int cur = x + D <= W ? D : W - x;        // SM_kernel.cu:327
    const float *r = Rs + tid;
    const float *l0 = Ls + tid;                    // l0[d] = L[., x+d]
    for (int d = 0; d < cur; ++d) {
        const int j = tid + d;
        float w = 0.f;
        if (Ms[j] != 0.f) {
            float cost = dot_lds(l0 + d, RW, r, TW, C);
            float e = expf(cost - Xs[j]);
            if (VAR) {
                float dd = (float)d - Us[j];
                w = Qs[j] * e * fmaf(dd, dd, -Os[j]);   // SV_kernel.cu:262
            } else {
                w = Qs[j] * e * ((float)d - Os[j]);     // SM_kernel.cu:346
            }
        }
        Wt[d * TW + tid] = w;
    }
    for (int c = 0; c < C; ++c) {
        const float *l = l0 + c * RW;
        float acc = 0.f;
        for (int d = 0; d < cur; ++d) acc = fmaf(Wt[d * TW + tid], l[d], acc);
        g_tar[c * plane] = acc;
    }
}

// Pick the tile width: widest of 256/128/64 threads whose LDS footprint fits.
int pick_tw(int W, size_t per_tw_floats, size_t fixed_floats, size_t *lds_bytes) {
    const int cands[3] = {256, 128, 64};
    for (int pass = 0; pass < 2; ++pass) {
        size_t budget = pass == 0 ? DECNET_LDS_BUDGET : DECNET_LDS_BYTES;
        for (int i = 0; i < 3; ++i) {
            int tw = cands[i];
            if (tw > 64 && tw >= 2 * W) continue;          // do not idle most of the tile
            size_t bytes = 4 * (per_tw_floats * tw + fixed_floats);
            if (bytes <= budget) { *lds_bytes = bytes; return tw; }
        }
    }
    return 0;
}

int check_args(const void *const *ptrs, int n, int B, int C, int H, int W, int max_disp) {
    for (int i = 0; i < n; ++i)
        if (!ptrs[i]) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 1 || W < 1 || max_disp < 1) return DECNET_ERR_BAD_SHAPE;
    if ((double)B * C * H * W >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    return DECNET_OK;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return DECNET_OK;
}

// Last resort of the forward pass (C x max_disp beyond any LDS tile, e.g. C = 216 with max_disp = 270):
// the arithmetic of spamat_fwd_rowtile straight from global memory, one thread per left pixel.
template <int MODE>
__global__ __launch_bounds__(256) void spamat_fwd_generic(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int B, int C,
    int H, int W, int D) {
    const size_t plane = (size_t)H * W, pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (size_t)B * plane) return;
    if (rmask[pix] == 0.f) {
        if (MODE != MODE_VAR) out[pix] = 0.f;
        if (MODE != MODE_MAT) var_out[pix] = 0.f;
        sum_sim[pix] = 0.f;
        max_cost[pix] = 0.f;
        return;
    }
    const size_t b = pix / plane, rem = pix - b * plane;
    const int x = (int)(rem % W);
    const float *l = ref + b * C * plane + rem, *r = tar + b * C * plane + rem;
    const int cur = x - D + 1 >= 0 ? D : x + 1;
    auto cost_at = [&](int d) {
        float cost = 0.f;
        for (int c = 0; c < C; ++c) cost = fmaf(l[(size_t)c * plane], r[(size_t)c * plane - d], cost);
        return cost;
    };
    float m = 0.000001f;
    for (int d = 0; d < cur; ++d) {
        if (tmask[pix - d] == 0.f) continue;
        const float cost = cost_at(d);
        if (m < cost) m = cost;
    }
    float S = 0.000001f, sd = 0.000001f, mu = 0.f;
    if (MODE == MODE_VAR) mu = disparity[pix];
    if (MODE != MODE_VAR) {
        for (int d = 0; d < cur; ++d) {
            if (tmask[pix - d] == 0.f) continue;
            const float e = expf(cost_at(d) - m);
            sd = fmaf(e, (float)d, sd);
            S += e;
        }
        mu = sd / S;
        out[pix] = mu;
    }
    if (MODE != MODE_MAT) {
        float S2 = 0.000001f, sv = 0.000001f;
        for (int d = 0; d < cur; ++d) {
            if (tmask[pix - d] == 0.f) continue;
            const float e = expf(cost_at(d) - m), dd = (float)d - mu;
            sv = fmaf(e * dd, dd, sv);
            S2 += e;
        }
        var_out[pix] = sv / S2;
        S = S2;
    }
    sum_sim[pix] = S;
    max_cost[pix] = m;
}

// Last resort of the backward pass: no LDS tile, so no limit on C x max_disp (the row-tile kernels need
// (C+1)*(D-1) floats of LDS: C = 72 with max_disp = 270 does not fit; the shipped configurations never
// get here).  One thread per pixel of the own side (SIDE 0: left pixel -> grad_ref (, grad_disparity);
// SIDE 1: right pixel -> grad_tar), 8 channels of gradient per pass over the disparities, the cost
// recomputed from global memory in each pass: the reference's arithmetic (SM_kernel.cu:143-195, 300-355;
// SV_kernel.cu:142-325) with C/8 instead of C redundant cost sweeps.
template <bool VAR, int SIDE>
__global__ __launch_bounds__(256) void spamat_bwd_generic(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, const float *__restrict__ out,
    const float *__restrict__ sum_sim, const float *__restrict__ max_cost,
    const float *__restrict__ grad_out, float *__restrict__ grad_own, float *__restrict__ grad_disp, int B,
    int C, int H, int W, int D) {
    const size_t plane = (size_t)H * W, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * plane) return;
    const size_t b = idx / plane, rem = idx - b * plane;
    const int x = (int)(rem % W);
    const size_t base3 = b * C * plane + rem;
    const bool on = (SIDE == 0 ? rmask[idx] : tmask[idx]) != 0.f;
    if (!on) {
        for (int c = 0; c < C; ++c) grad_own[base3 + (size_t)c * plane] = 0.f;
        if (VAR && SIDE == 0) grad_disp[idx] = 0.f;
        return;
    }
    const int cur = SIDE == 0 ? (x - D + 1 >= 0 ? D : x + 1) : (x + D <= W ? D : W - x);
    float gd = 0.f;
    for (int c0 = 0; c0 < C; c0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < cur; ++d) {
            const size_t l2 = SIDE == 0 ? idx : idx + d, r2 = SIDE == 0 ? idx - d : idx;   // left / right pixel
            if ((SIDE == 0 ? tmask[r2] : rmask[l2]) == 0.f) continue;
            const float *lp = ref + (SIDE == 0 ? base3 : base3 + d), *rp = tar + (SIDE == 0 ? base3 - d : base3);
            float cost = 0.f;
            for (int c = 0; c < C; ++c) cost = fmaf(lp[(size_t)c * plane], rp[(size_t)c * plane], cost);
            const float e = expf(cost - max_cost[l2]);
            float wgt;
            if (VAR) {
                const float dd = (float)d - disparity[l2];
                wgt = fmaf(dd, dd, -out[l2]);
                if (SIDE == 0 && c0 == 0) gd = fmaf(e, dd, gd);
            } else {
                wgt = (float)d - out[l2];
            }
            const float w = SIDE == 0 ? e * wgt : grad_out[l2] * e * wgt / sum_sim[l2];
            const float *op = SIDE == 0 ? rp : lp;                                          // the other side's features
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < C) acc[k] = fmaf(w, op[(size_t)(c0 + k) * plane], acc[k]);
        }
        const float sc = SIDE == 0 ? grad_out[idx] / sum_sim[idx] : 1.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < C) grad_own[base3 + (size_t)(c0 + k) * plane] = acc[k] * sc;
    }
    if (VAR && SIDE == 0) grad_disp[idx] = -2.f * grad_out[idx] * gd / sum_sim[idx];
}

}  // namespace

// ------------------------------- host launchers (internal) ------------------------------
// mode: 0 SpaMat, 1 SpaVar, 2 fused.  Used by capi.hip.

int decnet_rowtile_forward(int mode, const float *ref, const float *tar, const float *rmask,
                           const float *tmask, const float *disparity, float *out, float *var_out,
                           float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                           hipStream_t stream) {
    const int D = max_disp;
    size_t lds = 0;
    // Ls C*TW + Rs C*(TW+D-1) + Ts (TW+D-1)
    int TW = pick_tw(W, (size_t)2 * C + 1, (size_t)(C + 1) * (D - 1), &lds);
    if (!TW) {                                          // no LDS tile fits: global-memory kernel
        const size_t n = (size_t)B * H * W;
        const dim3 g((unsigned)((n + 255) / 256)), blk(256);
        if (mode == MODE_MAT)
            hipLaunchKernelGGL(spamat_fwd_generic<MODE_MAT>, g, blk, 0, stream, ref, tar, rmask, tmask, disparity, out,
                               var_out, sum_sim, max_cost, B, C, H, W, D);
        else if (mode == MODE_VAR)
            hipLaunchKernelGGL(spamat_fwd_generic<MODE_VAR>, g, blk, 0, stream, ref, tar, rmask, tmask, disparity, out,
                               var_out, sum_sim, max_cost, B, C, H, W, D);
        else
            hipLaunchKernelGGL(spamat_fwd_generic<MODE_FUSED>, g, blk, 0, stream, ref, tar, rmask, tmask, disparity, out,
                               var_out, sum_sim, max_cost, B, C, H, W, D);
        return decnet_launch_status();
    }
    int tiles = ceil_div(W, TW);
    dim3 grid((unsigned)((size_t)B * H * tiles)), block(TW);
    int rc;
#define LAUNCH(M)                                                                              \
    rc = set_lds(spamat_fwd_rowtile<M>, lds);                                                  \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(spamat_fwd_rowtile<M>, grid, block, lds, stream, ref, tar, rmask, tmask, \
                       disparity, out, var_out, sum_sim, max_cost, C, H, W, D, tiles)
    if (mode == MODE_MAT) { LAUNCH(MODE_MAT); }
    else if (mode == MODE_VAR) { LAUNCH(MODE_VAR); }
    else { LAUNCH(MODE_FUSED); }
#undef LAUNCH
    return decnet_launch_status();
}

int decnet_rowtile_backward(int var, const float *ref, const float *tar, const float *rmask,
                            const float *tmask, const float *disparity, const float *out,
                            const float *sum_sim, const float *max_cost, const float *grad_out,
                            float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                            int W, int max_disp, hipStream_t stream) {
    const int D = max_disp;
    size_t lds_r = 0, lds_t = 0;
    int TWr = pick_tw(W, (size_t)2 * C + 1 + D, (size_t)(C + 1) * (D - 1), &lds_r);
    int planes = var ? 5 : 4;
    int TWt = pick_tw(W, (size_t)2 * C + planes + D, (size_t)(C + planes) * (D - 1), &lds_t);
    if (!TWr || !TWt) {                                 // C x max_disp beyond any LDS tile: global-memory kernels
        const size_t n = (size_t)B * H * W;
        if (n >= ((size_t)1 << 31) * 256) return DECNET_ERR_BAD_SHAPE;
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (var) {
            hipLaunchKernelGGL((spamat_bwd_generic<true, 0>), grid, block, 0, stream, ref, tar, rmask, tmask, disparity,
                               out, sum_sim, max_cost, grad_out, grad_ref, grad_disp, B, C, H, W, D);
            hipLaunchKernelGGL((spamat_bwd_generic<true, 1>), grid, block, 0, stream, ref, tar, rmask, tmask, disparity,
                               out, sum_sim, max_cost, grad_out, grad_tar, nullptr, B, C, H, W, D);
        } else {
            hipLaunchKernelGGL((spamat_bwd_generic<false, 0>), grid, block, 0, stream, ref, tar, rmask, tmask, disparity,
                               out, sum_sim, max_cost, grad_out, grad_ref, nullptr, B, C, H, W, D);
            hipLaunchKernelGGL((spamat_bwd_generic<false, 1>), grid, block, 0, stream, ref, tar, rmask, tmask, disparity,
                               out, sum_sim, max_cost, grad_out, grad_tar, nullptr, B, C, H, W, D);
        }
        return decnet_launch_status();
    }
    int rc;
    {
        int tiles = ceil_div(W, TWr);
        dim3 grid((unsigned)((size_t)B * H * tiles)), block(TWr);
        if (var) {
            if ((rc = set_lds(spamat_bwd_ref_rowtile<true>, lds_r))) return rc;
            hipLaunchKernelGGL(spamat_bwd_ref_rowtile<true>, grid, block, lds_r, stream, ref, tar,
                               rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref,
                               grad_disp, C, H, W, D, tiles);
        } else {
            if ((rc = set_lds(spamat_bwd_ref_rowtile<false>, lds_r))) return rc;
            hipLaunchKernelGGL(spamat_bwd_ref_rowtile<false>, grid, block, lds_r, stream, ref, tar,
                               rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref,
                               grad_disp, C, H, W, D, tiles);
        }
        if ((rc = decnet_launch_status())) return rc;
    }
    {
        int tiles = ceil_div(W, TWt);
        dim3 grid((unsigned)((size_t)B * H * tiles)), block(TWt);
        if (var) {
            if ((rc = set_lds(spamat_bwd_tar_rowtile<true>, lds_t))) return rc;
            hipLaunchKernelGGL(spamat_bwd_tar_rowtile<true>, grid, block, lds_t, stream, ref, tar,
                               rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_tar,
                               C, H, W, D, tiles);
        } else {
            if ((rc = set_lds(spamat_bwd_tar_rowtile<false>, lds_t))) return rc;
            hipLaunchKernelGGL(spamat_bwd_tar_rowtile<false>, grid, block, lds_t, stream, ref, tar,
                               rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_tar,
                               C, H, W, D, tiles);
        }
    }
    return decnet_launch_status();
}

int decnet_check_spamat_args(const void *const *ptrs, int n, int B, int C, int H, int W,
                             int max_disp) {
    return check_args(ptrs, n, B, C, H, W, max_disp);
}
