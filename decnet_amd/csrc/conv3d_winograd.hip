// decnet_amd/csrc/conv3d_winograd.hip -- Conv3d(k3,s1,p1)+BN+ReLU by Winograd F(2x2x2, 3x3x3) in
// fp32 on the matrix cores (gfx950).  Same operator as stage0.hip:conv3d_k3_igemm -- one
// Conv3dUnit of CostRegNetNoDown in eval mode (submodule.py:115-123, 608-662) -- with 3.375x
// fewer multiplications: every 2x2x2 block of outputs is computed from a 4x4x4 input tile as
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        (applied along D, H and W)
// so the 27-tap implicit GEMM (K = 27*Ci) becomes 64 independent GEMMs with K = Ci:
//     M[xi][tile][co] = sum_ci V[xi][tile][ci] * U[xi][ci][co],   xi = 0..63
//   V = B^T-transformed input tiles (adds only), U = G-transformed weights (once per weight
//   version), Y = A^T-transformed M (adds only) -> BN scale/shift -> ReLU -> (+ residual).
// fp32 throughout.  Two tile shapes (the `variant` argument of the entry points):
//   0  F(2,3) on D, H and W: 4x4x4 input tile -> 2x2x2 outputs, 64 transform points, 8 multiplies
//      per output (27 direct).  Benign constants (0, +-1, +-1/2): on the 8-layer stack the
//      regularised volume differs from the direct convolution by 1e-6 relative, the disparity by
//      2e-5 px max.
//   1  F(2,3) on D, F(4,3) on H and W: 4x6x6 tile -> 2x4x4 outputs, 144 points, 4.5 multiplies
//      per output; constants up to 8 and 1/24: 6e-6 relative on the volume, 1.2e-4 px max /
//      1e-5 px mean on the disparity (still 10x / 100x inside the 1e-3 px budget).
// tests/test_stage0_gpu.py checks every algorithm against the same oracle.
//
// Tiles are processed in chunks of at most 1 GiB of V + M (64 x tiles x C floats each); chunks
// small enough to stay in the 256 MiB Infinity Cache between the three kernels were measured
// and gain nothing (the transforms already run at 4.3-5.3 TB/s), while one big chunk lets a GEMM
// workgroup pipeline 8 transform points back to back.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#ifndef DECNET_WINO_ABLATE
#define DECNET_WINO_ABLATE 0     // timing-only builds (tools/ablate.sh): 1 no stores | 2 no global loads |
#endif                           // 3 no MFMA | 5 neither loads nor stores

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int W_BN = 224;      // co tile of the GEMM, 14 MFMA tiles of 16 (as conv3d_k3_igemm)
constexpr int WB_PITCH = 240;  // == 16 (mod 32)

// 1-D transforms of F(m,3), m = O outputs, tile T = O + 2 (Lavin & Gray, arXiv:1509.09308)
template <int T> __device__ __forceinline__ void bt_1d(float (&v)[T]);
template <> __device__ __forceinline__ void bt_1d<4>(float (&v)[4]) {
    const float t0 = v[0] - v[2], t1 = v[1] + v[2], t2 = v[2] - v[1], t3 = v[1] - v[3];
    v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;
}
template <> __device__ __forceinline__ void bt_1d<6>(float (&v)[6]) {
    const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5];
    v[0] = 4.f * a0 - 5.f * a2 + a4;
    v[1] = -4.f * (a1 + a2) + a3 + a4;
    v[2] = 4.f * (a1 - a2) - a3 + a4;
    v[3] = 2.f * (a3 - a1) - a2 + a4;
    v[4] = 2.f * (a1 - a3) - a2 + a4;
    v[5] = 4.f * a1 - 5.f * a3 + a5;
}
template <int T> __device__ __forceinline__ void g_1d(const float (&g)[3], float (&o)[T]);
template <> __device__ __forceinline__ void g_1d<4>(const float (&g)[3], float (&o)[4]) {
    o[0] = g[0]; o[1] = 0.5f * (g[0] + g[1] + g[2]); o[2] = 0.5f * (g[0] - g[1] + g[2]); o[3] = g[2];
}
template <> __device__ __forceinline__ void g_1d<6>(const float (&g)[3], float (&o)[6]) {
    o[0] = 0.25f * g[0];
    o[1] = (-1.f / 6.f) * (g[0] + g[1] + g[2]);
    o[2] = (-1.f / 6.f) * (g[0] - g[1] + g[2]);
    o[3] = (1.f / 24.f) * g[0] + (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
    o[4] = (1.f / 24.f) * g[0] - (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
    o[5] = g[2];
}
template <int T> __device__ __forceinline__ void at_1d(const float (&m)[T], float (&o)[T - 2]);
template <> __device__ __forceinline__ void at_1d<4>(const float (&m)[4], float (&o)[2]) {
    o[0] = m[0] + m[1] + m[2];
    o[1] = m[1] - m[2] - m[3];
}
template <> __device__ __forceinline__ void at_1d<6>(const float (&m)[6], float (&o)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}

// ------------------------------ weight transform (once) --------------------------------
// w [Co][Ci][3][3][3] (torch) -> U [TD*TH*TW][Ci][224], U = G w G^T along the three axes, co padded.
// kt != 0: U^T [points][224][Ci] (k contiguous) for wino_gemm_reg.
template <int TD, int TH, int TW>
__global__ void wino_weight_transform(const float *__restrict__ w, float *__restrict__ U, int Co,
                                      int Ci, int kt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (ci, co)
    if (idx >= Ci * W_BN) return;
    const int co = idx % W_BN, ci = idx / W_BN;
    float t1[3][3][TW], t2[3][TH][TW];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            float g[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = co < Co ? w[((size_t)co * Ci + ci) * 27 + (i * 3 + jj) * 3 + k] : 0.f;
            g_1d<TW>(g, t1[i][jj]);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < TW; ++k) {
            const float g[3] = {t1[i][0][k], t1[i][1][k], t1[i][2][k]};
            float o[TH];
            g_1d<TH>(g, o);
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) t2[i][jj][k] = o[jj];
        }
#pragma unroll
    for (int jj = 0; jj < TH; ++jj)
#pragma unroll
        for (int k = 0; k < TW; ++k) {
            const float g[3] = {t2[0][jj][k], t2[1][jj][k], t2[2][jj][k]};
            float o[TD];
            g_1d<TD>(g, o);
#pragma unroll
            for (int i = 0; i < TD; ++i) {
                const size_t pt = (size_t)((i * TH + jj) * TW + k);
                U[kt ? (pt * W_BN + co) * Ci + ci : (pt * Ci + ci) * W_BN + co] = o[i];
            }
        }
}

struct Tiling {
    int D, H, W, Td, Th, Tw;
};
template <int OD, int OH, int OW>
__device__ __forceinline__ void tile_coords(int t, const Tiling &g, int &b, int &z0, int &y0, int &x0) {
    const int tw = t % g.Tw; t /= g.Tw;
    const int th = t % g.Th; t /= g.Th;
    const int td = t % g.Td;
    b = t / g.Td;
    z0 = OD * td; y0 = OH * th; x0 = OW * tw;
}

// ------------------------------ input transform -----------------------------------------
// x [B,D,H,W,C] -> V[xi][tile - t_lo][c], V = B^T d B along D, H, W.  One thread per (tile, channel),
// channel fastest (coalesced on both sides).
template <int TD, int TH, int TW>
__global__ __launch_bounds__(256) void wino_input_transform(const float *__restrict__ x,
                                                            float *__restrict__ V, Tiling g, int C,
                                                            int t_lo, int nt) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nt * C) return;
    const int c = (int)(idx % C), tl = (int)(idx / C);
    int b, z0, y0, x0;
    tile_coords<TD - 2, TH - 2, TW - 2>(t_lo + tl, g, b, z0, y0, x0);
    float d[TD][TH][TW];
#pragma unroll
    for (int i = 0; i < TD; ++i) {
        const int z = z0 - 1 + i;
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) {
            const int y = y0 - 1 + jj;
            const bool okzy = (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H;
            const float *row = x + (((size_t)b * g.D + z) * g.H + y) * g.W * C + c;
#pragma unroll
            for (int k = 0; k < TW; ++k) {
                const int xx = x0 - 1 + k;
                d[i][jj][k] = (okzy && (unsigned)xx < (unsigned)g.W) ? row[(size_t)xx * C] : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) bt_1d<TW>(d[i][jj]);
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int k = 0; k < TW; ++k) {
            float v[TH];
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) v[jj] = d[i][jj][k];
            bt_1d<TH>(v);
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) d[i][jj][k] = v[jj];
        }
#pragma unroll
    for (int jj = 0; jj < TH; ++jj)
#pragma unroll
        for (int k = 0; k < TW; ++k) {
            float v[TD];
#pragma unroll
            for (int i = 0; i < TD; ++i) v[i] = d[i][jj][k];
            bt_1d<TD>(v);
#pragma unroll
            for (int i = 0; i < TD; ++i) d[i][jj][k] = v[i];
        }
    float *o = V + (size_t)tl * C + c;
    const size_t xs = (size_t)nt * C;
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int jj = 0; jj < TH; ++jj)
#pragma unroll
            for (int k = 0; k < TW; ++k) o[(size_t)((i * TH + jj) * TW + k) * xs] = d[i][jj][k];
}

// ------------------------------ output transform + epilogue -----------------------------
// M[xi][tile - t_lo][co] -> y: A^T along W, H, D, then BN scale/shift, ReLU, + residual
// (CostRegNetNoDown.forward submodule.py:656).
template <int TD, int TH, int TW>
__global__ __launch_bounds__(256) void wino_output_transform(
    const float *__restrict__ M, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ residual, float *__restrict__ y, Tiling g, int Co, int relu, int t_lo,
    int nt) {
    constexpr int OD = TD - 2, OH = TH - 2, OW = TW - 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nt * Co) return;
    const int co = (int)(idx % Co), tl = (int)(idx / Co);
    int b, z0, y0, x0;
    tile_coords<OD, OH, OW>(t_lo + tl, g, b, z0, y0, x0);
    const float *mp = M + (size_t)tl * Co + co;
    const size_t xs = (size_t)nt * Co;
    float a[TD][TH][OW], bb[TD][OH][OW];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) {
            float m[TW];
#pragma unroll
            for (int k = 0; k < TW; ++k) m[k] = mp[(size_t)((i * TH + jj) * TW + k) * xs];
            at_1d<TW>(m, a[i][jj]);
        }
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int k = 0; k < OW; ++k) {
            float m[TH], o[OH];
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) m[jj] = a[i][jj][k];
            at_1d<TH>(m, o);
#pragma unroll
            for (int jj = 0; jj < OH; ++jj) bb[i][jj][k] = o[jj];
        }
    const float sc = scale[co], sh = shift[co];
#pragma unroll
    for (int jj = 0; jj < OH; ++jj)
#pragma unroll
        for (int k = 0; k < OW; ++k) {
            float m[TD], o[OD];
#pragma unroll
            for (int i = 0; i < TD; ++i) m[i] = bb[i][jj][k];
            at_1d<TD>(m, o);
#pragma unroll
            for (int i = 0; i < OD; ++i) {
                const int z = z0 + i, yy = y0 + jj, xx = x0 + k;
                if (z < g.D && yy < g.H && xx < g.W) {
                    float v = fmaf(o[i], sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    const size_t off = ((((size_t)b * g.D + z) * g.H + yy) * g.W + xx) * Co + co;
                    if (residual) v += residual[off];
                    y[off] = v;
                }
            }
        }
}

// ------------------------------ batched GEMM  M[xi] = V[xi] * U[xi] ----------------------
// blockIdx.y = xi.  Rows = tiles of the chunk.  Same block / wave tiling, LDS layout and
// prefetch scheme as conv3d_k3_igemm (WM x 2 waves, 48 x 112 per wave, double-buffered LDS,
// bounds-checked buffer loads), with K = Ci instead of 27*Ci.
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

template <int WM, int BK>
__global__ __launch_bounds__(WM * 128) void wino_gemm(const float *__restrict__ Vb,
                                                     const float *__restrict__ Ub,
                                                     float *__restrict__ Mb, int nt, int Ci, int Co,
                                                     int xg, int np, int swz) {
    // blockIdx.y owns xg consecutive transform points xi and runs them as ONE software pipeline
    // (the first K chunk of point xi+1 is prefetched during the last K chunk of point xi, the
    // accumulators are stored and cleared at the boundary): K = Ci alone is only 6 chunks, too
    // short to hide a pipeline fill per point.
    constexpr int THREADS = WM * 128, BM = WM * 48, TM = 3, TN = 7;
    constexpr int A_PITCH = BK + 2;
    constexpr int A_F4 = BM * (BK / 4);
    constexpr int A_PER_T = (A_F4 + THREADS - 1) / THREADS;
    constexpr int B_F4 = BK * (W_BN / 4);
    constexpr int B_PER_T = (B_F4 + THREADS - 1) / THREADS;
    constexpr int A_TILE = BM * A_PITCH, B_TILE = BK * WB_PITCH;
    constexpr int OOB = 0x7fffffff;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * A_TILE;

    // XCD-aware task order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // workgroup ids that are equal mod 8 take the M blocks of the SAME transform points: one XCD's
    // L2 then holds U of one point group at a time instead of every XCD streaming all of U.
    const int mblocks = gridDim.x, ngroups = gridDim.y;
    int pg = blockIdx.y, mb = blockIdx.x;
    if (swz) {
        const int id = blockIdx.y * mblocks + blockIdx.x, per8 = 8 * mblocks;
        const int r = id / per8, q = id - r * per8;
        if (8 * (r + 1) <= ngroups) { pg = 8 * r + (q & 7); mb = q >> 3; }     // tail (< 8 groups) unswizzled
    }
    const int xi0 = pg * xg;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, np * nt * Ci * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)Ub, 0, np * Ci * W_BN * 4, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kq = lane >> 4;
    const int m_block = mb * BM;

    int a_lds[A_PER_T], a_off[A_PER_T], a_k[A_PER_T];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
        const int idx = tid + i * THREADS;
        const int ml = idx / (BK / 4), q = idx - ml * (BK / 4);
        const bool ok = idx < A_F4 && m_block + ml < nt;
        a_lds[i] = idx < A_F4 ? ml * A_PITCH + 4 * q : -1;
        a_off[i] = ok ? ((m_block + ml) * Ci + 4 * q) * 4 : OOB;
        a_k[i] = 4 * q;
    }
    int b_lds[B_PER_T], b_off[B_PER_T], b_k[B_PER_T];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
        const int idx = tid + i * THREADS;
        const int kk = idx / (W_BN / 4), q = idx - kk * (W_BN / 4);
        b_lds[i] = idx < B_F4 ? kk * WB_PITCH + 4 * q : -1;
        b_off[i] = idx < B_F4 ? (kk * W_BN + 4 * q) * 4 : OOB;
        b_k[i] = kk;
    }
    const int nchunk = (Ci + BK - 1) / BK;
    const int nstep = xg * nchunk;
    const int v_point = nt * Ci * 4, u_point = Ci * W_BN * 4;         // bytes per transform point

    float4 ra[A_PER_T], rb[B_PER_T];
    auto prefetch = [&](int s) {
        const int p = s / nchunk, ci0 = (s - p * nchunk) * BK;
        const int va = (xi0 + p) * v_point + ci0 * 4, ua = (xi0 + p) * u_point + ci0 * W_BN * 4;
#if DECNET_WINO_ABLATE == 2 || DECNET_WINO_ABLATE == 5
        if (s > 0) return;
#endif
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const bool ok = a_off[i] != OOB && ci0 + a_k[i] < Ci;
            ra[i] = buf_load4(vr, ok ? a_off[i] + va : OOB);
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const bool ok = b_off[i] != OOB && ci0 + b_k[i] < Ci;
            rb[i] = buf_load4(ur, ok ? b_off[i] + ua : OOB);
        }
    };
    auto stage = [&](int buf) {
        float *a = As + buf * A_TILE, *b = Bs + buf * B_TILE;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i)
            if (a_lds[i] >= 0) {
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
    const int b_col0 = kq * WB_PITCH + wn * (W_BN / 2) + i16;
    int chunk = 0, point = xi0;
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
            for (int j = 0; j < TN; ++j) bv[j] = b[kk * 4 * WB_PITCH + j * 16];
#if DECNET_WINO_ABLATE == 3
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][0] += av[i] * bv[j];
#else
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
#endif
        }
        if (s + 1 < nstep) stage(buf ^ 1);
        if (++chunk == nchunk) {                       // transform point finished: store and clear
            float *Mo = Mb + (size_t)point * nt * Co;
#if DECNET_WINO_ABLATE == 1 || DECNET_WINO_ABLATE == 5
            if (nt > 0) Mo = nullptr;
            if (Mo || acc[0][0][0] == 12345.f)
#endif
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = wn * (W_BN / 2) + j * 16 + i16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m_block + wm * 48 + i * 16 + kq * 4 + r;
                        if (co < Co && m < nt) Mo[(size_t)m * Co + co] = acc[i][j][r];
                    }
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            chunk = 0;
            ++point;
        }
        __syncthreads();
    }
}

// ------------------------------ the same GEMM without LDS ---------------------------------
// K = Ci is short (216), so the LDS-staged kernel above spends a quarter of its time filling and
// draining its pipeline around barriers.  Here every wave is independent: both MFMA operands go
// HBM/L2/L1 -> registers as one 16-byte load per lane, no LDS, no barrier.  That works because the
// order of the K axis inside an MFMA is free as long as both operands agree: lane (i16, kq) loads
// the four consecutive k = 16c + 4kq + {0..3} of ITS row (V row m, or U^T row co: weights are kept
// k-contiguous, [point][224][Ci]) and MFMA step t of chunk c multiplies element t of those.
// The operands are swapped (A = U^T rows co, B = V rows m) so that a lane's four accumulator
// registers are four consecutive co of one tile row m: the result leaves as 16-byte stores.
// Duplicate operand reads between the waves of a workgroup (V twice, U^T WM times) hit in L1/L2.
// NCH > 0 (even): Ci spans exactly NCH 16-wide chunks and the chunk loop is unrolled, so every
// s_waitcnt counts exactly the loads that must have landed (and never the stores of the previous
// point); NCH == 0: any Ci, runtime chunk loop.
template <int WM, int NCH>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_gemm_reg(
    const float *__restrict__ Vb, const float *__restrict__ Utb, float *__restrict__ Mb, int nt, int Ci,
    int Co, int xg, int np, int swz) {
    constexpr int TM = 3, TN = 7, BM = WM * 48, OOB = 0x7fffffff;
    const int mblocks = gridDim.x, ngroups = gridDim.y;
    int pg = blockIdx.y, mb = blockIdx.x;
    if (swz) {                                          // XCD-aware order, as in wino_gemm
        const int id = blockIdx.y * mblocks + blockIdx.x, per8 = 8 * mblocks;
        const int r = id / per8, q = id - r * per8;
        if (8 * (r + 1) <= ngroups) { pg = 8 * r + (q & 7); mb = q >> 3; }
    }
    const int xi0 = pg * xg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, i16 = lane & 15, kq = lane >> 4;
    const int m0 = mb * BM + wm * 48;
    if (m0 >= nt) return;                               // no barriers: idle waves of the M tail just leave
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, np * nt * Ci * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)Utb, 0, np * Ci * W_BN * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void *)Mb, 0, np * nt * Co * 4, 0x00020000);
    int v_off[TM], u_off[TN], m_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + i16;
        v_off[i] = m < nt ? (m * Ci + kq * 4) * 4 : OOB;
        m_off[i] = m < nt ? (m * Co + wn * (W_BN / 2) + 4 * kq) * 4 : OOB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) u_off[j] = ((wn * (W_BN / 2) + j * 16 + i16) * Ci + kq * 4) * 4;
    const int nch = NCH > 0 ? NCH : (Ci + 15) >> 4;
    const int v_point = nt * Ci * 4, u_point = W_BN * Ci * 4, m_point = nt * Co * 4;   // bytes per point

    f32x4 v0[TM], u0[TN], v1[TM], u1[TN];
    // (p, c) = point (relative) and chunk; beyond the last point the offsets go out of range: zeros, no traffic
    auto load = [&](f32x4(&v)[TM], f32x4(&u)[TN], int p, int c) {
        const bool ok = c * 16 + kq * 4 < Ci && p < xg;               // K tail: whole 16-byte groups (Ci % 4 == 0)
#if DECNET_WINO_ABLATE == 2 || DECNET_WINO_ABLATE == 5                  // always the same (cached) lines
        const int vb = xi0 * v_point, ub = xi0 * u_point;
#else
        const int vb = (xi0 + p) * v_point + c * 64, ub = (xi0 + p) * u_point + c * 64;
#endif
#if DECNET_WINO_ABLATE == 6                                            // no loads at all (after the first)
        if (p + c > 0) return;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4 t = buf_load4(vr, ok && v_off[i] != OOB ? v_off[i] + vb : OOB);
            v[i] = f32x4{t.x, t.y, t.z, t.w};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float4 t = buf_load4(ur, ok ? u_off[j] + ub : OOB);
            u[j] = f32x4{t.x, t.y, t.z, t.w};
        }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const f32x4(&v)[TM], const f32x4(&u)[TN]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#if DECNET_WINO_ABLATE == 3
                    acc[j][i][0] += u[j][t] * v[i][t];
#else
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][t], v[i][t], acc[j][i], 0, 0, 0);
#endif
    };
    const bool co4 = (Co & 3) == 0;
    auto finish = [&](int p) {                          // transform point finished: store and clear
#if DECNET_WINO_ABLATE == 1 || DECNET_WINO_ABLATE == 5
        const int pb = OOB - 0x1000000;
#else
        const int pb = (xi0 + p) * m_point;
#endif
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = wn * (W_BN / 2) + j * 16 + 4 * kq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int base = m_off[i] != OOB ? m_off[i] + pb + j * 64 : OOB;
                if (co4) {
                    const f32x4 a = acc[j][i];
                    __builtin_amdgcn_raw_buffer_store_b128(
                        i32x4{__float_as_int(a[0]), __float_as_int(a[1]), __float_as_int(a[2]), __float_as_int(a[3])},
                        mr, co < Co ? base : OOB, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[j][i][r]), mr,
                                                              base != OOB && co + r < Co ? base + 4 * r : OOB, 0, 0);
                }
                acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    // two register sets: the next chunk is in flight while the current one is multiplied, and the
    // chunk sequence runs across the xg points of this workgroup without draining
    load(v0, u0, 0, 0);
    if constexpr (NCH > 0) {
        static_assert(NCH % 2 == 0, "set parity must repeat per point");
        for (int p = 0; p < xg; ++p) {
#pragma unroll
            for (int c = 0; c < NCH; c += 2) {
                // sched_barrier: keep the loads ABOVE the MFMA block they overlap with (the machine
                // scheduler otherwise sinks each load to just before its first use)
                load(v1, u1, p, c + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(v0, u0);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < NCH) load(v0, u0, p, c + 2); else load(v0, u0, p + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                compute(v1, u1);
                __builtin_amdgcn_sched_barrier(0);
            }
            finish(p);
        }
    } else {
        int p = 0, c = 0;                               // step being LOADED
        auto advance = [&]() { if (++c == nch) { c = 0; ++p; } };
        advance();
        int cc = 0, pc = 0;                             // step being COMPUTED
        const int nstep = xg * nch;
        for (int s = 0; s < nstep; s += 2) {
            load(v1, u1, p, c); advance();
            __builtin_amdgcn_sched_barrier(0);
            compute(v0, u0);
            __builtin_amdgcn_sched_barrier(0);
            if (++cc == nch) { finish(pc); cc = 0; ++pc; }
            if (s + 1 >= nstep) break;
            load(v0, u0, p, c); advance();
            __builtin_amdgcn_sched_barrier(0);
            compute(v1, u1);
            __builtin_amdgcn_sched_barrier(0);
            if (++cc == nch) { finish(pc); cc = 0; ++pc; }
        }
    }
}

static int gemm_kind() {          // 0: LDS-staged wino_gemm, 1: register-direct wino_gemm_reg
    static const int k = [] { const char *e = getenv("DECNET_WINO_GEMM"); return e && !strcmp(e, "lds") ? 0 : 1; }();
    return k;
}

template <int WM>
int launch_gemm_reg(const float *V, const float *U, float *M, int nt, int Ci, int Co, int np,
                    hipStream_t stream) {
    const int mblocks = ceil_div(nt, WM * 48);
    static const int xg_env = [] { const char *e = getenv("DECNET_WINO_XG"); return e ? atoi(e) : 0; }();
    int xg = 1;
    if (xg_env > 0 && np % xg_env == 0) xg = xg_env;
    static const int swz = [] { const char *e = getenv("DECNET_WINO_SWZ"); return e ? atoi(e) : 1; }();
    if ((Ci + 15) / 16 == 14)
        hipLaunchKernelGGL((wino_gemm_reg<WM, 14>), dim3(mblocks, np / xg), dim3(WM * 128), 0, stream, V, U, M,
                           nt, Ci, Co, xg, np, swz);
    else
        hipLaunchKernelGGL((wino_gemm_reg<WM, 0>), dim3(mblocks, np / xg), dim3(WM * 128), 0, stream, V, U, M,
                           nt, Ci, Co, xg, np, swz);
    return decnet_launch_status();
}

template <int WM, int BK>
int launch_gemm(const float *V, const float *U, float *M, int nt, int Ci, int Co, int np,
                hipStream_t stream) {
    constexpr int BM = WM * 48;
    const size_t lds = 4 * (size_t)(2 * BM * (BK + 2) + 2 * BK * WB_PITCH);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)wino_gemm<WM, BK>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // transform points per workgroup (a divisor of np): fill whole rounds of the 256 CUs, and
    // prefer more points per workgroup (one pipeline fill amortised over more K chunks)
    const int mblocks = ceil_div(nt, BM);
    int xg = 1;
    double best = -1.0;
    for (int c = 1; c <= np; ++c) {
        if (np % c) continue;
        const long blocks = (long)mblocks * (np / c);
        const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256) * (1.0 - 0.12 / c);
        if (eff > best + 1e-9) { best = eff; xg = c; }
    }
    static const int xg_env = [] { const char *e = getenv("DECNET_WINO_XG"); return e ? atoi(e) : 0; }();
    if (xg_env > 0 && np % xg_env == 0) xg = xg_env;
    static const int swz = [] { const char *e = getenv("DECNET_WINO_SWZ"); return e ? atoi(e) : 1; }();
    hipLaunchKernelGGL((wino_gemm<WM, BK>), dim3(mblocks, np / xg), dim3(WM * 128), lds, stream, V, U, M,
                       nt, Ci, Co, xg, np, swz);
    return decnet_launch_status();
}

int gemm_dispatch(const float *V, const float *U, float *M, int nt, int Ci, int Co, int np,
                  hipStream_t s) {
    static const int tile_env = [] { const char *e = getenv("DECNET_WINO_TILE"); return e ? atoi(e) : 0; }();
    if (gemm_kind() == 1)
        return tile_env == 96 ? launch_gemm_reg<2>(V, U, M, nt, Ci, Co, np, s)
                              : launch_gemm_reg<4>(V, U, M, nt, Ci, Co, np, s);
    if (tile_env == 96 || !(Ci % 36 == 0 && (long)ceil_div(nt, 192) * np >= 256))
        return launch_gemm<2, 24>(V, U, M, nt, Ci, Co, np, s);
    return launch_gemm<4, 36>(V, U, M, nt, Ci, Co, np, s);
}

// tiles per chunk: V + M of one chunk (2 * np * nt * C floats) <= DECNET_WINO_CHUNK_MB (1 GiB); equal chunks
int chunk_tiles(int T, int C, int np) {
    static const double cap_mb = [] {
        const char *e = getenv("DECNET_WINO_CHUNK_MB");       // experiments: V+M bytes per chunk
        return e ? atof(e) : 1024.0;
    }();
    long cap = (long)(cap_mb * 1024 * 1024 / (2.0 * np * 4 * C));
    if (cap < 192) cap = 192;
    const long nchunks = (T + cap - 1) / cap;          // equal chunks
    return (int)((T + nchunks - 1) / nchunks);
}

template <int OD, int OH, int OW>
int conv_variant(const float *x, const float *u, const float *scale, const float *shift,
                 const float *residual, float *y, float *workspace, int B, int D, int H, int W, int Ci,
                 int Co, int relu, hipStream_t s) {
    constexpr int TD = OD + 2, TH = OH + 2, TW = OW + 2, NP = TD * TH * TW;
    Tiling g{D, H, W, ceil_div(D, OD), ceil_div(H, OH), ceil_div(W, OW)};
    const double Td = (double)B * g.Td * g.Th * g.Tw;
    if (Td >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    const int T = (int)Td, cmax = Ci > Co ? Ci : Co;
    const int ct = chunk_tiles(T, cmax, NP);
    if ((double)ct * cmax * 4 * NP >= 2147483647.0) return DECNET_ERR_UNSUPPORTED;   // 32-bit offsets
    float *V = workspace, *M = workspace + (size_t)NP * ct * Ci;
    for (int t_lo = 0; t_lo < T; t_lo += ct) {
        const int nt = T - t_lo < ct ? T - t_lo : ct;
        size_t n = (size_t)nt * Ci;
        hipLaunchKernelGGL((wino_input_transform<TD, TH, TW>), dim3((unsigned)((n + 255) / 256)), dim3(256),
                           0, s, x, V, g, Ci, t_lo, nt);
        int rc = decnet_launch_status();
        if (rc) return rc;
        if ((rc = gemm_dispatch(V, u, M, nt, Ci, Co, NP, s))) return rc;
        n = (size_t)nt * Co;
        hipLaunchKernelGGL((wino_output_transform<TD, TH, TW>), dim3((unsigned)((n + 255) / 256)), dim3(256),
                           0, s, M, scale, shift, residual, y, g, Co, relu, t_lo, nt);
        if ((rc = decnet_launch_status())) return rc;
    }
    return DECNET_OK;
}

int variant_points(int variant) { return variant == 0 ? 64 : variant == 1 ? 144 : -1; }

}  // namespace

extern "C" {

/* variant: 0 = F(2,3)^3 (64 transform points), 1 = F(2,3) on D x F(4,3) on H, W (144 points) */
size_t decnet_conv3d_wino_weight_floats(int Ci, int variant) {
    const int np = variant_points(variant);
    return np < 0 || Ci < 1 ? 0 : (size_t)np * Ci * W_BN;
}

int decnet_conv3d_wino_pack_weight(const float *w, float *u, int Co, int Ci, int variant, void *stream) {
    if (!w || !u) return DECNET_ERR_NULL_POINTER;
    if (Co < 1 || Ci < 1 || variant_points(variant) < 0) return DECNET_ERR_BAD_SHAPE;
    if (Co > W_BN) return DECNET_ERR_UNSUPPORTED;
    const int n = Ci * W_BN;
    if (variant == 0)
        hipLaunchKernelGGL((wino_weight_transform<4, 4, 4>), dim3(ceil_div(n, 128)), dim3(128), 0,
                           (hipStream_t)stream, w, u, Co, Ci, gemm_kind());
    else
        hipLaunchKernelGGL((wino_weight_transform<4, 6, 6>), dim3(ceil_div(n, 128)), dim3(128), 0,
                           (hipStream_t)stream, w, u, Co, Ci, gemm_kind());
    return decnet_launch_status();
}

/* The batched GEMM stage alone (measurement / composition): M[xi] = V[xi] * U[xi], xi < np points,
 * V [np][nt][Ci], U from decnet_conv3d_wino_pack_weight, M [np][nt][Co]. */
int decnet_conv3d_wino_gemm(const float *V, const float *u, float *M, int nt, int Ci, int Co,
                            int variant, void *stream) {
    const int np = variant_points(variant);
    if (!V || !u || !M) return DECNET_ERR_NULL_POINTER;
    if (nt < 1 || Ci < 1 || Co < 1 || np < 0) return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0 || Co > W_BN || (double)nt * (Ci > Co ? Ci : Co) * 4 * np >= 2147483647.0)
        return DECNET_ERR_UNSUPPORTED;
    return gemm_dispatch(V, u, M, nt, Ci, Co, np, (hipStream_t)stream);
}

size_t decnet_conv3d_wino_workspace_floats(int B, int D, int H, int W, int Ci, int Co, int variant) {
    const int np = variant_points(variant);
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1 || np < 0) return 0;
    const int oh = variant == 0 ? 2 : 4;
    const double T = (double)B * ((D + 1) / 2) * ((H + oh - 1) / oh) * ((W + oh - 1) / oh);
    if (T >= 2147483648.0) return 0;
    const int nt = chunk_tiles((int)T, Ci > Co ? Ci : Co, np);
    return (size_t)np * nt * ((size_t)Ci + Co);
}

int decnet_conv3d_wino_bn_act(const float *x, const float *u, const float *scale, const float *shift,
                              const float *residual, float *y, float *workspace, int B, int D, int H,
                              int W, int Ci, int Co, int relu, int variant, void *stream) {
    if (!x || !u || !scale || !shift || !y || !workspace) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1 || variant_points(variant) < 0)
        return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0 || Co > W_BN) return DECNET_ERR_UNSUPPORTED;
    if ((double)B * D * H * W * (Ci > Co ? Ci : Co) >= 2147483648.0 * 4) return DECNET_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0)
        return conv_variant<2, 2, 2>(x, u, scale, shift, residual, y, workspace, B, D, H, W, Ci, Co, relu, s);
    return conv_variant<2, 4, 4>(x, u, scale, shift, residual, y, workspace, B, D, H, W, Ci, Co, relu, s);
}

}  // extern "C"
