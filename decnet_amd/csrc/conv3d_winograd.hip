// decnet_amd/csrc/conv3d_winograd.hip -- Conv3d(k3,s1,p1)+BN+ReLU by Winograd minimal filtering in
// fp32 on the matrix cores (gfx950).  Same operator as stage0.hip:conv3d_k3_igemm -- one Conv3dUnit
// of CostRegNetNoDown in eval mode (submodule.py:115-123, 608-662) -- with 3.4x / 6x fewer
// multiplications: every block of outputs is computed from an input tile as
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        (applied along D, H and W)
// so the 27-tap implicit GEMM (K = 27*Ci) becomes one small GEMM with K = Ci per transform point:
//     M[xi][tile][co] = sum_ci V[xi][tile][ci] * U[xi][ci][co]
//   V = B^T-transformed input tiles (adds only), U = G-transformed weights (once per weight
//   version), Y = A^T-transformed M (adds only) -> BN scale/shift -> ReLU -> (+ residual).
// fp32 throughout.  Two tile shapes (the `variant` argument of the entry points):
//   0  F(2,3) on D, H and W: 4x4x4 input tile -> 2x2x2 outputs, 64 transform points, 8 multiplies
//      per output (27 direct).  Constants 0, +-1, +-1/2: as accurate as the direct convolution
//      (tests/conv_numerics.py, against float64: 1.8e-6 relative on the regularised volume, the
//      direct kernel 2.9e-6, torch CPU 1.5e-6).
//   1  F(2,3) on D, F(4,3) on H and W: 4x6x6 tile -> 2x4x4 outputs, 144 points, 4.5 multiplies
//      per output; constants up to 8 and 1/24: 6.0e-6 relative on the volume, 2.3e-5 px mean on
//      the disparity (direct: 1.1e-5) -- 40x inside the 1e-3 px budget.
// tests/test_stage0_gpu.py checks every algorithm against the same oracle.
//
// Layout of the three intermediates ("chunk major": 16 channels = 64 bytes are the unit):
//     V   [point][ceil(Ci/16)][tile][16]      U^T [point][ceil(Ci/16)][224 co][16]
//     M   [point][ceil(Co/16)][tile][16]
// so that (a) a wave of the GEMM reads the 16 rows x 16 k of an MFMA operand as ONE contiguous
// 1 KiB load and writes a 16x16 result tile as one contiguous 1 KiB store, and (b) a wave of the
// transform kernels (16 channels x 4 tiles) moves 256 contiguous bytes per transform point.
//
// Tiles are processed in chunks of at most 1 GiB of V + M; chunks small enough to stay in the
// 256 MiB Infinity Cache between the three kernels were measured and gain nothing.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int W_BN = 224;      // co rows of U^T per transform point: 14 MFMA tiles of 16
__host__ __device__ constexpr int pad16(int c) { return (c + 15) & ~15; }

// 1-D transforms of F(m,3), m = O outputs, tile T = O + 2 (Lavin & Gray, arXiv:1509.09308)
template <int T> __device__ __forceinline__ void bt_1d(float (&v)[T]);
template <> __device__ __forceinline__ void bt_1d<4>(float (&v)[4]) {
    const float t0 = v[0] - v[2], t1 = v[1] + v[2], t2 = v[2] - v[1], t3 = v[1] - v[3];
    v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;
}
// F(4,3) on the points {0, +-3/4, +-3/2, inf} instead of Lavin & Gray's {0, +-1, +-2, inf}: every entry of
// B^T and A^T is still exact in binary, and the fp32 error of a 216-channel 3-D layer drops from
// 6.1e-6 to 2.2e-6 of max|y| for F(2,3)xF(4,3)^2 and from 2.6e-5 to 4.2e-6 for F(4,3)^3 (numpy
// emulation against float64; cf. Barabasz et al., arXiv:1803.10986, on point selection).
template <> __device__ __forceinline__ void bt_1d<6>(float (&v)[6]) {
    const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5];
    const float e = a4 - 2.25f * a2, o = 0.75f * a3 - 1.6875f * a1;
    const float f = a4 - 0.5625f * a2, q = 1.5f * a3 - 0.84375f * a1;
    v[0] = 1.265625f * a0 - 2.8125f * a2 + a4;
    v[1] = e + o;
    v[2] = e - o;
    v[3] = f + q;
    v[4] = f - q;
    v[5] = 1.265625f * a1 - 2.8125f * a3 + a5;
}
template <int T> __device__ __forceinline__ void g_1d(const float (&g)[3], float (&o)[T]);
template <> __device__ __forceinline__ void g_1d<4>(const float (&g)[3], float (&o)[4]) {
    o[0] = g[0]; o[1] = 0.5f * (g[0] + g[1] + g[2]); o[2] = 0.5f * (g[0] - g[1] + g[2]); o[3] = g[2];
}
template <> __device__ __forceinline__ void g_1d<6>(const float (&g)[3], float (&o)[6]) {
    // rows [1, p, p^2] / N_p, N_p = prod_{k != p} (p - k): N_0 = 81/64, N_{+-3/4} = -243/128, N_{+-3/2} = 243/32
    const float ea = (-128.f / 243.f) * (g[0] + 0.5625f * g[2]), oa = (-96.f / 243.f) * g[1];
    const float eb = (32.f / 243.f) * (g[0] + 2.25f * g[2]), ob = (48.f / 243.f) * g[1];
    o[0] = (64.f / 81.f) * g[0];
    o[1] = ea + oa;
    o[2] = ea - oa;
    o[3] = eb + ob;
    o[4] = eb - ob;
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
    o[1] = 0.75f * d12 + 1.5f * d34;
    o[2] = 0.5625f * s12 + 2.25f * s34;
    o[3] = 0.421875f * d12 + 3.375f * d34 + m[5];
}

// ------------------------------ weight transform (once) --------------------------------
// w [Co][Ci][3][3][3] (torch) -> U^T [TD*TH*TW][ceil(Ci/16)][224][16], U = G w G^T along the three
// axes; co >= Co rows are zero, ci >= Ci slots of the last chunk are never read.
template <int TD, int TH, int TW>
__global__ void wino_weight_transform(const float *__restrict__ w, float *__restrict__ U, int Co,
                                      int Ci) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (co, ci), ci fastest
    if (idx >= Ci * W_BN) return;
    const int ci = idx % Ci, co = idx / Ci;
    const int KC = (Ci + 15) >> 4;
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
    float *up = U + ((size_t)(ci >> 4) * W_BN + co) * 16 + (ci & 15);
    const size_t ps = (size_t)KC * W_BN * 16;
#pragma unroll
    for (int jj = 0; jj < TH; ++jj)
#pragma unroll
        for (int k = 0; k < TW; ++k) {
            const float g[3] = {t2[0][jj][k], t2[1][jj][k], t2[2][jj][k]};
            float o[TD];
            g_1d<TD>(g, o);
#pragma unroll
            for (int i = 0; i < TD; ++i) up[(size_t)((i * TH + jj) * TW + k) * ps] = o[i];
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
// x [B,D,H,W,C] -> V[xi][c/16][tile - t_lo][c%16], V = B^T d B along D, H, W.  One thread per
// (tile, channel).  Measured alternatives that were not faster: 4 tiles x 16 channels per wave
// (256-byte V stores, 64-byte x loads), and staging a row of tiles through LDS (phases serialise:
// two workgroups per CU are not enough to overlap them).
// KS = 2 (tiles 6 wide): two threads share a (tile, channel); each loads all of it, keeps only ITS
// three W-transformed columns and finishes those.  A 6x6x6 tile in one thread is 216 live values:
// 257 registers, one wave per SIMD, 2.8 lock-step rounds of waves (0.053 ms for 179 MB); halved it
// is 108 values and three waves per SIMD; the doubled loads hit in L1/L2.
template <int TD, int TH, int TW, int KS, int HALF>
__device__ __forceinline__ void wino_input_body(const float *__restrict__ x, float *__restrict__ V, const Tiling &g,
                                                int C, int t_lo, int nt, int x_bytes, int tl, int c) {
    constexpr int KW = TW / KS, K0 = HALF * KW;
    const int KC = (C + 15) >> 4, kc = c >> 4;
    int b, z0, y0, x0;
    tile_coords<TD - 2, TH - 2, TW - 2>(t_lo + tl, g, b, z0, y0, x0);
    // branch-free halo: out-of-volume taps are sent past the end of the buffer and read as zeros, so
    // the loads of a thread are in flight together (with branches the compiler waits for every row
    // of loads before the next)
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    float d[TD][TH][KW];
#pragma unroll
    for (int i = 0; i < TD; ++i) {
        const int z = z0 - 1 + i;
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) {
            const int y = y0 - 1 + jj;
            const bool okzy = (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H;
            const int row = ((((b * g.D + z) * g.H + y) * g.W + x0 - 1) * C + c) * 4;
            float r[TW];
#pragma unroll
            for (int k = 0; k < TW; ++k) {
                const bool ok = okzy && (unsigned)(x0 - 1 + k) < (unsigned)g.W;
                r[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, ok ? row + k * C * 4 : OOB, 0, 0));
            }
            bt_1d<TW>(r);
#pragma unroll
            for (int k = 0; k < KW; ++k) d[i][jj][k] = r[K0 + k];
        }
    }
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int k = 0; k < KW; ++k) {
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
        for (int k = 0; k < KW; ++k) {
            float v[TD];
#pragma unroll
            for (int i = 0; i < TD; ++i) v[i] = d[i][jj][k];
            bt_1d<TD>(v);
#pragma unroll
            for (int i = 0; i < TD; ++i) d[i][jj][k] = v[i];
        }
    float *o = V + ((size_t)kc * nt + tl) * 16 + (c & 15);
    const size_t xs = (size_t)KC * nt * 16;
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int jj = 0; jj < TH; ++jj)
#pragma unroll
            for (int k = 0; k < KW; ++k) o[(size_t)((i * TH + jj) * TW + K0 + k) * xs] = d[i][jj][k];
}

template <int TD, int TH, int TW, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wino_input_transform(const float *__restrict__ x,
                                                            float *__restrict__ V, Tiling g, int C,
                                                            int t_lo, int nt, int x_bytes) {
    // workgroup = one tile (one half of it for KS = 2), all channels (a wave = 64 consecutive channels:
    // whole cache lines of x); XCD aware: ids equal mod 8 (one XCD, one L2) walk a contiguous range of
    // tiles, so the halo shared by neighbouring tiles, the two halves of a tile and the second half
    // of V's 128-byte lines meet in the same L2
    const int per_xcd = (nt + 7) >> 3;
    const int tl = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int half = blockIdx.y % KS, cblk = blockIdx.y / KS;
    const int c = cblk * blockDim.x + threadIdx.x;                  // cblk > 0 only if C > 256
    if ((int)(blockIdx.x >> 3) >= per_xcd || tl >= nt || c >= C) return;
    if (KS == 1 || half == 0) wino_input_body<TD, TH, TW, KS, 0>(x, V, g, C, t_lo, nt, x_bytes, tl, c);
    else wino_input_body<TD, TH, TW, KS, KS - 1>(x, V, g, C, t_lo, nt, x_bytes, tl, c);
}

// M[xi][co/16][tile - t_lo][co%16] -> y: A^T along W, H, D, then BN scale/shift, ReLU, + residual
// (CostRegNetNoDown.forward submodule.py:656).  Thread mapping as the input transform.
template <int TD, int TH, int TW, int KS, int HALF>
__device__ __forceinline__ void wino_output_body(
    const float *__restrict__ M, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ residual, float *__restrict__ y, const Tiling &g, int Co, int relu, int t_lo,
    int nt, int y_bytes, int cg, int tl, int lane16) {
    constexpr int OD = TD - 2, OH = TH - 2, OW = TW - 2, OWH = OW / KS, X0 = HALF * OWH;
    const int CG = (Co + 15) >> 4, co = cg * 16 + lane16;
    int b, z0, y0, x0;
    tile_coords<OD, OH, OW>(t_lo + tl, g, b, z0, y0, x0);
    const float *mp = M + ((size_t)cg * nt + tl) * 16 + lane16;
    const size_t xs = (size_t)CG * nt * 16;
    float a[TD][TH][OWH], bb[TD][OH][OWH];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) {
            float m[TW], o[OW];
#pragma unroll
            for (int k = 0; k < TW; ++k) m[k] = mp[(size_t)((i * TH + jj) * TW + k) * xs];
            at_1d<TW>(m, o);
#pragma unroll
            for (int k = 0; k < OWH; ++k) a[i][jj][k] = o[X0 + k];
            // at most two depth planes of loads in flight: 216 loads at once need 380 registers
            if (TD * TH * TW > 160 && jj == TH - 1 && (i & 1)) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int k = 0; k < OWH; ++k) {
            float m[TH], o[OH];
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) m[jj] = a[i][jj][k];
            at_1d<TH>(m, o);
#pragma unroll
            for (int jj = 0; jj < OH; ++jj) bb[i][jj][k] = o[jj];
        }
    const float sc = scale[co], sh = shift[co];
    // branch-free epilogue (buffer accesses past the end are dropped / read as zero): the residual
    // loads of all outputs are in flight together
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(residual ? residual : y), 0, y_bytes, 0x00020000);
    // the epilogue runs in groups of JG output rows: the residual loads of a group are in flight
    // together, and a 4x4x4 tile does not need 64 outputs + 64 residuals live next to bb
    constexpr int JG = (OD * OH * OWH > 32) ? 2 : OH;
#pragma unroll
    for (int j0 = 0; j0 < OH; j0 += JG) {
        float out[OD][JG][OWH], res[OD][JG][OWH];
#pragma unroll
        for (int jj = 0; jj < JG; ++jj)
#pragma unroll
            for (int k = 0; k < OWH; ++k) {
                float m[TD], o[OD];
#pragma unroll
                for (int i = 0; i < TD; ++i) m[i] = bb[i][j0 + jj][k];
                at_1d<TD>(m, o);
#pragma unroll
                for (int i = 0; i < OD; ++i) {
                    const int z = z0 + i, yy = y0 + j0 + jj, xx = x0 + X0 + k;
                    const int off = z < g.D && yy < g.H && xx < g.W
                                        ? ((((b * g.D + z) * g.H + yy) * g.W + xx) * Co + co) * 4 : OOB;
                    res[i][jj][k] = residual ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, off, 0, 0)) : 0.f;
                    float v = fmaf(o[i], sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    out[i][jj][k] = v;
                }
            }
#pragma unroll
        for (int i = 0; i < OD; ++i)
#pragma unroll
            for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                for (int k = 0; k < OWH; ++k) {
                    const int z = z0 + i, yy = y0 + j0 + jj, xx = x0 + X0 + k;
                    const int off = z < g.D && yy < g.H && xx < g.W
                                        ? ((((b * g.D + z) * g.H + yy) * g.W + xx) * Co + co) * 4 : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(out[i][jj][k] + res[i][jj][k]), yr, off, 0, 0);
                }
        if (JG < OH) __builtin_amdgcn_sched_barrier(0);
    }
}

// A^T of F(4,3) / F(2,3) as a table, for the plane-streaming output transform below
__constant__ float AT6_TAB[6][4] = {{1.f, 0.f, 0.f, 0.f},       {1.f, 0.75f, 0.5625f, 0.421875f},
                                    {1.f, -0.75f, 0.5625f, -0.421875f}, {1.f, 1.5f, 2.25f, 3.375f},
                                    {1.f, -1.5f, 2.25f, -3.375f},  {0.f, 0.f, 0.f, 1.f}};

// 6x6x6 tiles: the depth planes of M stream through two 36-value buffers (plane i + 1 in flight while
// plane i is reduced 36 -> 16 by the W and H passes) and the D pass is an accumulation
// out[z] += A^T[z][i] * plane_i, so a thread never holds more than 64 accumulators + 72 loads
// (the all-in-registers form above needs 337 registers for 216 points: one wave per SIMD).
template <int TH, int TW>
__device__ __forceinline__ void wino_output_stream6(
    const float *__restrict__ M, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ residual, float *__restrict__ y, const Tiling &g, int Co, int relu, int t_lo,
    int nt, int y_bytes, int cg, int tl, int lane16) {
    constexpr int TD = 6, OD = 4, OH = TH - 2, OW = TW - 2;
    const int CG = (Co + 15) >> 4, co = cg * 16 + lane16;
    int b, z0, y0, x0;
    tile_coords<OD, OH, OW>(t_lo + tl, g, b, z0, y0, x0);
    const float *mp = M + ((size_t)cg * nt + tl) * 16 + lane16;
    const size_t xs = (size_t)CG * nt * 16, ps = xs * TH * TW;       // point stride, plane stride
    float acc[OD][OH][OW];
#pragma unroll
    for (int i = 0; i < OD; ++i)
#pragma unroll
        for (int jj = 0; jj < OH; ++jj)
#pragma unroll
            for (int k = 0; k < OW; ++k) acc[i][jj][k] = 0.f;
    float p0[TH][TW], p1[TH][TW];
    auto load_plane = [&](float (&p)[TH][TW], const float *base) {
#pragma unroll
        for (int jj = 0; jj < TH; ++jj)
#pragma unroll
            for (int k = 0; k < TW; ++k) p[jj][k] = base[(size_t)(jj * TW + k) * xs];
    };
    auto consume = [&](float (&p)[TH][TW], int i) {
        float a[TH][OW], h[OH][OW];
#pragma unroll
        for (int jj = 0; jj < TH; ++jj) at_1d<TW>(p[jj], a[jj]);
#pragma unroll
        for (int k = 0; k < OW; ++k) {
            float m[TH], o[OH];
#pragma unroll
            for (int jj = 0; jj < TH; ++jj) m[jj] = a[jj][k];
            at_1d<TH>(m, o);
#pragma unroll
            for (int jj = 0; jj < OH; ++jj) h[jj][k] = o[jj];
        }
        const float c0 = AT6_TAB[i][0], c1 = AT6_TAB[i][1], c2 = AT6_TAB[i][2], c3 = AT6_TAB[i][3];
#pragma unroll
        for (int jj = 0; jj < OH; ++jj)
#pragma unroll
            for (int k = 0; k < OW; ++k) {
                acc[0][jj][k] = fmaf(c0, h[jj][k], acc[0][jj][k]);
                acc[1][jj][k] = fmaf(c1, h[jj][k], acc[1][jj][k]);
                acc[2][jj][k] = fmaf(c2, h[jj][k], acc[2][jj][k]);
                acc[3][jj][k] = fmaf(c3, h[jj][k], acc[3][jj][k]);
            }
    };
    load_plane(p0, mp);
#pragma unroll 1
    for (int i = 0; i < TD; i += 2) {
        load_plane(p1, mp + (size_t)(i + 1) * ps);
        __builtin_amdgcn_sched_barrier(0);
        consume(p0, i);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 2 < TD) load_plane(p0, mp + (size_t)(i + 2) * ps);
        __builtin_amdgcn_sched_barrier(0);
        consume(p1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const float sc = scale[co], sh = shift[co];
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(residual ? residual : y), 0, y_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < OD; ++i) {                      // one output depth plane at a time: 16 residual loads in flight
        float res[OH][OW];
        int off[OH][OW];
#pragma unroll
        for (int jj = 0; jj < OH; ++jj)
#pragma unroll
            for (int k = 0; k < OW; ++k) {
                const int z = z0 + i, yy = y0 + jj, xx = x0 + k;
                off[jj][k] = z < g.D && yy < g.H && xx < g.W ? ((((b * g.D + z) * g.H + yy) * g.W + xx) * Co + co) * 4 : OOB;
                res[jj][k] = residual ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, off[jj][k], 0, 0)) : 0.f;
            }
#pragma unroll
        for (int jj = 0; jj < OH; ++jj)
#pragma unroll
            for (int k = 0; k < OW; ++k) {
                float v = fmaf(acc[i][jj][k], sc, sh);
                if (relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v + res[jj][k]), yr, off[jj][k], 0, 0);
            }
    }
}

// KS = 2: two threads per (tile, co), each finishing one half of the output columns (see the input
// transform); blockIdx.y = half.
template <int TD, int TH, int TW, int KS>
__global__ __launch_bounds__(256) void wino_output_transform(
    const float *__restrict__ M, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ residual, float *__restrict__ y, Tiling g, int Co, int relu, int t_lo,
    int nt, int y_bytes) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CG = (Co + 15) >> 4;
    const size_t q = idx >> 4;
    const int cg = (int)(q / nt), tl = (int)(q - (size_t)cg * nt);
    if (cg >= CG || cg * 16 + (int)(idx & 15) >= Co) return;
    if (TD == 6 && KS == 1) {
        wino_output_stream6<TH, TW>(M, scale, shift, residual, y, g, Co, relu, t_lo, nt, y_bytes, cg, tl, (int)(idx & 15));
        return;
    }
    if (KS == 1 || blockIdx.y == 0)
        wino_output_body<TD, TH, TW, KS, 0>(M, scale, shift, residual, y, g, Co, relu, t_lo, nt, y_bytes, cg, tl, (int)(idx & 15));
    else
        wino_output_body<TD, TH, TW, KS, KS - 1>(M, scale, shift, residual, y, g, Co, relu, t_lo, nt, y_bytes, cg, tl, (int)(idx & 15));
}

// ------------------------------ output transform of layer i + input transform of layer i + 1 ----
// Between two Winograd layers the activation y never has to reach HBM: one workgroup owns one sample x four
// channels, turns the sample's M tiles into y = relu(M A ... * scale + shift) (+ residual) IN LDS (the whole
// D x H x W volume of four channels: 92 KB at 8 x 20 x 36) and, after one barrier, reads the overlapping 6^3
// input tiles of the next layer back out of LDS and writes their B^T transforms as V.  HBM traffic per
// layer boundary: M once in, V once out (2 x 134 MB at config 2) instead of M in, y out, y in (through L2,
// 3.4 x overlapped) and V out.  Both intermediates are in the QUAD-major layout here
//     V, M   [point][ceil(C/16)][4 quads][tile][4 channels]
// so that a workgroup's share of a transform point is one contiguous run (tiles of a sample x 16 bytes) and
// a wave moves 256 contiguous bytes per instruction; wino_gemm_bf16x3 takes either layout (strides).
// The residual connection (CostRegNetNoDown.forward submodule.py:656-658: output0 is added back three layers
// later) travels in a private planar layout R [sample][quad][D][H][4][W | 1] = a dump of the workgroup's LDS.
// Phase 1 is wino_output_stream6 per (tile, channel); phase 2 is wino_input_body with two threads per
// (tile, channel), LDS instead of x.
constexpr int MID_THREADS = 512;

template <int HALF, int NCH = 4>
__device__ __forceinline__ void mid_input_half(const float *__restrict__ ys, __amdgpu_buffer_rsrc_t vr, int voff, int xs,
                                               const Tiling &g, int Wp, int ch, int z0, int y0, int x0) {
    constexpr int T = 6, KW = 3, K0 = HALF * KW;
    float d[T][T][KW];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int z = z0 - 1 + i;
#pragma unroll
        for (int jj = 0; jj < T; ++jj) {
            const int y = y0 - 1 + jj;
            const bool okzy = (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H;
            const float *row = ys + ((okzy ? z * g.H + y : 0) * NCH + ch) * Wp;
            float r[T];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const int x = x0 - 1 + k;
                const bool ok = okzy && (unsigned)x < (unsigned)g.W;
                const float v = row[ok ? x : 0];
                r[k] = ok ? v : 0.f;
            }
            bt_1d<T>(r);
#pragma unroll
            for (int k = 0; k < KW; ++k) d[i][jj][k] = r[K0 + k];
        }
        // one depth plane of LDS reads at a time: hoisting all 216 reads (+ their predicates) spills
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            float v[T];
#pragma unroll
            for (int jj = 0; jj < T; ++jj) v[jj] = d[i][jj][k];
            bt_1d<T>(v);
#pragma unroll
            for (int jj = 0; jj < T; ++jj) d[i][jj][k] = v[jj];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jj = 0; jj < T; ++jj) {
        // (opaque copy of the point stride per row: otherwise all 108 scalar offsets are computed up front and
        // spill out of the scalar registers into vector lanes)
        int xo = xs;
        asm volatile("" : "+s"(xo));
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            float v[T];
#pragma unroll
            for (int i = 0; i < T; ++i) v[i] = d[i][jj][k];
            bt_1d<T>(v);
#pragma unroll
            for (int i = 0; i < T; ++i)                 // the point offset is uniform: scalar operand of the store
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v[i]), vr, voff, ((i * T + jj) * T + K0 + k) * xo, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Phase 1 of wino_mid_transform: the sample's M tiles of four channels -> y = relu(A^T M A * scale +
// shift) (+ what the LDS volume held before: the preloaded residual) into the LDS volume [D][H][4][Wp].  A^T along W and H
// per depth plane of M, the D pass as an accumulation; one thread per (tile, channel), two planes of M in flight.
// NCH channels c0 .. c0 + NCH - 1 of the quad (4, or 2 for the half-quad workgroups of wino_mid_transform2), NTHR threads.
template <int NCH = 4, int NTHR = MID_THREADS>
__device__ __forceinline__ void wino_output_phase(__amdgpu_buffer_rsrc_t mr, float *__restrict__ ys, const Tiling &g, int Wp,
                                                  int cq, int b, int nts, int nt, int xs, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, int C, int relu, bool res_in, int c0 = 0) {
    constexpr int T = 6, O = 4;
    const int items = nts * NCH;
    for (int it = threadIdx.x; it < items; it += NTHR) {
        const int ch = it % NCH, tl = it / NCH;
        const int co = cq * 4 + c0 + ch;
        int bb, z0, y0, x0;
        tile_coords<O, O, O>(tl, g, bb, z0, y0, x0);
        const int moff = ((cq * nt + b * nts + tl) * 4 + c0 + ch) * 4;
        float acc[O][O][O];
#pragma unroll
        for (int i = 0; i < O; ++i)
#pragma unroll
            for (int jj = 0; jj < O; ++jj)
#pragma unroll
                for (int k = 0; k < O; ++k) acc[i][jj][k] = 0.f;
        float p0[T][T], p1[T][T];
        auto load_plane = [&](float (&p)[T][T], int i) {
            int xo = xs;                                // opaque per plane, see mid_input_half
            asm volatile("" : "+s"(xo));
            const int so = i * (T * T) * xo;
#pragma unroll
            for (int jj = 0; jj < T; ++jj)
#pragma unroll
                for (int k = 0; k < T; ++k)
                    p[jj][k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(mr, moff, so + (jj * T + k) * xo, 0));
        };
        auto consume = [&](float (&p)[T][T], int i) {
            float a[T][O], h[O][O];
#pragma unroll
            for (int jj = 0; jj < T; ++jj) at_1d<T>(p[jj], a[jj]);
#pragma unroll
            for (int k = 0; k < O; ++k) {
                float m[T], o[O];
#pragma unroll
                for (int jj = 0; jj < T; ++jj) m[jj] = a[jj][k];
                at_1d<T>(m, o);
#pragma unroll
                for (int jj = 0; jj < O; ++jj) h[jj][k] = o[jj];
            }
            const float c0 = AT6_TAB[i][0], c1 = AT6_TAB[i][1], c2 = AT6_TAB[i][2], c3 = AT6_TAB[i][3];
#pragma unroll
            for (int jj = 0; jj < O; ++jj)
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    acc[0][jj][k] = fmaf(c0, h[jj][k], acc[0][jj][k]);
                    acc[1][jj][k] = fmaf(c1, h[jj][k], acc[1][jj][k]);
                    acc[2][jj][k] = fmaf(c2, h[jj][k], acc[2][jj][k]);
                    acc[3][jj][k] = fmaf(c3, h[jj][k], acc[3][jj][k]);
                }
        };
        load_plane(p0, 0);
#pragma unroll 1
        for (int i = 0; i < T; i += 2) {
            load_plane(p1, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(p0, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < T) load_plane(p0, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            consume(p1, i + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float sc = co < C ? scale[co] : 0.f, sh = co < C ? shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < O; ++i)
#pragma unroll
            for (int jj = 0; jj < O; ++jj) {
                const int z = z0 + i, yy = y0 + jj;
                float *row = ys + ((z * g.H + yy) * NCH + ch) * Wp + x0;
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    if (z < g.D && yy < g.H && x0 + k < g.W) {
                        float v = fmaf(acc[i][jj][k], sc, sh);
                        if (relu) v = fmaxf(v, 0.f);
                        if (res_in) v += row[k];
                        row[k] = v;
                    }
                }
            }
    }
}

__global__ __launch_bounds__(MID_THREADS) void wino_mid_transform(
    const float *__restrict__ M, float *__restrict__ V, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ res_in, float *__restrict__ res_out, Tiling g,
    int C, int nt, int relu) {
    extern __shared__ float ys[];                       // [D][H][4][Wp]
    constexpr int O = 4;
    const int Wp = g.W | 1, vol = g.D * g.H * 4 * Wp;
    const int nq = (C + 3) >> 2, Q = pad16(C) >> 2;     // quads with data, quads per transform point
    const int b = blockIdx.x / nq, cq = blockIdx.x - b * nq;
    const int nts = g.Td * g.Th * g.Tw, items = nts * 4;
    const int xs = Q * nt * 16;                         // bytes between transform points
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void *)M, 0, 216 * xs, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, 216 * xs, 0x00020000);
    const size_t blk = ((size_t)b * nq + cq) * vol;     // this workgroup's block of R
    if (res_in) {
        for (int i = threadIdx.x * 4; i < vol; i += MID_THREADS * 4) {
            if (i + 4 <= vol) *reinterpret_cast<f32x4 *>(ys + i) = *reinterpret_cast<const f32x4 *>(res_in + blk + i);
            else for (int e = i; e < vol; ++e) ys[e] = res_in[blk + e];
        }
        __syncthreads();
    }
    // ---- phase 1: M -> y ----
    wino_output_phase(mr, ys, g, Wp, cq, b, nts, nt, xs, scale, shift, C, relu, res_in != nullptr);
    __syncthreads();
    if (res_out) {
        for (int i = threadIdx.x * 4; i < vol; i += MID_THREADS * 4) {
            if (i + 4 <= vol) *reinterpret_cast<f32x4 *>(res_out + blk + i) = *reinterpret_cast<const f32x4 *>(ys + i);
            else for (int e = i; e < vol; ++e) res_out[blk + e] = ys[e];
        }
    }
    // ---- phase 2: y -> V (B^T along W, H, D), two threads per (tile, channel): each keeps three of the six
    // W-transformed columns ----
    const int itp = (items + 63) & ~63;                 // the half is uniform within a wave
    for (int it = threadIdx.x; it < 2 * itp; it += MID_THREADS) {
        const int half = it >= itp, id = it - half * itp;
        if (id >= items) continue;
        const int ch = id & 3, tl = id >> 2;
        int bb, z0, y0, x0;
        tile_coords<O, O, O>(tl, g, bb, z0, y0, x0);
        const int voff = ((cq * nt + b * nts + tl) * 4 + ch) * 4;
        if (half == 0) mid_input_half<0>(ys, vr, voff, xs, g, Wp, ch, z0, y0, x0);
        else mid_input_half<1>(ys, vr, voff, xs, g, Wp, ch, z0, y0, x0);
    }
}

// pitch of the warped-row table of wino_head_transform: >= W + D - 1, == 8 (mod 32)
__host__ __device__ inline int head_rw_pitch(int W, int D) { return ((W + D - 1 + 23) & ~31) + 8; }

// ---- cost volume + input transform of the first layer (the cost volume never reaches HBM) ----
// stage0.hip:costvol_cor_ndhwc computes cost[b,d,y,x,c] = (x >= d ? L[b,c,y,x] : 0) * bilinear(R[b,c]; x - d, y)
// (GetCostVolume, submodule.py:479-522: warp_ope "homgrp", cost_func "cor") for decnet_costvol_forward; here the same
// values -- the same fp32 operation sequence, fp contraction off -- are formed for one sample x four channels straight
// into the LDS volume of wino_mid_transform, and that kernel's phase 2 writes their B^T transforms as V (quad major).
// The bilinear coordinates depend on x - d and on y only: two small LDS tables (W + D and H entries).
// CF: the cost function (common.h:decnet_cost: cor, ssd, or sum = the "cat" volume behind conv_pre).
template <int CF>
__global__ __launch_bounds__(MID_THREADS) void wino_head_transform(
    const float *__restrict__ left, const float *__restrict__ right, float *__restrict__ V, Tiling g, int C, int nt) {
#pragma clang fp contract(off)
    extern __shared__ float ys[];                       // [D][H][4][Wp] | L [4][H][W] | R [4][H][W] | tables | RW
    constexpr int O = 4;
    const int D = g.D, H = g.H, W = g.W, Wp = W | 1, vol = D * H * 4 * Wp, plane = H * W;
    const int nq = (C + 3) >> 2, Q = pad16(C) >> 2;
    const int b = blockIdx.x / nq, cq = blockIdx.x - b * nq;
    const int nts = g.Td * g.Th * g.Tw, items = nts * 4;
    const int xs = Q * nt * 16;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, 216 * xs, 0x00020000);
    float *Ls = ys + vol, *Rs = Ls + 4 * plane;
    float *tx = Rs + 4 * plane;                          // [W + D][3]: x0 (as float), wx0, wx1 of x - d = i - (D - 1)
    float *ty = tx + 3 * (W + D);                        // [H][3]: y0, wy0, wy1
    const int nch = C - cq * 4 < 4 ? C - cq * 4 : 4;
    const size_t src = ((size_t)b * C + cq * 4) * plane; // the four channel planes are contiguous in NCHW
    // (eight positions per pass: their sixteen loads are in flight together instead of one round trip per position)
    for (int i0 = threadIdx.x; i0 < 4 * plane; i0 += 8 * MID_THREADS) {
        float lv[8], rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * MID_THREADS;
            const bool ok = i < nch * plane;
            lv[u] = ok ? left[src + i] : 0.f;
            rv[u] = ok ? right[src + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * MID_THREADS;
            if (i < 4 * plane) { Ls[i] = lv[u]; Rs[i] = rv[u]; }
        }
    }
    for (int i = threadIdx.x; i < W + D; i += MID_THREADS) {
        const int xd = i - (D - 1);                      // x - d
        const float cx = (float)xd / ((float)(W - 1.0) / 2.0f) - 1.0f;
        const float ix = ((cx + 1.0f) * (float)W - 1.0f) / 2.0f;
        const float fx = floorf(ix);
        tx[3 * i] = fx; tx[3 * i + 2] = ix - fx; tx[3 * i + 1] = 1.0f - (ix - fx);
    }
    for (int y = threadIdx.x; y < H; y += MID_THREADS) {
        const float cy = (float)y / ((float)(H - 1.0) / 2.0f) - 1.0f;
        const float iy = ((cy + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float fy = floorf(iy);
        ty[3 * y] = fy; ty[3 * y + 2] = iy - fy; ty[3 * y + 1] = 1.0f - (iy - fy);
    }
    __syncthreads();
    // The warped right value depends on (channel, y, x - d) only: W + D - 1 bilinear samples per (channel, row) instead of
    // W * D (round 6: the cost phase was 19 of the kernel's 56 us with one evaluation per cell -- timing-only builds: loads
    // 6.5, costs 19.3, transform + stores 30.5, additive at one workgroup per CU).  RW [H][4][SP], SP == 8 (mod 32): the
    // half-wave of 8 positions x 4 channels reads 32 banks.
    const int S = W + D - 1, SP = head_rw_pitch(W, D);
    float *RW = ty + 3 * H;
    for (int e = threadIdx.x; e < 4 * H * S; e += MID_THREADS) {
        const int c = e & 3, rest = e >> 2, y = rest / S, i = rest - y * S;
        const int y0 = (int)ty[3 * y], y1 = y0 + 1;
        const float wy0 = ty[3 * y + 1], wy1 = ty[3 * y + 2];
        const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        const float *R0 = Rs + c * plane + (vy0 ? y0 : 0) * W, *R1 = Rs + c * plane + (vy1 ? y1 : 0) * W;
        const int x0 = (int)tx[3 * i], x1 = x0 + 1;
        const float wx0 = tx[3 * i + 1], wx1 = tx[3 * i + 2];
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        const float a00 = R0[vx0 ? x0 : 0], a01 = R0[vx1 ? x1 : 0], a10 = R1[vx0 ? x0 : 0], a11 = R1[vx1 ? x1 : 0];
        float rr = 0.f;                                 // same tap order as grid_sample
        if (vy0 && vx0) rr += a00 * w00;
        if (vy0 && vx1) rr += a01 * w01;
        if (vy1 && vx0) rr += a10 * w10;
        if (vy1 && vx1) rr += a11 * w11;
        RW[(y * 4 + c) * SP + i] = rr;
    }
    __syncthreads();
    // a thread owns (y, x, channel) positions and walks the disparities
    for (int e = threadIdx.x; e < 4 * plane; e += MID_THREADS) {
        const int c = e & 3, yx = e >> 2, y = yx / W, x = yx - y * W;
        const float lv = Ls[c * plane + yx];
        float *cell = ys + (y * 4 + c) * Wp + x;        // + d * H * 4 * Wp
        const float *rw = RW + (y * 4 + c) * SP + x + D - 1;     // - d: the sample at x - d
#pragma unroll 4
        for (int d = 0; d < D; ++d) {
            const float l = x >= d ? lv : 0.f;          // submodule.py:506-508
            cell[d * H * 4 * Wp] = decnet_cost<CF>(l, rw[-d]);   // submodule.py:511-530
        }
    }
    __syncthreads();
    const int itp = (items + 63) & ~63;
    for (int it = threadIdx.x; it < 2 * itp; it += MID_THREADS) {
        const int half = it >= itp, id = it - half * itp;
        if (id >= items) continue;
        const int ch = id & 3, tl = id >> 2;
        int bb, z0, y0, x0;
        tile_coords<O, O, O>(tl, g, bb, z0, y0, x0);
        const int voff = ((cq * nt + b * nts + tl) * 4 + ch) * 4;
        if (half == 0) mid_input_half<0>(ys, vr, voff, xs, g, Wp, ch, z0, y0, x0);
        else mid_input_half<1>(ys, vr, voff, xs, g, Wp, ch, z0, y0, x0);
    }
}

// ------------------------------ batched GEMM  M[xi] = V[xi] * U[xi] ----------------------
// K = Ci is short (216), so an LDS-staged tile pipeline (the first version of this kernel: 192x224
// tiles, double-buffered LDS as conv3d_k3_igemm) spent a quarter of its time filling and draining
// around barriers.  Here every wave is independent: both MFMA operands go HBM/L2/L1 -> registers
// as one 16-byte load per lane, no LDS, no barrier.  That works because the order of the K axis
// inside an MFMA is free as long as both operands agree: lane (i16, kq) loads the four consecutive
// k = 16c + 4kq + {0..3} of ITS row (tile m of V, or co of U^T) and MFMA step t of chunk c
// multiplies element t of those.  With the chunk-major layouts a wave-wide operand load is 1 KiB
// contiguous.  The operands are swapped (A = U^T rows co, B = V rows m) so that a lane's four
// accumulator registers are four consecutive co of one tile m: the result leaves as 16-byte
// stores, again 1 KiB contiguous per MFMA tile.  A wave owns 48 tiles x 112 co (3 x 7 MFMA tiles,
// 84 accumulator registers) of one transform point; two waves per SIMD.  Duplicate operand reads
// between the waves of a workgroup (V twice, U^T WM times) hit in L1/L2.
//
// NFULL/TAIL > 0: Ci = 16*NFULL + 4*TAIL exactly (216 = 16*13 + 4*2): the chunk loop is unrolled
// (every s_waitcnt then counts exactly the loads that must have landed, never the stores of the
// previous point) and the last chunk takes TAIL k per lane -- no K padding.  NFULL = 0: any Ci
// (multiple of 4), runtime loop, last chunk zero-padded through the buffer bounds check.
template <int N>
__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    if constexpr (N == 4) {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        return f32x4{__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)};
    } else if constexpr (N == 2) {
        const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
        return f32x4{__int_as_float(v.x), __int_as_float(v.y), 0.f, 0.f};
    } else {
        static_assert(N == 1, "tail of 1, 2 or 4 k per lane");
        return f32x4{__int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0)), 0.f, 0.f, 0.f};
    }
}

// One wave's work: TM x 7 MFMA tiles (16*TM tiles m from m0, 112 co) of the xg points from xi0.
// (TN x J0: the co tiles J0 .. J0+TN-1 of the wave's 112-co half; 7 x 0 = all of it)
template <int NFULL, int TAIL, int TM, int TN = 7, int J0 = 0>
__device__ __forceinline__ void wino_gemm_wave(const float *__restrict__ Vb, const float *__restrict__ Ub,
                                               float *__restrict__ Mb, int nt, int Ci, int Co, int xg,
                                               int np, int xi0, int m0, int wn, int v_shared) {
    constexpr int OOB = 0x7fffffff;
    const int lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    if (m0 >= nt) return;                               // no barriers: idle waves of the M tail just leave
    const int KC = (Ci + 15) >> 4, CG = (Co + 15) >> 4;
    const int v_chunk = nt * 64, u_chunk = W_BN * 64;                 // bytes per 16-channel chunk
    // v_shared: every "point" multiplies the SAME V (decnet_tap_gemm: the taps of a dilated convolution)
    const int v_point = v_shared ? 0 : KC * v_chunk, u_point = KC * u_chunk, m_point = CG * v_chunk;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
        (void *)Vb, 0, v_shared ? KC * v_chunk : np * v_point, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)Ub, 0, np * u_point, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void *)Mb, 0, np * m_point, 0x00020000);
    int v_row[TM], u_row[TN];                           // byte offset of this lane's row inside a chunk
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + i16;
        v_row[i] = m < nt ? m * 64 : OOB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) u_row[j] = (wn * (W_BN / 2) + (J0 + j) * 16 + i16) * 64;

    f32x4 v0[TM], u0[TN], v1[TM], u1[TN];
    // (p, c) = point (relative) and chunk; beyond the last point the offsets go out of range: zeros, no traffic
    auto load = [&](f32x4(&v)[TM], f32x4(&u)[TN], int p, int c, auto nk) {
        constexpr int NK = decltype(nk)::value;                       // k per lane in this chunk
        bool ok = p < xg;
        if (NFULL == 0) ok = ok && c * 16 + kq * 4 < Ci;              // K tail: whole 16-byte groups (Ci % 4 == 0)
        const int vb = (xi0 + p) * v_point + c * v_chunk + kq * (4 * NK);
        const int ub = (xi0 + p) * u_point + c * u_chunk + kq * (4 * NK);
#pragma unroll
        for (int i = 0; i < TM; ++i) v[i] = buf_load<NK>(vr, ok && v_row[i] != OOB ? v_row[i] + vb : OOB);
#pragma unroll
        for (int j = 0; j < TN; ++j) u[j] = buf_load<NK>(ur, ok ? u_row[j] + ub : OOB);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const f32x4(&v)[TM], const f32x4(&u)[TN], auto nk) {
        constexpr int NK = decltype(nk)::value;
#pragma unroll
        for (int t = 0; t < NK; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][t], v[i][t], acc[j][i], 0, 0, 0);
    };
    auto finish = [&](int p) {                          // transform point finished: store and clear
        const int pb = (xi0 + p) * m_point + kq * 16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cg = wn * 7 + J0 + j;             // 16-co group: rows 4kq + r of MFMA tile j
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 a = acc[j][i];
                __builtin_amdgcn_raw_buffer_store_b128(
                    i32x4{__float_as_int(a[0]), __float_as_int(a[1]), __float_as_int(a[2]), __float_as_int(a[3])},
                    mr, cg < CG && v_row[i] != OOB ? v_row[i] + pb + cg * v_chunk : OOB, 0, 0);
                acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    using K4 = std::integral_constant<int, 4>;
    // two register sets: the next chunk is in flight while the current one is multiplied, and the
    // chunk sequence runs across the xg points of this workgroup without draining.
    // sched_barrier: keep the loads ABOVE the MFMA block they overlap with (the machine scheduler
    // otherwise sinks each load to just before its first use).
    load(v0, u0, 0, 0, K4{});
    if constexpr (NFULL > 0) {
        constexpr int NS = NFULL + (TAIL > 0);
        static_assert(NS % 2 == 0, "register set parity must repeat per point");
        using KT = std::integral_constant<int, (TAIL > 0 ? TAIL : 4)>;
        for (int p = 0; p < xg; ++p) {
#pragma unroll
            for (int c = 0; c < NS; c += 2) {
                if (c + 1 == NS - 1) load(v1, u1, p, c + 1, KT{}); else load(v1, u1, p, c + 1, K4{});
                __builtin_amdgcn_sched_barrier(0);
                compute(v0, u0, K4{});
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < NS) load(v0, u0, p, c + 2, K4{}); else load(v0, u0, p + 1, 0, K4{});
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 == NS - 1) compute(v1, u1, KT{}); else compute(v1, u1, K4{});
                __builtin_amdgcn_sched_barrier(0);
            }
            finish(p);
        }
    } else {
        const int nch = KC, nstep = xg * nch;
        int p = 0, c = 0;                               // step being LOADED
        auto advance = [&]() { if (++c == nch) { c = 0; ++p; } };
        advance();
        int cc = 0, pc = 0;                             // step being COMPUTED
        for (int s = 0; s < nstep; s += 2) {
            load(v1, u1, p, c, K4{}); advance();
            __builtin_amdgcn_sched_barrier(0);
            compute(v0, u0, K4{});
            __builtin_amdgcn_sched_barrier(0);
            if (++cc == nch) { finish(pc); cc = 0; ++pc; }
            if (s + 1 >= nstep) break;
            load(v0, u0, p, c, K4{}); advance();
            __builtin_amdgcn_sched_barrier(0);
            compute(v1, u1, K4{});
            __builtin_amdgcn_sched_barrier(0);
            if (++cc == nch) { finish(pc); cc = 0; ++pc; }
        }
    }
}

// Workgroup = WM x 2 waves (WM row groups of 48 tiles x the two halves of co).  Measured and not
// faster: levelling the last round of workgroups with smaller tail tasks (the dispatcher already
// fills the gaps), and one wave per SIMD with a 96 x 112 tile (35 % fewer operand loads per MFMA,
// but nothing left to hide the chunk boundaries: 0.204 vs 0.189 ms).
template <int WM, int NFULL, int TAIL>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_gemm(
    const float *__restrict__ Vb, const float *__restrict__ Ub, float *__restrict__ Mb, int nt, int Ci,
    int Co, int xg, int np, int swz) {
    // XCD-aware task order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // workgroup ids that are equal mod 8 take the M blocks of the SAME transform points: one XCD's
    // L2 then holds U of one point group at a time instead of every XCD streaming all of U.
    const int mblocks = gridDim.x, ngroups = gridDim.y;
    int pg = blockIdx.y, mb = blockIdx.x;
    if (swz & 1) {
        const int id = blockIdx.y * mblocks + blockIdx.x, per8 = 8 * mblocks;
        const int r = id / per8, q = id - r * per8;
        if (8 * (r + 1) <= ngroups) { pg = 8 * r + (q & 7); mb = q >> 3; }     // tail (< 8 groups) unswizzled
    }
    const int wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    wino_gemm_wave<NFULL, TAIL, 3>(Vb, Ub, Mb, nt, Ci, Co, xg, np, pg * xg, (mb * WM + wm) * 48, wn, swz & 2);
}

// The same GEMM with the machine filled ONCE and the work shared out by hand (default for Ci = 216).
// A point's GEMM is cut into wave tasks of 48 tiles x 112 co; with 216 points x 15 x 2 = 6480 equal tasks on
// the 2048 wave slots (2 per SIMD) the launch above runs 3.16 "rounds" of tasks as 4: the last quarter of the
// time 5 % of the work.  Here 512 workgroups of 4 waves stay for the whole launch; the waves of XCD x (workgroup
// id mod 8, the dispatcher's round robin) own the points x, x + 8, ...: each takes floor(tasks / waves) whole
// tasks, and what is left over is cut in three along M (16 tiles x 112 co, TM = 1) so that the last step takes
// a third of a task's MFMAs instead of a whole one -- 3.5 task times instead of 4.  Static assignment: no atomics,
// nothing to reset between launches, re-entrant; if fewer workgroups are resident (another stream shares the
// GPU) the rest simply run afterwards.
template <int NFULL, int TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_gemm_persist(
    const float *__restrict__ Vb, const float *__restrict__ Ub, float *__restrict__ Mb, int nt, int Ci,
    int Co, int np) {
    const int xcd = blockIdx.x & 7, wx = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int nwx = (gridDim.x >> 3) * 4;                           // waves per XCD
    const int MB = (nt + 47) / 48, npx = (np - xcd + 7) >> 3;       // this XCD's points: xcd, xcd + 8, ...
    const int tasks = npx * MB * 2, full = tasks / nwx, rem = tasks - full * nwx;
    // task id -> (point, M block, co half); consecutive ids = the two co halves of one V block, then the next
    // M block of the same point: the waves of an XCD share few points' U^T at any time
    auto big = [&](int t) {
        const int pl = t / (MB * 2), r = t - pl * (MB * 2);
        wino_gemm_wave<NFULL, TAIL, 3>(Vb, Ub, Mb, nt, Ci, Co, 1, np, xcd + 8 * pl, (r >> 1) * 48, r & 1, 0);
    };
    for (int r = 0; r < full; ++r) big(r * nwx + wx);
    if (6 * rem <= nwx + nwx / 4) {                                 // sixths: thirds along M x (4 | 3 co tiles)
        for (int u = wx; u < 6 * rem; u += nwx) {
            const int t = full * nwx + u / 6, part = (u % 6) >> 1, half = u & 1;
            const int pl = t / (MB * 2), r = t - pl * (MB * 2);
            if (half == 0)
                wino_gemm_wave<NFULL, TAIL, 1, 4, 0>(Vb, Ub, Mb, nt, Ci, Co, 1, np, xcd + 8 * pl, (r >> 1) * 48 + part * 16, r & 1, 0);
            else
                wino_gemm_wave<NFULL, TAIL, 1, 3, 4>(Vb, Ub, Mb, nt, Ci, Co, 1, np, xcd + 8 * pl, (r >> 1) * 48 + part * 16, r & 1, 0);
        }
    } else if (3 * rem <= 2 * nwx) {                                // thirds of the left-over tasks
        for (int u = wx; u < 3 * rem; u += nwx) {
            const int t = full * nwx + u / 3, part = u - (u / 3) * 3;
            const int pl = t / (MB * 2), r = t - pl * (MB * 2);
            wino_gemm_wave<NFULL, TAIL, 1>(Vb, Ub, Mb, nt, Ci, Co, 1, np, xcd + 8 * pl, (r >> 1) * 48 + part * 16, r & 1, 0);
        }
    } else if (wx < rem) {
        big(full * nwx + wx);
    }
}

// ------------------------------ the same GEMM on the bf16 matrix cores, fp32 accurate ------------
// DECNET_WINO_GEMM=bf16x3 (experimental).  tools/ubench/mfma_rate.hip: v_mfma_f32_16x16x4_f32 issues
// every ~35 cycles, v_mfma_f32_16x16x32_bf16 every ~18.  Each fp32 operand is split into three bf16
// terms x = hi + mid + lo (truncations with exact residuals, 24 mantissa bits together) and the six
// products above 2^-24 are accumulated in fp32:
//     u*v ~ uh*vh + uh*vm + um*vh + uh*vl + um*vm + ul*vh        (dropped: um*vl, ul*vm, ul*vl)
// i.e. 6 x 18 MFMA cycles per 32 k instead of 8 x 35.  U^T is split once at weight-packing time
// ([point][pair of 16-chunks][term][224 co][4 kq][8 bf16]: a wave's operand load stays 1 KiB
// contiguous), V is split in registers.  Two waves per SIMD: the U^T terms stream through a ring of
// four operand tiles (three tiles = ~1000 MFMA cycles ahead), the fp32 V of the next pair is in
// flight during the whole current pair.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// {bf16(b), bf16(a)} as one register, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ int pack_bf16(float a, float b) {
    return __builtin_bit_cast(int, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

__device__ __forceinline__ void split3(const f32x4 &a, const f32x4 &b, i32x4 &hi, i32x4 &mid, i32x4 &lo) {
    // 8 values (a: chunk 2p, b: chunk 2p+1) -> three packed 8 x bf16 operands; round-to-nearest terms (see
    // csrc/conv2d_mfma.hip).  Two values per conversion: the packed pair IS the operand register, and the float value
    // of a term is a shift / a mask of it
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = x[2 * e], x1 = x[2 * e + 1];
        const int h = pack_bf16(x0, x1);
        const float r0 = x0 - __int_as_float(h << 16), r1 = x1 - __int_as_float(h & 0xffff0000);
        const int m = pack_bf16(r0, r1);
        const float s0 = r0 - __int_as_float(m << 16), s1 = r1 - __int_as_float(m & 0xffff0000);
        hi[e] = h;
        mid[e] = m;
        lo[e] = pack_bf16(s0, s1);
    }
}

// fp32 U^T [point][kc][224][16] -> bf16 terms [point][pair][term][224][kq][8]
__global__ void wino_split_weights(const float *__restrict__ U, int *__restrict__ Ub, int np, int KC, int Ci) {
    const int NP2 = (KC + 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (point, pair, co, kq)
    if (idx >= np * NP2 * W_BN * 4) return;
    const int kq = idx & 3, co = (idx >> 2) % W_BN, pp = (idx >> 2) / W_BN;
    const int pair = pp % NP2, pt = pp / NP2;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    const float *u0 = U + (((size_t)pt * KC + 2 * pair) * W_BN + co) * 16 + 4 * kq;
    // k slots >= Ci of the last chunk are never written in the fp32 layout: zeros here
    if (32 * pair + 4 * kq < Ci) a = *reinterpret_cast<const f32x4 *>(u0);
    if (2 * pair + 1 < KC && 32 * pair + 16 + 4 * kq < Ci) b = *reinterpret_cast<const f32x4 *>(u0 + (size_t)W_BN * 16);
    i32x4 t[3];
    split3(a, b, t[0], t[1], t[2]);
#pragma unroll
    for (int term = 0; term < 3; ++term)
        *reinterpret_cast<i32x4 *>(Ub + ((((size_t)pt * NP2 + pair) * 3 + term) * W_BN + co) * 16 + 4 * kq) = t[term];
}

template <int WM, int NPAIR, int TM = 3>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(TM > 3 ? 1 : TM == 2 ? 3 : 2, TM > 3 ? 1 : TM == 2 ? 3 : 2))) void wino_gemm_bf16x3(
    const float *__restrict__ Vb, const int *__restrict__ Ubb, float *__restrict__ Mb, int nt, int Ci,
    int Co, int np, int swz, int v_ms, int v_kqs, int m_ms, int m_kqs, int v_shared) {
    // v_shared: every point multiplies the SAME V (decnet_tap_gemm).  v_ms / v_kqs, m_ms / m_kqs: bytes between consecutive tiles and between the four 4-channel groups of a
    // 16-channel chunk in V and in M: (64, 16) = the chunk-major layout [chunk][tile][16] of the head of this
    // file, (16, 16 nt) = the quad-major layout [chunk][4 quads][tile][4] of wino_mid_transform
    constexpr int TN = 7, OOB = 0x7fffffff, RING = 4;   // ring of 4 operand tiles: tile s + 3 (AHEAD) in flight while s is multiplied
    const int mblocks = gridDim.x, ngroups = gridDim.y;
    int pt = blockIdx.y, mb = blockIdx.x;
    if (swz) {
        const int id = blockIdx.y * mblocks + blockIdx.x, per8 = 8 * mblocks;
        const int r = id / per8, q = id - r * per8;
        if (8 * (r + 1) <= ngroups) { pt = 8 * r + (q & 7); mb = q >> 3; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, i16 = lane & 15, kq = lane >> 4;
    const int m0 = (mb * WM + wm) * (TM * 16);
    if (m0 >= nt) return;
    const int KC = (Ci + 15) >> 4, CG = (Co + 15) >> 4;
    const int v_chunk = nt * 64, v_point = v_shared ? 0 : KC * v_chunk, m_point = CG * v_chunk;
    const int u_term = W_BN * 64, u_pair = 3 * u_term, u_point = NPAIR * u_pair;
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (v_shared ? 1 : np) * KC * v_chunk, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)Ubb, 0, np * u_point, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void *)Mb, 0, np * m_point, 0x00020000);
    int v_row[TM], m_row[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + i16;
        v_row[i] = m < nt ? m * v_ms : OOB;
        m_row[i] = m < nt ? m * m_ms : OOB;
    }
    const int u_lane = (wn * (W_BN / 2) + i16) * 64 + kq * 16 + pt * u_point;

    // operand tiles are consumed in the order s = pair * 7 + j; the three bf16 terms of tile s + AHEAD
    // are in flight while tile s is multiplied (ring of 4 x 12 registers), the fp32 V of the next pair
    // during the whole current pair
    f32x4 vf[TM][2];
    i32x4 ub[RING][3], vs[TM][3];
    auto load_u = [&](int slot, int s) {
        const int p = s / TN, j = s - p * TN;
#pragma unroll
        for (int term = 0; term < 3; ++term)
            ub[slot][term] = __builtin_amdgcn_raw_buffer_load_b128(
                ur, p < NPAIR ? u_lane + p * u_pair + term * u_term + j * 1024 : OOB, 0, 0);
    };
    auto load_v = [&](int p) {
        const int vb = pt * v_point + kq * v_kqs;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 2 * p + h;
                const bool ok = p < NPAIR && v_row[i] != OOB && c * 16 + kq * 4 < Ci;   // whole 16-byte groups
                const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(vr, ok ? v_row[i] + vb + c * v_chunk : OOB, 0, 0);
                vf[i][h] = f32x4{__int_as_float(t.x), __int_as_float(t.y), __int_as_float(t.z), __int_as_float(t.w)};
            }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (u term, v term): the six products above 2^-24, small ones first
    constexpr int UT[6] = {2, 1, 0, 1, 0, 0}, VT[6] = {0, 1, 2, 0, 1, 0};

    load_v(0);
    load_u(0, 0);
    load_u(1, 1);
    // operand tiles two at a time (six independent accumulators between two MFMAs on the same one);
    // the next two tiles load into the other half of the ring meanwhile
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) split3(vf[i][0], vf[i][1], vs[i][0], vs[i][1], vs[i][2]);
        load_v(p + 1);                                  // vf is free again
#pragma unroll
        for (int g = 0; g < (TN + 1) / 2; ++g) {        // tiles 2g, 2g+1 (the last group has one tile + a dummy)
            const int s = p * 8 + 2 * g;                // ring position: 8 slots per pair (7 tiles + 1 dummy)
            {
                const int sn = s + 2, pn = sn / 8, jn = sn % 8;
                if (jn < TN) load_u(sn % RING, pn * TN + jn);
                if (jn + 1 < TN) load_u((sn + 1) % RING, pn * TN + jn + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int j = 2 * g + jj;
                        if (j < TN)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, ub[(s + jj) % RING][UT[t]]),
                                __builtin_bit_cast(bf16x8, vs[i][VT[t]]), acc[j][i], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int pb = pt * m_point + kq * m_kqs;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cg = wn * TN + j;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f32x4 a = acc[j][i];
            __builtin_amdgcn_raw_buffer_store_b128(
                i32x4{__float_as_int(a[0]), __float_as_int(a[1]), __float_as_int(a[2]), __float_as_int(a[3])},
                mr, cg < CG && m_row[i] != OOB ? m_row[i] + pb + cg * v_chunk : OOB, 0, 0);
        }
    }
}

// 1: bf16x3 GEMM (needs the split copy of U^T behind the fp32 one, see decnet_conv3d_wino_pack_weight).  The default
// since round 3 (the same accuracy, 0.113 instead of 0.131 ms per layer at config 2); DECNET_WINO_GEMM=fp32 (or
// static / any other value) selects the fp32 MFMA kernels.  Read once per process: it fixes the packed weights' size.
static int gemm_bf16x3() {
    static const int k = [] { const char *e = getenv("DECNET_WINO_GEMM"); return !e || !e[0] || !strcmp(e, "bf16x3") ? 1 : 0; }();
    return k;
}

template <int WM>
int launch_gemm(const float *V, const float *U, float *M, int nt, int Ci, int Co, int np,
                hipStream_t stream, int v_shared = 0) {
    const int xg = 1;                                   // points per workgroup (measured: 1 is best)
    const int swz = 1 | (v_shared ? 2 : 0);             // XCD-aware block order on
    const dim3 grid(ceil_div(nt, WM * 48), np / xg), block(WM * 128);
    if (Ci == 216)
        hipLaunchKernelGGL((wino_gemm<WM, 13, 2>), grid, block, 0, stream, V, U, M, nt, Ci, Co, xg, np, swz);
    else
        hipLaunchKernelGGL((wino_gemm<WM, 0, 0>), grid, block, 0, stream, V, U, M, nt, Ci, Co, xg, np, swz);
    return decnet_launch_status();
}

// v_quad / m_quad: V / M in the quad-major layout (wino_gemm_bf16x3 only)
int gemm_dispatch(const float *V, const float *U, float *M, int nt, int Ci, int Co, int np,
                  hipStream_t s, int v_quad = 0, int m_quad = 0) {
    if (gemm_bf16x3() && Ci == 216) {
        constexpr int swz = 1;                          // XCD-aware block order
        const int *Ub = reinterpret_cast<const int *>(U + (size_t)np * pad16(Ci) * W_BN);   // split copy behind U^T
        hipLaunchKernelGGL((wino_gemm_bf16x3<2, 7>), dim3(ceil_div(nt, 96), np), dim3(256), 0, s, V, Ub, M, nt, Ci,
                           Co, np, swz, v_quad ? 16 : 64, v_quad ? 16 * nt : 16, m_quad ? 16 : 64, m_quad ? 16 * nt : 16, 0);
        return decnet_launch_status();
    }
    if (v_quad || m_quad) return DECNET_ERR_UNSUPPORTED;
    if (Ci == 216 && np >= 8) {
        hipLaunchKernelGGL((wino_gemm_persist<13, 2>), dim3(512), dim3(256), 0, s, V, U, M, nt, Ci, Co, np);
        return decnet_launch_status();
    }
    return launch_gemm<2>(V, U, M, nt, Ci, Co, np, s);
}

// tiles per chunk: V + M of one chunk <= DECNET_WINO_CHUNK_MB (1 GiB); equal chunks
int chunk_tiles(int T, int C, int np) {
    static const double cap_mb = [] {
        const char *e = getenv("DECNET_WINO_CHUNK_MB");       // experiments: V+M bytes per chunk
        return e ? atof(e) : 1024.0;
    }();
    long cap = (long)(cap_mb * 1024 * 1024 / (2.0 * np * 4 * pad16(C)));
    if (cap < 192) cap = 192;
    const long nchunks = (T + cap - 1) / cap;          // equal chunks
    return (int)((T + nchunks - 1) / nchunks);
}

template <int OD, int OH, int OW>
int conv_variant(const float *x, const float *u, const float *scale, const float *shift,
                 const float *residual, float *y, float *workspace, int B, int D, int H, int W, int Ci,
                 int Co, int relu, hipStream_t s) {
    constexpr int TD = OD + 2, TH = OH + 2, TW = OW + 2, NP = TD * TH * TW;
    constexpr int KS = 1;      // threads per (tile, channel) in the transforms (2: measured slower, round 3)
    Tiling g{D, H, W, ceil_div(D, OD), ceil_div(H, OH), ceil_div(W, OW)};
    const double Td = (double)B * g.Td * g.Th * g.Tw;
    if (Td >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    const int T = (int)Td, cmax = Ci > Co ? Ci : Co;
    const int ct = chunk_tiles(T, cmax, NP);
    if ((double)ct * pad16(cmax) * 4 * NP >= 2147483647.0) return DECNET_ERR_UNSUPPORTED;   // 32-bit offsets
    float *V = workspace, *M = workspace + (size_t)NP * ct * pad16(Ci);
    for (int t_lo = 0; t_lo < T; t_lo += ct) {
        const int nt = T - t_lo < ct ? T - t_lo : ct;
        const int x_bytes = (int)((size_t)B * D * H * W * Ci * 4);
        const int ith = Ci >= 256 ? 256 : (Ci + 63) / 64 * 64;
        hipLaunchKernelGGL((wino_input_transform<TD, TH, TW, KS>),
                           dim3((unsigned)(8 * ((nt + 7) / 8)), (unsigned)(KS * ceil_div(Ci, ith))), dim3(ith), 0, s,
                           x, V, g, Ci, t_lo, nt, x_bytes);
        int rc = decnet_launch_status();
        if (rc) return rc;
        if ((rc = gemm_dispatch(V, u, M, nt, Ci, Co, NP, s))) return rc;
        const size_t n = (size_t)nt * pad16(Co);
        hipLaunchKernelGGL((wino_output_transform<TD, TH, TW, KS>), dim3((unsigned)((n + 255) / 256), KS),
                           dim3(256), 0, s, M, scale, shift, residual, y, g, Co, relu, t_lo, nt,
                           (int)((size_t)B * D * H * W * Co * 4));
        if ((rc = decnet_launch_status())) return rc;
    }
    return DECNET_OK;
}

// ---- a stack of C -> C layers with the activations between them kept on chip (wino_mid_transform) ----
size_t stack_lds_bytes(int D, int H, int W) { return (size_t)D * H * 4 * (W | 1) * 4; }
size_t stack_residual_floats(int B, int D, int H, int W, int C) { return (size_t)B * ((C + 3) / 4) * D * H * 4 * (W | 1); }

// 1 when the fused stack can run this shape: F(4,3)^3, the bf16x3 GEMM (C = 216), every tile in one chunk,
// one sample x four channels in LDS
bool stack_ok(int B, int D, int H, int W, int C, int variant) {
    static const int off = [] { const char *e = getenv("DECNET_WINO_STACK"); return e && !strcmp(e, "0") ? 1 : 0; }();
    if (off || variant != 2 || !gemm_bf16x3() || C != 216) return false;
    if (stack_lds_bytes(D, H, W) > 160 * 1024) return false;
    const double T = (double)B * ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 4);
    if (T >= 2147483648.0 || chunk_tiles((int)T, C, 216) < (int)T) return false;
    return (double)T * pad16(C) * 4 * 216 < 2147483647.0 && (double)B * D * H * W * C * 4 < 2147483647.0;
}

size_t head_lds_bytes(int D, int H, int W) {
    return stack_lds_bytes(D, H, W) + ((size_t)8 * H * W + 3 * (W + D) + 3 * H + 4 * H * head_rw_pitch(W, D)) * 4;
}

// (Round 5: the stack as two half batches on two streams, shifted by half a layer so that one half's transform --
// then ONE round of <= 256 workgroups -- runs beside the other half's GEMM, was built and measured: 1.262 ms per seven
// layers as one batch, 1.268 free-running, 1.432 as a strict pipeline.  The two kernels do not share a CU: the GEMM's two
// waves per SIMD hold the whole register file, so the streams interleave at CU granularity instead of overlapping.
// Commit 82b86c1 has the code; profiles/r05c_wino_split_two_halves.txt the numbers.)
// x != nullptr: the stack's input is a channels-last volume; x == nullptr: it is the cost volume of (left, right),
// formed on chip by wino_head_transform
int conv_stack(const float *x, const float *left, const float *right, const float *const *u,
               const float *const *scale, const float *const *shift,
               int n_layers, int res_src, int res_dst, float *y, float *workspace, float *R, int B, int D, int H,
               int W, int C, hipStream_t s, int cost_func = DECNET_COST_COR) {
    constexpr int NP = 216;
    Tiling g{D, H, W, ceil_div(D, 4), ceil_div(H, 4), ceil_div(W, 4)};
    const int nt = B * g.Td * g.Th * g.Tw;
    float *V = workspace, *M = workspace + (size_t)NP * nt * pad16(C);
    const int bytes = (int)((size_t)B * D * H * W * C * 4);
    const size_t lds = stack_lds_bytes(D, H, W);
    // more than 64 KiB of dynamic LDS needs the attribute; it is per device, so it is set per call (cheap) rather than cached
    if (hipFuncSetAttribute((const void *)wino_mid_transform, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return DECNET_ERR_UNSUPPORTED;
    }
    if (x) {
        const int ith = C >= 256 ? 256 : (C + 63) / 64 * 64;
        hipLaunchKernelGGL((wino_input_transform<6, 6, 6, 1>), dim3((unsigned)(8 * ((nt + 7) / 8)), (unsigned)ceil_div(C, ith)),
                           dim3(ith), 0, s, x, V, g, C, 0, nt, bytes);
    } else {
        const size_t hl = head_lds_bytes(D, H, W);
        const void *fn = cost_func == DECNET_COST_COR   ? (const void *)wino_head_transform<DECNET_COST_COR>
                         : cost_func == DECNET_COST_SSD ? (const void *)wino_head_transform<DECNET_COST_SSD>
                                                        : (const void *)wino_head_transform<DECNET_COST_SUM>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl) != hipSuccess) {
            (void)hipGetLastError();
            return DECNET_ERR_UNSUPPORTED;
        }
        const dim3 grid((unsigned)(B * ((C + 3) / 4)));
        if (cost_func == DECNET_COST_COR)
            hipLaunchKernelGGL(wino_head_transform<DECNET_COST_COR>, grid, dim3(MID_THREADS), hl, s, left, right, V, g, C, nt);
        else if (cost_func == DECNET_COST_SSD)
            hipLaunchKernelGGL(wino_head_transform<DECNET_COST_SSD>, grid, dim3(MID_THREADS), hl, s, left, right, V, g, C, nt);
        else
            hipLaunchKernelGGL(wino_head_transform<DECNET_COST_SUM>, grid, dim3(MID_THREADS), hl, s, left, right, V, g, C, nt);
    }
    int rc = decnet_launch_status();
    if (rc) return rc;
    for (int i = 0; i < n_layers; ++i) {
        const bool last = i == n_layers - 1;
        // V of layer 0 comes from wino_input_transform (chunk major) unless the head kernel forms it, M of the last layer
        // goes to wino_output_transform (chunk major); everything between is quad major
        if ((rc = gemm_dispatch(V, u[i], M, nt, C, C, NP, s, i > 0 || !x, !last))) return rc;
        if (last) {
            const size_t n = (size_t)nt * pad16(C);
            hipLaunchKernelGGL((wino_output_transform<6, 6, 6, 1>), dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, s,
                               M, scale[i], shift[i], (const float *)nullptr, y, g, C, 1, 0, nt, bytes);
        } else {
            hipLaunchKernelGGL(wino_mid_transform, dim3((unsigned)(B * ((C + 3) / 4))), dim3(MID_THREADS), lds, s, M, V,
                               scale[i], shift[i], i == res_dst ? R : (const float *)nullptr,
                               i == res_src ? R : (float *)nullptr, g, C, nt, 1);
        }
        if ((rc = decnet_launch_status())) return rc;
    }
    return DECNET_OK;
}

int variant_points(int variant) { return variant == 0 ? 64 : variant == 1 ? 144 : variant == 2 ? 216 : -1; }

}  // namespace

extern "C" {

/* variant: 0 = F(2,3)^3 (64 transform points), 1 = F(2,3) on D x F(4,3) on H, W (144 points),
 * 2 = F(4,3)^3 (216 points) */
size_t decnet_conv3d_wino_weight_floats(int Ci, int variant) {
    const int np = variant_points(variant);
    if (np < 0 || Ci < 1) return 0;
    const size_t f32 = (size_t)np * pad16(Ci) * W_BN;
    // + the bf16 hi/mid/lo split of U^T for the bf16x3 GEMM: 3 terms x 2 bytes per (padded to pairs of chunks) k
    const size_t split = gemm_bf16x3() ? (size_t)np * ((pad16(Ci) / 16 + 1) / 2) * 3 * W_BN * 16 : 0;
    return f32 + split;
}

int decnet_conv3d_wino_pack_weight(const float *w, float *u, int Co, int Ci, int variant, void *stream) {
    if (!w || !u) return DECNET_ERR_NULL_POINTER;
    if (Co < 1 || Ci < 1 || variant_points(variant) < 0) return DECNET_ERR_BAD_SHAPE;
    if (Co > W_BN) return DECNET_ERR_UNSUPPORTED;
    const int n = Ci * W_BN;
    if (variant == 0)
        hipLaunchKernelGGL((wino_weight_transform<4, 4, 4>), dim3(ceil_div(n, 128)), dim3(128), 0,
                           (hipStream_t)stream, w, u, Co, Ci);
    else if (variant == 1)
        hipLaunchKernelGGL((wino_weight_transform<4, 6, 6>), dim3(ceil_div(n, 128)), dim3(128), 0,
                           (hipStream_t)stream, w, u, Co, Ci);
    else
        hipLaunchKernelGGL((wino_weight_transform<6, 6, 6>), dim3(ceil_div(n, 128)), dim3(128), 0,
                           (hipStream_t)stream, w, u, Co, Ci);
    int rc = decnet_launch_status();
    if (rc || !gemm_bf16x3()) return rc;
    const int np = variant_points(variant), KC = pad16(Ci) / 16;
    const int nthr = np * ((KC + 1) / 2) * W_BN * 4;
    hipLaunchKernelGGL(wino_split_weights, dim3(ceil_div(nthr, 256)), dim3(256), 0, (hipStream_t)stream, u,
                       reinterpret_cast<int *>(u + (size_t)np * pad16(Ci) * W_BN), np, KC, Ci);
    return decnet_launch_status();
}

/* The batched GEMM stage alone (measurement / composition): M[xi] = V[xi] * U[xi], xi < points;
 * V [points][ceil(Ci/16)][nt][16], u from decnet_conv3d_wino_pack_weight,
 * M [points][ceil(Co/16)][nt][16]. */
int decnet_conv3d_wino_gemm(const float *V, const float *u, float *M, int nt, int Ci, int Co,
                            int variant, void *stream) {
    const int np = variant_points(variant);
    if (!V || !u || !M) return DECNET_ERR_NULL_POINTER;
    if (nt < 1 || Ci < 1 || Co < 1 || np < 0) return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0 || Co > W_BN || (double)nt * pad16(Ci > Co ? Ci : Co) * 4 * np >= 2147483647.0)
        return DECNET_ERR_UNSUPPORTED;
    return gemm_dispatch(V, u, M, nt, Ci, Co, np, (hipStream_t)stream);
}

/* M[t] = V * U[t] for t < ntaps: ONE V [ceil(Ci/16)][P][16] against ntaps weight matrices
 * u [ntaps][ceil(Ci/16)][224][16] -> M [ntaps][ceil(Co/16)][P][16] (decnet_tapconv_*: dilated convolutions as a
 * per-tap product + gather).  split: the bf16-term copy behind u has been written (decnet_tapconv_split_weight). */
int decnet_tap_gemm(const float *V, const float *u, float *M, int P, int Ci, int Co, int ntaps, int split, void *stream) {
    if (!V || !u || !M) return DECNET_ERR_NULL_POINTER;
    if (P < 1 || Ci < 1 || Co < 1 || ntaps < 1) return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0 || Co > W_BN || (double)P * pad16(Ci > Co ? Ci : Co) * 4 * ntaps >= 2147483647.0 ||
        (double)ntaps * pad16(Ci) * W_BN * 4 >= 2147483647.0)
        return DECNET_ERR_UNSUPPORTED;
    // split != 0 is the caller's statement that decnet_tapconv_split_weight ran after the last pack: without it the
    // region behind the fp32 matrices is unwritten, so the fp32 kernel is the only correct choice
    if (split && gemm_bf16x3() && Ci == 216) {         // the split copy behind u: decnet_tapconv_split_weight
        const int *Ub = reinterpret_cast<const int *>(u + (size_t)ntaps * pad16(Ci) * W_BN);
        hipLaunchKernelGGL((wino_gemm_bf16x3<2, 7>), dim3(ceil_div(P, 96), ntaps), dim3(256), 0, (hipStream_t)stream, V, Ub,
                           M, P, Ci, Co, ntaps, 1, 64, 16, 64, 16, 1);
        return decnet_launch_status();
    }
    return launch_gemm<2>(V, u, M, P, Ci, Co, ntaps, (hipStream_t)stream, 1);
}

/* After the last decnet_tapconv_pack_weight: the bf16-term copy of all ntaps weight matrices behind the fp32 ones
 * (u holds decnet_tapconv_weight_floats(Ci, ntaps) floats); decnet_tap_gemm reads it when Ci = 216. */
int decnet_tapconv_split_weight(float *u, int Ci, int ntaps, void *stream) {
    if (!u) return DECNET_ERR_NULL_POINTER;
    if (Ci < 1 || ntaps < 1) return DECNET_ERR_BAD_SHAPE;
    const int KC = (Ci + 15) >> 4;
    const long nthr = (long)ntaps * ((KC + 1) / 2) * W_BN * 4;
    if (nthr >= 2147483647L) return DECNET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wino_split_weights, dim3(ceil_div((int)nthr, 256)), dim3(256), 0, (hipStream_t)stream, u,
                       reinterpret_cast<int *>(u + (size_t)ntaps * pad16(Ci) * W_BN), ntaps, KC, Ci);
    return decnet_launch_status();
}

size_t decnet_conv3d_wino_workspace_floats(int B, int D, int H, int W, int Ci, int Co, int variant) {
    const int np = variant_points(variant);
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1 || np < 0) return 0;
    const int oh = variant == 0 ? 2 : 4, od = variant == 2 ? 4 : 2;
    const double T = (double)B * ((D + od - 1) / od) * ((H + oh - 1) / oh) * ((W + oh - 1) / oh);
    if (T >= 2147483648.0) return 0;
    const int nt = chunk_tiles((int)T, Ci > Co ? Ci : Co, np);
    return (size_t)np * nt * ((size_t)pad16(Ci) + pad16(Co));
}

/* n_layers >= 2 consecutive Conv3d(k3,s1,p1) + BN + ReLU units C -> C (u[i] / scale[i] / shift[i] as for
 * decnet_conv3d_wino_bn_act) with the activations BETWEEN the layers kept on chip: the output transform of layer
 * i and the input transform of layer i + 1 are one kernel, so x is read once, y written once, and per layer
 * boundary only the transformed tiles touch HBM.  res_src / res_dst (or -1, -1): the output of layer res_src is
 * added to the output of layer res_dst (after its ReLU), 0 <= res_src < res_dst < n_layers - 1
 * (CostRegNetNoDown.forward submodule.py:653-658).  workspace: decnet_conv3d_wino_stack_workspace_floats floats.
 * Returns DECNET_ERR_UNSUPPORTED (nothing launched) for shapes the fused kernels do not cover -- callers then run the
 * layers one by one (decnet_stage0_forward does). */
size_t decnet_conv3d_wino_stack_workspace_floats(int B, int D, int H, int W, int C, int variant) {
    if (B < 1 || D < 1 || H < 1 || W < 1 || C < 1 || !stack_ok(B, D, H, W, C, variant)) return 0;
    const size_t w = decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, variant);
    return w ? ((w + 63) & ~(size_t)63) + stack_residual_floats(B, D, H, W, C) : 0;
}

int decnet_conv3d_wino_stack_bn_act(const float *x, const float *const *u, const float *const *scale,
                                    const float *const *shift, int n_layers, int res_src, int res_dst, float *y,
                                    float *workspace, int B, int D, int H, int W, int C, int variant, void *stream) {
    if (!x || !u || !scale || !shift || !y || !workspace) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || C < 1 || n_layers < 1) return DECNET_ERR_BAD_SHAPE;
    for (int i = 0; i < n_layers; ++i)
        if (!u[i] || !scale[i] || !shift[i]) return DECNET_ERR_NULL_POINTER;
    if ((res_src < 0) != (res_dst < 0)) return DECNET_ERR_BAD_SHAPE;
    if (res_src >= 0 && !(res_src < res_dst && res_dst < n_layers - 1)) return DECNET_ERR_UNSUPPORTED;
    if (n_layers < 2 || !stack_ok(B, D, H, W, C, variant)) return DECNET_ERR_UNSUPPORTED;
    const size_t w = (decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, variant) + 63) & ~(size_t)63;
    return conv_stack(x, nullptr, nullptr, u, scale, shift, n_layers, res_src, res_dst, y, workspace, workspace + w, B, D, H,
                      W, C, (hipStream_t)stream);
}

/* The same stack fed by the stage-0 cost volume of (left, right) [B,C,H,W] (decnet_costvol_forward's values, D
 * disparities) without writing that volume: its input transform is formed on chip.  Same workspace; additionally needs
 * H, W >= 2 and the two feature planes of four channels next to the volume in LDS (DECNET_ERR_UNSUPPORTED otherwise). */
int decnet_costvol_wino_stack_bn_act(const float *left, const float *right, const float *const *u,
                                     const float *const *scale, const float *const *shift, int n_layers, int res_src,
                                     int res_dst, float *y, float *workspace, int B, int C, int H, int W, int D,
                                     int variant, void *stream) {
    return decnet_costvol_wino_stack_bn_act_cf(left, right, u, scale, shift, n_layers, res_src, res_dst, y, workspace, B, C, H,
                                               W, D, variant, DECNET_COST_COR, stream);
}

int decnet_costvol_wino_stack_bn_act_cf(const float *left, const float *right, const float *const *u,
                                        const float *const *scale, const float *const *shift, int n_layers, int res_src,
                                        int res_dst, float *y, float *workspace, int B, int C, int H, int W, int D,
                                        int variant, int cost_func, void *stream) {
    if (!left || !right || !u || !scale || !shift || !y || !workspace) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || C < 1 || n_layers < 1) return DECNET_ERR_BAD_SHAPE;
    if (cost_func != DECNET_COST_COR && cost_func != DECNET_COST_SSD && cost_func != DECNET_COST_SUM)
        return DECNET_ERR_BAD_SHAPE;
    if (H < 2 || W < 2) return DECNET_ERR_UNSUPPORTED;   // the stretched warp of one row / column: per-layer path
    for (int i = 0; i < n_layers; ++i)
        if (!u[i] || !scale[i] || !shift[i]) return DECNET_ERR_NULL_POINTER;
    if ((res_src < 0) != (res_dst < 0)) return DECNET_ERR_BAD_SHAPE;
    if (res_src >= 0 && !(res_src < res_dst && res_dst < n_layers - 1)) return DECNET_ERR_UNSUPPORTED;
    static const int off = [] { const char *e = getenv("DECNET_WINO_HEAD"); return e && !strcmp(e, "0") ? 1 : 0; }();
    if (off || n_layers < 2 || !stack_ok(B, D, H, W, C, variant) || head_lds_bytes(D, H, W) > 160 * 1024)
        return DECNET_ERR_UNSUPPORTED;
    const size_t w = (decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, variant) + 63) & ~(size_t)63;
    return conv_stack(nullptr, left, right, u, scale, shift, n_layers, res_src, res_dst, y, workspace, workspace + w, B, D, H,
                      W, C, (hipStream_t)stream, cost_func);
}

int decnet_conv3d_wino_bn_act(const float *x, const float *u, const float *scale, const float *shift,
                              const float *residual, float *y, float *workspace, int B, int D, int H,
                              int W, int Ci, int Co, int relu, int variant, void *stream) {
    if (!x || !u || !scale || !shift || !y || !workspace) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || D < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1 || variant_points(variant) < 0)
        return DECNET_ERR_BAD_SHAPE;
    if (Ci % 4 != 0 || Co > W_BN) return DECNET_ERR_UNSUPPORTED;
    if ((double)B * D * H * W * (Ci > Co ? Ci : Co) * 4 >= 2147483647.0) return DECNET_ERR_UNSUPPORTED;  // 32-bit offsets
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0)
        return conv_variant<2, 2, 2>(x, u, scale, shift, residual, y, workspace, B, D, H, W, Ci, Co, relu, s);
    if (variant == 1)
        return conv_variant<2, 4, 4>(x, u, scale, shift, residual, y, workspace, B, D, H, W, Ci, Co, relu, s);
    return conv_variant<4, 4, 4>(x, u, scale, shift, residual, y, workspace, B, D, H, W, Ci, Co, relu, s);
}

}  // extern "C"
