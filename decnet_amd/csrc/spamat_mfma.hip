// decnet_amd/csrc/spamat_mfma.hip -- SpaMat / SpaVar forward, banded cost tiles on the
// matrix cores (gfx950).  Replaces get_max_cost + sparse_matching_forward
// (SM_kernel.cu:22-125) and get_max_cost + sparse_var_forward (SV_kernel.cu:22-124), and
// fuses the two the way the model uses them (SparseDenseNetRefinementMask.py:183-192).
//
// Why the matrix cores for an HBM-shaped op: at stage 3 (C=8, D=216) every byte of L/R
// feeds 36 fp32 MACs + 4.5 exp, above the chip's fp32 ridge.  cost[x'][x] = sum_c R[c][x'] L[c][x]
// over a 16x16 tile of (right pixel, left pixel) IS a K=C matrix product, and
// v_mfma_f32_16x16x4_f32 evaluates it as the same c-ordered fp32 fma chain the reference
// binary runs (exact fp32, no reduced precision) with one operand register per 64 MACs
// instead of one LDS read per MAC.
//
// One workgroup (8 waves) owns one segment (normally the whole row) of ONE image row.
//   phase 1  both mask rows -> LDS; block-wide prefix counts of the active pixels.
//   phase 2  R[C][HALO+SW] -> LDS with 16-byte row loads (all issued before the first store).
//   phase 3  per row, one of two paths:
//     DENSE    (>= 80 % of the candidate pairs active): a wave owns 16 consecutive left pixels;
//              NT = ceil((D-1)/16)+1 cost tiles of the disparity band, 4*NT costs per lane in
//              registers (lane&15 = left pixel, 4*(lane>>4)+reg = right pixel); d is affine
//              in (tile, register, lane); the right mask is an additive 0 / -1e30 bias.
//     COMPACT  (sparse masks): left pixels are grouped in aligned spans of S pixels (S chosen
//              per row so that a span holds <= ~16 active pixels and its disparity window holds
//              <= 16*(NT+1) active right pixels); a wave multiplies the 16 gathered active
//              left pixels against 16-wide tiles of the COMPACTED list of active right pixels
//              in the window, so work scales with density^2 and the pass becomes HBM-bound.
//   Both paths then run max / exp-sum / variance passes over registers and a 4-lane exchange.
// Rows with at most 256 active pixels per side are taken by spamat_fwd_sparse (further down) before
// this kernel runs: it never stages a whole row and reaches 55-93 % of the HBM roofline.
//
// <= 128 VGPRs: 4 waves share a SIMD (fp32 MFMA and VALU share the FP32 units on gfx950 --
// measured, see DESIGN.md -- so occupancy hides latency, it does not add throughput).
// Compiled with -fno-honor-nans (build.py): otherwise every fmaxf on an MFMA result costs an
// extra canonicalising v_max.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

enum { MODE_MAT = 0, MODE_VAR = 1, MODE_FUSED = 2 };

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int THREADS = 512, NWAVE = THREADS / 64;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 4 consecutive floats row[x .. x+3], zeros outside [0, W); one 16-byte load when possible.
// Four consecutive values row[x .. x + 3] (zeros outside [0, W)), x a multiple of 4, in two forms chosen by a
// WORKGROUP-UNIFORM flag `al` (uniform_flag(): the rows start on 16-byte boundaries; with W a multiple of 4 a group lies
// inside or outside its row as a whole, otherwise the caller does the row's last group with load4s after its batch):
//   load4f  ONE unconditional 16-byte load -- lanes that are outside (or whose channel does not exist: `ok`) read `safe`,
//           any aligned readable address -- and a select (SEL = false: no select; for callers whose outside lanes are
//           finite-but-ignored: right pixels outside the row carry the -1e30 bias, left pixels outside it are not stored);
//   load4s  four guarded 4-byte loads.
// Round 5, from the ISA: the earlier per-lane `if (inside) wide load else four guarded loads` has both sides writing the
// same registers, and the compiler orders them with `s_waitcnt vmcnt(0)` in front of EVERY wide load -- a staging item's
// eight loads were eight serial memory round trips (stage 3 dense rows: 0.10 of 0.42 ms; stages 1 - 2 and the mid-density
// body likewise).  Call sites branch ONCE on the flag around all of their loads.
__device__ __forceinline__ bool uniform_flag(bool f) { return __builtin_amdgcn_readfirstlane((int)f) != 0; }
template <bool SEL = true>
__device__ __forceinline__ float4 load4f(const float *__restrict__ safe, const float *__restrict__ row, int x, int W, bool ok) {
    const bool in = ok && x >= 0 && x < W;
    const float4 t = *reinterpret_cast<const float4 *>(in ? row + x : safe);
    if (!SEL) return t;                                 // the caller never uses the outside lanes' values as numbers that count
    return make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
}
// the per-lane form (sparse-row bodies only: two mask loads, or loads whose lanes are all inside)
__device__ __forceinline__ float4 load4_lanes(const float *__restrict__ row, int x, int W, bool aligned) {
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
__device__ __forceinline__ float4 load4s(const float *__restrict__ row, int x, int W) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < W) v.x = row[x];
    if (x + 1 >= 0 && x + 1 < W) v.y = row[x + 1];
    if (x + 2 >= 0 && x + 2 < W) v.z = row[x + 2];
    if (x + 3 >= 0 && x + 3 < W) v.w = row[x + 3];
    return v;
}

// 4 activity flags of the pixels x .. x + 3 (x a multiple of 4) out of a row of bit-packed masks (bit i of word w =
// pixel 64 w + i, zero past W: decnet_detail_mask's layout); positions outside the row read as inactive.
__device__ __forceinline__ int mask4_bits(const unsigned long long *__restrict__ rowbits, int x, int W) {
    if (x < 0 || x >= W) return 0;
    return (int)((rowbits[x >> 6] >> (x & 63)) & 15ull);
}

// the same out of a row of 32-bit words (bit i of word w = pixel 32 w + i): the masks the sparse-row kernel hands over to the
// band kernel in the row's own max_cost entries (mbits = 2, see spamat_fwd_sparse)
__device__ __forceinline__ int mask4_words(const unsigned *__restrict__ roww, int x, int W) {
    if (x < 0 || x >= W) return 0;
    return (int)((roww[x >> 5] >> (x & 31)) & 15u);
}

// fp32 -> three bf16 terms x = hi + mid + lo (truncations with exact residuals: 24 mantissa bits
// together), 8 values -> three packed 8 x bf16 MFMA operands.
__device__ __forceinline__ void split3x8(const float (&x)[8], i32x4 &hi, i32x4 &mid, i32x4 &lo) {
    int h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = __float_as_int(x[e]) & 0xffff0000;
        const float r1 = x[e] - __int_as_float(h[e]);
        m[e] = __float_as_int(r1) & 0xffff0000;
        l[e] = __float_as_int(r1 - __int_as_float(m[e]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                       // {x[2e+1][31:16], x[2e][31:16]}
        hi[e] = __builtin_amdgcn_perm(h[2 * e + 1], h[2 * e], 0x07060302);
        mid[e] = __builtin_amdgcn_perm(m[2 * e + 1], m[2 * e], 0x07060302);
        lo[e] = __builtin_amdgcn_perm(l[2 * e + 1], l[2 * e], 0x07060302);
    }
}

// LDS layout (4-byte words).  The channel pitch RP is == 16 (mod 32) so that the four channel
// rows an MFMA operand fetch touches (lanes 0-15 / 16-31 / 32-47 / 48-63) never share a bank.
//   Rs [Cq][RP]   R[c][xs - HALO + j];  column RP-1 is kept zero (target of padded gathers)
//   BX [RP]       dense: 0 / -1e30 bias of the right mask;  compact: compacted right indices
//   RK [RP+4]     exclusive prefix count of active right pixels
//   LM [SW+16]    dense: left mask;  compact: exclusive prefix count of active left pixels
//   XL [SW]       compacted left positions
//   WT [32]       scan scratch
// dense16_body's term planes keep position p at 16-byte slot (p % 4) * P + p / 4: a staging thread's four positions go to
// four runs of consecutive slots (its neighbours' next to them: no bank conflicts on the ds_write_b128; with position-major
// slots the lanes were 64 bytes apart, a 4-way conflict = the 15 % conflict cycles the SQ counters showed), and the sixteen
// lanes of an operand read touch sixteen different slots mod 16 when P % 8 == 4.
__host__ __device__ inline int d16_pitch(int quads) { return quads + ((4 - quads) & 7); }
struct Layout {
    int SW, HALO, RP, Cq;
    int offR, offBX, offRK, offLM, offXL, offWT, total;
};
__host__ __device__ inline Layout make_layout(int C, int NT, int XT, bool d16 = false) {
    Layout l;
    l.SW = XT * 16;
    l.HALO = (NT - 1) * 16;
    l.Cq = (C + 3) & ~3;
    l.RP = ((l.HALO + l.SW + 31) & ~31) + 16;
    l.offR = 0;
    // (D16: the same region holds both views as bf16 terms, dense16_body)
    const int swh = ((XT + 1) / 2) * 16;                             // dense16_body runs on half a segment at a time
    const int rt = d16 ? 3 * ((C + 7) / 8) * 16 * (d16_pitch((l.HALO + swh) >> 2) + d16_pitch(swh >> 2)) : 0;
    l.offBX = l.offR + (l.Cq * l.RP > rt ? l.Cq * l.RP : rt);
    l.offRK = l.offBX + l.RP;
    l.offLM = l.offRK + l.RP + 4;
    l.offXL = l.offLM + l.SW + 16;
    l.offWT = l.offXL + l.SW;
    l.total = l.offWT + 32;
    return l;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// Softmax passes over NTL cost tiles held in acc[].  DENSE (COMPACT = 0): d = 16*m + dl - r (affine).
// COMPACT = 1: d = xlj - XR[16*(t0+m) + 4q + r], XR an int list in LDS (band kernel's compact path).
// COMPACT = 2 (sparse-row bodies): the list holds the positions as FLOATS (exact: < 2^24), so d = xlf - xr is one
// subtraction instead of an integer subtraction and a conversion in passes 2 and 3, and the range test of pass 1
// (0 <= d < D  <=>  |(xlf - h) - xr| <= h, h = (D - 1) / 2) is three operations like the integer one.
// COMPACT = 3 (the mid-density body): as 2, and the range test runs only on the tiles outside [m_lo, m_hi): both lists
// are sorted, so every slot from the first one the chunk's LAST left pixel may match up to the last one its FIRST left
// pixel may match is in range for all 16 left pixels -- at stage 3, density 0.3 that is ~60 % of a chunk's tiles.  (The
// per-tile scalar branches cost ~20 registers: spills in the 80-register sparse-row kernel, which keeps COMPACT = 2.)
// Same values as COMPACT = 1 bit for bit.  Returns through references.
// Tiles 0 .. ntile - 1 are the live ones.  COMPACT paths leave the tile sequence with ONE taken branch at
// the first dead tile -- `if (m < ntile)` around every tile made the compiler move each tile's body out of line:
// two taken branches per live tile and pass, ~50 per chunk (round 5, from the ISA).  The sparse-row bodies (COMPACT >= 2)
// test only every second tile: a dead odd tile holds slots behind the chunk's window (right pixels beyond the last left
// pixel, or the list's padding), which the range test of pass 1 turns into -1e30 like any other out-of-range candidate
// -- half as many tests, and two tiles' loads and arithmetic to interleave.
// live_tiles<0, N, STEP>(ntile, f): f(integral_constant<m>) for the live tiles m < ntile, as NESTED ifs (compile-time
// recursion) -- one forward branch out at the first dead tile, every live tile's body on the fall-through path, acc[m]
// indexed statically.  (A `break` out of the unrolled loop does the same on paper, but at 16 tiles the unroller gives up
// and the accumulators go to scratch memory.)  STEP = 2 tests every second tile only; STEP = 0: no test, all N tiles.
template <int M, int N, int STEP, class F>
__device__ __forceinline__ void live_tiles(int ntile, F &&f) {
    if constexpr (M < N) {
        if (STEP > 0 && M >= ntile) return;
        f(std::integral_constant<int, M>{});
        if constexpr (STEP == 2 && M + 1 < N) f(std::integral_constant<int, M + 1>{});
        live_tiles<M + (STEP > 1 ? STEP : 1), N, STEP>(ntile, f);
    }
}
template <int NTL, int MODE, int COMPACT>
__device__ __forceinline__ void softmax_passes(f32x4 (&acc)[NTL], int ntile, int D, int dl,
                                               const float *__restrict__ lds, int bias_off, int xr_off,
                                               int xlj, float mu_in,
                                               float &mx_o, float &S_o, float &mu_o, float &var_o,
                                               int m_lo = 0, int m_hi = 0) {
    constexpr int GS = COMPACT >= 2 ? 2 : COMPACT;      // tile gate: pairs (sparse-row bodies), every tile, none (dense)
    int dlv = dl;
    asm volatile("" : "+v"(dlv));   // opaque: otherwise LICM hoists every range compare out of the
                                    // tile loop and spills their lane masks
    // ---- pass 1: range 0 <= d < min(D, x+1) (SM_kernel.cu:42,46), max_cost (SM_kernel.cu:45-59)
    float mx0 = 0.000001f, mx1 = 0.000001f;
    // COMPACT: the right positions are re-read from LDS in every pass through a pointer the
    // compiler cannot identify with the previous pass's (otherwise GVN keeps all 4*NTL of them
    // live across the passes and the kernel spills)
    // (LDS word offsets, made opaque per pass: see the COMPACT note above)
    (void)bias_off;                 // the dense path's right-mask bias is the MFMAs' initial accumulator
    int xo1 = xr_off;
    asm volatile("" : "+v"(xo1));
    const int *xp1 = reinterpret_cast<const int *>(lds) + xo1;
    const float xlf = (float)xlj, hD = 0.5f * (float)(D - 1), xlh = xlf - hD;      // (COMPACT = 2)
    live_tiles<0, NTL, GS>(ntile, [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        {
            if (COMPACT >= 2) {
                if (COMPACT == 2 || m < m_lo || m >= m_hi) {   // wave-uniform: a tile on the edge of the chunk's window
                    if (COMPACT == 3) asm volatile("" ::: "memory");   // keep a real scalar branch (no if-conversion)
                    const float4 p = *reinterpret_cast<const float4 *>(xp1 + 16 * m);
                    acc[m][0] = fabsf(xlh - p.x) <= hD ? acc[m][0] : NEG_BIG;
                    acc[m][1] = fabsf(xlh - p.y) <= hD ? acc[m][1] : NEG_BIG;
                    acc[m][2] = fabsf(xlh - p.z) <= hD ? acc[m][2] : NEG_BIG;
                    acc[m][3] = fabsf(xlh - p.w) <= hD ? acc[m][3] : NEG_BIG;
                }
            } else if (COMPACT) {
                const int4 p = *reinterpret_cast<const int4 *>(xp1 + 16 * m);
                acc[m][0] = (unsigned)(xlj - p.x) < (unsigned)D ? acc[m][0] : NEG_BIG;
                acc[m][1] = (unsigned)(xlj - p.y) < (unsigned)D ? acc[m][1] : NEG_BIG;
                acc[m][2] = (unsigned)(xlj - p.z) < (unsigned)D ? acc[m][2] : NEG_BIG;
                acc[m][3] = (unsigned)(xlj - p.w) < (unsigned)D ? acc[m][3] : NEG_BIG;
            } else {
                // (the right mask, SM_kernel.cu:48, is already in: the cost MFMAs start from the
                // 0 / -1e30 bias tile instead of zeros)
                if (m == 0 || 16 * m + 15 >= D) {
                    asm volatile("" ::: "memory");  // keep a real scalar branch (no if-conversion)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = 16 * m + dlv - r;
                        acc[m][r] = (unsigned)d >= (unsigned)D ? NEG_BIG : acc[m][r];
                    }
                }
            }
            mx0 = fmaxf(fmaxf(mx0, acc[m][0]), acc[m][1]);
            mx1 = fmaxf(fmaxf(mx1, acc[m][2]), acc[m][3]);
        }
    });
    float mx = fmaxf(mx0, mx1);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!COMPACT && MODE != MODE_MAT) {
        // ---- dense rows, variance wanted: passes 2 + 3 through per-tile moments.  A lane's four values of
        // tile m sit at d = A - r, A = 16 m + dl, r = 0..3.  Pass 2 keeps s = sum e_r, t2 = 2 sum r e_r,
        // q = sum r^2 e_r per tile (3 registers instead of the 4 exponentials); then
        //   S = sum s,   T = sum (16 m) s - sum t2 / 2 + dl S,   V = sum_m (A - mu) ((A - mu) s - t2) + q
        // -- 3.5 VALU operations per candidate instead of 5, and no cancellation: the expansion is around
        // the tile's own position, every term is of the size of e (d - mu)^2 itself.
        // (measured in tools/ubench/softmax_rate.hip: 2440 vs 2900 cycles per 60 candidates per SIMD)
        const float nmv = -mx * LOG2E;
        float Sa = 0.f, Sb = 0.f, T16a = 0.f, T16b = 0.f, Tt = 0.f;
#pragma unroll
        for (int m = 0; m < NTL; ++m) {
            if (m < ntile) {
                const float e0 = fast_exp2(fmaf(acc[m][0], LOG2E, nmv));
                const float e1 = fast_exp2(fmaf(acc[m][1], LOG2E, nmv));
                const float e2 = fast_exp2(fmaf(acc[m][2], LOG2E, nmv));
                const float e3 = fast_exp2(fmaf(acc[m][3], LOG2E, nmv));
                const float s4 = (e0 + e1) + (e2 + e3);
                const float t2 = fmaf(6.f, e3, fmaf(4.f, e2, e1 + e1));
                const float qq = fmaf(9.f, e3, fmaf(4.f, e2, e1));
                acc[m][0] = s4;
                acc[m][1] = t2;
                acc[m][2] = qq;
                if (m & 1) { Sb += s4; T16b = fmaf(s4, (float)(16 * m), T16b); }
                else { Sa += s4; T16a = fmaf(s4, (float)(16 * m), T16a); }
                Tt += t2;
            }
        }
        const float dlf = (float)dl;
        float Sl = Sa + Sb;
        float Tl = fmaf(dlf, Sl, fmaf(-0.5f, Tt, T16a + T16b));
        Sl += __shfl_xor(Sl, 16);
        Sl += __shfl_xor(Sl, 32);
        const float S = Sl + 0.000001f;
        float mu = mu_in;
        if (MODE != MODE_VAR) {
            Tl += __shfl_xor(Tl, 16);
            Tl += __shfl_xor(Tl, 32);
            mu = (Tl + 0.000001f) / S;
        }
        const float c0 = dlf - mu;
        float V0 = 0.f, V1 = 0.f;
#pragma unroll
        for (int m = 0; m < NTL; ++m) {
            if (m < ntile) {
                const float A = (float)(16 * m) + c0;
                const float u = fmaf(A, acc[m][0], -acc[m][1]);
                if (m & 1) V1 = fmaf(A, u, V1) + acc[m][2];
                else V0 = fmaf(A, u, V0) + acc[m][2];
            }
        }
        float Vl = V0 + V1;
        Vl += __shfl_xor(Vl, 16);
        Vl += __shfl_xor(Vl, 32);
        mx_o = mx; S_o = S; mu_o = mu; var_o = (Vl + 0.000001f) / S;
        return;
    }
    // ---- pass 2: e = exp(cost - max) (one fma: cost*log2e - max*log2e), S, T (SM_kernel.cu:100-122)
    const float nm = -mx * LOG2E;
    float S0 = 0.f, S1 = 0.f, T0 = 0.f, T1 = 0.f;
    int xo2 = xr_off;
    asm volatile("" : "+v"(xo2));
    const int *xp2 = reinterpret_cast<const int *>(lds) + xo2;
    live_tiles<0, NTL, GS>(ntile, [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        {
            int4 p;
            float4 pf;
            if (COMPACT >= 2) pf = *reinterpret_cast<const float4 *>(xp2 + 16 * m);
            else if (COMPACT) p = *reinterpret_cast<const int4 *>(xp2 + 16 * m);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                float e0 = fast_exp2(fmaf(acc[m][r], LOG2E, nm));
                float e1 = fast_exp2(fmaf(acc[m][r + 1], LOG2E, nm));
                acc[m][r] = e0;
                acc[m][r + 1] = e1;
                S0 += e0;
                S1 += e1;
                if (MODE != MODE_VAR) {
                    if (COMPACT >= 2) {
                        T0 = fmaf(e0, xlf - (r == 0 ? pf.x : pf.z), T0);
                        T1 = fmaf(e1, xlf - (r == 0 ? pf.y : pf.w), T1);
                    } else if (COMPACT) {
                        T0 = fmaf(e0, (float)(xlj - (r == 0 ? p.x : p.z)), T0);
                        T1 = fmaf(e1, (float)(xlj - (r == 0 ? p.y : p.w)), T1);
                    } else {
                        T0 = fmaf(e0, (float)(16 * m - r), T0);
                        T1 = fmaf(e1, (float)(16 * m - r - 1), T1);
                    }
                }
            }
        }
    });
    const float dlf = (float)dl;
    float Sl = S0 + S1;
    float Tl = COMPACT ? T0 + T1 : fmaf(dlf, Sl, T0 + T1);      // dense: d = (16m - r) + dl
    Sl += __shfl_xor(Sl, 16);
    Sl += __shfl_xor(Sl, 32);
    const float S = Sl + 0.000001f;
    float mu = mu_in;
    if (MODE != MODE_VAR) {
        Tl += __shfl_xor(Tl, 16);
        Tl += __shfl_xor(Tl, 32);
        mu = (Tl + 0.000001f) / S;
    }
    // ---- pass 3: V = sum e*(d-mu)^2  (SV_kernel.cu:100-121)
    float var = 0.f;
    if (MODE != MODE_MAT) {
        const float c0 = (COMPACT ? (float)xlj : dlf) - mu;
        float V0 = 0.f, V1 = 0.f;
        int xo3 = xr_off;
        asm volatile("" : "+v"(xo3));
        const int *xp3 = reinterpret_cast<const int *>(lds) + xo3;
        live_tiles<0, NTL, GS>(ntile, [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            {
                int4 p;
                float4 pf;
                if (COMPACT >= 2) pf = *reinterpret_cast<const float4 *>(xp3 + 16 * m);
                else if (COMPACT) p = *reinterpret_cast<const int4 *>(xp3 + 16 * m);
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float d0, d1;
                    if (COMPACT >= 2) {
                        d0 = c0 - (r == 0 ? pf.x : pf.z);
                        d1 = c0 - (r == 0 ? pf.y : pf.w);
                    } else if (COMPACT) {
                        d0 = c0 - (float)(r == 0 ? p.x : p.z);
                        d1 = c0 - (float)(r == 0 ? p.y : p.w);
                    } else {
                        d0 = (float)(16 * m - r) + c0;
                        d1 = (float)(16 * m - r - 1) + c0;
                    }
                    V0 = fmaf(acc[m][r] * d0, d0, V0);
                    V1 = fmaf(acc[m][r + 1] * d1, d1, V1);
                }
            }
        });
        float Vl = V0 + V1;
        Vl += __shfl_xor(Vl, 16);
        Vl += __shfl_xor(Vl, 32);
        var = (Vl + 0.000001f) / S;
    }
    mx_o = mx; S_o = S; mu_o = mu; var_o = var;
}


// ---------------------------------------------------------------------------------------------
// Dense rows on the bf16 matrix cores (dense path of spamat_fwd_mfma<.., D16 = true>).
//
// The fp32 MFMAs of the band kernel above run on the SIMDs' FP32 lanes: their 960 cycles per wave-tile
// (30 x v_mfma_f32_16x16x4_f32) ADD to the softmax VALU passes (ablation builds), a quarter of the
// dense pass.  Here cost[x'][x] = sum_c R[c][x'] L[c][x] is computed with both operands as three bf16 terms
// (hi + mid + lo = the 24 mantissa bits, truncations with exact residuals) and every term pair except lo*lo
// (2^-32) in the K axis of two v_mfma_f32_16x16x32_bf16 per 8 channels:
//     k group (lane >> 4)      0        1        2        3
//     "big"   MFMA  (A, B)   (hi,hi)  (hi,mid) (mid,hi) (mid,mid)
//     "small" MFMA  (A, B)   (hi,lo)  (lo,hi)  (mid,lo) (lo,mid)
// Products of bf16 are exact and the accumulation is fp32: fp32 accuracy, on a pipe that works beside the other
// waves' VALU passes and holds the issue port 8 cycles per MFMA instead of 32.  What made this pay (a first
// version that split the left features per 16-pixel tile in registers was no faster): BOTH feature rows are
// split ONCE per workgroup while they are staged -- RT / LT [3 terms][C/8][position][8 ch x bf16], 16 bytes
// per term, channel group and position -- so an MFMA operand is one ds_read_b128 and the splitting costs
// ~5 % of the softmax work instead of ~20 %.  LDS: 48 bytes per staged position and channel group for each
// side, so a 972-pixel row of stage 3 is two segments (63 KB each, two workgroups per CU).
// BX (right-mask bias) and LM (left mask) are filled by the caller's phase 1.
// 8 channel rows x 4 positions of one view.  Positions right of the row are NOT zeroed on the aligned path (they read the
// row's first group): a right pixel there has the -1e30 bias (BX: its mask reads as 0) and only negative disparities, a left
// pixel outside the row is never stored.  Positions left of the row and channels that do not exist are zeros.
__device__ __forceinline__ void dense16_loads8(float4 (&v)[8], const float *__restrict__ src, size_t plane, int g, int C,
                                               int x, int W, bool al) {
    if (al) {
        const bool whole = x + 3 < W;                   // W % 4 != 0: the row's last group is done after the batch
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v[c] = load4f<false>(src, src + (size_t)(8 * g + c) * plane, x, W, whole && 8 * g + c < C);
        __builtin_amdgcn_sched_barrier(0);              // all eight requests before anything that waits for one
        // positions LEFT of the row (the first segment's halo) read the row's first group: finite-but-ignored under the
        // -1e30 bias as long as that group is finite -- a NaN / Inf there would reach candidates (x' < 0, d <= D - 1) of left
        // pixels whose candidate set does not hold it, against the header's contract: zeros instead (one wave-uniform
        // branch; only the waves that stage such positions take it).  Positions right of the row need nothing: every tile
        // row there has d < 0 and the range select replaces it.
        if (__ballot(x < 0) != 0ull) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (x < 0) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (C & 7) {                                    // (uniform) a partial channel group: its missing channels are zeros
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bool ch = 8 * g + c < C;
                v[c] = make_float4(ch ? v[c].x : 0.f, ch ? v[c].y : 0.f, ch ? v[c].z : 0.f, ch ? v[c].w : 0.f);
            }
        }
        if ((W & 3) && x < W && !whole) {               // (W & 3: uniform) one group per row: guarded loads
#pragma unroll
            for (int c = 0; c < 8; ++c)
                v[c] = 8 * g + c < C ? load4s(src + (size_t)(8 * g + c) * plane, x, W) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v[c] = 8 * g + c < C ? load4s(src + (size_t)(8 * g + c) * plane, x, W) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the loads of staging item `it` of dense16_body(xs, SW, HALO, nRw): 8 channels x 4 positions of one view
__device__ __forceinline__ void dense16_item_loads(float4 (&v)[8], int it, const float *__restrict__ lrow,
                                                   const float *__restrict__ rrow, size_t plane, int C, int W, int xs,
                                                   int SW, int HALO, int nRw) {
    const bool al = uniform_flag(((((uintptr_t)rrow) | ((uintptr_t)lrow) | ((uintptr_t)(plane * 4))) & 15) == 0);
    const int cg_n = (C + 7) >> 3, nqR = nRw >> 2, nqL = SW >> 2;
    const int nR_items = cg_n * nqR;
    const bool isL = it >= nR_items;
    const int k = isL ? it - nR_items : it, nq = isL ? nqL : nqR;
    const int g = k / nq, jq = k - g * nq;
    const float *src = isL ? lrow : rrow;
    const int x = isL ? xs + 4 * jq : xs - HALO + 4 * jq;
    dense16_loads8(v, src, plane, g, C, x, W, al);
}

// pre: the loads of this thread's first staging item (it = tid), requested by the caller before the mask phase (PRE)
template <int NT, int MODE, int CGT, bool PRE = false>
__device__ __forceinline__ void dense16_body(int *RT, const float *BX, const float *LM, const float *smem,
                                             const float *__restrict__ lrow, const float *__restrict__ rrow,
                                             const float *__restrict__ disparity, float *__restrict__ out,
                                             float *__restrict__ var_out, float *__restrict__ sum_sim,
                                             float *__restrict__ max_cost, size_t plane, size_t rowpix, int C, int W,
                                             int D, int xs, int XT, int SW, int HALO, int nRw, const float4 *pre = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    struct { int CG; } lo = {(C + 7) >> 3};
    const int PR = d16_pitch(nRw >> 2), PL = d16_pitch(SW >> 2);          // slots per (term, group, p % 4) run
    int *LT = RT + 3 * lo.CG * 16 * PR;
    // ---- features of both views: 8 channels x 4 positions per item, split into the three bf16 terms
    {
        const bool al = uniform_flag(((((uintptr_t)rrow) | ((uintptr_t)lrow) | ((uintptr_t)(plane * 4))) & 15) == 0);
        const int cg_n = lo.CG, nqR = nRw >> 2, nqL = SW >> 2;
        const int nR_items = cg_n * nqR, n_items = nR_items + cg_n * nqL;
#pragma unroll 1
        for (int it = tid; it < n_items; it += THREADS) {
            const bool isL = it >= nR_items;
            const int k = isL ? it - nR_items : it, nq = isL ? nqL : nqR;
            const int g = k / nq, jq = k - g * nq;
            const float *src = isL ? lrow : rrow;
            const int x = isL ? xs + 4 * jq : xs - HALO + 4 * jq;
            float4 v[8];
            if (PRE && it == tid) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = pre[c];
            } else {
                dense16_loads8(v, src, plane, g, C, x, W, al);
            }
            const int P = isL ? PL : PR;
            int *dst = (isL ? LT : RT) + (g * 4 * P + jq) * 4;
            const int tstride = cg_n * 16 * P;                            // words between terms
#pragma unroll
            for (int pz = 0; pz < 4; ++pz) {
                float xv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) xv[c] = pz == 0 ? v[c].x : pz == 1 ? v[c].y : pz == 2 ? v[c].z : v[c].w;
                i32x4 th, tm, tl;
                split3x8(xv, th, tm, tl);
                *reinterpret_cast<i32x4 *>(dst + pz * 4 * P) = th;
                *reinterpret_cast<i32x4 *>(dst + tstride + pz * 4 * P) = tm;
                *reinterpret_cast<i32x4 *>(dst + 2 * tstride + pz * 4 * P) = tl;
            }
        }
    }
    __syncthreads();

    const int j = lane & 15, q = lane >> 4;
    const int dl = j - 4 * q;                                            // d = 16 m + dl - r
    constexpr int CGB = CGT ? CGT : 1;
    const int cg_n = CGT ? CGT : lo.CG;
    // term of this lane's k group: A (right) big {hi,hi,mid,mid}, small {hi,lo,mid,lo}; B (left) big {hi,mid,hi,mid},
    // small {lo,hi,lo,mid}
    const int tA_big = q >> 1, tA_small = q == 0 ? 0 : q == 2 ? 1 : 2;
    const int tB_big = q & 1, tB_small = q == 1 ? 0 : q == 3 ? 1 : 2;
    const int gsR = 16 * PR, gsL = 16 * PL;                              // words between channel groups
    for (int xt = wave; xt < XT; xt += NWAVE) {
        const int x0 = xs + xt * 16;
        if (x0 >= W) break;
        const int x = x0 + j;
        const size_t pix = rowpix + x;
        const bool inside = x < W;
        const float rm = LM[xt * 16 + j];
        if (__ballot(rm != 0.f) == 0ull) {                               // no active left pixel in this tile
            if (inside && q == 0) {
                if (MODE != MODE_VAR) out[pix] = 0.f;
                if (MODE != MODE_MAT) var_out[pix] = 0.f;
                sum_sim[pix] = 0.f;
                max_cost[pix] = 0.f;
            }
            continue;
        }
        f32x4 acc[NT];
        // right-mask bias (0 / -1e30 per right pixel = tile row) as the initial accumulator: 0 + x is exact and
        // -1e30 + x = -1e30 (SM_kernel.cu:48 skips those candidates; positions left of the image are -1e30 too)
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const float4 bz = *reinterpret_cast<const float4 *>(BX + (HALO + xt * 16) + 4 * q - 16 * m);
            acc[m] = f32x4{bz.x, bz.y, bz.z, bz.w};
        }
        // tile m holds right pixels x0 - 16 m + (0..15); the lowest tile (m = NT-1) is the base address
        const int sa = (j & 3) * PR + 4 * xt + (j >> 2), sb = (j & 3) * PL + 4 * xt + (j >> 2);   // this lane's slots
        const int *a_small = RT + (tA_small * cg_n * 4 * PR + sa) * 4;
        const int *a_big = RT + (tA_big * cg_n * 4 * PR + sa) * 4;
        const int *b_small = LT + (tB_small * cg_n * 4 * PL + sb) * 4;
        const int *b_big = LT + (tB_big * cg_n * 4 * PL + sb) * 4;
        // (round 5: an explicit software pipeline of this section -- operand reads 3 / 4 / 6 MFMAs ahead, the bias read of
        // a tile two MFMAs ahead of its first MFMA, pinned with sched_barrier -- measured 0.385 - 0.39 ms against 0.385 - 0.39:
        // the compiler's one-read-ahead order is not what the pass waits for; profiles/r05z2_*)
#pragma unroll
        for (int g = 0; g < CGB; ++g) {
            for (int gg = g; gg < cg_n; gg += CGB) {                     // CGT > 0: exactly one trip
                const i32x4 bs = *reinterpret_cast<const i32x4 *>(b_small + gg * gsL);
                const i32x4 bb = *reinterpret_cast<const i32x4 *>(b_big + gg * gsL);
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const i32x4 av = *reinterpret_cast<const i32x4 *>(a_small + gg * gsR + (NT - 1 - m) * 16);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                                     __builtin_bit_cast(bf16x8, bs), acc[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const i32x4 av = *reinterpret_cast<const i32x4 *>(a_big + gg * gsR + (NT - 1 - m) * 16);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                                     __builtin_bit_cast(bf16x8, bb), acc[m], 0, 0, 0);
                }
            }
        }
        float mx, S, mu, var;
        const float mu_in = (MODE == MODE_VAR && inside) ? disparity[pix] : 0.f;
        softmax_passes<NT, MODE, 0>(acc, NT, D, dl, smem, 0, 0, 0, mu_in, mx, S, mu, var);
        if (inside && q == 0) {
            const bool on = rm != 0.f;
            if (MODE != MODE_VAR) out[pix] = on ? mu : 0.f;
            if (MODE != MODE_MAT) var_out[pix] = on ? var : 0.f;
            sum_sim[pix] = on ? S : 0.f;
            max_cost[pix] = on ? mx : 0.f;
        }
    }

}

// KQ = number of K=4 channel steps when known at compile time (C <= 4*KQ), 0 = runtime loop.
// D16: dense rows go through dense16_body (bf16 matrix cores) instead of the fp32 MFMA band path.
template <int NT, int MODE, int KQ, int PPT, int NTHR, int CAP, int NTCMAX>
__device__ __forceinline__ int sparse_row_body(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int row, int dense_pct, int mbits, int *fr_out = nullptr, int *fl_out = nullptr);
constexpr int MID_CAP = 640;                            // active pixels per side of a "mid-density" row (-2 marker): 55 KB of LDS

template <int NT, int MODE, int KQ, bool D16>
__device__ __forceinline__ void spamat_fwd_segment(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int XT, int allow_compact, int marker, int seg, int row, int compact_pct, int mbits) {
    // marker: this launch follows spamat_fwd_sparse, which left -1 in sum_sim[row start] of exactly
    // the rows it did not take, at the first pixel of every segment (a real sum_similarities is never
    // negative)
    const float mark = marker ? sum_sim[(size_t)row * W + (size_t)seg * (XT * 16)] : -1.0f;
    if (marker && !(mark < 0.f))
        return;
    if constexpr (KQ == 2) {
        // marker == 2 (whole rows per workgroup): a row the sparse-row kernel handed over may still have <= 512 active
        // pixels per side (densities 0.25-0.5 at stage 3) -- this workgroup runs the same sparse-row algorithm on it,
        // with 512 slots and NT + 1 tiles, instead of the compact path below
        if (marker == 2 && mark == -2.0f) {
            if (sparse_row_body<NT, MODE, KQ, 4, THREADS, MID_CAP, NT + 1>(ref, tar, rmask, tmask, disparity, out, var_out,
                                                                          sum_sim, max_cost, C, H, W, D, row, 100,
                                                                          mbits) == 1)      // (the -2 marker means the sparse-row kernel's density test passed)
                return;
            __syncthreads();                            // its LDS arrays are free again
        }
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Layout lo = make_layout(C, NT, XT, D16);
    float *Rs = smem + lo.offR;
    float *BX = smem + lo.offBX;
    int *XR = reinterpret_cast<int *>(BX);
    int *RK = reinterpret_cast<int *>(smem + lo.offRK);
    float *LM = smem + lo.offLM;
    int *RKL = reinterpret_cast<int *>(LM);
    int *XL = reinterpret_cast<int *>(smem + lo.offXL);
    int *WT = reinterpret_cast<int *>(smem + lo.offWT);
    const int SW = lo.SW, HALO = lo.HALO, RP = lo.RP;
    const int kq_n = KQ ? KQ : lo.Cq / 4;
    constexpr int KB = KQ ? KQ : 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = row / H, y = row - b * H;
    const int xs = seg * SW;
    const int nRw = HALO + SW;                       // staged right positions (multiple of 16)
    const size_t plane = (size_t)H * W;
    const float *lrow = ref + ((size_t)b * C * H + y) * W;
    const float *rrow = tar + ((size_t)b * C * H + y) * W;
    const float *trow = tmask + (size_t)row * W;
    const float *mrow = rmask + (size_t)row * W;
    const size_t rowpix = (size_t)row * W;
    // mbits: the mask arguments are bit-packed rows of ceil(W / 64) 64-bit words instead of float planes
    const unsigned long long *tbits = reinterpret_cast<const unsigned long long *>(tmask) + (size_t)row * ((W + 63) >> 6);
    const unsigned long long *lbits = reinterpret_cast<const unsigned long long *>(rmask) + (size_t)row * ((W + 63) >> 6);

    // bf16 layout (stage 3 dense rows): the loads of this thread's first staging item of the first half pass, requested
    // before the mask phase too (a compact row wastes them)
    // dense16 kernels: the handed-over activity words are requested BEFORE the first half pass's features (loads return in
    // order: behind the features the mask phase would wait for all of them; - 1.3 % on dense rows, profiles/r05z2_*)
    unsigned mf_wr = 0, mf_wl = 0;
    if constexpr (D16) {
        if (mbits == 2) {
            const int q4 = tid * 4, xr_ = xs - HALO + q4, xl_ = xs + q4;
            const unsigned *hw = reinterpret_cast<const unsigned *>(max_cost + rowpix);
            mf_wr = hw[(q4 < nRw && xr_ >= 0 && xr_ < W) ? xr_ >> 5 : 0];
            mf_wl = hw[((W + 31) >> 5) + ((q4 < SW && xl_ < W) ? xl_ >> 5 : 0)];
        }
    }
    float4 d16_pre[D16 ? 8 : 1];
    if constexpr (D16) {
        const int xta = (XT + 1) / 2;
        const int n_items = ((C + 7) >> 3) * (((HALO + xta * 16) >> 2) + ((xta * 16) >> 2));
        if (tid < n_items) dense16_item_loads(d16_pre, tid, lrow, rrow, plane, C, W, xs, xta * 16, HALO, HALO + xta * 16);
    }
    // fp32 layouts (stages 1, 2): the first pass of the R staging and the left operand of this wave's first tile are
    // requested BEFORE the mask phase -- neither depends on it -- so that a dense row pays one memory round trip where it
    // paid three (masks, then R, then, behind the barrier, the left operand)
    const bool al_r = ((((uintptr_t)rrow) | ((uintptr_t)(plane * 4))) & 15) == 0;
    const int st_nq = nRw >> 2;                      // 16-byte groups per channel row of the staged window (<= THREADS)
    const int st_rpp = THREADS / st_nq;              // channel rows per pass
    const int st_r0 = tid / st_nq, st_jq = tid - st_r0 * st_nq;
    float4 st_v[D16 ? 1 : 8];
    float bfirst[(!D16 && KQ > 0) ? KQ : 1];
    if constexpr (!D16) {
        if (st_r0 < st_rpp) {
            const int x = xs - HALO + 4 * st_jq;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = st_r0 + u * st_rpp;
                st_v[u] = c < C ? load4_lanes(rrow + (size_t)c * plane, x, W, al_r) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (KQ > 0) {
            const int jl = lane & 15, ql = lane >> 4;
            const int x = xs + wave * 16 + jl;
            const bool ok = wave < XT && x < W;
#pragma unroll
            for (int s = 0; s < KQ; ++s) bfirst[s] = (ok && 4 * s + ql < C) ? lrow[(size_t)(4 * s + ql) * plane + x] : 0.f;
        }
    }

    // ---------------- phase 1: masks -> LDS, flags kept in registers, block-wide counts ----------
    const int p4 = tid * 4;                          // this thread's 4 positions (RP, SW <= 2048)
    int fr = 0, fl = 0;                              // 4 right / left activity bits
    {
        const bool alm = (((uintptr_t)trow) & 15) == 0 && (((uintptr_t)mrow) & 15) == 0;
        const bool almu = D16 && uniform_flag(alm && (W & 3) == 0);     // (dense16 kernels: branch-free, see load4f; the others measured no
                                                        // gain at stage 2 and lost two workgroups per CU at stage 1: 135 registers)
        if (p4 < nRw) {
            const int x = xs - HALO + p4;
            if (mbits == 2) {
                if (D16) fr = (x >= 0 && x < W) ? (int)((mf_wr >> (x & 31)) & 15u) : 0;
                else fr = mask4_words(reinterpret_cast<const unsigned *>(max_cost + rowpix), x, W);
            } else if (mbits) {
                fr = mask4_bits(tbits, x, W);
            } else {
                float4 tv = almu ? load4f(trow, trow, x, W, true) : load4_lanes(trow, x, W, alm);
                fr = (tv.x != 0.f) | ((tv.y != 0.f) << 1) | ((tv.z != 0.f) << 2) | ((tv.w != 0.f) << 3);
            }
            float4 bv4;
            bv4.x = (fr & 1) ? 0.f : NEG_BIG;
            bv4.y = (fr & 2) ? 0.f : NEG_BIG;
            bv4.z = (fr & 4) ? 0.f : NEG_BIG;
            bv4.w = (fr & 8) ? 0.f : NEG_BIG;
            *reinterpret_cast<float4 *>(BX + p4) = bv4;
        }
        if (p4 < SW) {
            float4 mv;
            if (mbits) {
                if (D16 && mbits == 2) fl = (xs + p4 < W) ? (int)((mf_wl >> ((xs + p4) & 31)) & 15u) : 0;
                else fl = mbits == 2 ? mask4_words(reinterpret_cast<const unsigned *>(max_cost + rowpix) + ((W + 31) >> 5), xs + p4, W)
                                : mask4_bits(lbits, xs + p4, W);
                mv = make_float4((fl & 1) ? 1.f : 0.f, (fl & 2) ? 1.f : 0.f, (fl & 4) ? 1.f : 0.f, (fl & 8) ? 1.f : 0.f);
            } else {
                mv = almu ? load4f(mrow, mrow, xs + p4, W, true) : load4_lanes(mrow, xs + p4, W, alm);
                fl = (mv.x != 0.f) | ((mv.y != 0.f) << 1) | ((mv.z != 0.f) << 2) | ((mv.w != 0.f) << 3);
            }
            *reinterpret_cast<float4 *>(LM + p4) = mv;
        }
    }
    {   // per-wave totals only: nothing of the scan stays live across the dense path
        const int ir = wave_incl_scan(__popc(fr), lane), il = wave_incl_scan(__popc(fl), lane);
        if (lane == 63) { WT[wave] = ir; WT[8 + wave] = il; }
    }

    int nR = 0, nL = 0;
    bool compact = false;
    if constexpr (D16) {
        // the row's path decides the LDS format of the features, so the counts come first
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { nR += WT[w]; nL += WT[8 + w]; }
        const int validL = min(SW, W - xs);
        const int validR = min(W, xs + SW) - max(0, xs - HALO);
        compact = allow_compact && ((long)nL * nR * 100 < (long)validL * validR * compact_pct);
        if (!compact) {
            // two passes over half the segment each: both views as bf16 terms need 96 bytes of LDS per pixel and
            // channel group, and the segment partition (= the number of workgroups of a marker launch, the
            // compact path's halos) stays the fp32 layout's
            const int xta = (XT + 1) / 2, xtb = XT - xta;
            dense16_body<NT, MODE, (KQ + 1) / 2, true>(reinterpret_cast<int *>(Rs), BX, LM, smem, lrow, rrow, disparity, out,
                                                       var_out, sum_sim, max_cost, plane, rowpix, C, W, D, xs, xta, xta * 16,
                                                       HALO, HALO + xta * 16, d16_pre);
            if (xtb > 0 && xs + xta * 16 < W) {
                __syncthreads();
                dense16_body<NT, MODE, (KQ + 1) / 2>(reinterpret_cast<int *>(Rs), BX + xta * 16, LM + xta * 16, smem, lrow,
                                                     rrow, disparity, out, var_out, sum_sim, max_cost, plane, rowpix, C,
                                                     W, D, xs + xta * 16, xtb, xtb * 16, HALO, HALO + xtb * 16);
            }
            return;
        }
    }
    // ---------------- phase 2: stage R.  Threads are spread over (channel row, group of 4 positions): a row
    // of the staged window takes nRw/4 threads, the rest of the workgroup takes further channel rows, and
    // every thread has up to 8 loads in flight before its stores (stage 1, C = 72 over 144 positions:
    // 6 loads per thread instead of 72 serial ones on 36 threads).
    {
        const bool al = al_r;
        const int rpp = st_rpp;
        const int r0 = st_r0, jq = st_jq;
        if (r0 < rpp) {
            const int jj = 4 * jq, x = xs - HALO + jj;
            for (int c0 = r0; c0 < lo.Cq; c0 += 8 * rpp) {
                float4 v[8];
                if (!D16 && c0 == r0) {                 // the first pass was requested at the top
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = st_v[D16 ? 0 : u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * rpp;
                        v[u] = c < C ? load4_lanes(rrow + (size_t)c * plane, x, W, al) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u * rpp;
                    if (c < lo.Cq) *reinterpret_cast<float4 *>(Rs + c * RP + jj) = v[u];
                }
            }
        }
        for (int c = tid; c < lo.Cq; c += THREADS) Rs[c * RP + RP - 1] = 0.f;    // zero column
    }
    __syncthreads();

    const int validL = min(SW, W - xs);
    const int validR = min(W, xs + SW) - max(0, xs - HALO);
    if constexpr (!D16) {
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { nR += WT[w]; nL += WT[8 + w]; }
        // compact when fewer than 80 % of the candidate pairs are active (block-uniform)
        compact = allow_compact && ((long)nL * nR * 100 < (long)validL * validR * compact_pct);
    }

    const int j = lane & 15, q = lane >> 4;

    if constexpr (!D16) if (!compact) {
        // =========================== DENSE path ===========================================
        // (Round 5, stages 2 / 1: ablation builds say the fp32 MFMAs below are 38 % / 52 % of the pass (0.060 -> 0.038 ms,
        // 0.0176 -> 0.0084 without them), but moving them to the bf16 pipe -- `dense_cb_body`, 32 channels of one bf16
        // term pair per MFMA, left operand split in registers, commit 6c42429 -- bought 0.0533 -> 0.0507 ms at stage 2
        // and lost at stage 1 (0.0133 -> 0.0141; 0.0551 / 0.0158 with operand prefetch and term-major MFMA order): a
        // workgroup's life is mask round trip -> staging round trip -> 3 tiles per wave, 18 us per row at two
        // workgroups per CU, and that chain, not the arithmetic, is what the 0.053 ms are.  profiles/r05g_*, r05hi_*.)
        const int dl = j - 4 * q;                       // d = 16*m + dl - r
        float bv[KQ <= 6 ? KB : 1], bcur[KB];
        float rm = 0.f;
        // (round 5: unconditional loads from clamped addresses + a select instead of these exec-mask branches were
        // measured SLOWER -- stage 2 0.053 -> 0.060 ms, stage 1 0.0134 -> 0.0177: the branch skips the loads of the
        // prefetch past the last tile and of the padded channel steps)
        auto fetch_left = [&](int xt, float (&dst)[KB]) {
            const int x = xs + xt * 16 + j;
            const bool ok = xt < XT && x < W;
            if (KQ) {
#pragma unroll
                for (int s = 0; s < KQ; ++s)
                    dst[s] = (ok && 4 * s + q < C) ? lrow[(size_t)(4 * s + q) * plane + x] : 0.f;
            }
        };
        constexpr bool PREF = KQ <= 6;                  // next tile's left operand in flight (C = 72: no room, and
                                                        // stage 1 has one tile per wave anyway)
        if constexpr (PREF && KQ > 0) {
#pragma unroll
            for (int s = 0; s < KB; ++s) bv[s] = bfirst[s];      // this wave's first tile: requested at the top
        } else if constexpr (PREF) {
            fetch_left(wave, bv);
        }
        for (int xt = wave; xt < XT; xt += NWAVE) {
            const int x0 = xs + xt * 16;
            if (x0 >= W) break;
            const int x = x0 + j;
            const size_t pix = rowpix + x;
            const bool inside = x < W;
            rm = LM[xt * 16 + j];
            if constexpr (PREF) {
#pragma unroll
                for (int s = 0; s < KB; ++s) bcur[s] = bv[s];
                fetch_left(xt + NWAVE, bv);             // prefetch the next tile's left operand
            } else if (KQ > 0 && xt == wave) {
#pragma unroll
                for (int s = 0; s < KB; ++s) bcur[s] = bfirst[KQ > 0 ? s : 0];
            } else {
                fetch_left(xt, bcur);
            }
            if (__ballot(rm != 0.f) == 0ull) {          // no active left pixel in this tile
                if (inside && q == 0) {
                    if (MODE != MODE_VAR) out[pix] = 0.f;
                    if (MODE != MODE_MAT) var_out[pix] = 0.f;
                    sum_sim[pix] = 0.f;
                    max_cost[pix] = 0.f;
                }
                continue;
            }
            constexpr int nact = NT;                    // tiles left of the image read the zero /
                                                        // -1e30 padding and come out as -1e30

            // banded costs: the c-ordered fp32 fma chain of SM_kernel.cu:52-55 on the matrix
            // cores, then + (0 | -1e30) for the right mask (SM_kernel.cu:48)
            f32x4 acc[NT];
            const float *ap = Rs + q * RP + (HALO + xt * 16) + j;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if (m < nact) {
                    f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    // accumulate on top of the right-mask bias (0 / -1e30 per right pixel = tile row):
                    // 0 + x is exact and -1e30 + x == -1e30, so this equals adding the bias afterwards
                    if (NT > 8) asm volatile("" ::: "memory");   // keep the 15 bias reads from being hoisted together
                                                                 // (spills); short bands: let the LDS reads run ahead
                    const float4 bz = *reinterpret_cast<const float4 *>(BX + (HALO + xt * 16) + 4 * q - 16 * m);
                    a4 = f32x4{bz.x, bz.y, bz.z, bz.w};
                    if (KQ) {
#pragma unroll
                        for (int s = 0; s < KQ; ++s)
                            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP - 16 * m], bcur[s], a4, 0, 0, 0);
                    } else {
                        const float *bp = lrow + (size_t)q * plane + x;   // generic C: L from L2/HBM
                        for (int s = 0; s < kq_n; ++s) {
                            const float lv = (inside && 4 * s + q < C) ? bp[(size_t)4 * s * plane] : 0.f;
                            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP - 16 * m], lv, a4, 0, 0, 0);
                        }
                    }
                    acc[m] = a4;
                }
            }
            float mx, S, mu, var;
            const float mu_in = (MODE == MODE_VAR && inside) ? disparity[pix] : 0.f;
            softmax_passes<NT, MODE, 0>(acc, nact, D, dl, smem, lo.offBX + (HALO + xt * 16) + 4 * q, 0, 0, mu_in, mx, S, mu, var);
            if (inside && q == 0) {
                const bool on = rm != 0.f;
                if (MODE != MODE_VAR) out[pix] = on ? mu : 0.f;
                if (MODE != MODE_MAT) var_out[pix] = on ? var : 0.f;
                sum_sim[pix] = on ? S : 0.f;
                max_cost[pix] = on ? mx : 0.f;
            }
        }
        return;
    }

    // =============================== COMPACT path ==========================================
    {
        // the activity flags are re-read from BX / LM (cheaper than keeping them and the scan
        // live across the dense path) before those arrays are reused for the compacted lists
        int fr2 = 0, fl2 = 0;
        if (p4 < nRw) {
            const float4 t = *reinterpret_cast<const float4 *>(BX + p4);
            fr2 = (t.x == 0.f) | ((t.y == 0.f) << 1) | ((t.z == 0.f) << 2) | ((t.w == 0.f) << 3);
        }
        if (p4 < SW) {
            const float4 t = *reinterpret_cast<const float4 *>(LM + p4);
            fl2 = (t.x != 0.f) | ((t.y != 0.f) << 1) | ((t.z != 0.f) << 2) | ((t.w != 0.f) << 3);
        }
        const int cr = __popc(fr2), cl = __popc(fl2);
        const int ir = wave_incl_scan(cr, lane), il = wave_incl_scan(cl, lane);
        int baseR = 0, baseL = 0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w)
            if (w < wave) { baseR += WT[w]; baseL += WT[8 + w]; }
        __syncthreads();                               // everyone is done reading WT / BX / LM
        const int fr = fr2, fl = fl2;
        int er = baseR + ir - cr, el = baseL + il - cl;     // exclusive counts at p4
        if (p4 < nRw) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                RK[p4 + k] = er;
                if (fr & (1 << k)) XR[er++] = p4 + k;
            }
        }
        if (p4 < SW) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                RKL[p4 + k] = el;
                if (fl & (1 << k)) XL[el++] = p4 + k;
            }
        }
        if (tid == 0) { RK[nRw] = nR; RKL[SW] = nL; }
        if (tid < 16 && nR + tid < RP) XR[nR + tid] = RP - 1;   // padded gathers hit the zero column
    }
    // span: a power-of-two number of pixels holding <= ~16 active left pixels ...
    int S = 128;
    while (S > 16 && (long)S * nL > 24L * validL) S >>= 1;
    __syncthreads();
    // ... whose disparity window holds at most 16*(NT+1) - 15 active right pixels everywhere
    constexpr int NTC = NT + 1;
    while (S > 16) {
        int bad = 0;
        for (int g = tid; g * S < SW; g += THREADS) {
            const int jlo = max(0, g * S + HALO - (D - 1)), jhi = min(nRw - 1, g * S + HALO + S - 1);
            if (RK[jhi + 1] - RK[jlo] > 16 * NTC - 15) bad = 1;
        }
        if (!__syncthreads_or(bad)) break;
        S >>= 1;
    }

    // zero fill of the masked-off left pixels (functions/SpaMat.py:25-27 semantics)
    for (int p = tid; p < validL; p += THREADS) {
        if (RKL[p + 1] == RKL[p]) {
            const size_t pix = rowpix + xs + p;
            if (MODE != MODE_VAR) out[pix] = 0.f;
            if (MODE != MODE_MAT) var_out[pix] = 0.f;
            sum_sim[pix] = 0.f;
            max_cost[pix] = 0.f;
        }
    }

    const int ngroups = (SW + S - 1) / S;
    for (int g = wave; g < ngroups; g += NWAVE) {
        const int gx = g * S;
        const int e0 = __builtin_amdgcn_readfirstlane(RKL[gx]), e1 = __builtin_amdgcn_readfirstlane(RKL[min(gx + S, SW)]);
        if (e1 == e0) continue;
        const int jlo = max(0, gx + HALO - (D - 1)), jhi = min(nRw - 1, gx + HALO + S - 1);
        const int r_lo = __builtin_amdgcn_readfirstlane(RK[jlo]), r_hi = __builtin_amdgcn_readfirstlane(RK[jhi + 1]);
        const int t0 = r_lo >> 4;
        const int ntile = r_hi > r_lo ? ((r_hi - 1) >> 4) - t0 + 1 : 0;      // <= NTC
        for (int e = e0; e < e1; e += 16) {
            const bool act = e + j < e1;
            const int xl = XL[act ? e + j : e1 - 1];           // local left position
            const int x = xs + xl;
            const size_t pix = rowpix + x;
            float bcur[KB];
            if (KQ) {
#pragma unroll
                for (int s = 0; s < KQ; ++s)
                    bcur[s] = (act && 4 * s + q < C) ? lrow[(size_t)(4 * s + q) * plane + x] : 0.f;
            }
            f32x4 acc[NTC];
            const int *xa = XR + 16 * t0 + j;                  // A-operand gather index
            live_tiles<0, NTC, 1>(ntile, [&](auto mc) {
                constexpr int m = decltype(mc)::value;
                {
                    const float *ap = Rs + q * RP + xa[16 * m];
                    f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (KQ) {
#pragma unroll
                        for (int s = 0; s < KQ; ++s)
                            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP], bcur[s], a4, 0, 0, 0);
                    } else {
                        const float *bp = lrow + (size_t)q * plane + x;
                        for (int s = 0; s < kq_n; ++s) {
                            const float lv = (act && 4 * s + q < C) ? bp[(size_t)4 * s * plane] : 0.f;
                            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * RP], lv, a4, 0, 0, 0);
                        }
                    }
                    acc[m] = a4;
                }
            });
            float mx, Ssum, mu, var;
            const float mu_in = (MODE == MODE_VAR && act) ? disparity[pix] : 0.f;
            // d = (xl + HALO) - XR[...]: both in staged-row coordinates
            softmax_passes<NTC, MODE, 1>(acc, ntile, D, 0, smem, 0, lo.offBX + 16 * t0 + 4 * q, xl + HALO, mu_in,
                                         mx, Ssum, mu, var);
            if (act && q == 0) {
                if (MODE != MODE_VAR) out[pix] = mu;
                if (MODE != MODE_MAT) var_out[pix] = var;
                sum_sim[pix] = Ssum;
                max_cost[pix] = mx;
            }
        }
    }
}

// The kernel: one workgroup per segment of a row.  (Taking the two segments the bf16 layout makes of a stage-3 row
// in one workgroup, one after the other, halves the exiting workgroups of an all-sparse marker launch (-2 us) but
// serialises a dense row's two staging phases: 0.41 -> 0.44 ms at density 1.0; a loop over segments around the
// inlined body costs ~20 VGPR spills on top.  Measured, dropped.)
template <int NT, int MODE, int KQ, bool D16>
__global__ __launch_bounds__(THREADS, (KQ > 6 ? 2 : 4)) void spamat_fwd_mfma(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int segs_per_row, int XT, int allow_compact, int marker, int compact_pct, int mbits) {
    spamat_fwd_segment<NT, MODE, KQ, D16>(ref, tar, rmask, tmask, disparity, out, var_out, sum_sim, max_cost, C, H, W,
                                          D, XT, allow_compact, marker, blockIdx.x % segs_per_row,
                                          blockIdx.x / segs_per_row, compact_pct, mbits);
}

// ---------------------------------------------------------------------------------------------
// Sparse rows (<= 256 active pixels on each side, i.e. up to ~26 % density at stage 3): nothing but
// the masks and the features of the ACTIVE pixels is touched.  The kernel above stages the whole
// right row (39 KB of LDS at stage 3: two workgroups per CU) and then makes a second, dependent
// round trip for the left features, so a CU never has more than ~1 row of HBM traffic in flight,
// well short of the ~90 KB that 8 TB/s x latency needs; here a row costs 256 threads, <= 128 VGPRs
// and ~28 KB of LDS (compacted index lists + compacted feature tiles), four workgroups per CU.
//   1. both mask rows -> activity bits, block-wide exclusive counts (RK / RKL) and index lists
//   2. gather the C features of the active right / left pixels into RF / LF [C][slot] (all loads of
//      a thread in flight together), zero-fill the outputs of inactive left pixels meanwhile
//   3. spans of S left pixels x 16-wide tiles of the compacted right list: MFMA cost tiles and the
//      softmax passes of the compact path above (d from the index lists)
// Rows that are not sparse enough are left to spamat_fwd_mfma, launched right after with
// marker = 1: this kernel writes -1 to sum_sim[row start] of exactly those rows.
constexpr int SP_THREADS = 256, SP_CAP = 256;
// LDS words of sparse_row_body<.., KQ, PPT, NTHR, CAP, ..>
constexpr size_t sparse_row_words(int kq, int ppt, int nthr, int cap) {
    return (size_t)(cap + 32) + 2 * (size_t)(nthr * ppt / 2 + 2) + cap + 32 + 2 * (size_t)4 * kq * (cap + 16);
}

// The body: ONE whole image row by a workgroup of NTHR threads (W <= NTHR * PPT), at most CAP active pixels per side and
// NTCMAX cost tiles per span.  Returns false -- with nothing written -- when the row is not sparse enough for these limits.
// Two users: spamat_fwd_sparse (256 threads, 256 slots, 8 tiles: six workgroups per CU, the sparse regime's roofline
// numbers) and, round 3, the band kernel's own workgroup on the rows that kernel hands over (512 threads, 512 slots,
// NT + 1 tiles): rows of 257-512 active pixels per side (densities 0.25-0.5 at stage 3) no longer fall onto the band
// kernel's compact path, which stages the whole right row and fetches the left features per 16-pixel chunk.
template <int NT, int MODE, int KQ, int PPT, int NTHR, int CAP, int NTCMAX>
__device__ __forceinline__ int sparse_row_body(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int row, int dense_pct, int mbits, int *fr_out, int *fl_out) {
    // at most 8 cost tiles per span (32 accumulator registers: six workgroups per CU); a row whose
    // disparity windows hold more than 8*16-15 active right pixels even for 16-pixel spans goes to
    // spamat_fwd_mfma like the dense ones
    constexpr int CQ = 4 * KQ, NTC = NT + 1 < NTCMAX ? NT + 1 : NTCMAX;
    constexpr int SP_NWAVE_ = NTHR / 64, SP_FP_ = CAP + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // words: XR [CAP+32] | RK [(NPX+4) x u16] | RKL [(NPX+4) x u16] | XL [CAP] | WT [16] | RF [CQ][FP] | LF [CQ][FP]
    constexpr int NPX = NTHR * PPT, RKW = NPX / 2 + 2;               // RK / RKL: NPX + 4 u16
    constexpr int offXR = 0, offRK = CAP + 32, offRKL = offRK + RKW, offXL = offRKL + RKW,
                  offWT = offXL + CAP, offRF = offWT + 32, offLF = offRF + CQ * SP_FP_;
    float *XR = smem + offXR;                          // positions of the active right pixels, as floats (softmax_passes<.., 2>)
    unsigned short *RK = reinterpret_cast<unsigned short *>(smem + offRK);
    unsigned short *RKL = reinterpret_cast<unsigned short *>(smem + offRKL);
    int *XL = reinterpret_cast<int *>(smem) + offXL;
    int *WT = reinterpret_cast<int *>(smem) + offWT;
    float *RF = smem + offRF, *LF = smem + offLF;
    // (Round 5: a launch of their own for the mid rows -- this body with 576 slots, 10 tiles per chunk, the left view
    // requested behind the right view's compaction: 51 KB and 80 registers without spills, THREE 8-wave workgroups per
    // CU instead of two -- gains 10 % at densities 0.3 - 0.4 (0.130 -> 0.117 ms) and loses everywhere else: rows it
    // cannot take are scanned a third time (0.6: 0.235 -> 0.271), and an input without mid rows pays 4.5 us for its
    // exits (0.1: 0.078 -> 0.082).  With registers spilling (12 tiles, two rows per workgroup) every launch of it pays
    // the scratch set-up: +25 us.  profiles/r05l_mid_rows_own_launch.txt.)
    // (Round 5: the mid-density body with its compacted features as bf16 terms -- cost tiles on v_mfma_f32_16x16x32_bf16
    // as in dense16_body, split where a pixel is compacted -- was built and measured SLOWER at every density below 0.6
    // (0.3: 0.128 -> 0.137 ms, 0.5: 0.189 -> 0.202): the split runs under the compaction's divergence, 350 vector
    // instructions per wave whatever the density, and an operand fetch is 16 bytes per lane instead of 4.
    // profiles/r05j_mid_body_bf16x3_tiles.txt.)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = row / H, y = row - b * H;
    const size_t plane = (size_t)H * W, rowpix = (size_t)row * W;
    const float *lrow = ref + ((size_t)b * C * H + y) * W;
    const float *rrow = tar + ((size_t)b * C * H + y) * W;
    const float *trow = tmask + rowpix, *mrow = rmask + rowpix;

    // ---- 0. (mid-density body) the features of ALL of this thread's pixels are requested with the masks: at 25 - 65 %
    // activity a gather of the active pixels behind the index lists touches nearly every sector anyway, and it is a
    // second, dependent memory round trip.  The 256-slot kernel keeps the gather: sparse rows never fetch inactive lines.
    constexpr bool ALL = CAP > 256 && PPT == 4;
    const int p4 = tid * PPT;                        // this thread's PPT consecutive pixels
    float4 lv[ALL ? CQ : 1], rv[ALL ? CQ : 1];
    if constexpr (ALL) {
        // (load4_lanes: here the per-lane form compiles to sixteen requests in flight -- checked in the ISA -- and the
        // branch-free form measured 2 - 4 % slower at densities 0.3 - 0.4)
        const bool alf = ((W & 3) == 0) && ((((uintptr_t)lrow) | ((uintptr_t)rrow) | ((uintptr_t)(plane * 4))) & 15) == 0;
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const bool ok = p4 < W && c < C;
            lv[c] = ok ? load4_lanes(lrow + (size_t)c * plane, p4, W, alf) : make_float4(0.f, 0.f, 0.f, 0.f);
            rv[c] = ok ? load4_lanes(rrow + (size_t)c * plane, p4, W, alf) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // ---- 1. masks -> bits, counts
    int fr = 0, fl = 0;
    {
        const bool alm = ((W & 3) == 0) && ((((uintptr_t)trow) | ((uintptr_t)mrow)) & 15) == 0;
#pragma unroll
        for (int u = 0; u < PPT; u += 4) {
            if (p4 + u < W) {
                if (mbits == 2) {                       // handed over by the sparse-row kernel (inside the band kernel only)
                    const unsigned *hw = reinterpret_cast<const unsigned *>(max_cost + rowpix);
                    fr |= mask4_words(hw, p4 + u, W) << u;
                    fl |= mask4_words(hw + ((W + 31) >> 5), p4 + u, W) << u;
                } else if (mbits) {
                    const int wpr = (W + 63) >> 6;
                    fr |= mask4_bits(reinterpret_cast<const unsigned long long *>(tmask) + (size_t)row * wpr, p4 + u, W) << u;
                    fl |= mask4_bits(reinterpret_cast<const unsigned long long *>(rmask) + (size_t)row * wpr, p4 + u, W) << u;
                } else {
                    const float4 tv = load4_lanes(trow, p4 + u, W, alm), mv = load4_lanes(mrow, p4 + u, W, alm);
                    fr |= ((tv.x != 0.f) | ((tv.y != 0.f) << 1) | ((tv.z != 0.f) << 2) | ((tv.w != 0.f) << 3)) << u;
                    fl |= ((mv.x != 0.f) | ((mv.y != 0.f) << 1) | ((mv.z != 0.f) << 2) | ((mv.w != 0.f) << 3)) << u;
                }
            }
        }
    }
    if (fr_out) { *fr_out = fr; *fl_out = fl; }
    const int cr = __popc(fr), cl = __popc(fl);
    const int ir = wave_incl_scan(cr, lane), il = wave_incl_scan(cl, lane);
    if (lane == 63) { WT[wave] = ir; WT[8 + wave] = il; }
    __syncthreads();
    int nR = 0, nL = 0, baseR = 0, baseL = 0;
#pragma unroll
    for (int w = 0; w < SP_NWAVE_; ++w) {
        if (w < wave) { baseR += WT[w]; baseL += WT[8 + w]; }
        nR += WT[w];
        nL += WT[8 + w];
    }
    // LDS reads land in vector registers: without this the compiler treats every count-dependent branch below (chunk
    // loop bound, `m < ntile` around each of the 16 tiles of each pass) as DIVERGENT and wraps it in exec-mask
    // save / restore sequences -- 468 scalar instructions per chunk in the mid-density body (round 5, from the ISA)
    nR = __builtin_amdgcn_readfirstlane(nR);
    nL = __builtin_amdgcn_readfirstlane(nL);
    // not sparse enough for this body: nothing has been written, the caller decides (the stand-alone kernel marks the
    // row for the band kernel, the band kernel goes on with its own paths).
    // (2: few enough active pixels for the 512-slot body of the band kernel; 0: a dense row)
    if ((long)nL * nR * 100 >= (long)W * W * dense_pct) return 0;
    if (nL > CAP || nR > CAP) return (nL <= MID_CAP && nR <= MID_CAP) ? 2 : 0;
    {
        int er = baseR + ir - cr, el = baseL + il - cl; // exclusive counts at p4
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            RK[p4 + k] = er;
            if (fr & (1 << k)) {
                if constexpr (ALL) {
#pragma unroll
                    for (int c = 0; c < CQ; ++c)
                        RF[c * SP_FP_ + er] = k == 0 ? rv[c].x : k == 1 ? rv[c].y : k == 2 ? rv[c].z : rv[c].w;
                }
                XR[er++] = (float)(p4 + k);
            }
            RKL[p4 + k] = el;
            if (fl & (1 << k)) {
                if constexpr (ALL) {
#pragma unroll
                    for (int c = 0; c < CQ; ++c)
                        LF[c * SP_FP_ + el] = k == 0 ? lv[c].x : k == 1 ? lv[c].y : k == 2 ? lv[c].z : lv[c].w;
                }
                XL[el++] = p4 + k;
            }
        }
        if (tid == 0) { RK[NPX] = nR; RKL[NPX] = nL; }
        if (tid < 32) XR[nR + tid] = 1048576.0f;        // padding of the last tile and of a dead odd tile: d < 0, out of range
        if constexpr (ALL) {                            // the slots behind the last active pixel read as zeros
            if (tid < 32) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) {
                    if (nR + tid < SP_FP_) RF[c * SP_FP_ + nR + tid] = 0.f;
                    if (nL + tid < SP_FP_) LF[c * SP_FP_ + nL + tid] = 0.f;
                }
            }
        }
    }
    __syncthreads();

    // ---- 2. features of the active pixels (loads first, LDS stores after the span selection)
    constexpr int SPT = (CAP + NTHR - 1) / NTHR;       // slots per thread (1, or 2 for the 256-thread mid-density kernel)
    float rf[ALL ? 1 : SPT][CQ], lf[ALL ? 1 : SPT][CQ];
#pragma unroll
    for (int u = 0; u < (ALL ? 0 : SPT); ++u) {
        const int slot = tid + u * NTHR;
        const int xr_own = slot < nR ? (int)XR[slot] : -1, xl_own = slot < nL ? XL[slot] : -1;
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            rf[u][c] = (xr_own >= 0 && c < C) ? rrow[(size_t)c * plane + xr_own] : 0.f;
            lf[u][c] = (xl_own >= 0 && c < C) ? lrow[(size_t)c * plane + xl_own] : 0.f;
        }
    }
    // chunks of 16 consecutive ACTIVE left pixels (whatever their positions): the window of a chunk is every active
    // right pixel one of them can match, [x_first - (D - 1), x_last]; it has to fit NTC tiles.  (Round 2 cut the row
    // into spans of a power-of-two number of pixels and chunked inside a span: at density 0.3 that is 30 chunks of
    // 9.6 active pixels on average instead of 19 full ones, and a search loop of block-wide votes for the span size.)
    const int nchunk = (nL + 15) >> 4;
    {
        int bad = 0;
        for (int k = tid; k < nchunk; k += NTHR) {
            const int xa = XL[16 * k], xb = XL[min(16 * k + 15, nL - 1)];
            if ((int)RK[xb + 1] - (int)RK[max(0, xa - (D - 1))] > 16 * NTC - 15) bad = 1;
        }
        if (__syncthreads_or(bad)) return 2;            // nothing written yet; more tiles per chunk may do
    }
    if constexpr (!ALL) {
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
#pragma unroll
            for (int u = 0; u < SPT; ++u) {
                const int slot = tid + u * NTHR;
                if (slot < CAP) {
                    RF[c * SP_FP_ + slot] = rf[u][c];    // slots >= nR hold zeros
                    LF[c * SP_FP_ + slot] = lf[u][c];
                }
            }
            if (tid < 16) RF[c * SP_FP_ + CAP + tid] = 0.f;
        }
        __syncthreads();
    }

    // ---- 3. matching
    const int j = lane & 15, q = lane >> 4;
    for (int k = wave; k < nchunk; k += SP_NWAVE_) {
        const int e1 = nL;
        for (int e = 16 * k; e < min(16 * k + 16, nL); e += 16) {      // (one trip: the loop shape the register allocator
            const int xa = __builtin_amdgcn_readfirstlane(XL[e]);     //  handled without spilling)
            const int xb = __builtin_amdgcn_readfirstlane(XL[min(e + 15, nL - 1)]);
            const int r_lo = __builtin_amdgcn_readfirstlane(RK[max(0, xa - (D - 1))]);
            const int r_hi = __builtin_amdgcn_readfirstlane(RK[xb + 1]);
            const int t0 = r_lo >> 4;
            const int ntile = r_hi > r_lo ? ((r_hi - 1) >> 4) - t0 + 1 : 0;      // <= NTC; wave-uniform (scalar branches)
            // tiles [m_lo, m_hi) hold only slots every left pixel of the chunk may match: positions >= xb - (D - 1)
            // (the last left pixel's lower bound, the tightest) and <= xa (the first one's upper bound)
            int m_lo = 0, m_hi = 0;
            if constexpr (ALL) {
                const int s_lo = RK[max(0, xb - (D - 1))], s_hi = RK[xa + 1];
                m_lo = __builtin_amdgcn_readfirstlane(((s_lo + 15) >> 4) - t0);
                m_hi = __builtin_amdgcn_readfirstlane((s_hi >> 4) - t0);
            }
            const bool act = e + j < e1;
            const int el = act ? e + j : e1 - 1;
            const int xl = XL[el];
            const size_t pix = rowpix + xl;
            float bcur[KQ];
#pragma unroll
            for (int s = 0; s < KQ; ++s) bcur[s] = act ? LF[(4 * s + q) * SP_FP_ + el] : 0.f;
            f32x4 acc[NTC];
            const float *ap = RF + q * SP_FP_ + 16 * t0 + j;
            live_tiles<0, NTC, 2>(ntile, [&](auto mc) {        // pairs of tiles: see softmax_passes
                constexpr int m = decltype(mc)::value;
                f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KQ; ++s)
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * SP_FP_ + 16 * m], bcur[s], a4, 0, 0, 0);
                acc[m] = a4;
            });
            float mx, Ssum, mu, var;
            const float mu_in = (MODE == MODE_VAR && act) ? disparity[pix] : 0.f;
            softmax_passes<NTC, MODE, ALL ? 3 : 2>(acc, ntile, D, 0, smem, 0, offXR + 16 * t0 + 4 * q, xl, mu_in, mx,
                                                   Ssum, mu, var, m_lo, m_hi);
            // results go to rows 0..3 of this chunk's own LF slots (their features were consumed
            // into bcur above and no other wave reads them); the row is written once at the end
            if (act && q == 0) {
                if (MODE != MODE_VAR) LF[el] = mu;
                if (MODE != MODE_MAT) LF[SP_FP_ + el] = var;
                LF[2 * SP_FP_ + el] = Ssum;
                LF[3 * SP_FP_ + el] = mx;
            }
        }
    }
    __syncthreads();
    // ---- 4. one coalesced sweep over the row: results of the active left pixels, zeros elsewhere
    // (functions/SpaMat.py:25-27 semantics).  Every output line is written exactly once: scattered
    // result stores behind a zero fill cost 1.65x the algorithmic write traffic (WRITE_SIZE).
    {
        int slot = baseL + il - cl;                     // exclusive count at p4
        const bool alo = ((W & 3) == 0) && ((((uintptr_t)out) | ((uintptr_t)var_out) | ((uintptr_t)sum_sim) |
                                             ((uintptr_t)max_cost)) & 15) == 0;
#pragma unroll
        for (int u = 0; u < PPT; u += 4) {
            if (p4 + u >= W) break;
            float r[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool on = fl & (1 << (u + k));
#pragma unroll
                for (int o = 0; o < 4; ++o) r[o][k] = on ? LF[o * SP_FP_ + slot] : 0.f;
                slot += on;
            }
            const size_t pix = rowpix + p4 + u;
            if (alo) {
                if (MODE != MODE_VAR) *reinterpret_cast<float4 *>(out + pix) = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]);
                if (MODE != MODE_MAT) *reinterpret_cast<float4 *>(var_out + pix) = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]);
                *reinterpret_cast<float4 *>(sum_sim + pix) = make_float4(r[2][0], r[2][1], r[2][2], r[2][3]);
                *reinterpret_cast<float4 *>(max_cost + pix) = make_float4(r[3][0], r[3][1], r[3][2], r[3][3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (p4 + u + k < W) {
                        if (MODE != MODE_VAR) out[pix + k] = r[0][k];
                        if (MODE != MODE_MAT) var_out[pix + k] = r[1][k];
                        sum_sim[pix + k] = r[2][k];
                        max_cost[pix + k] = r[3][k];
                    }
                }
            }
        }
    }
    return 1;
}

// PPT = pixels per thread of the mask scan: 4 (W <= 1024) or 8 (W <= 2048).  seg_w = pixels per
// segment of the marker launch (it splits rows wider than 1024 pixels): one marker per segment.
template <int NT, int MODE, int KQ, int PPT>
__global__ __launch_bounds__(SP_THREADS, PPT == 4 ? 6 : 5) void spamat_fwd_sparse(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity, float *__restrict__ out,
    float *__restrict__ var_out, float *__restrict__ sum_sim, float *__restrict__ max_cost, int C,
    int H, int W, int D, int seg_w, int dense_pct, int mbits, int handover) {
    const int row = blockIdx.x;
    int fr = 0, fl = 0;
    const int rc = sparse_row_body<NT, MODE, KQ, PPT, SP_THREADS, SP_CAP, 8>(ref, tar, rmask, tmask, disparity, out, var_out,
                                                                            sum_sim, max_cost, C, H, W, D, row, dense_pct, mbits,
                                                                            &fr, &fl);
    if (rc != 1) {
        // left to spamat_fwd_mfma (marker launch): a negative value at the first pixel of every segment of the row (a
        // real sum_similarities is never negative); -2: <= 512 active pixels per side, worth the 512-slot body there
        const size_t rowpix = (size_t)row * W;
        const float mark = rc == 2 ? -2.0f : -1.0f;
        for (int x = threadIdx.x * seg_w; x < W; x += SP_THREADS * seg_w) sum_sim[rowpix + x] = mark;
        // handover (round 5): this workgroup has read both float mask rows (7.8 KB at 972 pixels) to find out that the row
        // is not its; the band kernel would read them again (PMC: 1.12 x the algorithmic bytes at mid densities, 1.08 x
        // on dense rows).  The activity bits go along instead -- 2 ceil(W / 32) words at the start of the row's own
        // max_cost entries, which nobody needs before the band kernel's workgroup of this row writes its results there.
        if (handover) {
            extern __shared__ __attribute__((aligned(16))) float smem[];
            unsigned *wb = reinterpret_cast<unsigned *>(smem);
            const int nw = (W + 31) >> 5, p4 = threadIdx.x * PPT;
            __syncthreads();                              // the body's LDS arrays are dead
            for (int i = threadIdx.x; i < 2 * nw; i += SP_THREADS) wb[i] = 0u;
            __syncthreads();
            if (p4 < W) {
                if (fr) atomicOr(&wb[p4 >> 5], (unsigned)fr << (p4 & 31));
                if (fl) atomicOr(&wb[nw + (p4 >> 5)], (unsigned)fl << (p4 & 31));
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * nw; i += SP_THREADS) reinterpret_cast<unsigned *>(max_cost + rowpix)[i] = wb[i];
        }
    }
}

template <int NT, int KQ>
int launch_nt(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
              const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost,
              int B, int C, int H, int W, int D, int allow_compact, int mbits, hipStream_t stream) {
    const int xt_row = ceil_div(W, 16);
    // dense rows on the bf16 matrix cores (dense16_body) for the shipped channel counts (C = 8, 24) unless
    // DECNET_SPAMAT_DENSE=fp32 or the compaction paths are pinned off; it needs more LDS per staged position
    static const int dense_fp32 = [] { const char *e = getenv("DECNET_SPAMAT_DENSE"); return e && !strcmp(e, "fp32"); }();
    // (C = 24, three channel groups to split per staged position: 0.080 vs 0.062 ms at stage 2 -- stays on fp32 MFMA)
    const bool d16 = KQ == 2 && !dense_fp32;
    auto bytes = [&](int xt) { return (size_t)4 * make_layout(C, NT, xt, d16).total; };
    // whole row per workgroup when two workgroups (16 waves) still fit a CU's LDS; otherwise
    // equal segments that do; otherwise whatever fits once.  Segments hold <= 64 tiles so that
    // one thread covers 4 pixels of the mask scan.
    const size_t budget2 = (DECNET_LDS_BYTES - 2048) / 2, budget1 = DECNET_LDS_BYTES - 1024;
    int XT = xt_row > 64 ? ceil_div(xt_row, ceil_div(xt_row, 64)) : xt_row;
    if (bytes(XT) > budget2) {
        int segs = ceil_div(xt_row, XT) + 1;
        while (segs < xt_row && bytes(ceil_div(xt_row, segs)) > budget2) ++segs;
        int xt2 = ceil_div(xt_row, segs);
        if (bytes(xt2) <= budget2) XT = xt2;
        else {
            while (XT > 1 && bytes(XT) > budget1) --XT;
        }
    }
    XT = (XT + 1) & ~1;                               // segment starts stay 32-float aligned
    if (XT > 64) XT = 64;
    const size_t lds = bytes(XT);
    if (lds > budget1 || make_layout(C, NT, XT, d16).RP > 2048) return DECNET_ERR_UNSUPPORTED;
    const int segs = ceil_div(xt_row, XT);
    dim3 block(THREADS);
    // rows go through the compaction path when fewer than compact_pct % of their candidate pairs are active
    // (80 in round 1 -- but the dense path is faster down to ~35 %, measured)
    constexpr int compact_pct = 35;
    // ... and through the sparse-row algorithm (spamat_fwd_sparse, and the MID_CAP-slot body behind its -2 marker) below
    // sparse_pct %: with chunks of 16 active pixels that algorithm beats the dense path up to ~45 % (stage 3, density
    // 0.6 = 36 % of the pairs: 0.30 vs 0.42 ms)
    constexpr int sparse_pct = 45;
    dim3 grid((unsigned)((size_t)B * H * segs));
    // sparse rows first (KQ > 0: C <= 24; rows of <= 2048 pixels), the rest by the marker launch
    // (C = 24, stage 2: rows are short and in practice 40-100 % dense -- there the sparse-row pre-launch costs 6 us
    // of a 57 us pass and only wins below ~20 % density: C <= 8 only)
    int marker = allow_compact && KQ == 2 && W <= 2048;
    // rows of 257-512 active pixels per side: the sparse-row algorithm inside the band kernel's workgroup (whole rows
    // per workgroup only; DECNET_SPAMAT_MID=0 switches it off)
    // DECNET_SPAMAT_MID=0 switches it off (a separate 256-thread launch for these rows was measured slower: tools/experiments)
    static const bool mid_off = [] { const char *e = getenv("DECNET_SPAMAT_MID"); return e && atoi(e) == 0; }();
    size_t lds_launch = lds;
    // the sparse-row kernel hands the activity bits of the rows it leaves to the band kernel (whole rows per workgroup,
    // float masks, room for 2 ceil(W / 32) words in a row of max_cost; DECNET_SPAMAT_HANDOVER=0: the band kernel reads
    // the float planes again)
    static const bool handover_off = [] { const char *e = getenv("DECNET_SPAMAT_HANDOVER"); return e && atoi(e) == 0; }();
    const int handover = (KQ > 0 && KQ <= 6 && marker && segs == 1 && !mbits && 2 * ((W + 31) / 32) <= W && !handover_off) ? 1 : 0;
    if (KQ == 2 && marker && segs == 1 && !mid_off) {
        const size_t need = 4 * sparse_row_words(KQ, 4, THREADS, MID_CAP);
        if (need <= budget2 + 8192) {
            marker = 2;
            if (need > lds_launch) lds_launch = need;
        }
    }
    if constexpr (KQ > 0 && KQ <= 6) if (marker) {
        const int ppt = W <= 1024 ? 4 : 8;
        const size_t slds = 4 * sparse_row_words(KQ, ppt, SP_THREADS, SP_CAP);
#define LAUNCHS(M, P)                                                                              \
    do {                                                                                           \
        if (slds > 64 * 1024) {                                                                    \
            hipError_t e = hipFuncSetAttribute((const void *)spamat_fwd_sparse<NT, M, (KQ ? KQ : 1), P>, \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds); \
            if (e != hipSuccess) return (int)e;                                                    \
        }                                                                                          \
        hipLaunchKernelGGL((spamat_fwd_sparse<NT, M, (KQ ? KQ : 1), P>), dim3((unsigned)(B * H)),  \
                           dim3(SP_THREADS), slds, stream, ref, tar, rmask, tmask, disparity, out, \
                           var_out, sum_sim, max_cost, C, H, W, D, XT * 16, sparse_pct, mbits, handover); \
    } while (0)
#define LAUNCHSP(M)                                                                                \
    do {                                                                                           \
        if (ppt == 4) LAUNCHS(M, 4);                                                               \
        else LAUNCHS(M, 8);                                                                        \
    } while (0)
        if (mode == MODE_MAT) LAUNCHSP(MODE_MAT);
        else if (mode == MODE_VAR) LAUNCHSP(MODE_VAR);
        else LAUNCHSP(MODE_FUSED);
#undef LAUNCHSP
#undef LAUNCHS
        int rc = decnet_launch_status();
        if (rc) return rc;
    }
#define LAUNCH1(M, DD)                                                                             \
    do {                                                                                           \
        if (lds_launch > 64 * 1024) {                                                              \
            hipError_t e = hipFuncSetAttribute((const void *)spamat_fwd_mfma<NT, M, KQ, DD>,       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch); \
            if (e != hipSuccess) return (int)e;                                                    \
        }                                                                                          \
        hipLaunchKernelGGL((spamat_fwd_mfma<NT, M, KQ, DD>), grid, block, lds_launch, stream, ref, tar, rmask, \
                           tmask, disparity, out, var_out, sum_sim, max_cost, C, H, W, D, segs, XT, \
                           allow_compact, marker, compact_pct, handover ? 2 : mbits);              \
    } while (0)
#define LAUNCH(M)                                                                                  \
    do {                                                                                           \
        if constexpr (KQ == 2) {                                                                   \
            if (d16) LAUNCH1(M, true);                                                             \
            else LAUNCH1(M, false);                                                                \
        } else {                                                                                   \
            LAUNCH1(M, false);                                                                     \
        }                                                                                          \
    } while (0)
    if (mode == MODE_MAT) LAUNCH(MODE_MAT);
    else if (mode == MODE_VAR) LAUNCH(MODE_VAR);
    else LAUNCH(MODE_FUSED);
#undef LAUNCH1
#undef LAUNCH
    return decnet_launch_status();
}

template <int NT>
int launch_kq(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
              const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost,
              int B, int C, int H, int W, int D, int allow_compact, int mbits, hipStream_t stream) {
#define GO(K)                                                                                      \
    return launch_nt<NT, K>(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim,        \
                            max_cost, B, C, H, W, D, allow_compact, mbits, stream)
    if (C <= 8 && C > 4) GO(2);        // stage 3 of the shipped network (C = 8)
    if (C <= 24 && C > 20) GO(6);      // stage 2 (C = 24)
    if (C <= 72 && C > 68) GO(18);     // stage 1 (C = 72)
    GO(0);                             // anything else: runtime K loop, left operand read straight
                                       // from L2/HBM per K-step
#undef GO
}

}  // namespace

// mode: 0 SpaMat, 1 SpaVar, 2 fused.  Returns DECNET_ERR_UNSUPPORTED when the band needs more
// than 18 tiles (max_disp > 272) or a tile does not fit LDS; the dispatcher in capi.hip then
// uses the row-tile kernel.  allow_compact = 0 pins the dense path (A/B benchmarks, tests).
// mbits = 1: rmask / tmask point at bit-packed masks ([B,H,ceil(W/64)] 64-bit words, decnet_detail_mask's layout).
int decnet_mfma_forward(int mode, const float *ref, const float *tar, const float *rmask,
                        const float *tmask, const float *disparity, float *out, float *var_out,
                        float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                        int allow_compact, int mbits, hipStream_t stream) {
    const int D = max_disp;
    const int need = D <= 1 ? 1 : (D - 1 + 15) / 16 + 1;
#define GO(N)                                                                                     \
    return launch_kq<N>(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim, max_cost, \
                        B, C, H, W, D, allow_compact, mbits, stream)
    if (need <= 3) GO(3);       // D <= 32   (stage 1: 24, 30)
    if (need <= 6) GO(6);       // D <= 80   (stage 2: 72)
    if (need <= 8) GO(8);       // D <= 112  (stage 2 at max_disp 270: 90)
    if (need <= 11) GO(11);     // D <= 160
    if (need <= 15) GO(15);     // D <= 224  (stage 3: 216)
    if (need <= 18) GO(18);     // D <= 272  (stage 3 at max_disp 270)
    return DECNET_ERR_UNSUPPORTED;
#undef GO
}
