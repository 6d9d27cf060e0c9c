// decnet_amd/csrc/spamat_bwd_mfma.hip -- SpaMat / SpaVar backward on the matrix cores (gfx950).
//
// Replaces sparse_matching_ref_backward / sparse_matching_tar_backward (SM_kernel.cu:143-195,
// 300-355) and sparse_var_{ref,tar,dis}_backward (SV_kernel.cu:142-325).  The reference runs one
// thread per (b,c,y,x) and recomputes the whole C-channel dot product for every disparity in
// every one of those C threads (O(C^2) per candidate).  Here a candidate's cost
// and softmax weight are formed ONCE (cost tile by v_mfma_f32_16x16x4_f32, same c-ordered fp32
// fma chain as the forward), and the weighted sum over disparities
//     gL[c][x]  = g/S * sum_d  w(x,d)   * R[c][x-d]          w = e*(d-out)   (SpaMat)
//     gR[c][x'] =       sum_d  w(x'+d,d)* L[c][x'+d] * g/S   w = e*((d-mu)^2-out)  (SpaVar)
// is a second small matrix product contracted over the band -- also on the matrix cores, with the
// weight tile used as an MFMA operand straight from the accumulator registers (step r of the
// K loop takes accumulator register r: lane quad q supplies row 4q+r, so no transpose and no
// LDS round trip).  The max of the forward is an input (max_cost), so no cost is ever stored:
// each 16x16 tile is cost -> weight -> contraction and then dropped.
//
// SIDE 0 ("ref"): lanes own LEFT pixels, the staged "other" row is R with a left halo.
// SIDE 1 ("tar"): lanes own RIGHT pixels, the staged "other" row is L with a right halo, and the
//                 per-left-pixel scalars (max, out, g/S, mu, mask) are rows of the tile: LDS planes.
// LDS: Os[Cq][OP] (OP == 4 mod 64: the contraction reads 4 consecutive pixels of one channel per
//      lane with ds_read_b128, 16 channels x 4 quads conflict-free) | planes[NPL][OW].
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int THREADS = 512, NWAVE = THREADS / 64;
constexpr int BWD_MARK = 0x7fc0dec0;     // a NaN payload no gradient takes: "row left to the band kernel"

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
// Round 5, from the ISA: several load4() calls in a row are SERIAL memory round trips -- both sides of its per-lane branch
// write the same registers and the compiler puts `s_waitcnt vmcnt(0)` in front of every wide load (80 of the 81 wide loads
// of the C = 72 instantiation).  Where a thread has a batch of them, the batch is issued as raw 16-byte loads from clamped
// addresses behind a WORKGROUP-UNIFORM flag (rows on 16-byte boundaries, W a multiple of 4: a group of four is inside or
// outside its row as a whole, x being a multiple of 4), a scheduling barrier, and the zeros are selected afterwards.
__device__ __forceinline__ bool uniform_flag(bool f) { return __builtin_amdgcn_readfirstlane((int)f) != 0; }
__device__ __forceinline__ float4 load4_raw(const float *__restrict__ safe, const float *__restrict__ at, bool ok) {
    return *reinterpret_cast<const float4 *>(ok ? at : safe);
}
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

struct BLayout {
    int SW, HALO, OW, OP, Cq, offP, total;
};
// cq: channel rows staged = 4 * KQ of the instantiation (>= C): the cost MFMAs read all of them (against zero own
// operands beyond C), so the rows between C and 4 KQ must exist and hold zeros, not whatever LDS held before
// nown: planes of per-OWN-pixel scalars behind the other side's planes (SW entries each)
__host__ __device__ inline BLayout make_blayout(int cq, int NT, int XT, int npl, int nown) {
    BLayout l;
    l.SW = XT * 16;
    l.HALO = (NT - 1) * 16;
    l.OW = l.HALO + l.SW;                       // staged other-side pixels (multiple of 16)
    l.OP = ((l.OW + 63) & ~63) + 4;
    l.Cq = cq;
    l.offP = l.Cq * l.OP;
    l.total = l.offP + npl * l.OW + nown * l.SW;
    return l;
}
// own planes.  SIDE 0 (lanes own left pixels): 0 -max*log2e (-1e30 where the left mask is off: every weight 0), 1 out,
// 2 g/S (0 where the mask is off), 3 mu (SpaVar).  SIDE 1 (lanes own right pixels): 0 the right mask.
// Round 6, from the ISA: these scalars were per-lane global loads inside the tile loop -- own mask, max, out in front of the
// band tiles (one exposed round trip per wave-tile) and mask / g / S again for the four pixels of the epilogue, each pair
// behind its own `s_waitcnt vmcnt(0)`: eight serial round trips per wave-tile, more than the tile's arithmetic at stage 2.
__host__ __device__ constexpr int bwd_nown(int side, bool var) { return side == 0 ? (var ? 4 : 3) : 1; }

// planes: 0 bias (0 / -1e30 of the other side's mask); SIDE 1 adds 1 -max*log2e, 2 out, 3 g/S, 4 mu
template <int NT, bool VAR, int KQ, int SIDE>
__device__ __forceinline__ void bwd_band_side(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity,
    const float *__restrict__ out, const float *__restrict__ sum_sim,
    const float *__restrict__ max_cost, const float *__restrict__ grad_out,
    float *__restrict__ grad_own, float *__restrict__ grad_disp, int C, int H, int W, int D,
    int segs_per_row, int XT, int marker, const int blk) {
    // marker: this launch follows spamat_bwd_sparse, which left BWD_MARK in channel 0 of grad_own at
    // the first pixel of every segment of exactly the rows it did not take
    if (marker) {
        const int sg = blk % segs_per_row, rw = blk / segs_per_row;
        const size_t at = ((size_t)(rw / H) * C * H + (rw % H)) * W + (size_t)sg * (XT * 16);
        if (__float_as_int(grad_own[at]) != BWD_MARK) return;
    }
    constexpr int NPL = SIDE == 0 ? 1 : (VAR ? 5 : 4), NOWN = bwd_nown(SIDE, VAR);
    constexpr int NCB = (4 * KQ + 15) / 16;          // 16-channel blocks of the contraction
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BLayout lo = make_blayout(4 * KQ, NT, XT, NPL, NOWN);
    float *Os = smem;
    float *PL = smem + lo.offP;
    float *OWNP = PL + NPL * lo.OW;                  // [NOWN][SW]
    const int SW = lo.SW, HALO = lo.HALO, OW = lo.OW, OP = lo.OP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blk % segs_per_row, row = blk / segs_per_row;
    const int b = row / H, y = row - b * H;
    const int xs = seg * SW;
    const int xo0 = SIDE == 0 ? xs - HALO : xs;      // image x of staged column 0
    const size_t plane = (size_t)H * W;
    const size_t rowpix = (size_t)row * W;
    const float *own_row = (SIDE == 0 ? ref : tar) + ((size_t)b * C * H + y) * W;
    const float *oth_row = (SIDE == 0 ? tar : ref) + ((size_t)b * C * H + y) * W;
    const float *oth_mask = (SIDE == 0 ? tmask : rmask) + rowpix;

    // ---- stage the other side's features and per-pixel planes ----------------------------------
    {
        const bool al = ((((uintptr_t)oth_row) | ((uintptr_t)(plane * 4))) & 15) == 0;
        const bool alp = (rowpix & 3) == 0 && ((((uintptr_t)oth_mask) | ((uintptr_t)out) |
                                                ((uintptr_t)sum_sim) | ((uintptr_t)max_cost) |
                                                ((uintptr_t)grad_out)) & 15) == 0;
        const bool alu = uniform_flag(al && (W & 3) == 0);      // batches of raw loads, see load4_raw
        const bool alpu = uniform_flag(alp && (W & 3) == 0 && (!VAR || (((uintptr_t)disparity) & 15) == 0));
        // features: threads spread over (channel row, group of 4 positions), up to 8 loads in flight each
        // (stage 1, C = 72 over 144 positions: 6 loads per thread instead of 72 serial ones on 36 threads)
        const int nq = OW >> 2;
        if (nq <= THREADS) {
            const int rpp = THREADS / nq, r0 = tid / nq, jq = tid - r0 * nq;
            if (r0 < rpp) {
                const int jj = 4 * jq, x = xo0 + jj;
                const bool inx = x >= 0 && x < W;
                for (int c0 = r0; c0 < lo.Cq; c0 += 8 * rpp) {
                    float4 v[8];
                    if (alu) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int c = c0 + u * rpp;
                            v[u] = load4_raw(oth_row, oth_row + (size_t)c * plane + x, inx && c < C);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = sel4(inx && c0 + u * rpp < C, v[u]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int c = c0 + u * rpp;
                            v[u] = c < C ? load4(oth_row + (size_t)c * plane, x, W, al) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * rpp;
                        if (c < lo.Cq) *reinterpret_cast<float4 *>(Os + c * OP + jj) = v[u];
                    }
                }
            }
        }
        for (int j = tid * 4; j < OW; j += THREADS * 4) {
            const int x = xo0 + j;
            if (nq > THREADS) {
                for (int c0 = 0; c0 < lo.Cq; c0 += 8) {
                    float4 v[8];
                    if (alu) {
                        const bool inx = x >= 0 && x < W;
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            v[c] = load4_raw(oth_row, oth_row + (size_t)(c0 + c) * plane + x, inx && c0 + c < C);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = sel4(inx && c0 + c < C, v[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            v[c] = c0 + c < C ? load4(oth_row + (size_t)(c0 + c) * plane, x, W, al)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c0 + c < lo.Cq) *reinterpret_cast<float4 *>(Os + (c0 + c) * OP + j) = v[c];
                }
            }
            float4 mk, mx, oo, gg, ss, dd;
            if (alpu) {
                const bool inx = x >= 0 && x < W;
                mk = load4_raw(oth_mask, oth_mask + x, inx);
                if (SIDE == 1) {
                    mx = load4_raw(oth_mask, max_cost + rowpix + x, inx);
                    oo = load4_raw(oth_mask, out + rowpix + x, inx);
                    gg = load4_raw(oth_mask, grad_out + rowpix + x, inx);
                    ss = load4_raw(oth_mask, sum_sim + rowpix + x, inx);
                    if (VAR) dd = load4_raw(oth_mask, disparity + rowpix + x, inx);
                }
                __builtin_amdgcn_sched_barrier(0);
                mk = sel4(inx, mk);
                if (SIDE == 1) {
                    mx = sel4(inx, mx); oo = sel4(inx, oo); gg = sel4(inx, gg); ss = sel4(inx, ss);
                    if (VAR) dd = sel4(inx, dd);
                }
            } else {
                mk = load4(oth_mask, x, W, alp);
                if (SIDE == 1) {
                    mx = load4(max_cost + rowpix, x, W, alp);
                    oo = load4(out + rowpix, x, W, alp);
                    gg = load4(grad_out + rowpix, x, W, alp);
                    ss = load4(sum_sim + rowpix, x, W, alp);
                    if (VAR) dd = load4(disparity + rowpix, x, W, alp);
                }
            }
            const bool on[4] = {x >= 0 && x < W && mk.x != 0.f, x + 1 >= 0 && x + 1 < W && mk.y != 0.f,
                                x + 2 >= 0 && x + 2 < W && mk.z != 0.f, x + 3 >= 0 && x + 3 < W && mk.w != 0.f};
            float4 bz = make_float4(on[0] ? 0.f : NEG_BIG, on[1] ? 0.f : NEG_BIG, on[2] ? 0.f : NEG_BIG,
                                    on[3] ? 0.f : NEG_BIG);
            *reinterpret_cast<float4 *>(PL + j) = bz;
            if (SIDE == 1) {
                *reinterpret_cast<float4 *>(PL + OW + j) =
                    make_float4(-mx.x * LOG2E, -mx.y * LOG2E, -mx.z * LOG2E, -mx.w * LOG2E);
                *reinterpret_cast<float4 *>(PL + 2 * OW + j) = oo;
                *reinterpret_cast<float4 *>(PL + 3 * OW + j) =          // g/S, 0 where the left mask is off
                    make_float4(on[0] ? gg.x / ss.x : 0.f, on[1] ? gg.y / ss.y : 0.f,
                                on[2] ? gg.z / ss.z : 0.f, on[3] ? gg.w / ss.w : 0.f);
                if (VAR) *reinterpret_cast<float4 *>(PL + 4 * OW + j) = dd;
            }
        }
        // the own pixels' scalars (see bwd_nown)
        const float *own_mask = (SIDE == 0 ? rmask : tmask) + rowpix;
        const bool alo = uniform_flag(alp && (W & 3) == 0 && (((uintptr_t)own_mask) & 15) == 0 &&
                                      (!VAR || (((uintptr_t)disparity) & 15) == 0));
        for (int j = tid * 4; j < SW; j += THREADS * 4) {
            const int x = xs + j;
            float4 mk, mx, oo, gg, ss, dd;
            if (alo) {
                const bool inx = x < W;
                mk = load4_raw(own_mask, own_mask + x, inx);
                if (SIDE == 0) {
                    mx = load4_raw(own_mask, max_cost + rowpix + x, inx);
                    oo = load4_raw(own_mask, out + rowpix + x, inx);
                    gg = load4_raw(own_mask, grad_out + rowpix + x, inx);
                    ss = load4_raw(own_mask, sum_sim + rowpix + x, inx);
                    if (VAR) dd = load4_raw(own_mask, disparity + rowpix + x, inx);
                }
                __builtin_amdgcn_sched_barrier(0);
                mk = sel4(inx, mk);
            } else {
                mk = load4(own_mask, x, W, false);
                if (SIDE == 0) {
                    mx = load4(max_cost + rowpix, x, W, false);
                    oo = load4(out + rowpix, x, W, false);
                    gg = load4(grad_out + rowpix, x, W, false);
                    ss = load4(sum_sim + rowpix, x, W, false);
                    if (VAR) dd = load4(disparity + rowpix, x, W, false);
                }
            }
            // (mk is 0 outside the row on both paths; the other planes are only used where mk != 0)
            if (SIDE == 0) {
                *reinterpret_cast<float4 *>(OWNP + j) =
                    make_float4(mk.x != 0.f ? -mx.x * LOG2E : NEG_BIG, mk.y != 0.f ? -mx.y * LOG2E : NEG_BIG,
                                mk.z != 0.f ? -mx.z * LOG2E : NEG_BIG, mk.w != 0.f ? -mx.w * LOG2E : NEG_BIG);
                *reinterpret_cast<float4 *>(OWNP + SW + j) = oo;
                *reinterpret_cast<float4 *>(OWNP + 2 * SW + j) =
                    make_float4(mk.x != 0.f ? gg.x / ss.x : 0.f, mk.y != 0.f ? gg.y / ss.y : 0.f,
                                mk.z != 0.f ? gg.z / ss.z : 0.f, mk.w != 0.f ? gg.w / ss.w : 0.f);
                if (VAR) *reinterpret_cast<float4 *>(OWNP + 3 * SW + j) = dd;
            } else {
                *reinterpret_cast<float4 *>(OWNP + j) = mk;
            }
        }
    }
    __syncthreads();

    const int j = lane & 15, q = lane >> 4;
    const int cj = j < lo.Cq ? j : lo.Cq - 1;        // channel this lane supplies to the contraction
    const bool al4 = (W & 3) == 0 && (((uintptr_t)grad_own) & 15) == 0;     // 16-byte stores of four pixels of a channel
    // the own features of the NEXT tile of this wave: requested unconditionally (clamped tile, pixel and channel) and
    // selected when they are used -- a masked load merged with a zero is waited for on the spot (see spamat_bwd_rowb)
    float bv[KQ], bcur[KQ];
    auto fetch_own = [&](int xt, float (&dst)[KQ]) {
        const int x = min(xs + min(xt, XT - 1) * 16 + j, W - 1);
#pragma unroll
        for (int s = 0; s < KQ; ++s) dst[s] = own_row[(size_t)min(4 * s + q, C - 1) * plane + x];
    };
    fetch_own(wave, bv);

    for (int xt = wave; xt < XT; xt += NWAVE) {
        const int x0 = xs + xt * 16;
        if (x0 >= W) break;
        const int x = x0 + j;
        const bool inside = x < W;
#pragma unroll
        for (int s = 0; s < KQ; ++s) bcur[s] = (inside && 4 * s + q < C) ? bv[s] : 0.f;
        fetch_own(xt + NWAVE, bv);
        // own-pixel scalars out of the own planes.  A masked-off own pixel takes part in no candidate (its costs were
        // never bounded by the forward's max, so exp could overflow): all its weights are forced to 0 -- SIDE 0 through
        // its -1e30 exponent offset, SIDE 1 by the select below
        float nm_own = 0.f, out_own = 0.f, mu_own = 0.f;
        bool own_on = true;
        if (SIDE == 0) {
            nm_own = OWNP[xt * 16 + j];
            out_own = OWNP[SW + xt * 16 + j];
            if (VAR) mu_own = OWNP[3 * SW + xt * 16 + j];
        } else {
            own_on = OWNP[xt * 16 + j] != 0.f;
        }
        f32x4 gacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) gacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        float gdis = 0.f;

#pragma unroll 1
        for (int m = 0; m < NT; ++m) {
            const int ob = SIDE == 0 ? HALO + xt * 16 - 16 * m : xt * 16 + 16 * m;   // other base
            // cost tile: rows = other pixels ob + 4q + r, columns (lanes) = own pixels
            const float *ap = Os + q * OP + ob + j;
            f32x4 cst = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bcur[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KQ; ++s)
                cst = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * OP], bcur[s], cst, 0, 0, 0);
            const float4 bz = *reinterpret_cast<const float4 *>(PL + ob + 4 * q);
            float4 nmr, outr, gsr, mur;
            if (SIDE == 1) {
                nmr = *reinterpret_cast<const float4 *>(PL + OW + ob + 4 * q);
                outr = *reinterpret_cast<const float4 *>(PL + 2 * OW + ob + 4 * q);
                gsr = *reinterpret_cast<const float4 *>(PL + 3 * OW + ob + 4 * q);
                if (VAR) mur = *reinterpret_cast<const float4 *>(PL + 4 * OW + ob + 4 * q);
            }
            const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
            f32x4 wt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = SIDE == 0 ? 16 * m + j - (4 * q + r) : 16 * m + (4 * q + r) - j;
                const float nm = SIDE == 0 ? nm_own : (r == 0 ? nmr.x : r == 1 ? nmr.y : r == 2 ? nmr.z : nmr.w);
                float cc = cst[r] + bzv[r];
                cc = ((unsigned)d < (unsigned)D && own_on) ? cc : NEG_BIG;
                const float e = __builtin_amdgcn_exp2f(fmaf(cc, LOG2E, nm));
                const float df = (float)d;
                float w;
                if (SIDE == 0) {
                    if (VAR) {
                        const float dd = df - mu_own;
                        w = e * fmaf(dd, dd, -out_own);        // SV_kernel.cu:191
                        gdis = fmaf(e, dd, gdis);              // SV_kernel.cu:321
                    } else {
                        w = e * (df - out_own);                // SM_kernel.cu:191
                    }
                } else {
                    const float o = r == 0 ? outr.x : r == 1 ? outr.y : r == 2 ? outr.z : outr.w;
                    const float gs = r == 0 ? gsr.x : r == 1 ? gsr.y : r == 2 ? gsr.z : gsr.w;
                    if (VAR) {
                        const float mu = r == 0 ? mur.x : r == 1 ? mur.y : r == 2 ? mur.z : mur.w;
                        const float dd = df - mu;
                        w = gs * e * fmaf(dd, dd, -o);         // SV_kernel.cu:262
                    } else {
                        w = gs * e * (df - o);                 // SM_kernel.cu:346
                    }
                }
                wt[r] = w;
            }
            // contraction over the 16 other pixels of the tile: K step r uses weight register r
            // (lane quad q <-> other pixel 4q + r) against Other[c][ob + 4q + r]
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int c = 16 * cb + cj < lo.Cq ? 16 * cb + cj : lo.Cq - 1;
                const float4 ov = *reinterpret_cast<const float4 *>(Os + c * OP + ob + 4 * q);
                gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0], ov.x, gacc[cb], 0, 0, 0);
                gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[1], ov.y, gacc[cb], 0, 0, 0);
                gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[2], ov.z, gacc[cb], 0, 0, 0);
                gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[3], ov.w, gacc[cb], 0, 0, 0);
            }
        }

        // gacc[cb][r]: channel c = 16*cb + (lane & 15), own pixel x0 + 4q + r
        const int xr = x0 + 4 * q;
        // grad_ref = g * sum / S (SM_kernel.cu:193); masked-off pixels stay 0 (SpaMat.py:42)
        const float4 s4 = *reinterpret_cast<const float4 *>(OWNP + (SIDE == 0 ? 2 * SW : 0) + xt * 16 + 4 * q);
        const float sc[4] = {SIDE == 0 ? s4.x : (s4.x != 0.f ? 1.f : 0.f), SIDE == 0 ? s4.y : (s4.y != 0.f ? 1.f : 0.f),
                             SIDE == 0 ? s4.z : (s4.z != 0.f ? 1.f : 0.f), SIDE == 0 ? s4.w : (s4.w != 0.f ? 1.f : 0.f)};
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int c = 16 * cb + j;
            if (c < C) {
                float *gp = grad_own + ((size_t)b * C + c) * plane + (size_t)y * W + xr;
                if (al4 && xr + 3 < W) {
                    *reinterpret_cast<float4 *>(gp) = make_float4(gacc[cb][0] * sc[0], gacc[cb][1] * sc[1],
                                                                  gacc[cb][2] * sc[2], gacc[cb][3] * sc[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (xr + r < W) gp[r] = gacc[cb][r] * sc[r];
                }
            }
        }
        if (SIDE == 0 && VAR) {
            gdis += __shfl_xor(gdis, 16);
            gdis += __shfl_xor(gdis, 32);
            // SV_kernel.cu:321-325: -2 g sum e (d - mu) / S; the own plane holds g / S (0 where the mask is off)
            if (inside && q == 0) grad_disp[rowpix + x] = -2.f * OWNP[2 * SW + xt * 16 + j] * gdis;
        }
    }
}

// Both sides from one launch (round 6): workgroups [0, n0) own left pixels (grad_ref, grad_disp), the rest right pixels
// (grad_tar).  As two launches each side left most of the chip idle at the small stages (stage 1: 240 workgroups of one
// latency chain each, 15 us + 15 us back to back); together they overlap.
template <int NT, bool VAR, int KQ>
__global__ __launch_bounds__(THREADS, 4) void spamat_bwd_mfma(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity,
    const float *__restrict__ out, const float *__restrict__ sum_sim,
    const float *__restrict__ max_cost, const float *__restrict__ grad_out,
    float *__restrict__ grad_ref, float *__restrict__ grad_tar, float *__restrict__ grad_disp, int C, int H, int W,
    int D, int n0, int segs0, int XT0, int segs1, int XT1, int marker) {
    const int blk = blockIdx.x;
    if (blk < n0)
        bwd_band_side<NT, VAR, KQ, 0>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref,
                                      grad_disp, C, H, W, D, segs0, XT0, marker, blk);
    else
        bwd_band_side<NT, VAR, KQ, 1>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_tar,
                                      nullptr, C, H, W, D, segs1, XT1, marker, blk - n0);
}

// tiles per segment of the band kernel for one side (0 = does not fit)
template <int NT, bool VAR, int SIDE>
int side_xt(int cq, int W) {
    constexpr int NPL = SIDE == 0 ? 1 : (VAR ? 5 : 4);
    const int xt_row = ceil_div(W, 16);
    auto bytes = [&](int xt) { return (size_t)4 * make_blayout(cq, NT, xt, NPL, bwd_nown(SIDE, VAR)).total; };
    const size_t budget2 = (DECNET_LDS_BYTES - 2048) / 2, budget1 = DECNET_LDS_BYTES - 1024;
    int XT = xt_row;
    if (bytes(XT) > budget2) {
        int segs = 2;
        while (segs < xt_row && bytes(ceil_div(xt_row, segs)) > budget2) ++segs;
        int xt2 = ceil_div(xt_row, segs);
        if (bytes(xt2) <= budget2) XT = xt2;
        else
            while (XT > 1 && bytes(XT) > budget1) --XT;
    }
    XT = (XT + 3) & ~3;                                  // segment starts stay 64-float aligned
    return bytes(XT) > budget1 ? 0 : XT;
}

template <int NT, bool VAR, int KQ>
int launch_sides(const float *ref, const float *tar, const float *rmask, const float *tmask,
                 const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
                 const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                 int W, int D, int XT0, int XT1, int marker, hipStream_t stream) {
    const size_t lds0 = (size_t)4 * make_blayout(4 * KQ, NT, XT0, 1, bwd_nown(0, VAR)).total;
    const size_t lds1 = (size_t)4 * make_blayout(4 * KQ, NT, XT1, VAR ? 5 : 4, bwd_nown(1, VAR)).total;
    const size_t lds = lds0 > lds1 ? lds0 : lds1;
    const int segs0 = ceil_div(ceil_div(W, 16), XT0), segs1 = ceil_div(ceil_div(W, 16), XT1);
    const size_t n0 = (size_t)B * H * segs0, n1 = (size_t)B * H * segs1;
    if (n0 + n1 >= 2147483648ull) return DECNET_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)spamat_bwd_mfma<NT, VAR, KQ>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((spamat_bwd_mfma<NT, VAR, KQ>), dim3((unsigned)(n0 + n1)), dim3(THREADS), lds, stream, ref,
                       tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar, grad_disp, C,
                       H, W, D, (int)n0, segs0, XT0, segs1, XT1, marker);
    return decnet_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Sparse rows (<= 256 active pixels on each side): as spamat_mfma.hip:spamat_fwd_sparse, nothing but
// the masks, the features of the ACTIVE pixels and the per-left-pixel scalars is read, and both
// gradients of a row come out of one workgroup:
//   1. mask rows -> bits, exclusive counts (RK / RKL), index lists XR / XL
//   2. gather RF / LF [C][slot] and NM = -max*log2e, OUT, GS = g/S (, MU) per active left pixel;
//      zero-fill the gradients of the inactive pixels meanwhile
//   3. gL: chunks of 16 consecutive ACTIVE left pixels x 16-wide tiles of the compacted right list
//      inside their disparity window: cost tile -> weight -> contraction, exactly the arithmetic of
//      spamat_bwd_mfma; then gR with the roles swapped (and grad_disparity for SpaVar).
// Nothing is kept per tile, so there is no limit on the number of tiles of a window.  Rows with more
// active pixels are left to the band kernels: BWD_MARK is written into channel 0 of both gradients
// at the first pixel of each of their segments (seg_w0 / seg_w1 pixels wide).
// Two instantiations: (CAP 256, 256 threads: up to six workgroups per CU, the sparse regime) and, round 3, (CAP 640, 512
// threads, MID: only the rows the first one marked) for the rows of 257 - 640 active pixels per side, which before fell onto
// the two band launches at the cost of dense rows (stage 3, density 0.5: 0.74 ms, the same as density 1.0).
constexpr int SB_THREADS = 256, SB_CAP = 256, SBM_THREADS = 512, SBM_CAP = 640;
__host__ __device__ constexpr int sb_lp(int cap) { return cap + 16; }
__host__ __device__ constexpr int sb_fp(int cap) { return ((cap + 16 + 63) & ~63) + 4; }      // feature pitch == 4 (mod 64)
__host__ __device__ constexpr size_t sb_words(int kq, int ppt, int cap, int nthr) {
    return (size_t)2 * sb_lp(cap) + 2 * (size_t)(nthr * ppt / 2 + 2) + 16 + 4 * (size_t)sb_lp(cap) +
           2 * (size_t)4 * kq * sb_fp(cap) + (size_t)(nthr / 64) * (4 * kq + 1) * 16;
}

__device__ __forceinline__ int wave_incl_scan_b(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

template <bool VAR, int KQ, int PPT, int CAP = SB_CAP, int NTHR = SB_THREADS, bool MID = false>
__global__ __launch_bounds__(NTHR, (NTHR == 256 ? 4 : 2)) void spamat_bwd_sparse(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ disparity,
    const float *__restrict__ out, const float *__restrict__ sum_sim,
    const float *__restrict__ max_cost, const float *__restrict__ grad_out,
    float *__restrict__ grad_ref, float *__restrict__ grad_tar, float *__restrict__ grad_disp, int C,
    int H, int W, int D, int seg_w0, int seg_w1) {
    constexpr int CQ = 4 * KQ, NCB = (CQ + 15) / 16, NPX = NTHR * PPT, RKW = NPX / 2 + 2;
    constexpr int SB_NWAVE = NTHR / 64, SB_LP = sb_lp(CAP), SB_FP = sb_fp(CAP), SB_CAP = CAP, SB_THREADS = NTHR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // words: XR [LP] | XL [LP] | RK, RKL [(NPX+4) x u16] | WT [16] | NM, OUT, GS, MU [LP] | RF, LF [CQ][FP]
    constexpr int offXR = 0, offXL = SB_LP, offRK = 2 * SB_LP, offRKL = offRK + RKW, offWT = offRKL + RKW,
                  offNM = offWT + 16, offOUT = offNM + SB_LP, offGS = offOUT + SB_LP, offMU = offGS + SB_LP,
                  offRF = offMU + SB_LP, offLF = offRF + CQ * SB_FP, offSC = offLF + CQ * SB_FP,
                  SCW = (CQ + 1) * 16;                   // per wave: [CQ gradient channels + grad_disparity][16 slots]
    int *XR = reinterpret_cast<int *>(smem) + offXR;
    int *XL = reinterpret_cast<int *>(smem) + offXL;
    unsigned short *RK = reinterpret_cast<unsigned short *>(smem + offRK);
    unsigned short *RKL = reinterpret_cast<unsigned short *>(smem + offRKL);
    int *WT = reinterpret_cast<int *>(smem) + offWT;
    float *NM = smem + offNM, *OUT = smem + offOUT, *GS = smem + offGS, *MU = smem + offMU;
    float *RF = smem + offRF, *LF = smem + offLF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *SC = smem + offSC + wave * SCW;
    const int row = blockIdx.x, b = row / H, y = row - b * H;
    if (MID) {      // only the rows the 256-slot launch left (BWD_MARK at the row's first pixel, channel 0 of grad_ref)
        if (__float_as_int(grad_ref[((size_t)b * C * H + y) * W]) != BWD_MARK) return;
    }
    const size_t plane = (size_t)H * W, rowpix = (size_t)row * W;
    const float *lrow = ref + ((size_t)b * C * H + y) * W;
    const float *rrow = tar + ((size_t)b * C * H + y) * W;
    float *glrow = grad_ref + ((size_t)b * C * H + y) * W;
    float *grrow = grad_tar + ((size_t)b * C * H + y) * W;
    const float *trow = tmask + rowpix, *mrow = rmask + rowpix;

    // ---- 1. masks -> bits, counts
    const int p0 = tid * PPT;
    int fr = 0, fl = 0;
    {
        const bool alm = ((W & 3) == 0) && ((((uintptr_t)trow) | ((uintptr_t)mrow)) & 15) == 0;
#pragma unroll
        for (int u = 0; u < PPT; u += 4) {
            if (p0 + u < W) {
                const float4 tv = load4(trow, p0 + u, W, alm), mv = load4(mrow, p0 + u, W, alm);
                fr |= ((tv.x != 0.f) | ((tv.y != 0.f) << 1) | ((tv.z != 0.f) << 2) | ((tv.w != 0.f) << 3)) << u;
                fl |= ((mv.x != 0.f) | ((mv.y != 0.f) << 1) | ((mv.z != 0.f) << 2) | ((mv.w != 0.f) << 3)) << u;
            }
        }
    }
    const int cr = __popc(fr), cl = __popc(fl);
    const int ir = wave_incl_scan_b(cr, lane), il = wave_incl_scan_b(cl, lane);
    if (lane == 63) { WT[wave] = ir; WT[8 + wave] = il; }
    __syncthreads();
    int nR = 0, nL = 0, baseR = 0, baseL = 0;
#pragma unroll
    for (int w = 0; w < SB_NWAVE; ++w) {
        if (w < wave) { baseR += WT[w]; baseL += WT[8 + w]; }
        nR += WT[w];
        nL += WT[8 + w];
    }
    if (nL > SB_CAP || nR > SB_CAP) {                  // left to the band kernels (marker launches)
        if (!MID) {                                     // (MID: the marks of the first launch stay)
            for (int x = tid * seg_w0; x < W; x += SB_THREADS * seg_w0) glrow[x] = __int_as_float(BWD_MARK);
            for (int x = tid * seg_w1; x < W; x += SB_THREADS * seg_w1) grrow[x] = __int_as_float(BWD_MARK);
        }
        return;
    }
    {
        int er = baseR + ir - cr, el = baseL + il - cl; // exclusive counts at p0
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            RK[p0 + k] = er;
            if (fr & (1 << k)) XR[er++] = p0 + k;
            RKL[p0 + k] = el;
            if (fl & (1 << k)) XL[el++] = p0 + k;
        }
        if (tid == 0) { RK[NPX] = nR; RKL[NPX] = nL; }
        if (tid < 16) {                                 // padding of the last tile: d out of range
            XR[nR + tid] = 1 << 20;
            XL[nL + tid] = -(1 << 20);
        }
    }
    __syncthreads();

    // ---- 2. gathers (all loads of a thread in flight), zero fill of the inactive pixels
    {
        constexpr int SPT = (SB_CAP + SB_THREADS - 1) / SB_THREADS;      // slots per thread (2 for the 640-slot launch)
        float rf[SPT][CQ], lf[SPT][CQ], nm[SPT], oo[SPT], gs[SPT], mu[SPT];
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int slot = tid + u * SB_THREADS;
            const int xr_own = slot < nR ? XR[slot] : -1, xl_own = slot < nL ? XL[slot] : -1;
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
                rf[u][c] = (xr_own >= 0 && c < C) ? rrow[(size_t)c * plane + xr_own] : 0.f;
                lf[u][c] = (xl_own >= 0 && c < C) ? lrow[(size_t)c * plane + xl_own] : 0.f;
            }
            nm[u] = oo[u] = gs[u] = mu[u] = 0.f;
            if (xl_own >= 0) {
                nm[u] = -max_cost[rowpix + xl_own] * LOG2E;
                oo[u] = out[rowpix + xl_own];
                gs[u] = grad_out[rowpix + xl_own] / sum_sim[rowpix + xl_own];
                if (VAR) mu[u] = disparity[rowpix + xl_own];
            }
        }
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int slot = tid + u * SB_THREADS;
            if (slot < SB_CAP) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) {
                    RF[c * SB_FP + slot] = rf[u][c];    // slots >= nR / nL hold zeros
                    LF[c * SB_FP + slot] = lf[u][c];
                }
                NM[slot] = nm[u]; OUT[slot] = oo[u]; GS[slot] = gs[u]; MU[slot] = mu[u];
            }
        }
#pragma unroll
        for (int c = 0; c < CQ; ++c)
            if (tid < 16) { RF[c * SB_FP + SB_CAP + tid] = 0.f; LF[c * SB_FP + SB_CAP + tid] = 0.f; }
        if (tid < 16) { NM[SB_CAP + tid] = 0.f; OUT[SB_CAP + tid] = 0.f; GS[SB_CAP + tid] = 0.f; MU[SB_CAP + tid] = 0.f; }
    }
    __syncthreads();

    // ---- 3. the two gradients
    const int j = lane & 15, q = lane >> 4;
    const int cj = j < CQ ? j : CQ - 1;               // channel this lane supplies to the contraction
    // (the side is a compile-time constant of two copies of this body: as a runtime loop variable every `side == 0 ? a : b`
    // inside the tile loop stayed a v_cndmask -- 20 of its 58 vector instructions per tile, round 6 ISA)
    auto one_side = [&](auto sidec) {
        constexpr int side = decltype(sidec)::value;
        const int n_own = side == 0 ? nL : nR;
        const int *XO = side == 0 ? XL : XR;           // own positions
        const int *XT = side == 0 ? XR : XL;           // other positions
        const float *FO = side == 0 ? LF : RF, *FT = side == 0 ? RF : LF;
        float *grow = side == 0 ? glrow : grrow;
        if (n_own == 0) {                               // no chunk writes anything: the row is all zeros
            for (int pp = tid; pp < W; pp += SB_THREADS) {
                for (int c = 0; c < C; ++c) grow[(size_t)c * plane + pp] = 0.f;
                if (VAR && side == 0) grad_disp[rowpix + pp] = 0.f;
            }
            return;
        }
        for (int e = 16 * wave; e < n_own; e += 16 * SB_NWAVE) {
            const bool act = e + j < n_own;
            const int xo = XO[act ? e + j : n_own - 1];
            const int x_lo = XO[e], x_hi = XO[min(e + 15, n_own - 1)];
            // other pixels that can pair with the chunk
            int i_lo, i_hi;
            if (side == 0) { i_lo = RK[max(0, x_lo - (D - 1))]; i_hi = RK[x_hi + 1]; }
            else { i_lo = RKL[x_lo]; i_hi = RKL[min(W, x_hi + D)]; }
            float bcur[KQ];
#pragma unroll
            for (int s = 0; s < KQ; ++s) bcur[s] = act ? FO[(4 * s + q) * SB_FP + e + j] : 0.f;
            float nm_own = 0.f, out_own = 0.f, mu_own = 0.f;
            if (side == 0) { nm_own = NM[e + j]; out_own = OUT[e + j]; mu_own = MU[e + j]; }
            f32x4 gacc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) gacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            float gdis = 0.f;
            if (i_hi > i_lo) {
#pragma unroll 1
                for (int t = i_lo >> 4; t <= (i_hi - 1) >> 4; ++t) {
                    // cost tile: rows = other slots 16t + 4q + r, columns (lanes) = own slots e + j
                    const float *ap = FT + q * SB_FP + 16 * t + j;
                    f32x4 cst = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bcur[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < KQ; ++s)
                        cst = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * SB_FP], bcur[s], cst, 0, 0, 0);
                    const int4 xt4 = *reinterpret_cast<const int4 *>(XT + 16 * t + 4 * q);
                    const int xtv[4] = {xt4.x, xt4.y, xt4.z, xt4.w};
                    float4 nmr, outr, gsr, mur;
                    if (side == 1) {
                        nmr = *reinterpret_cast<const float4 *>(NM + 16 * t + 4 * q);
                        outr = *reinterpret_cast<const float4 *>(OUT + 16 * t + 4 * q);
                        gsr = *reinterpret_cast<const float4 *>(GS + 16 * t + 4 * q);
                        if (VAR) mur = *reinterpret_cast<const float4 *>(MU + 16 * t + 4 * q);
                    }
                    f32x4 wt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = side == 0 ? xo - xtv[r] : xtv[r] - xo;
                        const float nm = side == 0 ? nm_own : (r == 0 ? nmr.x : r == 1 ? nmr.y : r == 2 ? nmr.z : nmr.w);
                        const float cc = ((unsigned)d < (unsigned)D && act) ? cst[r] : NEG_BIG;
                        const float ex = __builtin_amdgcn_exp2f(fmaf(cc, LOG2E, nm));
                        const float df = (float)d;
                        float w;
                        if (side == 0) {
                            if (VAR) {
                                const float dd = df - mu_own;
                                w = ex * fmaf(dd, dd, -out_own);       // SV_kernel.cu:191
                                gdis = fmaf(ex, dd, gdis);             // SV_kernel.cu:321
                            } else {
                                w = ex * (df - out_own);               // SM_kernel.cu:191
                            }
                        } else {
                            const float o = r == 0 ? outr.x : r == 1 ? outr.y : r == 2 ? outr.z : outr.w;
                            const float g = r == 0 ? gsr.x : r == 1 ? gsr.y : r == 2 ? gsr.z : gsr.w;
                            if (VAR) {
                                const float mu = r == 0 ? mur.x : r == 1 ? mur.y : r == 2 ? mur.z : mur.w;
                                const float dd = df - mu;
                                w = g * ex * fmaf(dd, dd, -o);         // SV_kernel.cu:262
                            } else {
                                w = g * ex * (df - o);                 // SM_kernel.cu:346
                            }
                        }
                        wt[r] = w;
                    }
                    // contraction over the 16 other slots of the tile (K step r <-> slot 4q + r)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int c = 16 * cb + cj < CQ ? 16 * cb + cj : CQ - 1;
                        const float4 ov = *reinterpret_cast<const float4 *>(FT + c * SB_FP + 16 * t + 4 * q);
                        gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0], ov.x, gacc[cb], 0, 0, 0);
                        gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[1], ov.y, gacc[cb], 0, 0, 0);
                        gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[2], ov.z, gacc[cb], 0, 0, 0);
                        gacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[3], ov.w, gacc[cb], 0, 0, 0);
                    }
                }
            }
            // gacc[cb][r]: channel 16*cb + (lane & 15), own slot e + 4q + r.  The chunk goes through
            // this wave's scratch and out as whole lines: the wave writes every pixel from its first
            // active one up to the next chunk's first (chunk 0 from pixel 0, the last one to W) --
            // gradients at the active pixels, zeros between them -- so every line of the row is
            // written once (a zero fill + scattered stores cost 1.6x the gradient bytes, WRITE_SIZE).
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int c = 16 * cb + j;
                if (c < CQ) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oi = e + 4 * q + r;
                        SC[c * 16 + 4 * q + r] = gacc[cb][r] * (side == 0 ? GS[oi] : 1.f);   // GS: padded to n + 16
                    }
                }
            }
            if (VAR && side == 0) {
                gdis += __shfl_xor(gdis, 16);
                gdis += __shfl_xor(gdis, 32);
                if (q == 0) SC[CQ * 16 + j] = -2.f * GS[e + j] * gdis;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned short *RO = side == 0 ? RKL : RK;       // exclusive own-side counts
            const int p_beg = e == 0 ? 0 : x_lo, p_end = e + 16 >= n_own ? W : XO[e + 16];
            for (int pp = p_beg + lane; pp < p_end; pp += 64) {
                const int r0 = RO[pp], sl = r0 - e;
                const bool on = RO[pp + 1] != r0;
                for (int c = 0; c < C; ++c) grow[(size_t)c * plane + pp] = on ? SC[c * 16 + sl] : 0.f;
                if (VAR && side == 0) grad_disp[rowpix + pp] = on ? SC[CQ * 16 + sl] : 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    one_side(std::integral_constant<int, 0>{});
    one_side(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------
// Dense rows, C <= 8, SpaMat: BOTH gradients from ONE pass over the cost / weight tiles (spamat_bwd_rowb below).
// The two band launches above form every 16 x 16 cost tile, its exponentials and weights twice (once per side) and each
// reads both feature rows: stage 3, B = 4: 0.75 ms, 1.63 x the algorithmic bytes.  One pass needs the weight tile
//   w = e (d - out)    (rows = right pixels, columns = left pixels; SM_kernel.cu:191)
// in both orientations:  gL[c][left] += sum_right w R[c][right]  and  gR[c][right] += sum_left w g/S L[c][left]
// (SM_kernel.cu:143-195, 300-355), and gR of a right tile collects from the NT left tiles xt .. xt + NT - 1.
// Work split (round 4, measured against one wave per row with a window of NT accumulator tiles: 0.55 vs 0.80 ms): a
// 256-thread workgroup owns the row and wave w owns the RIGHT tiles t = w (mod 4): for left tile xt it forms the (at most
// four) band tiles m = xt - t of its right tiles, so its right-gradient window is 4 accumulator tiles, not NT (a right
// tile keeps its owner while xt advances; the window slides when m0 = (xt - w) mod 4 wraps; the tile at m = NT - 1 is
// complete and stored), the per-wave dependency chain is 4 tiles long and 4 waves fit a SIMD without spills.  Shared by
// the workgroup: the ring of the last 16 right tiles, the left tile's operands (committed before barrier 1 from values
// fetched one tile ahead), and the four partial left gradients of a left tile (barrier 2; summed in a fixed wave order
// by wave xt mod 4, which also stores them).  Right tiles left of the row are skipped; nothing is atomic, every input
// byte is read once, every gradient byte written once.
// Round 4's kernel of this shape (spamat_bwd_roww: fp32 cost MFMAs, the weight tile through LDS as fp32 in both
// orientations, contractions on v_mfma_f32_4x4x1; 0.53 - 0.55 ms) is in the repository's history.
// ---- round 6: every matrix product of that pass on the bf16 pipe (bf16x3) -------------------------------------------------
// spamat_bwd_roww spent, per 16 x 16 tile and wave, 224 matrix-pipe cycles (two fp32 cost MFMAs + sixteen 4x4x1
// contractions), ~80 vector instructions and ~60 LDS cycles (the weight tile goes through LDS as fp32 in both orientations);
// the three pipes barely overlap: 0.53 ms at stage 3, B = 4 (profiles/r04c_pmc_sq_spamat_bwd_roww.txt).  Here
//   * cost tile: both views as three bf16 terms, two v_mfma_f32_16x16x32_bf16 (K = 8 channels x 4 term pairs: hh hm mh mm |
//     hl lh lm), the right mask's 0 / -1e30 bias as the initial accumulator -- 32 pipe cycles instead of 64;
//   * weights w = e (d - out) in the cost tile's own registers (lane = left pixel, registers = four right pixels), split into
//     three bf16 terms (truncations, exact residuals);
//   * left gradient  gL[left][c] = sum_right w R[c][right]: the weight terms ARE the A operand (K = 16 right pixels x 2 term
//     slots), B = the right view's image read column-wise with ds_read_b64_tr_b16 (N = 8 channels x 2 terms): two MFMAs
//     (w_h + w_m)(R_h | R_m) and w_h R_l + w_l (R_h | R_m), no LDS round trip for w;
//   * right gradient gR[right][c] = sum_left w g/S L[c][left]: the contraction runs over the LANE index of the weight tile, so
//     the three packed term planes go through a per-wave LDS scratch once (three ds_write_b64) and come back transposed
//     (four ds_read_b64_tr_b16); B = g/S-scaled left features, one image per left tile;
//   * the N halves (terms of the features) are added once per finished gradient tile, not per band tile.
// Products below 2^-24 relative are dropped (m l, l l on the weight side), like the forward's dense rows.  Work split, window of
// right tiles, barriers and the fixed summation order are described above.
// Images: one pixel = 64 bytes = four 16-byte chunks (h | m | l | zero) of 8 channels; chunk c of pixel p sits at
// 64 p + 16 (c ^ sw(p)), sw = 0, 2, 1, 3 for p / 4 = 0 .. 3: the cost operand reads (ds_read_b128, one chunk per lane) and
// the transposed reads (8 bytes per lane) are both bank-conflict free.
typedef __bf16 rb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rb_bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
typedef int rb_i32x4 __attribute__((ext_vector_type(4)));
typedef int rb_i32x2 __attribute__((ext_vector_type(2)));
#define RB_LDS __attribute__((address_space(3)))

constexpr int RB_NW = 4;
// LDS layout of spamat_bwd_rowb<NT, CB> (bytes).  CB = channel blocks of 8 (C <= 8 CB); a pixel of an image is CB x 64 bytes.
template <int NT, int CB>
struct RbLayout {
    static constexpr int RING = NT <= 8 ? 8 : 16;             // right tiles kept (a power of two >= NT)
    static constexpr int PXB = 64 * CB, TILEB = 16 * PXB;
    static constexpr int RIMG = 0;                            // ring of RING right tiles x 16 pixels x PXB
    static constexpr int LIMG = RIMG + RING * TILEB;          // left tile, terms of L            [16 px][PXB]
    static constexpr int GIMG = LIMG + TILEB;                 // left tile, terms of g/S L        [16 px][PXB]
    static constexpr int WT = GIMG + TILEB;                   // per wave: 3 term planes [16 left][16 right] bf16 (512 B each)
    static constexpr int BZ = WT + RB_NW * 1536;              // ring [RING][16] floats: 0 / -1e30 of the right mask
    static constexpr int SC = BZ + RING * 64;                 // NM [16] | OUT [16] | GS [16] floats
    static constexpr int GLP = SC + 256;                      // [4 waves][CB][64 lanes][4] partial left gradients
    static constexpr int BYTES = GLP + RB_NW * CB * 1024;
};

__device__ __forceinline__ int rb_sw(int px) { return ((px >> 1) & 2) | ((px >> 3) & 1); }
// weight-plane row of left pixel j: rows {0-3, 8-11} and {4-7, 12-15} (the two halves of a transposed read) on 8 distinct
// 32-byte slots each; 8-byte piece q of a row at (q ^ (slot >> 2)): the ds_write_b64 of 16 lanes are conflict free too
__device__ __forceinline__ int rb_wt_addr(int j, int piece) {
    const int s = (j & 3) | ((j & 8) >> 1) | ((j & 4) << 1);
    return s * 32 + 8 * (piece ^ (s >> 2));
}
__device__ __forceinline__ rb_i32x2 rb_tr16(const unsigned char *p) {
    return __builtin_bit_cast(rb_i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RB_LDS rb_bf16x4 *)(p)));
}
__device__ __forceinline__ void rb_split3(float x, int &h, int &m, int &l) {
    h = __float_as_int(x) & 0xffff0000;
    const float r1 = x - __int_as_float(h);
    m = __float_as_int(r1) & 0xffff0000;
    l = __float_as_int(r1 - __int_as_float(m));
}
__device__ __forceinline__ int rb_pack(int odd, int even) { return __builtin_amdgcn_perm(odd, even, 0x07060302); }   // {odd[31:16], even[31:16]}
__device__ __forceinline__ f32x4 rb_mfma(rb_i32x4 a, rb_i32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rb_bf16x8, a), __builtin_bit_cast(rb_bf16x8, b), c, 0, 0, 0);
}

// CB > 1 (round 6: stage 2, C = 24): the same pass with CB channel blocks -- the cost tile sums CB x 2 MFMAs, each
// contraction is 2 MFMAs per block on the block's own image columns, the weight tile (its exponentials, its split and its
// trip through LDS) is formed ONCE for all blocks; replaces the two band launches there (each of which forms every weight
// tile again, on fp32 MFMAs).
template <int NT, int CB>
__global__ __launch_bounds__(64 * RB_NW, CB == 1 ? 4 : (CB == 3 && NT > 8) ? 2 : 3) void spamat_bwd_rowb(
    const float *__restrict__ ref, const float *__restrict__ tar, const float *__restrict__ rmask,
    const float *__restrict__ tmask, const float *__restrict__ out, const float *__restrict__ sum_sim,
    const float *__restrict__ max_cost, const float *__restrict__ grad_out, float *__restrict__ grad_ref,
    float *__restrict__ grad_tar, int C, int H, int W, int D, int marker) {
    using LO = RbLayout<NT, CB>;
    static_assert(NT <= LO::RING, "band wider than the ring");
    constexpr int NW = RB_NW, KS = (NT + NW - 1) / NW, NTHR = 64 * NW, RING = LO::RING, PXB = LO::PXB, TILEB = LO::TILEB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LO::BYTES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x, b = row / H, y = row - b * H;
    const size_t plane = (size_t)H * W, rowpix = (size_t)row * W;
    const size_t frow = ((size_t)b * C * H + y) * W;
    if (marker && __float_as_int(grad_ref[frow]) != BWD_MARK) return;    // (block-uniform) the sparse-row launches took this row
    const float *lrow = ref + frow, *rrow = tar + frow;
    float *glrow = grad_ref + frow, *grrow = grad_tar + frow;
    const int j = lane & 15, q = lane >> 4;              // tile coordinates: column (left pixel / output column n), K group
    const int qq = (lane & 15) >> 2, pp = lane & 3;      // address role in a transposed read: row qq of the group's four, piece pp
    const int XT = (W + 15) >> 4;
    const bool al4 = (W & 3) == 0 && ((((uintptr_t)grad_ref) | ((uintptr_t)grad_tar)) & 15) == 0;
    float *BZ = reinterpret_cast<float *>(smem + LO::BZ);
    float *NM = reinterpret_cast<float *>(smem + LO::SC), *OUT = NM + 16, *GS = NM + 32;
    unsigned char *WT = smem + LO::WT + wave * 1536;
    float *GLP = reinterpret_cast<float *>(smem + LO::GLP);

    for (int i = threadIdx.x; i < (LO::WT) / 16; i += NTHR)              // images: zero features, zero fourth chunks
        reinterpret_cast<float4 *>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < RING * 16; i += NTHR) BZ[i] = NEG_BIG;

    // per-lane byte offsets (constant over the row; channel block cb adds 64 cb)
    const int swj = rb_sw(j);
    const int a1off = PXB * j + 16 * ((q >> 1) ^ swj);                   // cost A, MFMA 1: right terms  h h m m
    const int a2off = PXB * j + 16 * (((q >> 1) * 2) ^ swj);             //         MFMA 2:              h h l l
    const int pxr = 4 * q + qq;                                          // transposed read of the right image: pixel row
    const int xoff = PXB * pxr + 16 * ((pp >> 1) ^ rb_sw(pxr)) + 8 * (pp & 1);         // columns (h | m)
    // (columns (l | 0): chunk index ^ 2 = byte address ^ 32; every other summand of the address is a multiple of 64)
    const int wtw = rb_wt_addr(j, q);                                    // this lane's 8-byte piece of a weight plane
    const int lr0 = 8 * (q & 1) + qq, lr1 = lr0 + 4;                     // transposed reads over LEFT pixels: rows of the two reads
    const int p1 = q < 2 ? 0 : 512, p2 = q < 2 ? 0 : 1024;              // weight plane of this lane's K group: MFMA 1 h h m m, MFMA 2 h h l l
    const int u1a = p1 + rb_wt_addr(lr0, pp), u1b = p1 + rb_wt_addr(lr1, pp), ud2 = p2 - p1;
    // staging roles: lane = (pixel j, channel pair q): channels 8 cb + 2q, + 1 of one view; word q of a 16-byte chunk
    const int imgw = PXB * j + 4 * q;
    // per-left-tile operand reads (lane constants held in registers: recomputing them cost ~25 vector instructions per left
    // tile and wave).  g/S L image, MFMA 1: columns (h | m) for every K group; MFMA 2: (l | 0) beside w_h (K groups 0, 1),
    // (h | m) beside w_l -- chunk index ^ 2 = byte address ^ 32.  Cost B operands: left terms h m h m | l 0 h m.
    const int g1a = LO::GIMG + PXB * lr0 + 16 * ((pp >> 1) ^ rb_sw(lr0)) + 8 * (pp & 1);
    const int g1b = LO::GIMG + PXB * lr1 + 16 * ((pp >> 1) ^ rb_sw(lr1)) + 8 * (pp & 1);
    const int l1o = LO::LIMG + PXB * j + 16 * ((q & 1) ^ swj);
    const int x32 = q < 2 ? 32 : 0;

    // values of the NEXT left tile, requested one tile ahead and not looked at before the next commit: every load is
    // unconditional (clamped pixel, channel and tile; what is outside is selected away at the commit) -- an exec-masked
    // load merged with a zero makes the compiler wait for memory right behind the request, i.e. one HBM round trip per
    // left tile in front of barrier (1) (the form of spamat_bwd_roww; found in the ISA, round 6).
    // wave 0: right view + right mask; wave 1: left view; wave 2: left view, g, S, left mask; wave 3: the per-pixel scalars
    const float *srcv = wave == 0 ? rrow : lrow;
    const float *src0[CB], *src1[CB];                                    // this lane's two channel rows per block
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        src0[cb] = srcv + (size_t)(8 * cb + 2 * q < C ? 8 * cb + 2 * q : 0) * plane;
        src1[cb] = srcv + (size_t)(8 * cb + 2 * q + 1 < C ? 8 * cb + 2 * q + 1 : 0) * plane;
    }
    const float *pm = (wave == 0 ? tmask : rmask) + rowpix;             // mask plane this wave looks at
    const float *pa = (wave == 2 ? grad_out : max_cost) + rowpix, *pb = (wave == 2 ? sum_sim : out) + rowpix;
    float f0[CB], f1[CB], fm, fa, fb;
    auto fetch = [&](int xt) {
        const int x = min(min(xt, XT - 1) * 16 + j, W - 1);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f0[cb] = src0[cb][x];
            f1[cb] = src1[cb][x];
        }
        fm = pm[x];
        fa = pa[x];
        fb = pb[x];
    };
    auto put_image = [&](unsigned char *img, float v0, float v1) {      // three term words of (channel 2q, 2q + 1) at pixel j
        int h0, m0, l0, h1, m1, l1;
        rb_split3(v0, h0, m0, l0);
        rb_split3(v1, h1, m1, l1);
        *reinterpret_cast<int *>(img + imgw + 16 * (0 ^ swj)) = rb_pack(h1, h0);
        *reinterpret_cast<int *>(img + imgw + 16 * (1 ^ swj)) = rb_pack(m1, m0);
        *reinterpret_cast<int *>(img + imgw + 16 * (2 ^ swj)) = rb_pack(l1, l0);
    };
    fetch(0);

    f32x4 gr[KS][CB];                    // gr[k]: right tile xt - (m0 + 4k); lane (n = (term, channel), q), register r: pixel 4q + r
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) gr[k][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Finished gradient tiles are STORED one left tile later (behind the next barrier (1)): vmcnt counts stores too and
    // in order, so a store issued just in front of the next commit's wait for the prefetched values would put its own
    // round trip into every left tile's chain.
    // One quad per lane and block is enough: a wave finishes a right tile where m0 + 4k = NT - 1, i.e. wave = xt - (NT - 1)
    // (mod 4), and sums the left tile as wave = xt (mod 4): never both for one xt while (NT - 1) % 4 != 0.
    static_assert((NT - 1) % RB_NW != 0, "a wave would finish a right tile and a left tile in one step");
    float4 pq[CB];
    int pq_x = -1;                                       // first pixel of this lane's pending quads (-1: none)
    bool pq_right = false;                               // (wave-uniform) which gradient they belong to
    auto store4 = [&](float *gp, int x, const float4 &o) {
        if (al4 && x + 3 < W) {
            *reinterpret_cast<float4 *>(gp) = o;
        } else {
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (x + r < W) gp[r] = ov[r];
        }
    };
    auto flush_pending = [&]() {
        if (pq_x >= 0) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                if (8 * cb + j < C) store4((pq_right ? grrow : glrow) + (size_t)(8 * cb + j) * plane + pq_x, pq_x, pq[cb]);
        }
        pq_x = -1;
    };
    auto finish_right = [&](int t, const f32x4 (&gpart)[CB]) { // right tile t complete -> pending grad_tar quads (0 where the right mask is off)
        if (t < 0) return;                               // (wave-uniform)
        pq_right = true;
        const int x = t * 16 + 4 * q;
        const float4 bz = *reinterpret_cast<const float4 *>(BZ + (t & (RING - 1)) * 16 + 4 * q);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x4 gsum;
#pragma unroll
            for (int r = 0; r < 4; ++r) gsum[r] = gpart[cb][r] + __shfl_xor(gpart[cb][r], 8);     // + the other term half of N
            pq[cb] = make_float4(bz.x == 0.f ? gsum[0] : 0.f, bz.y == 0.f ? gsum[1] : 0.f, bz.z == 0.f ? gsum[2] : 0.f,
                                 bz.w == 0.f ? gsum[3] : 0.f);
        }
        if (j < 8 && x < W) pq_x = x;
    };
    __syncthreads();                                     // the zeroed images

    for (int xt = 0; xt < XT; ++xt) {
        const int x0 = xt * 16, s0 = xt & (RING - 1);
        const int m0 = (xt - wave) & (NW - 1);           // this wave's band tiles: m0, m0 + NW, ...
        if (xt > 0 && m0 == 0) {                         // the window slides: a new right tile (t = xt) enters at k = 0
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                for (int k = KS - 1; k > 0; --k) gr[k][cb] = gr[k - 1][cb];
                gr[0][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // commit tile xt (ring slot s0 held right tile xt - RING: dead since left tile xt - 2).  One branch per wave role
        // (a shared body with `wave == 2 ? ... : ...` inside was if-converted: all four waves ran the division)
        {
            const bool okx = x0 + j < W;
            const bool on = okx && fm != 0.f;
            float v0[CB], v1[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                v0[cb] = (okx && 8 * cb + 2 * q < C) ? f0[cb] : 0.f;
                v1[cb] = (okx && 8 * cb + 2 * q + 1 < C) ? f1[cb] : 0.f;
            }
            if (wave == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) put_image(smem + LO::RIMG + s0 * TILEB + 64 * cb, v0[cb], v1[cb]);
                if (q == 0) BZ[s0 * 16 + j] = on ? 0.f : NEG_BIG;
            } else if (wave == 1) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) put_image(smem + LO::LIMG + 64 * cb, v0[cb], v1[cb]);
            } else if (wave == 2) {
                const float gsv = on ? fa / fb : 0.f;                  // g / S, 0 where the left mask is off
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)                        // SM_kernel.cu:346: g/S L[c][left]
                    put_image(smem + LO::GIMG + 64 * cb, v0[cb] * gsv, v1[cb] * gsv);
                if (q == 0) GS[j] = gsv;
            } else if (q == 0) {
                NM[j] = on ? -fa * LOG2E : NEG_BIG;                    // masked-off / out-of-row left pixel: weights 0
                OUT[j] = fb;
            }
        }
        fetch(xt + 1);
        __syncthreads();                                 // (1) tile xt is in the ring, the left tile's images and scalars are up
        flush_pending();                                 // the gradient quads finished during the previous left tile
        // cost B operands: left terms h m h m | l 0 h m of the K groups
        rb_i32x4 bl1[CB], bl2[CB], G1[CB], G2[CB];
        {
            const int g2a = g1a ^ x32, g2b = g1b ^ x32, l2o = l1o ^ x32;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                bl1[cb] = *reinterpret_cast<const rb_i32x4 *>(smem + l1o + 64 * cb);
                bl2[cb] = *reinterpret_cast<const rb_i32x4 *>(smem + l2o + 64 * cb);
                const rb_i32x2 a = rb_tr16(smem + g1a + 64 * cb), bq = rb_tr16(smem + g1b + 64 * cb),
                               c2 = rb_tr16(smem + g2a + 64 * cb), d2 = rb_tr16(smem + g2b + 64 * cb);
                G1[cb] = rb_i32x4{a[0], a[1], bq[0], bq[1]};
                G2[cb] = rb_i32x4{c2[0], c2[1], d2[0], d2[1]};
            }
        }
        const float nm_own = NM[j], out_own = OUT[j];
        const float dm0 = (float)(j - 4 * q) - out_own;                              // d - out = 16 m - r + dm0
        f32x4 gl[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) gl[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // One band tile.  CHECK = the tile may lie outside the band (m >= NT) or left of the row (m > xt) and is then skipped
        // by a wave-uniform branch.  Every such branch makes the compiler copy the window of right-gradient accumulators at
        // its merge (16 - 20 v_mov per tile, round 6 ISA), so the steady state (xt >= NT - 1) runs the tiles that are always
        // in band (4k + 3 < NT) without the test; the first NT - 1 left tiles and the last k keep it.
        auto band_tile = [&](auto kc, auto checkc) {
            constexpr int k = decltype(kc)::value;
            constexpr bool CHECK = decltype(checkc)::value;
            const int m = m0 + NW * k;
            // (wave-uniform) outside the band / left of the row.  Computing those tiles with all costs at -1e30 instead of
            // branching around them saves the 16 v_mov per tile the branch costs (the compiler copies the window of
            // right-gradient accumulators at every merge) but adds one tile in sixteen: measured 0.436 -> 0.46 ms, not kept.
            if (CHECK && (m >= NT || m > xt)) return;
            const int ob = ((xt - m) & (RING - 1)) * TILEB;
            const float4 bz = *reinterpret_cast<const float4 *>(BZ + ((xt - m) & (RING - 1)) * 16 + 4 * q);
            f32x4 cst = f32x4{bz.x, bz.y, bz.z, bz.w};
            if (m == 0 || 16 * m + 15 >= D) {            // (wave-uniform) edge tiles of the band: 0 <= d < D
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * m + j - (4 * q + r);
                    cst[r] = (unsigned)d < (unsigned)D ? cst[r] : NEG_BIG;
                }
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const rb_i32x4 a1 = *reinterpret_cast<const rb_i32x4 *>(smem + LO::RIMG + ob + a1off + 64 * cb);
                const rb_i32x4 a2 = *reinterpret_cast<const rb_i32x4 *>(smem + LO::RIMG + ob + a2off + 64 * cb);
                cst = rb_mfma(a1, bl1[cb], cst);
                cst = rb_mfma(a2, bl2[cb], cst);
            }
            const float dm = dm0 + (float)(16 * m);
            int wh[4], wm[4], wl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(cst[r], LOG2E, nm_own));
                rb_split3(e * (dm - (float)r), wh[r], wm[r], wl[r]);                 // SM_kernel.cu:191
            }
            const int h01 = rb_pack(wh[1], wh[0]), h23 = rb_pack(wh[3], wh[2]);
            const int m01 = rb_pack(wm[1], wm[0]), m23 = rb_pack(wm[3], wm[2]);
            const int l01 = rb_pack(wl[1], wl[0]), l23 = rb_pack(wl[3], wl[2]);
            // the previous tile's transposed reads of the planes (other lanes of this wave) come before these stores; the
            // LDS executes a wave's accesses in order, the fences (no instructions) keep the COMPILER from reordering
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<rb_i32x2 *>(WT + wtw) = rb_i32x2{h01, h23};
            *reinterpret_cast<rb_i32x2 *>(WT + 512 + wtw) = rb_i32x2{m01, m23};
            *reinterpret_cast<rb_i32x2 *>(WT + 1024 + wtw) = rb_i32x2{l01, l23};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // left gradient: A = own weight terms (K = right pixels 4q + r x 2 slots), B = right image columns
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const rb_i32x2 X = rb_tr16(smem + LO::RIMG + ob + xoff + 64 * cb),
                               Y = rb_tr16(smem + LO::RIMG + ((ob + xoff + 64 * cb) ^ 32));
                gl[cb] = rb_mfma(rb_i32x4{h01, h23, m01, m23}, rb_i32x4{X[0], X[1], X[0], X[1]}, gl[cb]);   // (w_h + w_m)(R_h | R_m)
                gl[cb] = rb_mfma(rb_i32x4{h01, h23, l01, l23}, rb_i32x4{Y[0], Y[1], X[0], X[1]}, gl[cb]);   // w_h (R_l | 0) + w_l (R_h | R_m)
            }
            // right gradient: A = the planes read back transposed (K = left pixels x 2 slots), B = g/S L image columns
            {
                const rb_i32x2 ua = rb_tr16(WT + u1a), ub = rb_tr16(WT + u1b), uc = rb_tr16(WT + u1a + ud2),
                               ud = rb_tr16(WT + u1b + ud2);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    gr[k][cb] = rb_mfma(rb_i32x4{ua[0], ua[1], ub[0], ub[1]}, G1[cb], gr[k][cb]);
                    gr[k][cb] = rb_mfma(rb_i32x4{uc[0], uc[1], ud[0], ud[1]}, G2[cb], gr[k][cb]);
                }
            }
            if (m == NT - 1) {                           // right tile xt - m has seen all its left tiles
                finish_right(xt - m, gr[k]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) gr[k][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto all_tiles = [&](auto steadyc) {
            constexpr bool STEADY = decltype(steadyc)::value;
            if constexpr (KS > 0) band_tile(std::integral_constant<int, 0>{}, std::integral_constant<bool, !(STEADY && 3 < NT)>{});
            if constexpr (KS > 1) band_tile(std::integral_constant<int, 1>{}, std::integral_constant<bool, !(STEADY && 7 < NT)>{});
            if constexpr (KS > 2) band_tile(std::integral_constant<int, 2>{}, std::integral_constant<bool, !(STEADY && 11 < NT)>{});
            if constexpr (KS > 3) band_tile(std::integral_constant<int, 3>{}, std::integral_constant<bool, !(STEADY && 15 < NT)>{});
        };
        static_assert(KS <= 4, "all_tiles lists four tiles");
        if constexpr (CB == 1) {                         // (CB > 1: two copies of the tile bodies spill registers)
            if (xt >= NT - 1) all_tiles(std::true_type{});
            else all_tiles(std::false_type{});
        } else {
            all_tiles(std::false_type{});
        }
        // this wave's share of the left tile's gradient -> LDS; summed in a fixed order by wave xt mod 4
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            *reinterpret_cast<float4 *>(GLP + (wave * CB + cb) * 256 + lane * 4) = make_float4(gl[cb][0], gl[cb][1], gl[cb][2], gl[cb][3]);
        const bool summer = wave == (xt & (NW - 1)) && j < 8;
        float4 gs4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (summer) gs4 = *reinterpret_cast<const float4 *>(GS + 4 * q);
        __syncthreads();                                 // (2) partials complete; the left tile's images may be overwritten
        if (wave == (xt & (NW - 1))) pq_right = false;
        if (summer) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {        // both term halves of N (lanes n, n + 8) of every wave
                    const float4 pa4 = *reinterpret_cast<const float4 *>(GLP + (w2 * CB + cb) * 256 + lane * 4);
                    const float4 pb4 = *reinterpret_cast<const float4 *>(GLP + (w2 * CB + cb) * 256 + (lane + 8) * 4);
                    acc.x += pa4.x + pb4.x; acc.y += pa4.y + pb4.y; acc.z += pa4.z + pb4.z; acc.w += pa4.w + pb4.w;
                }
                // lane (n = channel, q): left pixels x0 + 4q + r: grad_ref = g * sum / S (SM_kernel.cu:193)
                pq[cb] = make_float4(acc.x * gs4.x, acc.y * gs4.y, acc.z * gs4.z, acc.w * gs4.w);
            }
            if (x0 + 4 * q < W) pq_x = x0 + 4 * q;
        }
    }
    flush_pending();
    // right tiles still open after the last left tile (those at m = NT - 1 were stored in the loop)
    {
        const int m0 = (XT - 1 - wave) & (NW - 1);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int m = m0 + NW * k;
            if (m < NT - 1) {
                finish_right(XT - 1 - m, gr[k]);
                flush_pending();
            }
        }
    }
}

template <int NT, int CB>
int launch_row(const float *ref, const float *tar, const float *rmask, const float *tmask, const float *out,
               const float *sum_sim, const float *max_cost, const float *grad_out, float *grad_ref, float *grad_tar,
               int B, int C, int H, int W, int D, int marker, hipStream_t stream) {
    if ((double)C * H * W >= 2147483648.0) return DECNET_ERR_UNSUPPORTED;        // 32-bit element offsets inside a sample
    hipLaunchKernelGGL((spamat_bwd_rowb<NT, CB>), dim3((unsigned)((size_t)B * H)), dim3(64 * RB_NW), 0, stream, ref, tar, rmask,
                       tmask, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar, C, H, W, D, marker);
    return decnet_launch_status();
}

template <int NT, bool VAR, int KQ>
int launch_both(const float *ref, const float *tar, const float *rmask, const float *tmask,
                const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
                const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C,
                int H, int W, int D, hipStream_t stream) {
    const int xt0 = side_xt<NT, VAR, 0>(4 * KQ, W), xt1 = side_xt<NT, VAR, 1>(4 * KQ, W);
    if (!xt0 || !xt1) return DECNET_ERR_UNSUPPORTED;
    // sparse rows first (C <= 24, rows of <= 2048 pixels), the rest by the marker launches
    int marker = 0;
    if constexpr (KQ <= 6) {
        if (W <= 2048) {
            marker = 1;
            const int ppt = W <= 1024 ? 4 : 8;
            const size_t slds = 4 * sb_words(KQ, ppt, SB_CAP, SB_THREADS);
            if (slds > 64 * 1024) {
                hipError_t e = ppt == 4
                    ? hipFuncSetAttribute((const void *)spamat_bwd_sparse<VAR, KQ, 4>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds)
                    : hipFuncSetAttribute((const void *)spamat_bwd_sparse<VAR, KQ, 8>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
                if (e != hipSuccess) return (int)e;
            }
            if (ppt == 4)
                hipLaunchKernelGGL((spamat_bwd_sparse<VAR, KQ, 4>), dim3((unsigned)(B * H)), dim3(SB_THREADS), slds,
                                   stream, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out,
                                   grad_ref, grad_tar, grad_disp, C, H, W, D, xt0 * 16, xt1 * 16);
            else
                hipLaunchKernelGGL((spamat_bwd_sparse<VAR, KQ, 8>), dim3((unsigned)(B * H)), dim3(SB_THREADS), slds,
                                   stream, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out,
                                   grad_ref, grad_tar, grad_disp, C, H, W, D, xt0 * 16, xt1 * 16);
            int rc = decnet_launch_status();
            if (rc) return rc;
            // rows of 257 - 640 active pixels per side (C <= 8, whole rows of <= 1024 pixels): the same algorithm with
            // 640 slots on 512 threads, on the marked rows only (DECNET_SPAMAT_MID=0 leaves them to the band launches)
            static const int mid_off = [] { const char *e = getenv("DECNET_SPAMAT_MID"); return e && atoi(e) == 0; }();
            if constexpr (KQ == 2) {
                const size_t mlds = 4 * sb_words(KQ, 4, SBM_CAP, SBM_THREADS);
                if (!mid_off && ppt == 4 && mlds <= DECNET_LDS_BYTES / 2) {
                    if (mlds > 64 * 1024) {
                        hipError_t e = hipFuncSetAttribute((const void *)spamat_bwd_sparse<VAR, KQ, 4, SBM_CAP, SBM_THREADS, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds);
                        if (e != hipSuccess) return (int)e;
                    }
                    hipLaunchKernelGGL((spamat_bwd_sparse<VAR, KQ, 4, SBM_CAP, SBM_THREADS, true>), dim3((unsigned)(B * H)),
                                       dim3(SBM_THREADS), mlds, stream, ref, tar, rmask, tmask, disparity, out, sum_sim,
                                       max_cost, grad_out, grad_ref, grad_tar, grad_disp, C, H, W, D, xt0 * 16, xt1 * 16);
                    if ((rc = decnet_launch_status())) return rc;
                }
            }
        }
    }
    // dense rows at C <= 8 (SpaMat): both gradients from one pass, four waves per row (the band launch below stays
    // the path of SpaVar, of C > 8 and of rows narrower than the band)
    if constexpr (KQ == 2 && !VAR && NT <= 15) {
        if (W >= 16 * NT)
            return launch_row<NT, 1>(ref, tar, rmask, tmask, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar, B, C, H, W,
                                     D, marker, stream);
    }
    // ... and at 9 - 24 channels (stage 2) with two / three channel blocks (round 6)
    if constexpr (KQ == 6 && !VAR && NT <= 11) {
        if (W >= 16 * NT) {
            int rc = C <= 16 ? launch_row<NT, 2>(ref, tar, rmask, tmask, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar, B, C,
                                                H, W, D, marker, stream)
                             : launch_row<NT, 3>(ref, tar, rmask, tmask, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar, B, C,
                                                H, W, D, marker, stream);
            if (rc != DECNET_ERR_UNSUPPORTED) return rc;
        }
    }
    return launch_sides<NT, VAR, KQ>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref,
                                     grad_tar, grad_disp, B, C, H, W, D, xt0, xt1, marker, stream);
}

template <int NT, bool VAR>
int launch_c(const float *ref, const float *tar, const float *rmask, const float *tmask,
             const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
             const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C,
             int H, int W, int D, hipStream_t stream) {
#define GO(K)                                                                                       \
    return launch_both<NT, VAR, K>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost,      \
                                   grad_out, grad_ref, grad_tar, grad_disp, B, C, H, W, D, stream)
    if (C <= 8) GO(2);
    if (C <= 24) GO(6);
    if (C <= 72) GO(18);
#undef GO
    return DECNET_ERR_UNSUPPORTED;
}

}  // namespace

// var: 0 SpaMat, 1 SpaVar (also writes grad_disp).  DECNET_ERR_UNSUPPORTED (C > 72, band wider
// than 18 tiles, LDS overflow) makes capi.hip fall back to the row-tile kernels.
int decnet_mfma_backward(int var, const float *ref, const float *tar, const float *rmask,
                         const float *tmask, const float *disparity, const float *out,
                         const float *sum_sim, const float *max_cost, const float *grad_out,
                         float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                         int W, int max_disp, hipStream_t stream) {
    const int D = max_disp;
    const int need = D <= 1 ? 1 : (D - 1 + 15) / 16 + 1;
    if (need > 18) return DECNET_ERR_UNSUPPORTED;
    // the band loop is a runtime loop (nothing is kept per tile), so NT only sizes the halo
#define GO(N)                                                                                       \
    do {                                                                                            \
        if (var)                                                                                    \
            return launch_c<N, true>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost,     \
                                     grad_out, grad_ref, grad_tar, grad_disp, B, C, H, W, D, stream); \
        return launch_c<N, false>(ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost,        \
                                  grad_out, grad_ref, grad_tar, grad_disp, B, C, H, W, D, stream);  \
    } while (0)
    if (need <= 3) GO(3);
    if (need <= 6) GO(6);
    if (need <= 11) GO(11);
    if (need <= 15) GO(15);
    GO(18);
#undef GO
}
