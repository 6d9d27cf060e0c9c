// tools/experiments/chain2d.hip (EXPERIMENT, not in libdecnet_hip.so; was decnet_amd/csrc/chain2d.hip) -- chains of few-channel 3x3 convolutions as ONE kernel.
//
// The full-resolution parts of the trunk are chains of Conv2dUnit layers with <= 8 output channels
// (modules/submodule.py): FeatExtNet conv0 (:263-266), Deconv2dBlock of the finest level (:162-178),
// GenerateSparseMask (:347-372), SoftAttention (:593-604), the head of Refinement (:690-717).  One layer per
// kernel makes every intermediate [B,8,H,W] tensor (134 MB at 972x540, batch 8) a round trip through HBM and,
// at 576 MACs per pixel, leaves the fp32 VALU -- not the memory system -- as the bound.  Here a chain
//
//     source (channel concatenation of tensors | transposed conv k3 s3 of the coarser level | disparity warp)
//        -> conv 3x3 (dilation d1) -> [conv 3x3 (d2) -> [conv 3x3 (d3)]] -> sink (store | sigmoid blend | mask)
//
// is one launch: a workgroup owns a strip of TW output columns x R output rows of one image and walks down the
// rows.  Every level of the chain lives in LDS as a ring of 2 d + 2 rows ("line buffers"), so an input row is read
// from HBM once, intermediate rows never leave the CU, and the only halo that is recomputed is the 2 x (sum of
// dilations) columns / rows at the strip borders.
//
// Arithmetic: matrix cores at fp32 accuracy.  A level is stored split into three bf16 terms x = hi + mid + lo
// (truncations with exact residuals, 24 mantissa bits) as [term][8-channel group][ring row][pixel][8 x bf16]; one
// ds_read_b128 per lane is the B operand (32 k x 16 pixels) of v_mfma_f32_16x16x32_bf16 whose four 8-wide k groups
// are the terms (hi, mid, lo, hi) of one (tap, channel group).  The A operand (16 rows x 32 k) holds the weights of
// 8 output channels twice: rows "hi" = (w_hi, w_hi, w_hi, w_lo), rows "mid" = (w_mid, w_mid, w_mid, 0); a lane's
// accumulator registers (0, 2) and (1, 3) are the hi / mid rows of its two output channels, so
//     out = sum_k [ (w_hi + w_mid)(x_hi + x_mid + x_lo) + w_lo x_hi ]
// -- every product above 2^-24 |w x| -- comes out of ONE MFMA per (tap, channel group) and 16 pixels, with no
// cross-lane step.  The weights of a wave's layer stay in registers (<= 27 tiles): waves are specialised by layer
// (static split by MFMA count), all layers run skewed in the same step (layer j computes row s + off_j, off_j =
// off_{j+1} + d_{j+1} + 1), one barrier per row step.
#include "common.h"
#include "decnet_chain2d.h"
#include <stdio.h>

typedef float f32x4_h __attribute__((ext_vector_type(4)));
typedef int i32x4_h __attribute__((ext_vector_type(4)));
typedef int i32x2_h __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_h __attribute__((ext_vector_type(8)));

namespace {

constexpr int MAXL = DECNET_CHAIN_MAX_LAYERS, MAXPART = DECNET_CHAIN_MAX_PARTS, MAXG0 = 12;
constexpr int THREADS = 512;
constexpr int NCW = 6, NLW = 2;                      // compute / loader waves of a workgroup
constexpr int MAXKU = 4, MAXKA = 4, MAXAUX = 4;     // source units / aux loads per loader lane and row; aux planes

struct LayerK {                      // one layer, as the kernel sees it
    float sc[8], sh[8];
    const float *aux;                // DECNET_EPI_SUBSQ: [B, auxc, H, W]
    int G, KT, dil, relu, cout, epi, auxc, woff;
    int off, halo, c0, ntiles, w0, nw;
    int lds_in, nr_in, lds_out, nr_out;
};
struct PartK {
    const float *p, *p2, *aux, *sc, *sh;
    int c, kind, cp, relu, ch0;      // ch0: first channel of the part in the concatenation
};
struct PlanK {
    PartK dec, wrp;                                     // the (at most one each) DECONV / WARP part
    LayerK L[MAXL];
    const float *chp[MAXG0 * 8], *chp2[MAXG0 * 8];      // PLAIN channels: plane 0 of the channel (b < bsplit / b >= bsplit), or null
    int chc[MAXG0 * 8];                                 // channels of the tensor that holds it (batch stride = chc * H * W)
    int gkind[MAXG0], gpart[MAXG0], ustart[MAXG0 + 1];
    int nparts, NL, G0, bsplit, H, W, TW, R, H0, pitch, Wrow, Wpad, lds0, nr0, sink, cout_last, lds_tab;
    float *out;
    unsigned short *bits16;
    const float *sa, *sb;
    float m_w[3], m_s, m_b, thold;
    int wpr;
    const float *auxp[MAXAUX];                          // aux planes (SUBSQ: channel c of the aux tensor; BLEND: dense, sparse)
    int auxbs[MAXAUX];                                  // batch stride of each, floats
    int naux, aux_layer, lds_aux, debug;
};

__device__ __forceinline__ void split3(float x, int &h, int &m, int &l) {
    // round-to-nearest-even terms (v_cvt_pk_bf16_f32): |x - h| <= 2^-9 |x|, |x - h - m| <= 2^-18 |x|, and the residual
    // that l leaves is <= 2^-27 |x| -- truncated terms (round 2) left 2^-24 and, worse, always of the sign of x, so the
    // dropped m.l / l.m products of a K-long sum added up instead of averaging out (tests/test_inputdata_gpu.py measures
    // the network's distance to its float64 run: 1.35 x the reference's float32 distance before, 1.0 x after)
    h = __float_as_int((float)(__bf16)x);
    const float r1 = x - __int_as_float(h);
    m = __float_as_int((float)(__bf16)r1);
    l = __float_as_int((float)(__bf16)(r1 - __int_as_float(m)));
}
__device__ __forceinline__ int pack2(int hi_elem, int lo_elem) {       // (elem j+1, elem j) -> one dword of bf16 pairs
    return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302);
}
__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int ring_slot(int r, int nr) { return (int)((unsigned)(r + 64 * nr) % (unsigned)nr); }

// The plan is read straight from the kernel-argument segment (constant address space: scalar loads, dynamic indexing
// without a private copy of the 2.7 KB structure)
typedef const __attribute__((address_space(4))) PlanK CPlan;
typedef const __attribute__((address_space(4))) LayerK CLayer;
typedef const __attribute__((address_space(4))) PartK CPart;

// A tiles: w [Cout][Cin][KT] (Conv2d weight), sign[Cin] (+-1 or null) -> wp[(tap * G + c) * 64 + lane]
__global__ void chain2d_pack(const float *__restrict__ w, const float *__restrict__ sign, i32x4_h *__restrict__ wp,
                             int Cin, int Cout, int KT, int G) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KT * G * 64) return;
    const int lane = idx & 63, st = idx >> 6, c = st % G, tap = st / G;
    const int i = lane & 15, g = lane >> 4;
    const int co = 2 * (i >> 2) + (i & 1), wterm = (i & 3) >> 1;
    int t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = 8 * c + e;
        float v = (co < Cout && ch < Cin) ? w[((size_t)co * Cin + ch) * KT + tap] : 0.f;
        if (sign && ch < Cin) v *= sign[ch];
        int h, m, l;
        split3(v, h, m, l);
        t[e] = g < 3 ? (wterm == 0 ? h : m) : (wterm == 0 ? l : 0);
    }
    i32x4_h o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2(t[2 * e + 1], t[2 * e]);
    wp[idx] = o;
}

// ---- the MFMA body of one 16-pixel tile: KT taps x GG channel groups, weights in registers ----------------------
template <int GG, int KT>
__device__ __forceinline__ void tile_mfma(const i32x4_h *a, const unsigned char *lds, int lane_in, const int *rowoff,
                                          int col0, int dil, int gstride, f32x4_h &acc0, f32x4_h &acc1) {
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
        const int ky = KT == 9 ? tap / 3 : 1, kx = KT == 9 ? tap % 3 : 1;
        const int base = lane_in + rowoff[KT == 9 ? ky : 0] + (col0 + (kx - 1) * dil) * 16;
#pragma unroll
        for (int c = 0; c < GG; ++c) {
            const i32x4_h bq = *reinterpret_cast<const i32x4_h *>(lds + base + c * gstride);
            const int st = tap * GG + c;
            if (st & 1)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_h, a[st]),
                                                               __builtin_bit_cast(bf16x8_h, bq), acc1, 0, 0, 0);
            else
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_h, a[st]),
                                                               __builtin_bit_cast(bf16x8_h, bq), acc0, 0, 0, 0);
        }
    }
}

// global (not flat) loads for pointers that passed through LDS or the kernel-argument tables
typedef const __attribute__((address_space(1))) float *gfp;
__device__ __forceinline__ gfp as_global(const float *p) { return (gfp)(uintptr_t)p; }

// ---------------------------------------------------------------------------------------------------------------------
// Compute waves: the row loop of ONE layer (the wave's), GG channel groups, weights in registers (SLOW: any group count /
// 1x1 layers, weights streamed from L1 / L2).  No vector-memory load inside the loop.
// ---------------------------------------------------------------------------------------------------------------------
template <int GG, bool LAST, int SINK, bool SLOW>
__device__ __forceinline__ void compute_rows(CPlan &P, CLayer &Lj, const i32x4_h *__restrict__ wp,
                                             unsigned char *lds, int wave, int lane, int b, int xs, int y0, int jl) {
    const int n = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, pitch = P.pitch, R = P.R, TW = P.TW;
    const size_t HW = (size_t)H * W;
    const int xorg = xs - P.H0;
    const int G = SLOW ? Lj.G : GG, KT = SLOW ? Lj.KT : 9, dil = Lj.dil;
    const int off = Lj.off, halo = Lj.halo, c0 = Lj.c0, ntiles = Lj.ntiles, tw0 = wave - Lj.w0, nw = Lj.nw;
    const int nr_in = Lj.nr_in, nr_out = Lj.nr_out, relu = Lj.relu, cout = Lj.cout;
    const bool subsq = Lj.epi == DECNET_EPI_SUBSQ;
    const int gstride = nr_in * pitch * 16;             // bytes between channel groups of the input level
    // B operand: lane (n, q) reads term {hi, mid, lo, hi}[q] of pixel col0 + n
    const int lane_in = Lj.lds_in + ((q == 3 ? 0 : q) * G * nr_in * pitch + n) * 16;
    const int lane_out = Lj.lds_out + n * 16 + q * 4;
    const int tstride = nr_out * pitch * 16;            // bytes between the terms of the output level
    const int off0 = P.L[0].off + P.L[0].dil + 1, s_first = -P.H0 - off0;
    i32x4_h a[9 * GG];
    if (!SLOW) {
#pragma unroll
        for (int st = 0; st < 9 * GG; ++st) a[st] = wp[(size_t)(Lj.woff + st) * 64 + lane];
    }
    // (from the table in LDS: no per-lane indexing of the kernel-argument segment)
    const float *ctab = reinterpret_cast<const float *>(lds + P.lds_tab + MAXG0 * 8 * sizeof(void *)) + 16 * jl;
    float sc0 = ctab[2 * q], sc1 = ctab[2 * q + 1], sh0 = ctab[8 + 2 * q], sh1 = ctab[9 + 2 * q];
    // per-lane constants of the sink (kept in vector registers: they are used by vector instructions only)
    float *outp0 = nullptr, *outp1 = nullptr;
    float mw0 = 0.f, mw1 = 0.f, mw2 = 0.f, ms = 0.f, mb = 0.f, th = 0.f;
    if (LAST) {
        if (SINK == DECNET_SINK_STORE) {
            outp0 = P.out + ((size_t)b * P.cout_last + 2 * q) * HW;
            outp1 = outp0 + HW;
        } else {
            outp0 = P.out + (size_t)b * HW;
        }
        if (SINK == DECNET_SINK_MASK) { mw0 = P.m_w[0]; mw1 = P.m_w[1]; mw2 = P.m_w[2]; ms = P.m_s; mb = P.m_b; th = P.thold; }
    }
    const bool st0 = 2 * q < P.cout_last, st1 = 2 * q + 1 < P.cout_last;
    const int naux = P.naux, lds_aux = P.lds_aux;
    unsigned short *bits16 = P.bits16;
    const int wpr64 = P.wpr * 64, wpr4 = P.wpr * 4, dbg = P.debug;
    // everything loaded so far is complete from here on: the compiler's waitcnt bookkeeping must not make the first
    // MFMA of every row step wait on the vector-memory counter
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!SLOW) {
#pragma unroll
        for (int st = 0; st < 9 * GG; ++st) asm volatile("" : "+v"(a[st]));
    }
    asm volatile("" : "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1));

    for (int s = s_first; s < R; ++s) {
        const int r = s + off, gy = y0 + r;
        const bool act = r >= -halo && r < R + halo;
        const bool row_in = gy >= 0 && gy < H;
        if (act && !(LAST && !row_in) && !(dbg & 128)) {
            int rowoff[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rowoff[ky] = ring_slot(r + (ky - 1) * dil, nr_in) * pitch * 16;
            if (KT == 1) rowoff[0] = rowoff[1];
            const int out_row = lane_out + ring_slot(r, nr_out) * pitch * 16;
            const float *abuf = reinterpret_cast<const float *>(lds + lds_aux) + (s & 1) * naux * pitch;
            const bool rows_ok = r >= 0 && r < R;
            // two tiles per pass: their MFMA chains and epilogues are independent, so the scheduler interleaves them
            // (one wave per SIMD slot issues a dependent instruction only every ~8 cycles)
            auto tile_acc = [&](int t, f32x4_h &acc0, f32x4_h &acc1) {
                const int col0 = c0 + 16 * t;
                if (row_in && !(dbg & 1)) {
                    if (!SLOW) {
                        tile_mfma<GG, 9>(a, lds, lane_in, rowoff, col0, dil, gstride, acc0, acc1);
                    } else {
                        // more channel groups than the register file holds, or a 1x1 layer: weight tiles stream from L1 / L2
                        const i32x4_h *wl = wp + (size_t)Lj.woff * 64 + lane;
                        for (int tap = 0; tap < KT; ++tap) {
                            const int ky = KT == 9 ? tap / 3 : 0, kx = KT == 9 ? tap - 3 * (tap / 3) : 1;
                            const int base = lane_in + (ky == 0 ? rowoff[0] : ky == 1 ? rowoff[1] : rowoff[2]) +
                                             (col0 + (kx - 1) * dil) * 16;
                            for (int c = 0; c < G; ++c) {
                                const i32x4_h bq = *reinterpret_cast<const i32x4_h *>(lds + base + c * gstride);
                                const i32x4_h aw = wl[(size_t)(tap * G + c) * 64];
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_h, aw),
                                                                               __builtin_bit_cast(bf16x8_h, bq), acc0, 0, 0, 0);
                            }
                        }
                    }
                }
            };
            auto tile_epi = [&](int t, const f32x4_h &acc0, const f32x4_h &acc1) {
                const int col0 = c0 + 16 * t;
                // ---- epilogue: lane (n, q) holds output channels 2 q, 2 q + 1 of pixel col0 + n ----
                const int col = col0 + n, gx = xorg + col;
                const bool in = row_in && gx >= 0 && gx < W;
                float v0 = (acc0[0] + acc1[0]) + (acc0[2] + acc1[2]);
                float v1 = (acc0[1] + acc1[1]) + (acc0[3] + acc1[3]);
                v0 = fmaf(v0, sc0, sh0);
                v1 = fmaf(v1, sc1, sh1);
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                if (subsq) {
                    // (aux - v)^2: GenerateSparseMask's torch.pow(cur_fea - pre_fea, 2), submodule.py:369
                    const bool cin = col < pitch;
                    const float a0 = (cin && 2 * q < naux) ? abuf[(2 * q) * pitch + col] : 0.f;
                    const float a1 = (cin && 2 * q + 1 < naux) ? abuf[(2 * q + 1) * pitch + col] : 0.f;
                    v0 = a0 - v0; v0 = v0 * v0;
                    v1 = a1 - v1; v1 = v1 * v1;
                    if (2 * q >= cout) v0 = 0.f;
                    if (2 * q + 1 >= cout) v1 = 0.f;
                }
                if (!in) { v0 = 0.f; v1 = 0.f; }
                if (!LAST) {
                    if (col < pitch && !(dbg & 16)) {
                        int h0, m0, l0, h1, m1, l1;
                        split3(v0, h0, m0, l0);
                        split3(v1, h1, m1, l1);
                        unsigned char *dst = lds + out_row + col0 * 16;
                        *reinterpret_cast<int *>(dst) = pack2(h1, h0);
                        *reinterpret_cast<int *>(dst + tstride) = pack2(m1, m0);
                        *reinterpret_cast<int *>(dst + 2 * tstride) = pack2(l1, l0);
                    }
                } else {
                    const bool st = in && rows_ok && col >= c0 && col < c0 + TW && !(dbg & 8);
                    const int o = gy * W + gx;
                    if (SINK == DECNET_SINK_STORE) {
                        if (st && st0) outp0[o] = v0;
                        if (st && st1) outp1[o] = v1;
                    } else if (SINK == DECNET_SINK_BLEND) {
#pragma clang fp contract(off)
                        // SoftAttention + the fusion of the stage loop (SparseDenseNetRefinementMask.py:195-202):
                        // soft = sigmoid(v); fused = dense * (1 - soft) + soft * sparse.  naux == 1: out = aux + v
                        // (Refinement's disp + res, submodule.py:716)
                        if (st && q == 0) {
                            if (naux == 2) {
                                const float sft = 1.f / (1.f + expf(-v0));
                                const float dn = abuf[col], sp = abuf[pitch + col];
                                const float t1 = dn * (1.f - sft), t2 = sft * sp;
                                outp0[o] = t1 + t2;
                            } else {
                                outp0[o] = abuf[col] + v0;
                            }
                        }
                    } else {
                        // DECNET_SINK_MASK: the 1x1 unit (3 -> 1, folded BatchNorm) of GenerateSparseMask.conv, sigmoid and
                        // threshold (csrc/maskgen.hip's arithmetic); float 0/1 plane + bit-packed copy
                        const float t2v = __shfl(v0, (lane + 16) & 63);         // channel 2 lives on the q = 1 lanes
                        float z = fmaf(mw0, v0, 0.f);
                        z = fmaf(mw1, v1, z);
                        z = fmaf(mw2, t2v, z);
                        z = fmaf(z, ms, mb);
                        const float sg = 1.f / (1.f + expf(-z));
                        const bool vis = row_in && rows_ok && gx >= 0 && gx < W;
                        const bool on = vis && sg > th;
                        if (vis && q == 0) outp0[o] = on ? 1.f : 0.f;
                        const unsigned long long bal = __ballot(on);
                        const int xt = xs + 16 * t;                          // first pixel of the tile
                        if (bits16 && lane == 0 && row_in && rows_ok && xt < wpr64)
                            bits16[((size_t)b * H + gy) * wpr4 + (xt >> 4)] = (unsigned short)(bal & 0xffffull);
                    }
                }
            };
            constexpr bool PAIR = GG == 1 && !SLOW;                // (18+ weight tiles leave no room for a second set)
            for (int t = tw0; t < ntiles; t += (PAIR ? 2 : 1) * nw) {
                f32x4_h p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f}, q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
                const bool two = PAIR && t + nw < ntiles;           // wave-uniform
                tile_acc(t, p0, p1);
                if (two) tile_acc(t + nw, q0, q1);
                tile_epi(t, p0, p1);
                if (two) tile_epi(t + nw, q0, q1);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Loader waves (NLW of the 8): everything that touches HBM on the input side.  They run one row AHEAD with two register
// sets: in step s they issue the loads of row s + 1 + off0 and then split + store row s + off0 (issued one step earlier)
// into the level-0 ring -- so only these waves ever wait for memory, and only for loads that had a whole row step to
// arrive.  The compute waves never execute a vector-memory load inside the loop (stores are fire and forget): in-order
// vmcnt would otherwise make every MFMA block wait for the row loads issued in front of it.
// One instantiation per (source kind, units per lane): the row loop must stay small -- the 64 KB instruction cache is
// shared by the loader and the compute loops of two CUs, and a loop body that spans every source kind x 4 units x 2
// register sets measured 2.3 us per row step with NO loads in it (instruction fetch), against 0.25 us for the barrier.
// ---------------------------------------------------------------------------------------------------------------------
template <int SRC, int KU>
__device__ __forceinline__ void loader_rows(CPlan &P, unsigned char *lds, int lt, int b, int xs, int y0) {
    const int H = P.H, W = P.W, pitch = P.pitch, R = P.R, H0 = P.H0, Wrow = P.Wrow, Wpad = P.Wpad, G0 = P.G0, nr0 = P.nr0;
    const size_t HW = (size_t)H * W;
    const int xorg = xs - H0;
    const int off0 = P.L[0].off + P.L[0].dil + 1, s_first = -H0 - off0;
    const int naux = P.naux, aoff = naux ? P.L[P.aux_layer].off : 0;
    const int nunits = P.ustart[G0];
    const float **ptab = reinterpret_cast<const float **>(lds + P.lds_tab);
    const int tstr = G0 * nr0 * pitch * 16;              // bytes between the terms of level 0
    int u_p[KU], u_sub[KU], u_kind[KU], u_dst[KU], u_c8[KU];
    unsigned u_ok[KU];                                   // bit e: channel e exists
#pragma unroll
    for (int k = 0; k < KU; ++k) {
        const int u = lt + NLW * 64 * k;
        u_p[k] = 0; u_sub[k] = 0; u_kind[k] = -1; u_dst[k] = 0; u_ok[k] = 0; u_c8[k] = 0;
        int c = 0;
        if (u < nunits) {
            for (int cc = 1; cc < G0; ++cc)
                if (u >= P.ustart[cc]) c = cc;
            const int v = u - P.ustart[c];
            u_sub[k] = v / Wpad;
            u_p[k] = v - u_sub[k] * Wpad;
            u_kind[k] = DECNET_PART_PLAIN;
            if (SRC != DECNET_PART_PLAIN) {
#pragma unroll
                for (int g = 0; g < MAXG0; ++g)
                    if (g < G0 && c == g) u_kind[k] = P.gkind[g];
            }
            u_dst[k] = P.lds0 + (c * nr0 * pitch + u_p[k]) * 16;
        }
        u_c8[k] = 8 * c;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (ptab[8 * c + e] != nullptr) u_ok[k] |= 1u << e;
        if (u_kind[k] != DECNET_PART_PLAIN) u_ok[k] = 0;
    }
    int a_ch[MAXKA], a_p[MAXKA];
#pragma unroll
    for (int k = 0; k < MAXKA; ++k) {
        const int v = lt + NLW * 64 * k;
        a_ch[k] = -1; a_p[k] = 0;
        if (v < naux * Wpad) {
            a_ch[k] = v / Wpad;
            a_p[k] = v - a_ch[k] * Wpad;
        }
    }
    float rawA[KU][8], rawB[KU][8], axA[MAXKA], axB[MAXKA];
    const int dbg = P.debug;

    auto issue = [&](float (&raw)[KU][8], float (&ax)[MAXKA], int s) {
        const int r0 = s + off0, gy0 = y0 + r0;
        const bool row_in = r0 >= -H0 && r0 < R + H0 && gy0 >= 0 && gy0 < H;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (SRC != DECNET_PART_PLAIN && u_kind[k] != DECNET_PART_PLAIN) continue;     // wave-uniform
            const int gx = xorg + u_p[k];
            const bool in = row_in && gx >= 0 && gx < W && u_p[k] < Wrow;
            const int o = in ? gy0 * W + gx : 0;
            // a missing channel (tail of the last group) loads channel 0 of its group instead and is zeroed afterwards: the
            // loads of a row are unconditional (the plane pointers come from the table in LDS: no registers held)
            const float *cp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cp[e] = ptab[u_c8[k] + e];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[k][e] = (u_ok[k] && !(dbg & 2)) ? as_global(cp[e] != nullptr ? cp[e] : cp[0])[o] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!in || !((u_ok[k] >> e) & 1)) raw[k][e] = 0.f;
        }
        if (naux) {
            // auxiliary planes of the consuming layer's row one step ahead
            const int ra = s + 1 + aoff, gya = y0 + ra;
            const bool arow = gya >= 0 && gya < H;
#pragma unroll
            for (int k = 0; k < MAXKA; ++k) {
                float v = 0.f;
                if (a_ch[k] >= 0) {
                    const int gx = xorg + a_p[k];
                    if (arow && gx >= 0 && gx < W && a_p[k] < Wrow)
                        v = as_global(P.auxp[a_ch[k]])[(size_t)b * P.auxbs[a_ch[k]] + (size_t)gya * W + gx];
                }
                ax[k] = v;
            }
        }
    };

    auto commit = [&](float (&raw)[KU][8], float (&ax)[MAXKA], int s) {
        const int r0 = s + off0, gy0 = y0 + r0;
        const bool src_active = r0 >= -H0 && r0 < R + H0;
        const bool src_row_in = gy0 >= 0 && gy0 < H;
        if (src_active && !(dbg & 4)) {
            const int slot0 = ring_slot(r0, nr0) * pitch * 16;
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int kind = u_kind[k];
                if (kind < 0 || u_p[k] >= pitch) continue;
                const int p = u_p[k], gx = xorg + p;
                unsigned char *dst = lds + u_dst[k] + slot0;
                const bool in = src_row_in && gx >= 0 && gx < W && p < Wrow;
                if (SRC != DECNET_PART_DECONV || kind != DECNET_PART_DECONV) {
                    float v[8];
                    if (SRC != DECNET_PART_WARP || kind == DECNET_PART_PLAIN) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = raw[k][e];
                    } else {
#pragma clang fp contract(off)
                        // Refinement.get_warped_feats_by_homgrp (submodule.py:719-745), the arithmetic of
                        // csrc/conv2d_small.hip:warp_disparity
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;
                        if (in) {
                            CPart &pt = P.wrp;
                            const float d = as_global(pt.aux)[(size_t)b * HW + (size_t)gy0 * W + gx];
                            const float cx = ((float)gx - d) / ((float)(W - 1.0) / 2.0f) - 1.0f;
                            const float cy = (float)gy0 / ((float)(H - 1.0) / 2.0f) - 1.0f;
                            const float ix = ((cx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((cy + 1.0f) * (float)H - 1.0f) / 2.0f;
                            const float fx = floorf(ix), fy = floorf(iy);
                            const int x0 = (int)fx, yy0 = (int)fy, x1 = x0 + 1, yy1 = yy0 + 1;
                            const float nw = (fx + 1.0f - ix) * (fy + 1.0f - iy), ne = (ix - fx) * (fy + 1.0f - iy);
                            const float sw = (fx + 1.0f - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
                            const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
                            const bool vy0 = (unsigned)yy0 < (unsigned)H, vy1 = (unsigned)yy1 < (unsigned)H;
                            gfp rb = as_global(pt.p) + (size_t)b * pt.c * HW;
                            // clamped taps: every load is issued, invalid ones are dropped by the selects below
                            const int xa = vx0 ? x0 : 0, xb = vx1 ? x1 : 0, ya = vy0 ? yy0 : 0, yb = vy1 ? yy1 : 0;
                            const int o00 = ya * W + xa, o01 = ya * W + xb, o10 = yb * W + xa, o11 = yb * W + xb;
                            float t00[8], t01[8], t10[8], t11[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                gfp rp = rb + (size_t)(e < pt.c ? e : 0) * HW;
                                t00[e] = rp[o00]; t01[e] = rp[o01]; t10[e] = rp[o10]; t11[e] = rp[o11];
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float t = 0.f;
                                if (vy0 && vx0) t += t00[e] * nw;
                                if (vy0 && vx1) t += t01[e] * ne;
                                if (vy1 && vx0) t += t10[e] * sw;
                                if (vy1 && vx1) t += t11[e] * se;
                                v[e] = e < pt.c ? t : 0.f;
                            }
                        }
                    }
                    int hh[8], mm[8], ll[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) split3(v[e], hh[e], mm[e], ll[e]);
                    i32x4_h th, tm, tl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        th[e] = pack2(hh[2 * e + 1], hh[2 * e]);
                        tm[e] = pack2(mm[2 * e + 1], mm[2 * e]);
                        tl[e] = pack2(ll[2 * e + 1], ll[2 * e]);
                    }
                    *reinterpret_cast<i32x4_h *>(dst) = th;
                    *reinterpret_cast<i32x4_h *>(dst + tstr) = tm;
                    *reinterpret_cast<i32x4_h *>(dst + 2 * tstr) = tl;
                } else {
                    // DECNET_PART_DECONV: ConvTranspose2d k = 3, stride 3 of the coarser level (Deconv2dUnit, submodule.py:
                    // 48-87): out[co][y][x] = act(scale * sum_ci pre[ci][y/3][x/3] w[ci][co][y%3][x%3] + shift); this unit =
                    // output channels 4 sub .. 4 sub + 3 of one pixel (the fmaf chain of csrc/conv2d_small.hip:deconv2d_k3s3)
                    CPart &pt = P.dec;
                    const int sub = u_sub[k];
                    float o4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (in) {
                        const int Y = gy0 / 3, X = gx / 3, ky = gy0 - 3 * Y, kx = gx - 3 * X;
                        const int Hc = H / 3, Wc = W / 3;
                        gfp pp = as_global(pt.p) + (size_t)b * pt.cp * Hc * Wc + (size_t)Y * Wc + X;
                        gfp wq = as_global(pt.aux) + (ky * 3 + kx) * 8 + 4 * sub;
                        for (int ci = 0; ci < pt.cp; ++ci) {
                            const float xv = pp[(size_t)ci * Hc * Wc];
                            const float w0 = wq[ci * 72], w1 = wq[ci * 72 + 1], w2 = wq[ci * 72 + 2], w3 = wq[ci * 72 + 3];
                            o4[0] = fmaf(xv, w0, o4[0]); o4[1] = fmaf(xv, w1, o4[1]);
                            o4[2] = fmaf(xv, w2, o4[2]); o4[3] = fmaf(xv, w3, o4[3]);
                        }
                        gfp scp = as_global(pt.sc) + 4 * sub, shp = as_global(pt.sh) + 4 * sub;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o4[e] = fmaf(o4[e], scp[e], shp[e]);
                            if (pt.relu) o4[e] = fmaxf(o4[e], 0.f);
                        }
                    }
                    int hh[4], mm[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split3(o4[e], hh[e], mm[e], ll[e]);
                    *reinterpret_cast<i32x2_h *>(dst + 8 * sub) = i32x2_h{pack2(hh[1], hh[0]), pack2(hh[3], hh[2])};
                    *reinterpret_cast<i32x2_h *>(dst + tstr + 8 * sub) = i32x2_h{pack2(mm[1], mm[0]), pack2(mm[3], mm[2])};
                    *reinterpret_cast<i32x2_h *>(dst + 2 * tstr + 8 * sub) = i32x2_h{pack2(ll[1], ll[0]), pack2(ll[3], ll[2])};
                }
            }
        }
        if (naux) {
            // aux row s + 1 + aoff -> buffer (s + 1) & 1
            float *abuf = reinterpret_cast<float *>(lds + P.lds_aux) + ((s + 1) & 1) * naux * pitch;
#pragma unroll
            for (int k = 0; k < MAXKA; ++k)
                if (a_ch[k] >= 0 && a_p[k] < pitch) abuf[a_ch[k] * pitch + a_p[k]] = ax[k];
        }
    };

    if (dbg & 64) {                                      // timing experiment: barriers only
        for (int s = s_first; s < R; ++s) __syncthreads();
        return;
    }
    issue(rawA, axA, s_first);
    for (int s = s_first; s < R; s += 2) {
        issue(rawB, axB, s + 1);
        commit(rawA, axA, s);
        __syncthreads();
        if (s + 1 < R) {
            issue(rawA, axA, s + 2);
            commit(rawB, axB, s + 1);
            __syncthreads();
        }
    }
}

template <int SINK, int SRC, int OCC>
__global__ __launch_bounds__(THREADS, 2 * OCC) void chain2d(PlanK P_by_value, const i32x4_h *__restrict__ wp) {
    CPlan &P = *(CPlan *)__builtin_amdgcn_kernarg_segment_ptr();      // = P_by_value, at offset 0
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, xs = blockIdx.x * P.TW, y0 = blockIdx.y * P.R;
    const size_t HW = (size_t)P.H * P.W;

    // plane 0 of every PLAIN channel for this workgroup's sample (or null), as a table in LDS
    const float **ptab = reinterpret_cast<const float **>(lds + P.lds_tab);
    if (tid < MAXG0 * 8) {
        const float *c1 = P.chp[tid], *c2 = P.chp2[tid];         // (loads first, then selects of the loaded values)
        const int cc = P.chc[tid];
        const bool second = b >= P.bsplit && c2 != nullptr;
        const float *cp = second ? c2 : c1;
        if (cp != nullptr) cp += (size_t)(second ? b - P.bsplit : b) * cc * HW;
        ptab[tid] = cp;
    }
    // folded scale / shift of every layer: [layer][0..7 scale, 8..15 shift], written with uniform indices
    float *ctab = reinterpret_cast<float *>(lds + P.lds_tab + MAXG0 * 8 * sizeof(void *));
    if (tid == 0) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                ctab[l * 16 + c] = P.L[l].sc[c];
                ctab[l * 16 + 8 + c] = P.L[l].sh[c];
            }
    }
    __syncthreads();

    if (wave >= NCW) {
        const int lt = tid - NCW * 64;                   // 0 .. NLW * 64 - 1
        const int ku = ceil_div_dev(P.ustart[P.G0], NLW * 64);
        if (ku <= 1) loader_rows<SRC, 1>(P, lds, lt, b, xs, y0);
        else if (OCC == 2 || ku == 2) loader_rows<SRC, 2>(P, lds, lt, b, xs, y0);
        else loader_rows<SRC, 4>(P, lds, lt, b, xs, y0);
        return;
    }

    // ================================================= compute waves =================================================
    int j = 0;
#pragma unroll
    for (int l = 1; l < MAXL; ++l)
        if (l < P.NL && wave >= P.L[l].w0) j = l;
    const bool last = j == P.NL - 1;
    const int G = P.L[j].G;
    // one specialised copy of the row loop per (channel groups, inner / last layer): the group count fixes the register
    // file the weights occupy and the unrolled MFMA block, the position fixes the epilogue
    if (P.L[j].KT == 9 && G == 1) {
        if (last) compute_rows<1, true, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
        else compute_rows<1, false, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
    } else if (P.L[j].KT == 9 && G == 2) {
        if (last) compute_rows<2, true, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
        else compute_rows<2, false, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
    } else if (OCC == 1 && P.L[j].KT == 9 && G == 3) {
        if (last) compute_rows<OCC == 1 ? 3 : 1, true, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
        else compute_rows<OCC == 1 ? 3 : 1, false, SINK, false>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
    } else {
        if (last) compute_rows<1, true, SINK, true>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
        else compute_rows<1, false, SINK, true>(P, P.L[j], wp, lds, wave, lane, b, xs, y0, j);
    }
}

inline int round_up(int a, int m) { return (a + m - 1) / m * m; }

}  // namespace

extern "C" {

size_t decnet_chain2d_packed_bytes(int Cin, int k) {
    if (Cin < 1 || (k != 1 && k != 3)) return 0;
    return (size_t)k * k * ceil_div(Cin, 8) * 64 * 16;
}

int decnet_chain2d_pack_weight(const float *w, const float *sign, void *w_packed, int Cin, int Cout, int k,
                               void *stream) {
    if (!w || !w_packed) return DECNET_ERR_NULL_POINTER;
    if (Cin < 1 || Cout < 1 || Cout > 8 || (k != 1 && k != 3)) return DECNET_ERR_UNSUPPORTED;
    const int G = ceil_div(Cin, 8), n = k * k * G * 64;
    hipLaunchKernelGGL(chain2d_pack, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, sign,
                       (i32x4_h *)w_packed, Cin, Cout, k * k, G);
    return decnet_launch_status();
}

// Plans the strips / rows / waves of one chain and launches it.  See decnet_chain2d.h for the descriptor.
int decnet_chain2d_forward(const decnet_chain_desc *d, void *stream) {
    if (!d || !d->out) return DECNET_ERR_NULL_POINTER;
    const int NL = d->n_layers, B = d->B, H = d->H, W = d->W;
    if (NL < 1 || NL > MAXL || d->n_parts < 1 || d->n_parts > MAXPART) return DECNET_ERR_UNSUPPORTED;
    if (B < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    if (B > 65535 || (double)H * W >= 2147483648.0) return DECNET_ERR_UNSUPPORTED;
    PlanK P{};
    PartK part[MAXPART] = {};
    int ndec = 0, nwrp = 0;
    // ---- source parts -> channel groups ----
    int ch = 0;
    for (int i = 0; i < d->n_parts; ++i) {
        const decnet_chain_part &s = d->parts[i];
        if (!s.p) return DECNET_ERR_NULL_POINTER;
        PartK &k = part[i];
        k.p = s.p; k.p2 = s.p2; k.aux = s.aux; k.sc = s.scale; k.sh = s.shift;
        k.c = s.c; k.kind = s.kind; k.cp = s.cp; k.relu = s.relu; k.ch0 = ch;
        if (s.c < 1) return DECNET_ERR_BAD_SHAPE;
        if (s.kind == DECNET_PART_DECONV) {
            if ((ch & 7) || s.c != 8 || !s.aux || !s.scale || !s.shift || s.cp < 1 || H % 3 || W % 3) return DECNET_ERR_UNSUPPORTED;
        } else if (s.kind == DECNET_PART_WARP) {
            if ((ch & 7) || s.c > 8 || !s.aux || H < 2 || W < 2) return DECNET_ERR_UNSUPPORTED;
        } else if (s.kind != DECNET_PART_PLAIN) {
            return DECNET_ERR_UNSUPPORTED;
        }
        const int g0 = ch >> 3;
        if (s.kind != DECNET_PART_PLAIN) {
            if (g0 >= MAXG0) return DECNET_ERR_UNSUPPORTED;
            P.gkind[g0] = s.kind;
            P.gpart[g0] = i;
            if (s.kind == DECNET_PART_DECONV) { P.dec = k; ++ndec; } else { P.wrp = k; ++nwrp; }
            if (ndec > 1 || nwrp > 1 || (ndec && nwrp)) return DECNET_ERR_UNSUPPORTED;
            ch += 8;                                   // a generated part owns its whole group
        } else {
            ch += s.c;
        }
    }
    const int Cin0 = ch, G0 = ceil_div(Cin0, 8);
    if (G0 > MAXG0) return DECNET_ERR_UNSUPPORTED;
    for (int i = 0; i < d->n_parts; ++i)
        if (d->parts[i].kind == DECNET_PART_PLAIN)
            for (int c = 0; c < d->parts[i].c; ++c) {
                const size_t o = (size_t)c * H * W;
                P.chp[part[i].ch0 + c] = d->parts[i].p + o;
                P.chp2[part[i].ch0 + c] = d->parts[i].p2 ? d->parts[i].p2 + o : nullptr;
                P.chc[part[i].ch0 + c] = d->parts[i].c;
            }
    for (int i = 0; i < d->n_parts; ++i)               // a plain part may not share a group with a generated one
        if (d->parts[i].kind == DECNET_PART_PLAIN)
            for (int g = part[i].ch0 >> 3; g <= (part[i].ch0 + part[i].c - 1) >> 3; ++g)
                if (P.gkind[g] != DECNET_PART_PLAIN) return DECNET_ERR_UNSUPPORTED;
    P.nparts = d->n_parts; P.NL = NL; P.G0 = G0; P.bsplit = d->bsplit; P.H = H; P.W = W;
    // ---- layers ----
    int H0 = 0, woff = 0, maxg = 1;
    for (int l = 0; l < NL; ++l) {
        const decnet_chain_layer &s = d->layers[l];
        LayerK &k = P.L[l];
        const int cin = l == 0 ? Cin0 : d->layers[l - 1].cout;
        if (s.cin != cin || s.cout < 1 || s.cout > 8 || (s.k != 1 && s.k != 3) || s.dilation < 1) return DECNET_ERR_UNSUPPORTED;
        if (!s.scale || !s.shift) return DECNET_ERR_NULL_POINTER;
        k.G = ceil_div(cin, 8); k.KT = s.k * s.k; k.dil = s.k == 3 ? s.dilation : 0; k.relu = s.relu; k.cout = s.cout;
        k.epi = s.epilogue; k.aux = s.aux; k.auxc = s.aux_channels; k.woff = woff;
        if (k.epi == DECNET_EPI_SUBSQ && (!s.aux || s.aux_channels < 1)) return DECNET_ERR_NULL_POINTER;
        for (int c = 0; c < 8; ++c) { k.sc[c] = c < s.cout ? s.scale[c] : 0.f; k.sh[c] = c < s.cout ? s.shift[c] : 0.f; }
        woff += k.KT * k.G;
        H0 += k.dil;
        if (k.G <= 3 && k.G > maxg) maxg = k.G;
    }
    if (!d->w_packed) return DECNET_ERR_NULL_POINTER;
    for (int l = NL - 1, h = 0, off = 0; l >= 0; --l) {      // halo behind the layer, row skew
        P.L[l].halo = h; P.L[l].off = off;
        h += P.L[l].dil;
        off += P.L[l].dil + 1;
    }
    P.H0 = H0;
    if (d->sink < 0 || d->sink > DECNET_SINK_MASK) return DECNET_ERR_UNSUPPORTED;
    P.sink = d->sink; P.cout_last = d->layers[NL - 1].cout;
    P.out = d->out; P.bits16 = (unsigned short *)d->bits; P.sa = d->sink_a; P.sb = d->sink_b;
    P.wpr = (W + 63) / 64;
    if (d->sink == DECNET_SINK_BLEND && (!d->sink_a || !d->sink_b || P.cout_last != 1)) return DECNET_ERR_NULL_POINTER;
    if (d->sink == DECNET_SINK_ADD && (!d->sink_a || P.cout_last != 1)) return DECNET_ERR_NULL_POINTER;
    if (d->sink == DECNET_SINK_MASK) {
        if (P.cout_last != 3) return DECNET_ERR_UNSUPPORTED;
        for (int i = 0; i < 3; ++i) P.m_w[i] = d->mask_w[i];
        P.m_s = d->mask_scale; P.m_b = d->mask_shift; P.thold = d->thold;
    }

    // ---- auxiliary planes of an epilogue / sink: loaded by the loader waves into LDS one row ahead ----
    for (int l = 0; l < NL; ++l)
        if (P.L[l].epi == DECNET_EPI_SUBSQ) {
            if (P.naux || P.L[l].auxc > MAXAUX) return DECNET_ERR_UNSUPPORTED;
            P.naux = P.L[l].auxc; P.aux_layer = l;
            for (int c = 0; c < P.naux; ++c) { P.auxp[c] = P.L[l].aux + (size_t)c * H * W; P.auxbs[c] = P.L[l].auxc * H * W; }
        }
    if (d->sink == DECNET_SINK_BLEND || d->sink == DECNET_SINK_ADD) {
        if (P.naux) return DECNET_ERR_UNSUPPORTED;
        P.aux_layer = NL - 1;
        P.auxp[0] = d->sink_a; P.auxbs[0] = H * W; P.naux = 1;
        if (d->sink == DECNET_SINK_BLEND) { P.auxp[1] = d->sink_b; P.auxbs[1] = H * W; P.naux = 2; }
    }

    // ---- strip width: the cheapest candidate that fits LDS and the loader lanes' units per row ----
    const int upg_total = [&] { int u = 0; for (int g = 0; g < G0; ++g) u += P.gkind[g] == DECNET_PART_DECONV ? 2 : 1; return u; }();
    double best = 1e300;
    int bestTW = 0, occ = 1;
    // two workgroups per CU (16 waves, <= 128 registers) unless a layer needs 27 weight tiles in registers or the
    // source is too wide for two units per loader lane; else one
    int want_occ = 2;
    for (int l = 0; l < NL; ++l)
        if (P.L[l].KT == 9 && P.L[l].G == 3) want_occ = 1;
    if (d->debug & 32) want_occ = 1;
    const bool mask = d->sink == DECNET_SINK_MASK;
    const int wgran = (ndec || nwrp) ? 64 : 16;          // generated parts: the unit kind must be wave-uniform
    for (; want_occ >= 1 && !bestTW; --want_occ) {
        const size_t lds_cap = want_occ == 2 ? DECNET_LDS_BYTES / 2 : DECNET_LDS_BYTES;
        const int ku_cap = want_occ == 2 ? 2 : MAXKU;
        for (int tw = mask ? 64 : 16; tw <= 512; tw += mask ? 64 : 4) {
            const int wrow = tw + 2 * H0, wpad = round_up(wrow, wgran), pitch = round_up(wrow + 16, 16);
            if (upg_total * wpad > ku_cap * NLW * 64 || P.naux * wpad > MAXKA * NLW * 64) break;
            size_t lds = (size_t)3 * G0 * (2 * P.L[0].dil + 2) * pitch * 16;
            for (int l = 0; l + 1 < NL; ++l) lds += (size_t)3 * (2 * P.L[l + 1].dil + 2) * pitch * 16;
            if (lds + 512 + MAXG0 * 8 * sizeof(void *) + MAXL * 16 * 4 + (size_t)2 * P.naux * pitch * 4 > lds_cap) break;
            const int strips = ceil_div(W, tw);
            // per row step: the slowest wave group (the split is by MFMA count, see below) + a fixed share for the
            // barrier, the source and the epilogues
            double mf = 0;
            for (int l = 0; l < NL; ++l) mf += (double)ceil_div(tw + 2 * P.L[l].halo, 16) * P.L[l].KT * P.L[l].G;
            const double cost = strips * (mf / NCW * 1.25 + 12.0 + 0.02 * upg_total * wpad);
            if (cost < best) { best = cost; bestTW = tw; }
        }
        if (bestTW) { occ = want_occ; break; }
    }
    if (!bestTW) return DECNET_ERR_UNSUPPORTED;
    if (d->force_tw > 0) bestTW = d->force_tw;
    const int TW = bestTW;
    P.TW = TW; P.Wrow = TW + 2 * H0; P.Wpad = round_up(P.Wrow, wgran); P.pitch = round_up(P.Wrow + 16, 16);
    int us = 0;
    for (int g = 0; g < G0; ++g) { P.ustart[g] = us; us += (P.gkind[g] == DECNET_PART_DECONV ? 2 : 1) * P.Wpad; }
    P.ustart[G0] = us;
    if (us > MAXKU * NLW * 64 || P.naux * P.Wpad > MAXKA * NLW * 64) return DECNET_ERR_UNSUPPORTED;
    // ---- LDS levels ----
    size_t lds = 0;
    P.lds0 = 0; P.nr0 = 2 * P.L[0].dil + 2;
    lds += (size_t)3 * G0 * P.nr0 * P.pitch * 16;
    for (int l = 0; l < NL; ++l) {
        LayerK &k = P.L[l];
        k.lds_in = l == 0 ? P.lds0 : P.L[l - 1].lds_out;
        k.nr_in = l == 0 ? P.nr0 : P.L[l - 1].nr_out;
        if (l + 1 < NL) {
            k.lds_out = (int)lds; k.nr_out = 2 * P.L[l + 1].dil + 2;
            lds += (size_t)3 * k.nr_out * P.pitch * 16;
        } else {
            k.lds_out = 0; k.nr_out = 1;
        }
        k.c0 = H0 - k.halo;
        k.ntiles = ceil_div(TW + 2 * k.halo, 16);
    }
    lds += 512;                                             // the last tile of the last ring row reads past its row
    P.lds_tab = (int)lds;
    lds += MAXG0 * 8 * sizeof(void *) + MAXL * 16 * sizeof(float);
    P.lds_aux = (int)lds;
    lds += (size_t)2 * P.naux * P.pitch * 4;
    if (lds > DECNET_LDS_BYTES) return DECNET_ERR_UNSUPPORTED;
    if (lds > DECNET_LDS_BYTES / 2 || us > 2 * NLW * 64) occ = 1;          // (a pinned strip width may not fit twice)
    // ---- waves per layer: minimise the slowest group's MFMAs per row step ----
    {
        int nw[MAXL] = {1, 1, 1};
        for (int left = NCW - NL; left > 0; --left) {
            int worst = 0;
            double wv = -1;
            for (int l = 0; l < NL; ++l) {
                const double v = (double)ceil_div(P.L[l].ntiles, nw[l]) * P.L[l].KT * P.L[l].G;
                if (v > wv) { wv = v; worst = l; }
            }
            ++nw[worst];
        }
        for (int l = 0, w0 = 0; l < NL; ++l) { P.L[l].w0 = w0; P.L[l].nw = nw[l]; w0 += nw[l]; }
    }
    // ---- rows per workgroup: whole rounds of one workgroup per CU ----
    const int strips = ceil_div(W, TW), warm = 2 * H0 + NL + 2;
    int R = H;
    {
        double bc = 1e300;
        for (int r = 4; r <= H; ++r) {
            const double wgs = (double)strips * B * ceil_div(H, r);
            const double c = ceil(wgs / (256.0 * occ)) * (r + warm);
            if (c < bc - 1e-9) { bc = c; R = r; }
        }
    }
    if (d->force_rows > 0) R = d->force_rows;
    P.debug = d->debug;
    if (d->debug & 256) fprintf(stderr, "chain2d: TW %d R %d strips %d pitch %d lds %zu units %d waves/layer %d %d %d occ %d\n", TW, R, strips, P.pitch, lds, us, P.L[0].nw, P.L[1].nw, P.L[2].nw, occ);
    P.R = R;
    const dim3 grid((unsigned)strips, (unsigned)ceil_div(H, R), (unsigned)B);
    if (grid.y > 65535) return DECNET_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const i32x4_h *wp = (const i32x4_h *)d->w_packed;
#define GO(SK, SR, OC)                                                                                                  \
    do {                                                                                                                \
        if (lds > 64 * 1024) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void *)chain2d<SK, SR, OC>,                                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return (int)e;                                                                         \
        }                                                                                                               \
        hipLaunchKernelGGL((chain2d<SK, SR, OC>), grid, dim3(THREADS), lds, st, P, wp);                                \
    } while (0)
#define GOO(SK, SR)                                                                      \
    do {                                                                                 \
        if (occ == 2) GO(SK, SR, 2);                                                     \
        else GO(SK, SR, 1);                                                              \
    } while (0)
#define GOS(SR)                                                                          \
    do {                                                                                 \
        if (P.sink == DECNET_SINK_STORE) GOO(DECNET_SINK_STORE, SR);                     \
        else if (P.sink == DECNET_SINK_MASK) GOO(DECNET_SINK_MASK, SR);                  \
        else GOO(DECNET_SINK_BLEND, SR);                                                 \
    } while (0)
    if (ndec) GOS(DECNET_PART_DECONV);
    else if (nwrp) GOS(DECNET_PART_WARP);
    else GOS(DECNET_PART_PLAIN);
#undef GOS
#undef GOO
#undef GO
    return decnet_launch_status();
}

}  // extern "C"
