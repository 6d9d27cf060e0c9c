// decnet_amd/csrc/conv2d_mfma_acc2.hip -- conv2d_mfma.hip with TWO accumulator sets: the accuracy option of the bf16x3
// 2-D trunk (DECNET_CONV2D_ACC=2, read once per process by conv2d_mfma.hip, which forwards its entry points to the
// *_acc2 functions here; the packed weight format differs, so the switch covers packing and launching alike).
// Round 4 measured it (tools/experiments then): error against float64 0.58 x the one-accumulator kernel's, i.e. below an
// fp32 fma chain's, for 20 - 50 % of the layer time.  Round 5 ships it as the supported way to reference-grade fp32 in the
// many-channel layers (the alternative was DECNET_CONV2D_MFMA=0: the library's kernels).
// decnet_amd/csrc/conv2d_mfma.hip -- the many-channel Conv2dUnit layers of the 2-D trunk on the matrix cores.
//
// Replaces (eval mode) the library convolution behind
//   * FeatureExtraction conv1.* / conv2.* / conv3_2.* and the Deconv2dBlock convs   submodule.py:245-343, 162-178
//   * DynamicUpsampling.weight_learning (73/217/649 -> 81 -> 81 -> 81)               submodule.py:566-589
//   * Refinement convs at 24..72 channels                                            submodule.py:690-717
// i.e. Conv2d k = 3 (any dilation, padding = dilation) or k = 1, stride 1, followed by the folded BatchNorm and
// ReLU of Conv2dUnit.forward (submodule.py:15-45), optionally on the channel concatenation of up to six tensors.
//
// Arithmetic: implicit GEMM  Y[pixel][co] = sum_tap sum_ci X[pixel + tap][ci] * Wt[tap][ci][co]  on
// v_mfma_f32_16x16x32_bf16 at fp32 accuracy: every fp32 operand is split into three bf16 terms
// x = hi + mid + lo (round to nearest with exact residuals, 24 mantissa bits together) and the six products above 2^-24
// (hh hm mh mm hl lh) go into the K axis of three MFMAs per 16 input channels (8-channel halves 0 / 1, k groups
// q = lane >> 4):
//   j = 0 : A {h0 h0 h1 h1} x B {h0 m0 h1 m1}   -> acc   (hh, hm)
//   j = 1 : A {m0 m0 m1 m1} x the SAME B tile   -> small (mh, mm)
//   j = 2 : A {h0 l0 h1 l1} x B {l0 h0 l1 h1}   -> small (hl, lh)
// against 4 x v_mfma_f32_16x16x4_f32 of 32 cycles each for the same 16 channels: 3 x 16 cycles.
// Two accumulator sets (round 4).  The instruction rounds its fp32 accumulator after each of its four k groups
// (tools/ubench/bf16x3_grouping.hip: the error of a long sum follows the NUMBER of k groups added into the big accumulator,
// whatever their size), so all three MFMAs into one accumulator are 12 roundings per 16 channels at the sum's own size
// -- 1.16 x the error of an fp32 fma chain behind a ReLU (8 roundings), which is what tests/test_inputdata_gpu.py saw as
// 1.2 - 1.3 x the reference's distance to its float64 run.  Here only j = 0 meets the big accumulator (4 roundings); the
// ten small term groups (<= 2^-9 of it) collect in `small` and are added once at the end.  j = 0 and j = 1 share their
// weight tile: two tiles per (chunk, tap) instead of three.
//
// Tiling: a wave owns TM rows x 16 columns of output pixels x TNW x 16 output channels (TM x TNW accumulator tiles, twice);
// a workgroup is 4 NH waves: 4 row groups x NH channel halves over ONE halo tile of the input (NH = 2: 512 threads for
// layers of 49 - 96 output channels -- the second accumulator set halves the channel tiles a wave can hold, the second
// wave group keeps the staging work per output where it was).  Per 16-channel chunk the halo tile is split once into its
// bf16 terms and kept in LDS as [term][8-channel group][pixel][8 x bf16] (one ds_read_b128 per lane = one A operand; the
// k groups q = 0 / 1 and 2 / 3 of an operand are served in the same LDS cycles (MI355X_MICROARCH.md, LDS), so they read
// the same plane or planes 4 apart: 4 P x 16 bytes is a multiple of 256 for every tile shape); the weights are split at
// packing time and streamed from L2 as ready-made B operands (one 16-byte load per lane), each used for TM (j = 2) or
// 2 TM (j = 0, 1) MFMAs.
// Epilogue: relu((acc + small) * scale + shift) -> NCHW, 64-byte runs per (channel, row).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


namespace {

constexpr int MAXSEG = 6;
constexpr int THREADS = 256;

struct Segs {
    const float *p[MAXSEG];
    int c[MAXSEG];
    int n;
};

// output-channel tiles (of 16) per workgroup (2, 3: one wave group; 4, 6: two wave groups of 2 / 3 tiles), and the padded
// tile count of a layer
__host__ __device__ inline int pick_tn(int Cout) {
    const int nt = (Cout + 15) >> 4;
    if (nt <= 2) return 2;
    if (nt <= 4) return nt;
    if (nt <= 6) return 6;
    const int p6 = (nt + 5) / 6 * 6, p4 = (nt + 3) / 4 * 4;
    return p4 < p6 ? 4 : 6;
}
__host__ __device__ inline int padded_nt(int Cout) {
    const int tn = pick_tn(Cout), nt = (Cout + 15) >> 4;
    return (nt + tn - 1) / tn * tn;
}

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

// eight values -> the three packed operand registers sets {hi, mid, lo}[4] (pairs (x[2e], x[2e+1]) share a register):
// the same terms as split3, two per v_cvt_pk_bf16_f32 -- the packed pair IS the operand register, and a term's float
// value is a shift / a mask of it (no v_perm_b32, half the conversions, packed subtractions)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pack_bf16(float a, float b) {
    return __builtin_bit_cast(int, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void split3x8(const float (&x)[8], i32x4 &th, i32x4 &tm, i32x4 &tl) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = x[2 * e], x1 = x[2 * e + 1];
        const int h = pack_bf16(x0, x1);
        const float r0 = x0 - __int_as_float(h << 16), r1 = x1 - __int_as_float(h & 0xffff0000);
        const int m = pack_bf16(r0, r1);
        th[e] = h;
        tm[e] = m;
        tl[e] = pack_bf16(r0 - __int_as_float(m << 16), r1 - __int_as_float(m & 0xffff0000));
    }
}

// w [Cout][Cin][KT] -> wp[chunk][tap][X | Y][n tile][lane] (16 bytes: the lane's 8 bf16 of the B operand)
//   X = {h0 m0 h1 m1} (j = 0, 1)   Y = {l0 h0 l1 h1} (j = 2);   last chunk with <= 8 real channels (tail8; its j = 1 is
//   skipped): X = {h0 m0 h0 m0} against A {h0 h0 m0 m0}, Y = {l0 h0 0 0} against A {h0 l0 - -}
// tr: w is a ConvTranspose2d weight [Cin][Cout / 9][3][3] read as the 1 x 1 convolution to n = 9 co + 3 ky + kx
__global__ void conv2d_mfma_pack(const float *__restrict__ w, i32x4 *__restrict__ wp, int Cin, int Cout, int KT,
                                 int NT, long total, int tr, int nchunk, int tail8) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int nt = (int)(rest % NT); rest /= NT;
    const int xy = (int)(rest % 2); rest /= 2;
    const int tap = (int)(rest % KT);
    const int ck = (int)(rest / KT);
    const int n = nt * 16 + (lane & 15), q = lane >> 4;
    int term, grp;                                             // term 0 / 1 / 2 = hi / mid / lo, -1 = zero
    if (tail8 && ck == nchunk - 1) {
        grp = 0;
        term = xy == 0 ? (q & 1) : (q == 0 ? 2 : q == 1 ? 0 : -1);
    } else {
        grp = q >> 1;
        term = xy == 0 ? (q & 1) : ((q & 1) ? 0 : 2);
    }
    int t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 16 * ck + 8 * grp + e;
        const float v = (term >= 0 && n < Cout && c < Cin) ? (tr ? w[(size_t)c * Cout + n] : w[((size_t)n * Cin + c) * KT + tap])
                                                           : 0.f;
        int h, m, l;
        split3(v, h, m, l);
        t[e] = term == 0 ? h : term == 1 ? m : l;
    }
    i32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_perm(t[2 * e + 1], t[2 * e], 0x07060302);
    wp[idx] = o;
}

// Epilogue of both kernels: the lane holds output channel n = tile * 16 + r of the pixels x0 + 4 q .. + 3 of TM rows.
// shuf = 0: y [B,Cout,H,W] = act(acc * scale[n] + shift[n]).  shuf = C (transposed convolution k = 3, stride 3, as a
// 1 x 1 convolution to 9 C channels n = 9 co + 3 ky + kx): y [B,C,3H,3W] at (3 row + ky, 3 x + kx), scale / shift per co.
template <int TM, int TN>
__device__ __forceinline__ void conv2d_mfma_store(const f32x4 (&acc)[TM][TN], const float *__restrict__ scale,
                                                  const float *__restrict__ shift, float *__restrict__ y, int b, int Cout,
                                                  int H, int W, int relu, int nt0, int r, int q, int x0, int row0,
                                                  int shuf) {
    const size_t HW = (size_t)H * W;
    const int xq = x0 + 4 * q;
    const bool vec = (W & 3) == 0 && xq + 3 < W;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        const int n = (nt0 + nt) * 16 + r;
        if (n >= Cout) continue;
        if (shuf) {
            const int co = n / 9, kk = n - 9 * co, ky = kk / 3, kx = kk - 3 * ky;
            const float sc = scale[co], sh = shift[co];
            float *yp = y + ((size_t)b * shuf + co) * 9 * HW;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = row0 + mt;
                if (row >= H) break;
                float *dst = yp + ((size_t)(3 * row + ky) * 3 * W) + kx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = fmaf(acc[mt][nt][i], sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    if (xq + i < W) dst[3 * (xq + i)] = v;
                }
            }
            continue;
        }
        const float sc = scale[n], sh = shift[n];
        float *yp = y + ((size_t)b * Cout + n) * HW;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int row = row0 + mt;
            if (row >= H) break;
            f32x4 v = acc[mt][nt];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = fmaf(v[i], sc, sh);
                if (relu) v[i] = fmaxf(v[i], 0.f);
            }
            float *dst = yp + (size_t)row * W + xq;
            if (vec) {
                *reinterpret_cast<f32x4 *>(dst) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xq + i < W) dst[i] = v[i];
            }
        }
    }
}

// ---- the MFMAs of one 16-channel chunk for one wave ---------------------------------------------------------------------
// bx / by: the X / Y weight tiles of the current tap (all TNW columns resident); bx is re-loaded for the next tap behind its
// last use (j = 1; j = 0 in the tail chunk): at least TM TNW MFMAs ahead, by behind j = 2: two steps ahead.  The pixel
// operands a[] of the next step are re-read from LDS behind the last column of this one.
template <int TM, int TNW>
struct Acc {
    f32x4 big[TM][TNW], small[TM][TNW];
    i32x4 bx[TNW], by[TNW];
};

// A operand planes ([term][8-channel group]) per k group q of a lane, steps j = 0, 1, 2 and the tail chunk's j = 0, 2
__device__ __forceinline__ void operand_offsets(int lane, int P, int (&offA)[3], int (&offT)[2]) {
    const int r = lane & 15, q = lane >> 4, g = q >> 1;
    offA[0] = (0 * 2 + g) * P + r;                            // {h0 h0 h1 h1}
    offA[1] = (1 * 2 + g) * P + r;                            // {m0 m0 m1 m1}
    offA[2] = (((q & 1) ? 2 : 0) * 2 + g) * P + r;            // {h0 l0 h1 l1}
    offT[0] = (g * 2 + 0) * P + r;                            // {h0 h0 m0 m0}
    offT[1] = (((q & 1) ? 2 : 0) * 2 + 0) * P + r;            // {h0 l0 h0 l0} (the weights of q = 2, 3 are zero)
}

template <int TM, int TNW>
__device__ __forceinline__ void chunk_mfma(Acc<TM, TNW> &R, const i32x4 *__restrict__ cur, const i32x4 *__restrict__ &wb,
                                           int wblk, const int (&offA)[3], const int (&offT)[2], int rowbase, int PW, int KT,
                                           int dil, bool tail) {
    auto mf = [](const i32x4 &a, const i32x4 &b, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    const int o0 = tail ? offT[0] : offA[0], o2 = tail ? offT[1] : offA[2];
    i32x4 a[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) a[mt] = cur[o0 + (rowbase + mt) * PW];
#pragma unroll 1
    for (int tap = 0; tap < KT; ++tap) {
        const int tn_ = tap + 1 < KT ? tap + 1 : 0;
        const int tyn = KT == 9 ? tn_ / 3 : 0, txn = KT == 9 ? tn_ - 3 * tyn : 0;
        const int ty = KT == 9 ? tap / 3 : 0, tx = KT == 9 ? tap - 3 * ty : 0;
        const int tapoff = (rowbase + ty * dil) * PW + tx * dil;
        const int tapoff_n = (rowbase + tyn * dil) * PW + txn * dil;
        const i32x4 *wn = wb + 2 * wblk;                       // the next tap's (or chunk's) X tiles; Y = + wblk
        // ---- j = 0: (hh, hm) -> big
        {
            const int nxt = (tail ? o2 : offA[1]) + tapoff;
#pragma unroll
            for (int nt = 0; nt < TNW; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    R.big[mt][nt] = mf(a[mt], R.bx[nt], R.big[mt][nt]);
                    if (nt == TNW - 1) a[mt] = cur[nxt + mt * PW];
                }
                if (tail) R.bx[nt] = wn[nt * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- j = 1: (mh, mm) -> small, the same weight tiles (not in the tail chunk: its channels 8 - 15 are padding and
        // the four products of the channels 0 - 7 that are not small went into j = 0)
        if (!tail) {
            const int nxt = o2 + tapoff;
#pragma unroll
            for (int nt = 0; nt < TNW; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    R.small[mt][nt] = mf(a[mt], R.bx[nt], R.small[mt][nt]);
                    if (nt == TNW - 1) a[mt] = cur[nxt + mt * PW];
                }
                R.bx[nt] = wn[nt * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- j = 2: (hl, lh) -> small
        {
            const int nxt = o0 + tapoff_n;
#pragma unroll
            for (int nt = 0; nt < TNW; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    R.small[mt][nt] = mf(a[mt], R.by[nt], R.small[mt][nt]);
                    if (nt == TNW - 1) a[mt] = cur[nxt + mt * PW];
                }
                R.by[nt] = wn[wblk + nt * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wb = wn;
    }
}

// NU: staging units per thread, 128 NH NU >= pixels of the halo tile.  NH: wave groups (channel halves) per workgroup
template <int TM, int TNW, int NH, int NU>
__global__ __launch_bounds__(THREADS * NH, NH == 1 ? 2 : 1) void conv2d_mfma(
    Segs in, const i32x4 *__restrict__ wp, const float *__restrict__ scale, const float *__restrict__ shift,
    float *__restrict__ y, int Cout, int H, int W, int KT, int dil, int relu, int nchunk, int NT, int tiles_x,
    int tail8, int shuf) {
    constexpr int TH = 4 * TM, SGT = 128 * NH;                 // staging threads per 8-channel half
    extern __shared__ i32x4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pad = KT == 9 ? dil : 0;
    const int PW = 16 + 2 * pad, PH = TH + 2 * pad, P = PW * PH;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int y0 = tyi * TH, x0 = txi * 16;
    const int nt0 = (blockIdx.y * NH + (wave >> 2)) * TNW;     // this wave's first channel tile
    const int b = blockIdx.z;
    const size_t HW = (size_t)H * W;

    const int r = lane & 15, q = lane >> 4;
    int offA[3], offT[2];
    operand_offsets(lane, P, offA, offT);
    const int rowbase = (wave & 3) * TM;
    const bool wave_active = y0 + rowbase < H && nt0 * 16 < Cout;

    Acc<TM, TNW> R;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TNW; ++nt) R.big[mt][nt] = R.small[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wblk = NT * 64;                                  // X -> Y of a (chunk, tap); 2 wblk per (chunk, tap)
    const i32x4 *wb = wp + (size_t)nt0 * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < TNW; ++nt) { R.bx[nt] = wb[nt * 64]; R.by[nt] = wb[wblk + nt * 64]; }

    // ---- staging: the first half of the threads the channels 0-7 of a chunk, the second half the channels 8-15; a thread
    // owns the pixels (tid % SGT) + SGT u of the halo tile.  issue(): global loads of a chunk into registers; commit():
    // split + LDS
    const int sg = tid / SGT, st = tid - sg * SGT;
    int po[NU];                                                // pixel offset inside a channel plane, -1: zero
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int p = st + SGT * u;
        const int py = p / PW, px = p - py * PW;
        const int gy = y0 - pad + py, gx = x0 - pad + px;
        po[u] = (p < P && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    float raw[NU][8];
    auto issue = [&](int ck) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ck + 8 * sg + e;
            const float *cp = nullptr;
            int base = 0;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                if (s < in.n) {
                    if (c >= base && c < base + in.c[s]) cp = in.p[s] + ((size_t)b * in.c[s] + (c - base)) * HW;
                    base += in.c[s];
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float v = 0.f;
                if (cp != nullptr && po[u] >= 0) v = cp[po[u]];
                raw[u][e] = v;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int p = st + SGT * u;
            if (p >= P) break;
            i32x4 th, tm, tl;
            split3x8(raw[u], th, tm, tl);
            smem[(0 * 2 + sg) * P + p] = th;
            smem[(1 * 2 + sg) * P + p] = tm;
            smem[(2 * 2 + sg) * P + p] = tl;
        }
    };

    issue(0);
    commit();
    __syncthreads();
    for (int ck = 0; ck < nchunk; ++ck) {
        const bool more = ck + 1 < nchunk;
        if (more) issue(ck + 1);                               // in flight during this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (wave_active) chunk_mfma<TM, TNW>(R, smem, wb, wblk, offA, offT, rowbase, PW, KT, dil, !more && tail8);
        __syncthreads();                                       // the tile has been read by every wave
        if (more) {
            commit();
            __syncthreads();
        }
    }

    if (wave_active) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int nt = 0; nt < TNW; ++nt) R.big[mt][nt] += R.small[mt][nt];
        conv2d_mfma_store<TM, TNW>(R.big, scale, shift, y, b, Cout, H, W, relu, nt0, r, q, x0, y0 + rowbase, shuf);
    }
}

// ---- producer / consumer variant ---------------------------------------------------------------------------------------
// One workgroup per CU: waves 0 .. 4 NH - 1 issue only weight loads and MFMAs, the last four waves only stage (global loads
// of the next chunk's halo tile, bf16 split, LDS stores into the other of two tiles); the split's VALU work and the loads'
// latency run beside the matrix pipe instead of in front of it (in the kernel above the in-order vmcnt makes the first
// weight tile after issue() wait for the pixel loads too: measured 0.371 ms on the 81 -> 81 layer at 180 x 324 against
// 0.28 ms with staging compiled out).  One barrier per chunk.
template <int TM, int TNW, int NH, int NU>
__global__ __launch_bounds__(THREADS * NH + THREADS, 1) void conv2d_mfma_pc(
    Segs in, const i32x4 *__restrict__ wp, const float *__restrict__ scale, const float *__restrict__ shift,
    float *__restrict__ y, int Cout, int H, int W, int KT, int dil, int relu, int nchunk, int NT, int tiles_x,
    int tail8, int shuf) {
    constexpr int TH = 4 * TM;
    extern __shared__ i32x4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pad = KT == 9 ? dil : 0;
    const int PW = 16 + 2 * pad, PH = TH + 2 * pad, P = PW * PH;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int y0 = tyi * TH, x0 = txi * 16;
    const int b = blockIdx.z;
    const size_t HW = (size_t)H * W;
    const int tile_units = 6 * P;

    if (wave >= 4 * NH) {
        // ================= staging waves: the first two the channels 0-7 of a chunk, the other two the channels 8-15 ======
        const int lt = tid - 4 * NH * 64, sg = lt >> 7;
        int po[NU];                                            // pixel offset inside a channel plane, -1: zero
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int p = (lt & 127) + 128 * u;
            const int py = p / PW, px = p - py * PW;
            const int gy = y0 - pad + py, gx = x0 - pad + px;
            po[u] = (p < P && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
        }
        for (int ck = 0; ck <= nchunk; ++ck) {
            if (ck < nchunk) {
                float raw[NU][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 16 * ck + 8 * sg + e;
                    const float *cp = nullptr;
                    int base = 0;
#pragma unroll
                    for (int s = 0; s < MAXSEG; ++s) {
                        if (s < in.n) {
                            if (c >= base && c < base + in.c[s]) cp = in.p[s] + ((size_t)b * in.c[s] + (c - base)) * HW;
                            base += in.c[s];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        float v = 0.f;
                        if (cp != nullptr && po[u] >= 0) v = cp[po[u]];
                        raw[u][e] = v;
                    }
                }
                i32x4 *tile = smem + (ck & 1) * tile_units;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int p = (lt & 127) + 128 * u;
                    if (p >= P) break;
                    i32x4 th, tm, tl;
                    split3x8(raw[u], th, tm, tl);
                    tile[(0 * 2 + sg) * P + p] = th;
                    tile[(1 * 2 + sg) * P + p] = tm;
                    tile[(2 * 2 + sg) * P + p] = tl;
                }
            }
            // barrier ck: tile ck is complete, and the MFMA waves have finished with tile ck - 1 (= tile ck + 1's place)
            __syncthreads();
        }
        return;
    }

    // ================= MFMA waves =================
    const int nt0 = (blockIdx.y * NH + (wave >> 2)) * TNW;
    const int r = lane & 15, q = lane >> 4;
    int offA[3], offT[2];
    operand_offsets(lane, P, offA, offT);
    const int rowbase = (wave & 3) * TM;
    const bool wave_active = y0 + rowbase < H && nt0 * 16 < Cout;

    Acc<TM, TNW> R;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TNW; ++nt) R.big[mt][nt] = R.small[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wblk = NT * 64;
    const i32x4 *wb = wp + (size_t)nt0 * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < TNW; ++nt) { R.bx[nt] = wb[nt * 64]; R.by[nt] = wb[wblk + nt * 64]; }

    __syncthreads();                                           // barrier 0: tile 0
    for (int ck = 0; ck < nchunk; ++ck) {
        const i32x4 *cur = smem + (ck & 1) * tile_units;
        if (wave_active) chunk_mfma<TM, TNW>(R, cur, wb, wblk, offA, offT, rowbase, PW, KT, dil, ck + 1 == nchunk && tail8);
        __syncthreads();                                       // barrier ck + 1
    }

    if (wave_active) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int nt = 0; nt < TNW; ++nt) R.big[mt][nt] += R.small[mt][nt];
        conv2d_mfma_store<TM, TNW>(R.big, scale, shift, y, b, Cout, H, W, relu, nt0, r, q, x0, y0 + rowbase, shuf);
    }
}

// pixels of the halo tile of a TM variant; staging units per thread
constexpr int tile_pixels(int tm, int pad) { return (16 + 2 * pad) * (4 * tm + 2 * pad); }
constexpr int staging_units(int tm, int nh, bool pc, bool dilated) {
    return ((dilated ? 640 : tile_pixels(tm, 1)) + (pc ? 128 : 128 * nh) - 1) / (pc ? 128 : 128 * nh);
}
// two accumulator sets + pixel operands + two weight tile sets (+ staging registers) within 256 registers
constexpr bool fits(int tm, int tnw, int nh, bool pc, bool dilated = false) {
    return 2 * tm * tnw * 4 + tm * 4 + 2 * tnw * 4 + (pc ? 0 : 8 * staging_units(tm, nh, pc, dilated)) + 36 <= 256;
}
// Rows per wave.  A launch costs (rounds of resident workgroups) x (TM + a fixed share for prologue, staging and
// epilogue); e.g. H = 180: TM = 5 gives 9 exact row tiles.  One workgroup per CU except the 256-thread kernel (two).
inline int pick_tm(int B, int H, int W, int nchunkN, int tnw, int nh, int pad, bool pc) {
    const char *env = getenv("DECNET_CONV2D_MFMA_TM");                     // tests / experiments: pin the tile height
    const int forced = env ? atoi(env) : 0;
    static const int cand[5] = {8, 6, 5, 4, 2};
    int best = 2;
    double best_cost = 1e30;
    for (int i = 0; i < 5; ++i) {
        const int tm = cand[i];
        if (!fits(tm, tnw, nh, pc)) continue;
        if (pad > 1 && tm > 4) continue;
        if (forced == tm) return tm;
        const double wgs = (double)ceil_div(W, 16) * ceil_div(H, 4 * tm) * B * nchunkN;
        const double cost = ceil(wgs / (pc || nh == 2 ? 256.0 : 512.0)) * (tm + (pc ? 0.4 : 0.7));
        if (cost < best_cost) { best_cost = cost; best = tm; }
    }
    return best;
}

template <int TM, int TNW, int NH, bool DIL, bool PC>
int launch(const Segs &in, const i32x4 *wp, const float *scale, const float *shift, float *y, int B, int Cout, int H,
           int W, int KT, int dil, int relu, int nchunk, int NT, int shuf, hipStream_t stream) {
    constexpr int NU = staging_units(TM, NH, PC, DIL);
    const int pad = KT == 9 ? dil : 0;
    const size_t lds = (size_t)tile_pixels(TM, pad) * 6 * 16 * (PC ? 2 : 1);
    if (lds > DECNET_LDS_BYTES || tile_pixels(TM, pad) > NU * (PC ? 128 : 128 * NH)) return DECNET_ERR_UNSUPPORTED;
    int cin = 0;
    for (int i = 0; i < in.n; ++i) cin += in.c[i];
    const int tail8 = cin - 16 * (nchunk - 1) <= 8;              // the last chunk's channels 8-15 are padding
    const int tiles_x = ceil_div(W, 16), tiles_y = ceil_div(H, 4 * TM);
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(NT / (TNW * NH)), (unsigned)B);
    // more than 64 KiB of dynamic LDS needs the attribute; set per launch (it is per device, and cheap)
    if constexpr (PC) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)conv2d_mfma_pc<TM, TNW, NH, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv2d_mfma_pc<TM, TNW, NH, NU>), grid, dim3(THREADS * NH + THREADS), lds, stream, in, wp, scale,
                           shift, y, Cout, H, W, KT, dil, relu, nchunk, NT, tiles_x, tail8, shuf);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)conv2d_mfma<TM, TNW, NH, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv2d_mfma<TM, TNW, NH, NU>), grid, dim3(THREADS * NH), lds, stream, in, wp, scale, shift, y,
                           Cout, H, W, KT, dil, relu, nchunk, NT, tiles_x, tail8, shuf);
    }
    return decnet_launch_status();
}

template <int TNW, int NH, bool PC>
int launch_tm(int tm, const Segs &in, const i32x4 *wp, const float *scale, const float *shift, float *y, int B,
              int Cout, int H, int W, int KT, int dil, int relu, int nchunk, int NT, int shuf, hipStream_t stream) {
    const int pad = KT == 9 ? dil : 0;
#define ARGS in, wp, scale, shift, y, B, Cout, H, W, KT, dil, relu, nchunk, NT, shuf, stream
    if (pad <= 1) {
        if constexpr (fits(8, TNW, NH, PC)) {
            if (tm == 8) return launch<8, TNW, NH, false, PC>(ARGS);
        }
        if constexpr (fits(6, TNW, NH, PC)) {
            if (tm >= 6) return launch<6, TNW, NH, false, PC>(ARGS);
        }
        if constexpr (fits(5, TNW, NH, PC)) {
            if (tm >= 5) return launch<5, TNW, NH, false, PC>(ARGS);
        }
        if (tm >= 4) return launch<4, TNW, NH, false, PC>(ARGS);
        return launch<2, TNW, NH, false, PC>(ARGS);
    }
    // dilated taps: bigger halo, staging sized for 640 pixels
    if (tm >= 4 && tile_pixels(4, pad) <= 640) return launch<4, TNW, NH, true, PC>(ARGS);
    if (tile_pixels(2, pad) <= 640) return launch<2, TNW, NH, true, PC>(ARGS);
#undef ARGS
    return DECNET_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

size_t decnet_conv2d_mfma_packed_bytes_acc2(int Cin, int Cout, int k) {
    if (Cin < 1 || Cout < 1 || (k != 1 && k != 3)) return 0;
    const size_t blocks = (size_t)ceil_div(Cin, 16) * (k * k) * 2 + 2;       // + 2: the prefetch runs one (chunk, tap) ahead
    return blocks * padded_nt(Cout) * 64 * 16;
}

static int pack_impl(const float *w, void *w_packed, int Cin, int Cout, int k, int tr, void *stream) {
    if (!w || !w_packed) return DECNET_ERR_NULL_POINTER;
    const size_t bytes = decnet_conv2d_mfma_packed_bytes_acc2(Cin, Cout, k);
    if (!bytes) return DECNET_ERR_UNSUPPORTED;
    const int NT = padded_nt(Cout);
    const int nchunk = ceil_div(Cin, 16);
    const long total = (long)nchunk * (k * k) * 2 * NT * 64;
    hipError_t e = hipMemsetAsync((char *)w_packed + (size_t)total * 16, 0, bytes - (size_t)total * 16,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv2d_mfma_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (i32x4 *)w_packed, Cin, Cout, k * k, NT, total, tr, nchunk, Cin - 16 * (nchunk - 1) <= 8 ? 1 : 0);
    return decnet_launch_status();
}

int decnet_conv2d_mfma_pack_weight_acc2(const float *w, void *w_packed, int Cin, int Cout, int k, void *stream) {
    return pack_impl(w, w_packed, Cin, Cout, k, 0, stream);
}

size_t decnet_deconv2d_mfma_packed_bytes_acc2(int Cin, int Cout) {
    if (Cout < 1 || Cout > 7281) return 0;
    return decnet_conv2d_mfma_packed_bytes_acc2(Cin, 9 * Cout, 1);
}

int decnet_deconv2d_mfma_pack_weight_acc2(const float *w, void *w_packed, int Cin, int Cout, void *stream) {
    if (Cout < 1 || Cout > 7281) return DECNET_ERR_UNSUPPORTED;
    return pack_impl(w, w_packed, Cin, 9 * Cout, 1, 1, stream);
}

static int run_impl(const Segs &in, long Cin, const void *w_packed, const float *scale, const float *shift, float *y,
                    int B, int Cout, int H, int W, int k, int dilation, int relu, int shuf, void *stream) {
    if (B > 65535 || Cin > 65536 || (double)H * W >= 2147483648.0 / (shuf ? 9 : 1)) return DECNET_ERR_UNSUPPORTED;
    const int TN = pick_tn(Cout), NT = padded_nt(Cout), nchunk = ceil_div((int)Cin, 16);
    if (NT / TN > 65535 || (double)ceil_div(W, 16) * ceil_div(H, 8) >= 2.0e9) return DECNET_ERR_UNSUPPORTED;
    // the staging-in-line kernel when its grid fills the chip at least once, else the producer / consumer kernel
    // (two wave groups: measured 217 -> 81 at 60 x 108, 448 workgroups: 0.156 vs 0.104 ms; 81 -> 81 at 180 x 324,
    // 1512 workgroups: 0.371 vs 0.390 ms)
    const int pad = k == 3 ? dilation : 0;
    const int nh = TN >= 4 ? 2 : 1, tnw = TN / nh;
    int tm = pick_tm(B, H, W, NT / TN, tnw, nh, pad, false);
    const char *env = getenv("DECNET_CONV2D_MFMA_PC");                      // tests / experiments: 0 / 1 pins the kernel
    bool pc = nh == 2 && (double)ceil_div(W, 16) * ceil_div(H, 4 * tm) * B * (NT / TN) <= 512.0;
    if (env && nh == 2) pc = atoi(env) != 0;
    if (pc) tm = pick_tm(B, H, W, NT / tnw, tnw, 1, pad, true);
    const i32x4 *wp = (const i32x4 *)w_packed;
    hipStream_t st = (hipStream_t)stream;
#define GO(T, N, P) \
    return launch_tm<T, N, P>(tm, in, wp, scale, shift, y, B, Cout, H, W, k * k, dilation, relu, nchunk, NT, shuf, st)
    switch (TN) {
        case 2: GO(2, 1, false);
        case 3: GO(3, 1, false);
        // (producer / consumer: one wave group per workgroup -- 12 waves would cap a wave at 168 registers -- and the
        // channel halves as separate workgroups: these are the launches that do not fill the chip anyway)
        case 4: if (pc) GO(2, 1, true); else GO(2, 2, false);
        case 6: if (pc) GO(3, 1, true); else GO(3, 2, false);
    }
#undef GO
    return DECNET_ERR_UNSUPPORTED;
}

int decnet_conv2d_mfma_cat_bn_act_acc2(const float *const *xs, const int *cins, int nseg, const void *w_packed,
                                  const float *scale, const float *shift, float *y, int B, int Cout, int H, int W,
                                  int k, int dilation, int relu, void *stream) {
    if (!xs || !cins || !w_packed || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (nseg < 1 || nseg > MAXSEG || (k != 1 && k != 3)) return DECNET_ERR_UNSUPPORTED;
    if (B < 1 || Cout < 1 || H < 1 || W < 1 || dilation < 1) return DECNET_ERR_BAD_SHAPE;
    Segs in{};
    long Cin = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!xs[i]) return DECNET_ERR_NULL_POINTER;
        if (cins[i] < 1) return DECNET_ERR_BAD_SHAPE;
        in.p[i] = xs[i];
        in.c[i] = cins[i];
        Cin += cins[i];
    }
    in.n = nseg;
    return run_impl(in, Cin, w_packed, scale, shift, y, B, Cout, H, W, k, dilation, relu, 0, stream);
}

int decnet_deconv2d_mfma_k3s3_bn_act_acc2(const float *x, const void *w_packed, const float *scale, const float *shift,
                                     float *y, int B, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (!x || !w_packed || !scale || !shift || !y) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return DECNET_ERR_BAD_SHAPE;
    if (Cout > 7281) return DECNET_ERR_UNSUPPORTED;
    Segs in{};
    in.p[0] = x;
    in.c[0] = Cin;
    in.n = 1;
    return run_impl(in, Cin, w_packed, scale, shift, y, B, 9 * Cout, H, W, 1, 1, relu, Cout, stream);
}

}  // extern "C"
