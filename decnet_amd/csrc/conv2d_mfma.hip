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
// x = hi + mid + lo (truncations with exact residuals, 24 mantissa bits together) and the six products above 2^-24
// (hi.hi hi.mid mid.hi mid.mid hi.lo lo.hi) go into the K axis of three MFMAs per 16 input channels:
//   j = 0 / 1 : channels 0-7 / 8-15,  A k-groups {hi,hi,mid,mid} x B k-groups {hi,mid,hi,mid}
//   j = 2     : A {hi(0-7), lo(0-7), hi(8-15), lo(8-15)} x B {lo(0-7), hi(0-7), lo(8-15), hi(8-15)}
// against 4 x v_mfma_f32_16x16x4_f32 of 32 cycles each for the same 16 channels: 3 x 16 cycles.
//
// Tiling: a 256-thread workgroup owns 4 TM rows x 16 columns of output pixels x (up to) TN x 16 output channels; wave
// w owns rows w TM .. w TM + TM - 1, one 16-pixel MFMA tile per row, TM x TN accumulator tiles.  Per 16-channel
// chunk the input halo tile is split once into its bf16 terms and kept in LDS as [term][8-channel group][pixel][8 x
// bf16] (one ds_read_b128 per lane = one A operand, conflict free); the weights are split at packing time and
// streamed from L2 as ready-made B operands (one 16-byte load per lane), each used for TM MFMAs.
// Epilogue: relu(acc * scale + shift) -> NCHW, 64-byte runs per (channel, row).
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

// output-channel tiles (of 16) per workgroup, and the padded tile count of a layer
__host__ __device__ inline int pick_tn(int Cout) {
    const int nt = (Cout + 15) >> 4;
    return nt <= 2 ? 2 : nt <= 3 ? 3 : nt == 4 ? 4 : nt <= 6 ? nt : 5;
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

// w [Cout][Cin][KT] -> wp[chunk][tap][j][n tile][lane] (16 bytes: the lane's 8 bf16 of the B operand)
// tr: w is a ConvTranspose2d weight [Cin][Cout / 9][3][3] read as the 1 x 1 convolution to n = 9 co + 3 ky + kx
__global__ void conv2d_mfma_pack(const float *__restrict__ w, i32x4 *__restrict__ wp, int Cin, int Cout, int KT,
                                 int NT, long total, int tr) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long rest = idx >> 6;
    const int nt = (int)(rest % NT); rest /= NT;
    const int j = (int)(rest % 3); rest /= 3;
    const int tap = (int)(rest % KT);
    const int ck = (int)(rest / KT);
    const int n = nt * 16 + (lane & 15), b = lane >> 4;
    const int term = j < 2 ? (b & 1) : ((b & 1) ? 0 : 2);
    const int grp = j < 2 ? j : (b >> 1);
    int t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 16 * ck + 8 * grp + e;
        const float v = (n < Cout && c < Cin) ? (tr ? w[(size_t)c * Cout + n] : w[((size_t)n * Cin + c) * KT + tap]) : 0.f;
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

// NU: staging units per thread, 128 NU >= pixels of the halo tile
template <int TM, int TN, int NU>
__global__ __launch_bounds__(THREADS, 2) void conv2d_mfma(
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
    const int nt0 = blockIdx.y * TN;
    const int b = blockIdx.z;
    const size_t HW = (size_t)H * W;

    const int r = lane & 15, q = lane >> 4;
    // A operand of MFMA j: this lane's (term, channel group) plane of the LDS tile
    int offA[3];
    offA[0] = ((q >> 1) * 2 + 0) * P + r;
    offA[1] = ((q >> 1) * 2 + 1) * P + r;
    offA[2] = (((q & 1) ? 2 : 0) * 2 + (q >> 1)) * P + r;
    const int rowbase = wave * TM;
    const bool wave_active = y0 + rowbase < H;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const i32x4 *wb = wp + (size_t)nt0 * 64 + lane;          // advances NT * 64 per (chunk, tap, j)
    const int wstep = NT * 64;
    i32x4 bq[TN];                                              // B operands of the current step
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) bq[nt] = wb[nt * 64];

    // ---- staging: waves 0-1 the channels 0-7 of a chunk, waves 2-3 the channels 8-15; a thread owns the pixels
    // (tid & 127) + 128 u of the halo tile.  issue(): global loads of a chunk into registers; commit(): split + LDS
    const int sg = wave >> 1;
    int po[NU];                                                // pixel offset inside a channel plane, -1: zero
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int p = (tid & 127) + 128 * u;
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
            const int p = (tid & 127) + 128 * u;
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
        const bool skip1 = !more && tail8;
        if (more) issue(ck + 1);   // in flight during this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (wave_active) {
            // every operand tile is re-loaded for the NEXT (tap, j) step right behind its last MFMA of this one: the B
            // tiles (L2 latency) have most of a step to arrive, the A tiles (LDS) the TM MFMAs of the last column
            i32x4 a[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) a[mt] = smem[offA[0] + rowbase * PW + mt * PW];
            for (int tap = 0; tap < KT; ++tap) {
                const int tn_ = tap + 1 < KT ? tap + 1 : 0;
                const int tyn = KT == 9 ? tn_ / 3 : 0, txn = KT == 9 ? tn_ - 3 * tyn : 0;
                const int ty = KT == 9 ? tap / 3 : 0, tx = KT == 9 ? tap - 3 * ty : 0;
                const int tapoff = (rowbase + ty * dil) * PW + tx * dil;
                const int tapoff_n = (rowbase + tyn * dil) * PW + txn * dil;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    // last chunk with <= 8 real channels: its j = 1 step (channels 8-15) is all padding
                    if (j == 1 && skip1) continue;
                    wb += (j == 0 && skip1) ? 2 * wstep : wstep;   // -> the next step's tiles
                    const int nxt = j == 0 ? offA[skip1 ? 2 : 1] + tapoff : j == 1 ? offA[2] + tapoff : offA[0] + tapoff_n;
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[mt]), __builtin_bit_cast(bf16x8, bq[nt]), acc[mt][nt], 0, 0, 0);
                            if (nt == TN - 1) a[mt] = smem[nxt + mt * PW];
                        }
                        bq[nt] = wb[nt * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __syncthreads();             // the tile has been read by every wave
        if (more) {
            commit();
            __syncthreads();
        }
    }

    if (wave_active)
        conv2d_mfma_store<TM, TN>(acc, scale, shift, y, b, Cout, H, W, relu, nt0, r, q, x0, y0 + rowbase, shuf);
}

// ---- producer / consumer variant (TN >= 4) ---------------------------------------------------------------------------
// One 512-thread workgroup per CU: waves 0-3 issue only weight loads and MFMAs, waves 4-7 only stage (global loads of
// the next chunk's halo tile, bf16 split, LDS stores into the other of two tiles); a SIMD holds one wave of each kind,
// so the split's VALU work and the loads' latency run beside the matrix pipe instead of in front of it (in the
// 4-wave kernel above the in-order vmcnt makes the first weight tile after issue() wait for the pixel loads too:
// measured 0.371 ms on the 81 -> 81 layer at 180 x 324 against 0.28 ms with staging compiled out).  One barrier per
// chunk.  The weight tiles of the three steps (j) of a tap live in a ring of 3 x TN operand registers and are
// re-loaded three steps (~1400 cycles) ahead.
template <int TM, int TN, int NU>
__global__ __launch_bounds__(2 * THREADS, 1) void conv2d_mfma_pc(
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
    const int nt0 = blockIdx.y * TN;
    const int b = blockIdx.z;
    const size_t HW = (size_t)H * W;
    const int tile_units = 6 * P;

    if (wave >= 4) {
        // ================= staging waves: 4-5 the channels 0-7 of a chunk, 6-7 the channels 8-15 =================
        const int lt = tid - 4 * 64, sg = (wave - 4) >> 1;
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
    const int r = lane & 15, q = lane >> 4;
    int offA[3];
    offA[0] = ((q >> 1) * 2 + 0) * P + r;
    offA[1] = ((q >> 1) * 2 + 1) * P + r;
    offA[2] = (((q & 1) ? 2 : 0) * 2 + (q >> 1)) * P + r;
    const int rowbase = wave * TM;
    const bool wave_active = y0 + rowbase < H;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const i32x4 *wb = wp + (size_t)nt0 * 64 + lane;          // advances NT * 64 per (chunk, tap, j)
    const int wstep = NT * 64;
    i32x4 bq[3][TN];                                           // weight tiles of the steps j = 0, 1, 2 of a tap
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) bq[j][nt] = wb[j * wstep + nt * 64];

    __syncthreads();                                           // barrier 0: tile 0
    for (int ck = 0; ck < nchunk; ++ck) {
        const i32x4 *cur = smem + (ck & 1) * tile_units;
        const bool skip1 = ck + 1 == nchunk && tail8;
        if (wave_active) {
            i32x4 a[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) a[mt] = cur[offA[0] + rowbase * PW + mt * PW];
            for (int tap = 0; tap < KT; ++tap) {
                const int tn_ = tap + 1 < KT ? tap + 1 : 0;
                const int tyn = KT == 9 ? tn_ / 3 : 0, txn = KT == 9 ? tn_ - 3 * tyn : 0;
                const int ty = KT == 9 ? tap / 3 : 0, tx = KT == 9 ? tap - 3 * ty : 0;
                const int tapoff = (rowbase + ty * dil) * PW + tx * dil;
                const int tapoff_n = (rowbase + tyn * dil) * PW + txn * dil;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j == 1 && skip1) {                             // all-padding step of the last chunk
                        wb += wstep;
                        continue;
                    }
                    const int nxt = j == 0 ? offA[skip1 ? 2 : 1] + tapoff : j == 1 ? offA[2] + tapoff : offA[0] + tapoff_n;
                    // columns 0 .. TN-G-1 one at a time (weight tile re-loaded behind its TM MFMAs), the last G columns
                    // row by row so that a pixel tile's re-load has (TM - 1) G MFMAs to land before the next step
                    constexpr int G = TN < 3 ? TN : 3;
#pragma unroll
                    for (int nt = 0; nt < TN - G; ++nt) {
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[mt]), __builtin_bit_cast(bf16x8, bq[j][nt]), acc[mt][nt], 0,
                                0, 0);
                        bq[j][nt] = wb[3 * wstep + nt * 64];   // same j of the next tap (3 blocks of padding at the end)
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
                        for (int nt = TN - G; nt < TN; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[mt]), __builtin_bit_cast(bf16x8, bq[j][nt]), acc[mt][nt], 0,
                                0, 0);
                        a[mt] = cur[nxt + mt * PW];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int nt = TN - G; nt < TN; ++nt) bq[j][nt] = wb[3 * wstep + nt * 64];
                    __builtin_amdgcn_sched_barrier(0);
                    wb += wstep;
                }
            }
        }
        __syncthreads();                                       // barrier ck + 1
    }

    if (wave_active)
        conv2d_mfma_store<TM, TN>(acc, scale, shift, y, b, Cout, H, W, relu, nt0, r, q, x0, y0 + rowbase, shuf);
}

// Rows per wave.  A launch costs (rounds of resident workgroups) x (TM + a fixed share for prologue, staging and
// epilogue); e.g. H = 180: TM = 5 gives 9 exact row tiles.  pc: the producer / consumer kernel (one workgroup per CU).
constexpr bool fits(int tm, int tn, bool pc) {                       // accumulators + operands within 256 registers
    if (pc) return tm * tn * 4 + tm * 4 + tn * 12 + 30 <= 256;
    return tm * tn <= 36 || (tm == 8 && tn <= 4);
}
inline int pick_tm(int B, int H, int W, int nchunkN, int tn, int pad, bool pc) {
    const char *env = getenv("DECNET_CONV2D_MFMA_TM");                     // tests / experiments: pin the tile height
    const int forced = env ? atoi(env) : 0;
    static const int cand[5] = {8, 6, 5, 4, 2};
    int best = 2;
    double best_cost = 1e30;
    for (int i = 0; i < 5; ++i) {
        const int tm = cand[i];
        if (!fits(tm, tn, pc)) continue;
        if (pad > 1 && tm > 4) continue;
        if (forced == tm) return tm;
        const double wgs = (double)ceil_div(W, 16) * ceil_div(H, 4 * tm) * B * nchunkN;
        const double cost = ceil(wgs / (pc ? 256.0 : 512.0)) * (tm + (pc ? 0.4 : 0.7));
        if (cost < best_cost) { best_cost = cost; best = tm; }
    }
    return best;
}

template <int TM, int TN, int NU, bool PC>
int launch(const Segs &in, const i32x4 *wp, const float *scale, const float *shift, float *y, int B, int Cout, int H,
           int W, int KT, int dil, int relu, int nchunk, int NT, int shuf, hipStream_t stream) {
    const int pad = KT == 9 ? dil : 0;
    const size_t lds = (size_t)(16 + 2 * pad) * (4 * TM + 2 * pad) * 6 * 16 * (PC ? 2 : 1);
    if (lds > DECNET_LDS_BYTES) return DECNET_ERR_UNSUPPORTED;
    int cin = 0;
    for (int i = 0; i < in.n; ++i) cin += in.c[i];
    const int tail8 = cin - 16 * (nchunk - 1) <= 8;              // the last chunk's channels 8-15 are padding
    const int tiles_x = ceil_div(W, 16), tiles_y = ceil_div(H, 4 * TM);
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(NT / TN), (unsigned)B);
    // more than 64 KiB of dynamic LDS needs the attribute; set per launch (it is per device, and cheap)
    if constexpr (PC) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)conv2d_mfma_pc<TM, TN, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv2d_mfma_pc<TM, TN, NU>), grid, dim3(2 * THREADS), lds, stream, in, wp, scale, shift, y,
                           Cout, H, W, KT, dil, relu, nchunk, NT, tiles_x, tail8, shuf);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)conv2d_mfma<TM, TN, NU>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((conv2d_mfma<TM, TN, NU>), grid, dim3(THREADS), lds, stream, in, wp, scale, shift, y, Cout, H,
                           W, KT, dil, relu, nchunk, NT, tiles_x, tail8, shuf);
    }
    return decnet_launch_status();
}

// pixels of the halo tile of a TM variant
inline int tile_pixels(int tm, int pad) { return (16 + 2 * pad) * (4 * tm + 2 * pad); }

template <int TN, bool PC>
int launch_tm(int tm, const Segs &in, const i32x4 *wp, const float *scale, const float *shift, float *y, int B,
              int Cout, int H, int W, int KT, int dil, int relu, int nchunk, int NT, int shuf, hipStream_t stream) {
    const int pad = KT == 9 ? dil : 0;
#define ARGS in, wp, scale, shift, y, B, Cout, H, W, KT, dil, relu, nchunk, NT, shuf, stream
    if (pad <= 1) {
        if constexpr (fits(8, TN, PC)) {
            if (tm == 8) return launch<8, TN, 5, PC>(ARGS);
        }
        if constexpr (fits(6, TN, PC)) {
            if (tm == 6) return launch<6, TN, 4, PC>(ARGS);
        }
        if (tm == 5) return launch<5, TN, 4, PC>(ARGS);
        if (tm == 4) return launch<4, TN, 3, PC>(ARGS);
        return launch<2, TN, 2, PC>(ARGS);
    }
    // dilated taps: bigger halo, 5 staging units per thread (640 pixels)
    if (tm >= 4 && tile_pixels(4, pad) <= 640) return launch<4, TN, 5, PC>(ARGS);
    if (tile_pixels(2, pad) <= 640) return launch<2, TN, 5, PC>(ARGS);
#undef ARGS
    return DECNET_ERR_UNSUPPORTED;
}

}  // namespace

// ---- DECNET_CONV2D_ACC=2: every entry point below forwards to conv2d_mfma_acc2.hip (two accumulator sets) ----------
extern "C" {
size_t decnet_conv2d_mfma_packed_bytes_acc2(int Cin, int Cout, int k);
int decnet_conv2d_mfma_pack_weight_acc2(const float *w, void *w_packed, int Cin, int Cout, int k, void *stream);
size_t decnet_deconv2d_mfma_packed_bytes_acc2(int Cin, int Cout);
int decnet_deconv2d_mfma_pack_weight_acc2(const float *w, void *w_packed, int Cin, int Cout, void *stream);
int decnet_conv2d_mfma_cat_bn_act_acc2(const float *const *xs, const int *cins, int nseg, const void *w_packed,
                                       const float *scale, const float *shift, float *y, int B, int Cout, int H, int W,
                                       int k, int dilation, int relu, void *stream);
int decnet_deconv2d_mfma_k3s3_bn_act_acc2(const float *x, const void *w_packed, const float *scale, const float *shift,
                                          float *y, int B, int Cin, int Cout, int H, int W, int relu, void *stream);
}
static bool acc2_mode() {
    static const bool on = [] { const char *e = getenv("DECNET_CONV2D_ACC"); return e && atoi(e) == 2; }();
    return on;
}

extern "C" {

size_t decnet_conv2d_mfma_packed_bytes(int Cin, int Cout, int k) {
    if (acc2_mode()) return decnet_conv2d_mfma_packed_bytes_acc2(Cin, Cout, k);
    if (Cin < 1 || Cout < 1 || (k != 1 && k != 3)) return 0;
    const size_t blocks = (size_t)ceil_div(Cin, 16) * (k * k) * 3 + 3;       // + 3: the prefetch runs three blocks ahead
    return blocks * padded_nt(Cout) * 64 * 16;
}

static int pack_impl(const float *w, void *w_packed, int Cin, int Cout, int k, int tr, void *stream) {
    if (!w || !w_packed) return DECNET_ERR_NULL_POINTER;
    const size_t bytes = decnet_conv2d_mfma_packed_bytes(Cin, Cout, k);
    if (!bytes) return DECNET_ERR_UNSUPPORTED;
    const int NT = padded_nt(Cout);
    const long total = (long)ceil_div(Cin, 16) * (k * k) * 3 * NT * 64;
    hipError_t e = hipMemsetAsync((char *)w_packed + (size_t)total * 16, 0, bytes - (size_t)total * 16,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv2d_mfma_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (i32x4 *)w_packed, Cin, Cout, k * k, NT, total, tr);
    return decnet_launch_status();
}

int decnet_conv2d_mfma_pack_weight(const float *w, void *w_packed, int Cin, int Cout, int k, void *stream) {
    if (acc2_mode()) return decnet_conv2d_mfma_pack_weight_acc2(w, w_packed, Cin, Cout, k, stream);
    return pack_impl(w, w_packed, Cin, Cout, k, 0, stream);
}

size_t decnet_deconv2d_mfma_packed_bytes(int Cin, int Cout) {
    if (acc2_mode()) return decnet_deconv2d_mfma_packed_bytes_acc2(Cin, Cout);
    if (Cout < 1 || Cout > 7281) return 0;
    return decnet_conv2d_mfma_packed_bytes(Cin, 9 * Cout, 1);
}

int decnet_deconv2d_mfma_pack_weight(const float *w, void *w_packed, int Cin, int Cout, void *stream) {
    if (acc2_mode()) return decnet_deconv2d_mfma_pack_weight_acc2(w, w_packed, Cin, Cout, stream);
    if (Cout < 1 || Cout > 7281) return DECNET_ERR_UNSUPPORTED;
    return pack_impl(w, w_packed, Cin, 9 * Cout, 1, 1, stream);
}

static int run_impl(const Segs &in, long Cin, const void *w_packed, const float *scale, const float *shift, float *y,
                    int B, int Cout, int H, int W, int k, int dilation, int relu, int shuf, void *stream) {
    if (B > 65535 || Cin > 65536 || (double)H * W >= 2147483648.0 / (shuf ? 9 : 1)) return DECNET_ERR_UNSUPPORTED;
    const int TN = pick_tn(Cout), NT = padded_nt(Cout), nchunk = ceil_div((int)Cin, 16);
    if (NT / TN > 65535 || (double)ceil_div(W, 16) * ceil_div(H, 8) >= 2.0e9) return DECNET_ERR_UNSUPPORTED;
    // the 4-wave kernel (two workgroups per CU) when its grid fills the chip at least once, else the 8-wave
    // producer / consumer kernel (TN >= 4: measured 217 -> 81 at 60 x 108, 448 workgroups: 0.156 vs 0.104 ms;
    // 81 -> 81 at 180 x 324, 1512 workgroups: 0.371 vs 0.390 ms)
    const int pad = k == 3 ? dilation : 0;
    int tm = pick_tm(B, H, W, NT / TN, TN, pad, false);
    const char *env = getenv("DECNET_CONV2D_MFMA_PC");                      // tests / experiments: 0 / 1 pins the kernel
    bool pc = TN >= 4 && (double)ceil_div(W, 16) * ceil_div(H, 4 * tm) * B * (NT / TN) <= 512.0;
    if (env && TN >= 4) pc = atoi(env) != 0;
    if (pc) tm = pick_tm(B, H, W, NT / TN, TN, pad, true);
    const i32x4 *wp = (const i32x4 *)w_packed;
    hipStream_t st = (hipStream_t)stream;
#define GO(T, P) \
    return launch_tm<T, P>(tm, in, wp, scale, shift, y, B, Cout, H, W, k * k, dilation, relu, nchunk, NT, shuf, st)
    switch (TN) {
        case 2: GO(2, false);
        case 3: GO(3, false);
        case 4: if (pc) GO(4, true); else GO(4, false);
        case 5: if (pc) GO(5, true); else GO(5, false);
        case 6: if (pc) GO(6, true); else GO(6, false);
    }
#undef GO
    return DECNET_ERR_UNSUPPORTED;
}

int decnet_conv2d_mfma_cat_bn_act(const float *const *xs, const int *cins, int nseg, const void *w_packed,
                                  const float *scale, const float *shift, float *y, int B, int Cout, int H, int W,
                                  int k, int dilation, int relu, void *stream) {
    if (acc2_mode())
        return decnet_conv2d_mfma_cat_bn_act_acc2(xs, cins, nseg, w_packed, scale, shift, y, B, Cout, H, W, k, dilation, relu,
                                                  stream);
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

int decnet_deconv2d_mfma_k3s3_bn_act(const float *x, const void *w_packed, const float *scale, const float *shift,
                                     float *y, int B, int Cin, int Cout, int H, int W, int relu, void *stream) {
    if (acc2_mode())
        return decnet_deconv2d_mfma_k3s3_bn_act_acc2(x, w_packed, scale, shift, y, B, Cin, Cout, H, W, relu, stream);
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
