// decnet_amd/csrc/tapconv.hip -- dilated 3x3 / 1x1 convolutions on small images with many channels as a
// per-tap product on the matrix cores + a gather: the ASPP block of FeatExtNetChannelPlus
// (modules/submodule.py:225-241, 216 -> 4 x 216 channels at 1/27 resolution).
//
//   x [B,Ci,H,W]  --decnet_tapconv_to_chunks-->  V [ceil(Ci/16)][P = B*H*W][16]
//   T[t][co][p] = sum_ci x[p][ci] * w_t[co][ci]          decnet_tap_gemm (conv3d_winograd.hip:wino_gemm with
//                                                          one V shared by all taps t of all branches)
//   y[b, br*Co + co, y, x] = act(scale * sum_{t in branch br} T[t][co][p + offset(t, dilation_br)] + shift)
//                                                          decnet_tapconv_gather (taps outside the image skipped)
// All branches of a block read the same input, so one V and one batched GEMM serve them; the
// concatenated NCHW output is written directly.  The library runs each dilated branch as its own
// implicit GEMM at ~25 TFLOP/s on these 5760-pixel images.
#include "common.h"

namespace {

constexpr int T_BN = 224;          // co rows per tap in U^T (= W_BN of conv3d_winograd.hip)

__global__ __launch_bounds__(256) void nchw_to_chunks(const float *__restrict__ x, float *__restrict__ V, int C,
                                                      int HW, int P) {
    // thread = (position p, 16-channel chunk kc), p fastest: 16 coalesced plane reads, one 64-byte store
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int KC = (C + 15) >> 4;
    if (idx >= (size_t)P * KC) return;
    const int p = (int)(idx % P), kc = (int)(idx / P);
    const int b = p / HW, i = p - b * HW;
    const float *xp = x + ((size_t)b * C + kc * 16) * HW + i;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = kc * 16 + j < C ? xp[(size_t)j * HW] : 0.f;
    float4 *o = reinterpret_cast<float4 *>(V + ((size_t)kc * P + p) * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

// w [Co][Ci][kk] -> U^T [tap0 + t][ceil(Ci/16)][224][16]
__global__ void pack_tap_weights(const float *__restrict__ w, float *__restrict__ U, int Co, int Ci, int kk,
                                 int tap0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;           // (t, co, ci)
    if (idx >= kk * T_BN * Ci) return;
    const int ci = idx % Ci, co = (idx / Ci) % T_BN, t = idx / (Ci * T_BN);
    const int KC = (Ci + 15) >> 4;
    U[(((size_t)(tap0 + t) * KC + (ci >> 4)) * T_BN + co) * 16 + (ci & 15)] =
        co < Co ? w[((size_t)co * Ci + ci) * kk + t] : 0.f;
}

struct Branches {
    int n, tap0[4], kk[4], dil[4];
};

// block = 64 consecutive positions x one 16-channel group x one branch
__global__ __launch_bounds__(256) void tap_gather(const float *__restrict__ M, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, float *__restrict__ y,
                                                  Branches br, int B, int Co, int H, int W, int relu) {
    __shared__ float tile[16][65];
    const int P = B * H * W, HW = H * W, CG = (Co + 15) >> 4;
    const int p0 = blockIdx.x * 64, cg = blockIdx.y, ib = blockIdx.z;
    const int co_lo = threadIdx.x & 15, pq = threadIdx.x >> 4;        // 16 positions per pass
    const int kk = br.kk[ib], dil = br.dil[ib], k = kk == 9 ? 3 : 1;
    // all of a thread's requests (4 positions x up to 9 taps) before the first use -- round 5, from the ISA: `if (inside)
    // acc += M[..]` was one memory round trip per tap and pass, 36 in a row.  A tap outside the image (or past kk, or a
    // position past P) reads the position's own entry of the branch's first tap and adds 0: same sum, same order.
    float tv[4][9];
    bool ok[4][9];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pl = pass * 16 + pq, p = p0 + pl, pc = p < P ? p : P - 1;
        const int b = pc / HW, i = pc - b * HW, yy = i / W, xx = i - yy * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = (t / k - k / 2) * dil, dx = (t % k - k / 2) * dil;
            const int y2 = yy + dy, x2 = xx + dx;
            ok[pass][t] = t < kk && p < P && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W;
            tv[pass][t] = ok[pass][t] ? M[(((size_t)(br.tap0[ib] + t) * CG + cg) * P + (pc + dy * W + dx)) * 16 + co_lo]
                                      : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += ok[pass][t] ? tv[pass][t] : 0.f;
        tile[co_lo][pass * 16 + pq] = acc;
    }
    __syncthreads();
    const int co = cg * 16 + (threadIdx.x >> 4);
    if (co >= Co) return;
    const float sc = scale[ib * Co + co], sh = shift[ib * Co + co];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pl = pass * 16 + (threadIdx.x & 15), p = p0 + pl;
        if (p >= P) continue;
        const int b = p / HW, i = p - b * HW;
        float v = fmaf(tile[threadIdx.x >> 4][pl], sc, sh);
        if (relu) v = fmaxf(v, 0.f);
        y[((size_t)b * br.n * Co + (size_t)ib * Co + co) * HW + i] = v;
    }
}

}  // namespace

extern "C" {

size_t decnet_tapconv_chunk_floats(int B, int Ci, int H, int W) {
    if (B < 1 || Ci < 1 || H < 1 || W < 1) return 0;
    return (size_t)((Ci + 15) / 16) * 16 * B * H * W;
}

int decnet_tapconv_to_chunks(const float *x, float *V, int B, int Ci, int H, int W, void *stream) {
    if (!x || !V) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Ci < 1 || H < 1 || W < 1 || (double)B * H * W * Ci >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
    const size_t n = (size_t)B * H * W * ((Ci + 15) / 16);
    hipLaunchKernelGGL(nchw_to_chunks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, V, Ci,
                       H * W, B * H * W);
    return decnet_launch_status();
}

size_t decnet_tapconv_weight_floats(int Ci, int ntaps) {
    // fp32 U^T per tap + its bf16-term copy ([pair of chunks][3 terms][224][16] words, decnet_tapconv_split_weight)
    const size_t kc = (Ci + 15) / 16;
    return Ci < 1 || ntaps < 1 ? 0 : (size_t)ntaps * (kc * 16 * T_BN + ((kc + 1) / 2) * 3 * T_BN * 16);
}

/* one branch: w [Co,Ci,k,k] (k = 1 or 3) -> taps tap0 .. tap0 + k*k - 1 of u */
int decnet_tapconv_pack_weight(const float *w, float *u, int Co, int Ci, int k, int tap0, void *stream) {
    if (!w || !u) return DECNET_ERR_NULL_POINTER;
    if (Co < 1 || Ci < 1 || tap0 < 0 || (k != 1 && k != 3)) return DECNET_ERR_BAD_SHAPE;
    if (Co > T_BN) return DECNET_ERR_UNSUPPORTED;
    const int n = k * k * T_BN * Ci;
    hipLaunchKernelGGL(pack_tap_weights, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, u, Co, Ci, k * k,
                       tap0);
    return decnet_launch_status();
}

/* nbranch <= 4 branches of Co channels each; branch i uses taps tap0[i] .. + k[i]*k[i] - 1 with dilation
 * dil[i]; M from decnet_tap_gemm; scale, shift [nbranch*Co]; y [B, nbranch*Co, H, W]. */
int decnet_tapconv_gather(const float *M, const float *scale, const float *shift, float *y, int B, int Co, int H,
                          int W, int nbranch, const int *tap0, const int *k, const int *dil, int relu,
                          void *stream) {
    if (!M || !scale || !shift || !y || !tap0 || !k || !dil) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || Co < 1 || H < 1 || W < 1 || nbranch < 1) return DECNET_ERR_BAD_SHAPE;
    if (nbranch > 4 || (double)B * H * W >= 2147483648.0 / 64) return DECNET_ERR_UNSUPPORTED;
    Branches br{};
    br.n = nbranch;
    for (int i = 0; i < nbranch; ++i) {
        if ((k[i] != 1 && k[i] != 3) || dil[i] < 1 || tap0[i] < 0) return DECNET_ERR_BAD_SHAPE;
        br.tap0[i] = tap0[i]; br.kk[i] = k[i] * k[i]; br.dil[i] = dil[i];
    }
    const int P = B * H * W;
    hipLaunchKernelGGL(tap_gather, dim3((unsigned)ceil_div(P, 64), (unsigned)((Co + 15) / 16), (unsigned)nbranch),
                       dim3(256), 0, (hipStream_t)stream, M, scale, shift, y, br, B, Co, H, W, relu);
    return decnet_launch_status();
}

}  // extern "C"
