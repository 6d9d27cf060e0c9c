// tools/ubench/bf16x3_grouping.hip -- does the GROUPING of the six bf16x3 products inside v_mfma_f32_16x16x32_bf16
// matter for accuracy?  x = hi + mid + lo (round to nearest), products kept: hh hm mh mm hl lh.  Per 16 channels that is
// 12 k-groups of 8 products = 3 MFMAs; which four groups share an instruction is free:
//   grouping A (shipped in conv2d_mfma / wino_gemm_bf16x3): j0 = {hh mh hm mm}(ch 0-7), j1 = the same (ch 8-15),
//                                                           j2 = {hl lh}(0-7), {hl lh}(8-15): magnitudes 1 .. 2^-18 mixed
//   grouping B: j0 = {hh0 hh1 hm0 hm1}, j1 = {mh0 mh1 mm0 mm1}, j2 = {hl0 hl1 lh0 lh1}: at most 2^-9 apart
// If the instruction aligns its 32 products to the largest exponent and drops bits below some width, A loses the low bits
// of mm next to hh.  Reference: fp64 on the host; also the plain fp32 fma chain (what the few-channel kernels and the
// reference's library do).  Prints rms and mean (bias) of the error in units of 2^-24 * rms(result).
// hipcc --offload-arch=gfx950 -O3 bf16x3_grouping.hip -o bf16x3_grouping.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// A [blk][16 rows][K], B [blk][16 cols][K] (K-contiguous), out [blk][3][256]
__global__ __launch_bounds__(64) void run(const float *A, const float *B, float *out, int K) {
    const int lane = threadIdx.x, rc = lane & 15, q = lane >> 4;
    const float *a = A + ((size_t)blockIdx.x * 16 + rc) * K, *b = B + ((size_t)blockIdx.x * 16 + rc) * K;
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < K; c0 += 16) {
        __bf16 ah[2][8], am[2][8], al[2][8], bh[2][8], bm[2][8], bl[2][8];
        for (int g = 0; g < 2; ++g)
            for (int e = 0; e < 8; ++e) {
                split3(a[c0 + 8 * g + e], ah[g][e], am[g][e], al[g][e]);
                split3(b[c0 + 8 * g + e], bh[g][e], bm[g][e], bl[g][e]);
            }
        auto pick = [&](__bf16 (&t0)[8], __bf16 (&t1)[8], __bf16 (&t2)[8], __bf16 (&t3)[8]) {
            bf16x8 v;
            for (int e = 0; e < 8; ++e) v[e] = q == 0 ? t0[e] : q == 1 ? t1[e] : q == 2 ? t2[e] : t3[e];
            return v;
        };
        // grouping A
        for (int g = 0; g < 2; ++g)
            accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[g], ah[g], am[g], am[g]), pick(bh[g], bm[g], bh[g], bm[g]),
                                                           accA, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[0], al[0], ah[1], al[1]), pick(bl[0], bh[0], bl[1], bh[1]), accA,
                                                       0, 0, 0);
        // grouping B
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[0], ah[1], ah[0], ah[1]), pick(bh[0], bh[1], bm[0], bm[1]), accB,
                                                       0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(am[0], am[1], am[0], am[1]), pick(bh[0], bh[1], bm[0], bm[1]), accB,
                                                       0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[0], ah[1], al[0], al[1]), pick(bl[0], bl[1], bh[0], bh[1]), accB,
                                                       0, 0, 0);
        // grouping C: B's first MFMA into the accumulator, the two small ones into a zeroed temporary, one add
        accC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[0], ah[1], ah[0], ah[1]), pick(bh[0], bh[1], bm[0], bm[1]), accC,
                                                       0, 0, 0);
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(am[0], am[1], am[0], am[1]), pick(bh[0], bh[1], bm[0], bm[1]),
                                                          zero, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pick(ah[0], ah[1], al[0], al[1]), pick(bl[0], bl[1], bh[0], bh[1]), t, 0, 0, 0);
        accC += t;
    }
    // D: lane holds rows 4 q + r of column rc  (row = A's pixel, column = B's output channel)
    float *o = out + (size_t)blockIdx.x * 4 * 256;
    for (int r = 0; r < 4; ++r) {
        o[(4 * q + r) * 16 + rc] = accA[r];
        o[256 + (4 * q + r) * 16 + rc] = accB[r];
        o[768 + (4 * q + r) * 16 + rc] = accC[r];
    }
    // fp32 fma chain: output (row = lane & 15, col = 4 q + r)
    for (int r = 0; r < 4; ++r) {
        const float *bb = B + ((size_t)blockIdx.x * 16 + 4 * q + r) * K;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(a[k], bb[k], s);
        o[512 + rc * 16 + 4 * q + r] = s;
    }
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 720, NB = 256;
    const int relu = argc > 2 ? atoi(argv[2]) : 1;
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hA((size_t)NB * 16 * K), hB((size_t)NB * 16 * K), ho((size_t)NB * 4 * 256);
    for (auto &v : hA) { v = nd(gen); if (relu && v < 0.f) v = 0.f; }
    for (auto &v : hB) v = nd(gen) / std::sqrt((float)K);
    float *dA, *dB, *dO;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dO, ho.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    run<<<NB, 64>>>(dA, dB, dO, K);
    hipMemcpy(ho.data(), dO, ho.size() * 4, hipMemcpyDeviceToHost);
    double se[4] = {0, 0, 0, 0}, me[4] = {0, 0, 0, 0}, st = 0, mx[4] = {0, 0, 0, 0};
    for (int blk = 0; blk < NB; ++blk)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double t = 0;
                for (int k = 0; k < K; ++k) t += (double)hA[((size_t)blk * 16 + i) * K + k] * (double)hB[((size_t)blk * 16 + j) * K + k];
                st += t * t;
                for (int v = 0; v < 4; ++v) {
                    const double e = (double)ho[(size_t)blk * 1024 + v * 256 + i * 16 + j] - t;
                    se[v] += e * e; me[v] += e; if (std::fabs(e) > mx[v]) mx[v] = std::fabs(e);
                }
            }
    const double n = (double)NB * 256, rms = std::sqrt(st / n), u = rms * std::ldexp(1.0, -24);
    const char *nm[4] = {"grouping A (shipped)", "grouping B (by magnitude)", "fp32 fma chain", "grouping C (small terms apart)"};
    printf("K = %d, relu(A) = %d, rms(result) = %.4f; errors in units of 2^-24 rms(result)\n", K, relu, rms);
    for (int v = 0; v < 4; ++v)
        printf("  %-28s rms %.3f   mean %+.4f   max %.2f\n", nm[v], std::sqrt(se[v] / n) / u, me[v] / n / u, mx[v] / u);
    return 0;
}
