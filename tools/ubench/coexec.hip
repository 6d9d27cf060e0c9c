// Do the bf16 matrix pipe and the vector ALU of a SIMD work at the same time?  Per wave and iteration: a block of NM
// independent v_mfma_f32_16x16x32_bf16 and a block of NV fma + NE v_exp_f32 (the dense stage-3 tile: 30 / ~230 / 60).
//   mode 0: MFMA block only      mode 1: VALU block only      mode 2: both, one after the other, in every wave
//   mode 3: even waves MFMA only, odd waves VALU only (the same totals per SIMD as mode 2 with half the waves each)
// 4 waves per SIMD (1024-thread... 2 x 512), one workgroup pair per CU.  Prints cycles per iteration per SIMD @2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NM = 30, NV = 228, NE = 60;
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters) {
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(float)(threadIdx.x + i); bb[i] = (__bf16)(float)(i + 1); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 4) == 0);   // waves w and w + 4 share a SIMD
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 4) != 0);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int r = 0; r < (MODE == 3 ? 2 : 1); ++r)
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i % 6], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < (MODE == 3 ? 2 : 1); ++r) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i % 12] = __builtin_fmaf(v[i % 12], 0.999f, 0.5f);
#pragma unroll
                for (int i = 0; i < NE; ++i) v[i % 12] = __builtin_amdgcn_exp2f(v[i % 12] * 1e-3f);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name) {
    float *o; (void)hipMalloc(&o, 512 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    k<MODE><<<512, 512>>>(o, 50);
    (void)hipEventRecord(e0);
    k<MODE><<<512, 512>>>(o, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves; "one tile" = NM MFMAs + the VALU block of ONE wave; modes 0-2 run 4 per iteration, mode 3 also
    // (2 waves x 2 repeats of each kind)
    printf("%-58s %8.3f ms  %7.0f cycles per wave-tile per SIMD @2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / iters / 4);
}
int main() {
    run<0>("30 MFMA (bf16 16x16x32) per wave");
    run<1>("228 fma + 60 exp2 per wave");
    run<2>("both in every wave, one after the other");
    run<3>("waves 0-3 of a workgroup MFMA only, waves 4-7 VALU only");
    return 0;
}
