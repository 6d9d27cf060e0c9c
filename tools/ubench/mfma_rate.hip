// Issue rate of the MFMA flavours a bf16-split fp32 GEMM could use, per SIMD: N back-to-back
// independent MFMAs per wave, 1 or 2 waves per SIMD, wall-clock -> cycles per instruction at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    s16x4 sa = {(short)threadIdx.x, 1, 2, 3}, sb = {4, 5, 6, (short)threadIdx.x};
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(float)(threadIdx.x + i); bb[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char *name, double flop_per_instr) {
    float *o; (void)hipMalloc(&o, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 512}) {                 // 1 or 2 waves per SIMD, one workgroup per CU
        k<KIND><<<256, threads>>>(o, 100);
        (void)hipEventRecord(e0);
        k<KIND><<<256, threads>>>(o, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 8 * (threads / 256);          // MFMAs issued per SIMD
        printf("%-22s %d waves/SIMD: %.1f cycles/MFMA @2.4GHz, %.0f TFLOP/s\n", name, threads / 256,
               ms * 1e-3 * 2.4e9 / per_simd, per_simd * 1024 * flop_per_instr / (ms * 1e-3) / 1e12);
    }
}
int main() {
    run<0>("f32 16x16x4", 2048);
    run<1>("bf16 16x16x16 (_1k)", 8192);
    run<2>("bf16 16x16x32", 16384);
    return 0;
}
