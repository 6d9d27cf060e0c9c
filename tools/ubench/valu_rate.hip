// tools/ubench/valu_rate.hip -- what does one SIMD of gfx950 sustain for fp32 VALU ops?
// Measures wave-instructions per cycle per SIMD for v_fma_f32, v_pk_fma_f32, v_exp_f32,
// v_max3_f32 at 1/2/4/8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int OP>
__global__ void k(float *out, float a, float b) {
    float x[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = f2{x[i], x[i] + 1.f}; }
    f2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
            if (OP == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (OP == 3) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], a), x[(i + 1) & 7]);
            if (OP == 4) x[i] = x[i] + a;
            if (OP == 5) p[i] = p[i] + pa;
            if (OP == 6) p[i] = p[i] * pa;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, float *d) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        int threads = 256 * wps;      // wps waves per SIMD with one block per CU
        if (threads > 1024) { threads = 1024; }
        int blocks = 256 * (256 * wps / threads);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, threads>>>(d, 1.0001f, 0.5f);
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(d, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * (threads / 64) * ITERS * 8;
        double per_simd = winst / 1024.0;
        printf("%-14s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_exp_f32", d); run<3>("v_max3_f32", d);
    run<4>("v_add_f32", d); run<5>("v_pk_add_f32", d); run<6>("v_pk_mul_f32", d);
    return 0;
}
