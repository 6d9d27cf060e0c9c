// tools/ubench/mfma4x4.hip -- v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result layout and issue rate.
// 16 independent 4x4 blocks per wave, K = 1: block b = lane / 4; A[i] from lane 4b + i, B[j] from lane 4b + j,
// D[i][j] += A[i] * B[j].  Questions for a few-channel fp32 convolution on it (no padding waste at Cout = 4, 8, 24):
//   (1) which lane / register holds D[i][j];   (2) cycles per instruction with 1, 2, 4, 8 accumulator chains.
// hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float *a, const float *b, float *d) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float *out, float seed, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float *out, int wgs_per_cu) {
    const int iters = 4096, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate<NACC><<<blocks, 256>>>(out, 1.0f, 16);
    hipEventRecord(e0);
    rate<NACC><<<blocks, 256>>>(out, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wgs_per_cu waves (256 threads = 4 waves = one per SIMD), each iters * 32 MFMAs
    const double mfma_per_simd = (double)wgs_per_cu * iters * 32;
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    const double tf = (double)blocks * 4 * iters * 32 * 512 / (ms * 1e-3) / 1e12;
    printf("chains %d, %d wave(s)/SIMD: %.3f ms  %.2f cycles per MFMA per SIMD at 2.4 GHz  %.1f TFLOP/s\n", NACC, wgs_per_cu,
           ms, cyc, tf);
}

int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 100.f * (1 + i); }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane 4b + j, register i holds A[4b + i] * B[4b + j]
    int ok_h1 = 1, ok_h2 = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, x = l % 4;
            if (hd[l * 4 + r] != ha[4 * blk + r] * hb[4 * blk + x]) ok_h1 = 0;   // reg = A row, lane = B column
            if (hd[l * 4 + r] != ha[4 * blk + x] * hb[4 * blk + r]) ok_h2 = 0;   // reg = B column, lane = A row
        }
    printf("layout: lane 4b+j reg i = A[4b+i]*B[4b+j]: %s;  lane 4b+i reg j = A[4b+i]*B[4b+j]: %s\n", ok_h1 ? "YES" : "no",
           ok_h2 ? "YES" : "no");
    printf("lane 5: %g %g %g %g (A = 1 + lane, B = 100 (1 + lane))\n", hd[20], hd[21], hd[22], hd[23]);
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1>(out, 1); run<2>(out, 1); run<4>(out, 1); run<8>(out, 1);
    run<1>(out, 4); run<2>(out, 4); run<4>(out, 4); run<8>(out, 2);
    return 0;
}
