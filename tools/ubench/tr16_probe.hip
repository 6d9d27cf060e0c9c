// tools/ubench/tr16_probe.hip -- what ds_read_b64_tr_b16 delivers (gfx950), and the operand maps of
// v_mfma_f32_16x16x32_bf16 as spamat_bwd_rowb uses them.  Prints PASS / FAIL lines; exit code 0 iff all pass.
//
//   1. LDS holds a [16 rows][16 cols] bf16 matrix M[r][c] = 16 r + c (32-byte rows).  Lane 4q+p of each 16-lane group
//      supplies the address of row R0 + q, columns 4p .. 4p+3; expectation (cdna_hip_programming.md T10): lane i of the
//      group receives column i of the four rows, row R0 + q in element q.
//   2. D = A B with A[i][k], B[k][n] small integers through the operand maps "lane (i, kq) element e = A[i][8 kq + e]",
//      "lane (n, kq) element e = B[8 kq + e][n]", result lane (n, q) register r = D[4 q + r][n].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }

__global__ void probe_tr(unsigned short *out) {      // out[64 lanes][4]
    __shared__ __attribute__((aligned(16))) unsigned short M[16 * 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) M[i] = f2bf((float)i);
    __syncthreads();
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int R0 = 4 * g;                              // group g reads rows 4g .. 4g+3
    const unsigned short *ad = M + (R0 + q) * 16 + 4 * p;
    bf16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v *)(ad));
    s16x4 s = __builtin_bit_cast(s16x4, v);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)s[e];
}

__global__ void probe_mfma(float *out) {             // out[64][4]
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kq + e;
        a[e] = (__bf16)(float)((i + 2 * k) % 7);             // A[i][k]
        b[e] = (__bf16)(float)((3 * k + i) % 5);             // B[k][n = i]
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    unsigned short *d, h[256];
    float *df, hf[256];
    hipMalloc(&d, sizeof(h));
    hipMalloc(&df, sizeof(hf));
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, df);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAIL launch\n"); return 2; }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int g = lane >> 4, i = lane & 15;
            unsigned u = (unsigned)h[lane * 4 + e] << 16;
            float got;
            memcpy(&got, &u, 4);
            const float want = (float)(16 * (4 * g + e) + i);          // row 4g + e, column i
            if (got != want) {
                if (bad < 12) printf("tr16: lane %d elem %d got %g want %g\n", lane, e, got, want);
                ++bad;
            }
        }
    printf("%s ds_read_b64_tr_b16 map (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    int bad2 = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int n = lane & 15, q = lane >> 4, row = 4 * q + r;
            float want = 0.f;
            for (int k = 0; k < 32; ++k) want += (float)((row + 2 * k) % 7) * (float)((3 * k + n) % 5);
            if (hf[lane * 4 + r] != want) {
                if (bad2 < 12) printf("mfma: lane %d reg %d got %g want %g\n", lane, r, hf[lane * 4 + r], want);
                ++bad2;
            }
        }
    printf("%s v_mfma_f32_16x16x32_bf16 operand / result maps (%d mismatches)\n", bad2 ? "FAIL" : "PASS", bad2);
    return (bad || bad2) ? 1 : 0;
}
