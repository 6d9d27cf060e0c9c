// tools/ubench/bwd_tile_rate.hip -- instruction-issue floor of ONE 16 x 16 band tile of the dense-row SpaMat backward
// (csrc/spamat_bwd_mfma.hip:spamat_bwd_rowb) on gfx950: nothing but the tile's own arithmetic in a loop, 4 waves per SIMD,
// every CU busy.  Per lane and tile: four costs come out of the cost MFMAs; per cost
//     e = exp2(fma(c, log2e, -max log2e));  w = e (d - out);  w -> three bf16 terms (and, fma, and, sub) ; 6 packs per tile
// and six v_mfma_f32_16x16x32_bf16 (2 cost, 2 left-gradient, 2 right-gradient contractions).  Variants:
//   0  the tile's vector work + its six MFMAs (what the kernel issues per tile, without LDS and addressing)
//   1  the vector work alone                 2  the six MFMAs alone              3  inputs only (the loop's own overhead)
// bench.py (train.roofline_valu) prices a launch as  tiles per SIMD x (variant 0 - variant 3).
// hipcc --offload-arch=gfx950 -O3 bwd_tile_rate.hip -o bwd_tile_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, int &h, int &m, int &l) {
    h = __float_as_int(x) & 0xffff0000;
    const float r1 = x - __int_as_float(h);
    m = __float_as_int(r1) & 0xffff0000;
    l = __float_as_int(r1 - __int_as_float(m));
}
__device__ __forceinline__ int pack(int odd, int even) { return __builtin_amdgcn_perm(odd, even, 0x07060302); }
__device__ __forceinline__ f32x4 mfma(i32x4 a, i32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int V>
__global__ __launch_bounds__(256, 4) void k(float *out, float seed) {
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const float nm = -seed * 3.f, dm0 = (float)(j - 4 * q) - seed;
    i32x4 opa = {lane, lane * 3, lane * 5, lane * 7}, opb = {lane * 11, lane * 13, lane * 17, lane * 19};
    f32x4 gl = {0.f, 0.f, 0.f, 0.f}, gr = gl;
    float res = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        f32x4 cst;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = seed * (float)(r + 1) + res * 1e-9f + (float)lane * 0.01f;
            asm volatile("" : "+v"(v));
            cst[r] = v;
        }
        if (V == 3) { res += (cst[0] + cst[1]) + (cst[2] + cst[3]); continue; }
        if (V == 0 || V == 2) {                       // the two cost MFMAs
            cst = mfma(opa, opb, cst);
            cst = mfma(opb, opa, cst);
        }
        if (V == 2) {                                 // the four contractions on fixed operands
            gl = mfma(opa, opb, gl); gl = mfma(opb, opa, gl);
            gr = mfma(opa, opa, gr); gr = mfma(opb, opb, gr);
            res += cst[0];
            continue;
        }
        const float dm = dm0 + (float)(16 * (it & 15));
        int wh[4], wm[4], wl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(cst[r], 1.4426950408889634f, nm));
            split3(e * (dm - (float)r), wh[r], wm[r], wl[r]);
        }
        const i32x4 a1 = {pack(wh[1], wh[0]), pack(wh[3], wh[2]), pack(wm[1], wm[0]), pack(wm[3], wm[2])};
        const i32x4 a2 = {a1[0], a1[1], pack(wl[1], wl[0]), pack(wl[3], wl[2])};
        if (V == 0) {
            gl = mfma(a1, opb, gl); gl = mfma(a2, opa, gl);
            gr = mfma(a1, opa, gr); gr = mfma(a2, opb, gr);
        } else {
            res += __int_as_float((a1[0] ^ a1[1]) + (a1[2] ^ a1[3]) + (a2[2] ^ a2[3]));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res + gl[0] + gl[1] + gl[2] + gl[3] + gr[0] + gr[1] + gr[2] + gr[3];
}

template <int V>
void run(const char *name, float *d) {
    const int blocks = 256 * 4, threads = 256;     // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<blocks, threads>>>(d, 0.013f);
    hipEventRecord(e0);
    k<V><<<blocks, threads>>>(d, 0.013f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)blocks * (threads / 64) * ITERS / 1024.0;
    printf("%-34s %.3f ms  %.0f cycles per tile per SIMD @2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / tiles_per_simd);
}
int main() {
    float *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("tile: vector work + six MFMAs", d); run<1>("tile: vector work alone", d); run<2>("tile: six MFMAs alone", d);
    run<3>("inputs only (loop overhead)", d);
    return 0;
}
