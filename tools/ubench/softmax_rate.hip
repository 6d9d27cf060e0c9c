// tools/ubench/softmax_rate.hip -- VALU floor of the SpaMat/SpaVar softmax passes on gfx950.
// 60 cost values per lane in registers (15 tiles x 4, the stage-3 band), nothing but the VALU work of
// spamat_mfma.hip:softmax_passes in a loop; 4 waves per SIMD, every CU busy.  Variants:
//   0  the three passes as shipped (max3 | fma, exp2, S+=, T=fma | d=imm+c, t=e*d, V=fma)
//   1  per-tile moments: pass 2 keeps s = sum e, t = sum r e, q = sum r^2 e per tile (r = 0..3), S/T/V from them
//   2  as 0 with four accumulator chains instead of two
//   3  exp2 only (60 v_exp_f32)         4  60 independent v_fma_f32      8  the loop's overhead alone
//   5/6  pass 2 in separated phases     7  max + 60 x (fma, exp2, add)
// hipcc --offload-arch=gfx950 -O3 -fno-honor-nans softmax_rate.hip -o softmax_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define NT 15
#define ITERS 512
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int V>
__global__ __launch_bounds__(256, 4) void k(float *out, float seed, int D) {
    float acc[NT][4];
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const float dlf = (float)(j - 4 * q);
    float res = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = seed * (float)(m * 4 + r + 1) + res * 1e-9f + (float)lane * 0.01f;
                asm volatile("" : "+v"(v));
                acc[m][r] = v;
            }
        if (V == 3) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) res += ex2(acc[m][r]);   // (adds included: 60 exp + 60 add)
            continue;
        }
        if (V == 8) {                       // the loop's own overhead: making the 60 inputs, nothing else
#pragma unroll
            for (int m = 0; m < NT; ++m) res += (acc[m][0] + acc[m][1]) + (acc[m][2] + acc[m][3]);
            continue;
        }
        if (V == 4) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] = __builtin_fmaf(acc[m][r], seed, 0.5f);
#pragma unroll
            for (int m = 0; m < NT; ++m) res += acc[m][0];
            continue;
        }
        // pass 1
        float mx0 = 1e-6f, mx1 = 1e-6f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            mx0 = fmaxf(fmaxf(mx0, acc[m][0]), acc[m][1]);
            mx1 = fmaxf(fmaxf(mx1, acc[m][2]), acc[m][3]);
        }
        float mx = fmaxf(mx0, mx1);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float nm = -mx * 1.4426950408889634f;
        float S, mu, var;
        if (V == 5 || V == 6) {
            // phase-separated: all exponent arguments, then all exp2, then the sums (dependent pairs far apart)
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] = __builtin_fmaf(acc[m][r], 1.4426950408889634f, nm);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] = ex2(acc[m][r]);
            __builtin_amdgcn_sched_barrier(0);
            float Sa[4] = {0.f, 0.f, 0.f, 0.f}, Ta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Sa[r] += acc[m][r];
                    Ta[r] = __builtin_fmaf(acc[m][r], (float)(16 * m - r), Ta[r]);
                }
            float Sl = (Sa[0] + Sa[1]) + (Sa[2] + Sa[3]), Tl = (Ta[0] + Ta[1]) + (Ta[2] + Ta[3]);
            Tl = __builtin_fmaf(dlf, Sl, Tl);
            Sl += __shfl_xor(Sl, 16); Sl += __shfl_xor(Sl, 32);
            Tl += __shfl_xor(Tl, 16); Tl += __shfl_xor(Tl, 32);
            S = Sl + 1e-6f;
            mu = (Tl + 1e-6f) / S;
            const float c0 = dlf - mu;
            float Va[4] = {0.f, 0.f, 0.f, 0.f};
            if (V == 5) {
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d0 = (float)(16 * m - r) + c0;
                        Va[r] = __builtin_fmaf(acc[m][r] * d0, d0, Va[r]);
                    }
            } else {
                // (d - mu)^2 e as e * d0 * d0 with the squares formed first (independent of e)
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d0 = (float)(16 * m - r) + c0;
                        Va[r] = __builtin_fmaf(d0 * d0, acc[m][r], Va[r]);
                    }
            }
            float Vl = (Va[0] + Va[1]) + (Va[2] + Va[3]);
            Vl += __shfl_xor(Vl, 16); Vl += __shfl_xor(Vl, 32);
            var = (Vl + 1e-6f) / S;
        } else if (V == 7) {
            // cost of the pieces: pass 1 + exp arguments + exp2 only
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] = ex2(__builtin_fmaf(acc[m][r], 1.4426950408889634f, nm));
            S = 1.f; mu = 0.f; var = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) S += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
        } else if (V == 0 || V == 2) {
            constexpr int NA = V == 2 ? 4 : 2;
            float Sa[NA], Ta[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) Sa[a] = Ta[a] = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = ex2(__builtin_fmaf(acc[m][r], 1.4426950408889634f, nm));
                    acc[m][r] = e;
                    Sa[r % NA] += e;
                    Ta[r % NA] = __builtin_fmaf(e, (float)(16 * m - r), Ta[r % NA]);
                }
            float Sl = 0.f, Tl = 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a) { Sl += Sa[a]; Tl += Ta[a]; }
            Tl = __builtin_fmaf(dlf, Sl, Tl);
            Sl += __shfl_xor(Sl, 16); Sl += __shfl_xor(Sl, 32);
            Tl += __shfl_xor(Tl, 16); Tl += __shfl_xor(Tl, 32);
            S = Sl + 1e-6f;
            mu = (Tl + 1e-6f) / S;
            const float c0 = dlf - mu;
            float Va[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) Va[a] = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d0 = (float)(16 * m - r) + c0;
                    Va[r % NA] = __builtin_fmaf(acc[m][r] * d0, d0, Va[r % NA]);
                }
            float Vl = 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a) Vl += Va[a];
            Vl += __shfl_xor(Vl, 16); Vl += __shfl_xor(Vl, 32);
            var = (Vl + 1e-6f) / S;
        } else {
            // per-tile moments over r = 0..3 (d = 16 m + dl - r): s = e0+e1+e2+e3, t = e1+2e2+3e3, q = e1+4e2+9e3
            float sm[NT], tm[NT], qm[NT];
            float Sl = 0.f, T16 = 0.f, Tt = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const float e0 = ex2(__builtin_fmaf(acc[m][0], 1.4426950408889634f, nm));
                const float e1 = ex2(__builtin_fmaf(acc[m][1], 1.4426950408889634f, nm));
                const float e2 = ex2(__builtin_fmaf(acc[m][2], 1.4426950408889634f, nm));
                const float e3 = ex2(__builtin_fmaf(acc[m][3], 1.4426950408889634f, nm));
                const float s = (e0 + e1) + (e2 + e3);
                const float t = __builtin_fmaf(3.f, e3, __builtin_fmaf(2.f, e2, e1));
                const float qq = __builtin_fmaf(9.f, e3, __builtin_fmaf(4.f, e2, e1));
                sm[m] = s; tm[m] = t; qm[m] = qq;
                Sl += s;
                T16 = __builtin_fmaf(s, (float)(16 * m), T16);
                Tt += t;
            }
            float Tl = __builtin_fmaf(dlf, Sl, T16 - Tt);
            Sl += __shfl_xor(Sl, 16); Sl += __shfl_xor(Sl, 32);
            Tl += __shfl_xor(Tl, 16); Tl += __shfl_xor(Tl, 32);
            S = Sl + 1e-6f;
            mu = (Tl + 1e-6f) / S;
            const float c0 = dlf - mu;
            float V0 = 0.f, V1 = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) {          // sum_r e_r (A - r)^2 = A (A s - 2 t) + q,  A = 16 m + dl - mu
                const float A = (float)(16 * m) + c0;
                const float u = __builtin_fmaf(A, sm[m], -2.f * tm[m]);
                if (m & 1) V1 = __builtin_fmaf(A, u, V1) + qm[m];
                else V0 = __builtin_fmaf(A, u, V0) + qm[m];
            }
            float Vl = V0 + V1;
            Vl += __shfl_xor(Vl, 16); Vl += __shfl_xor(Vl, 32);
            var = (Vl + 1e-6f) / S;
        }
        res += mu + var + S;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int V>
void run(const char *name, float *d) {
    const int blocks = 256 * 4, threads = 256;     // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<blocks, threads>>>(d, 0.013f, 216);
    hipEventRecord(e0);
    k<V><<<blocks, threads>>>(d, 0.013f, 216);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = (double)blocks * (threads / 64) * ITERS / 1024.0;
    printf("%-34s %.3f ms  %.0f cycles per wave-tile (60 values) per SIMD @2.4 GHz\n", name, ms,
           ms * 1e-3 * 2.4e9 / tiles_per_simd);
}
int main() {
    float *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("three passes as shipped", d); run<1>("per-tile moments", d); run<2>("four accumulator chains", d);
    run<3>("60 x (exp2 + add)", d); run<4>("60 x fma", d);
    run<8>("inputs only (loop overhead)", d);
    run<5>("phase-separated", d); run<6>("phase-separated, squares first", d); run<7>("max + 60 x (fma, exp2, add)", d);
    return 0;
}
