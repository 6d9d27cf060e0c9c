// Achievable HBM bandwidth on this chip for the simplest kernels: streaming 16-byte reads (sum),
// streaming writes, and a copy, over 1 GiB (>> 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rd(const float4 *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void wr(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void cp(const float4 *a, float4 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b; float *o;
    if (hipMalloc(&a, bytes) || hipMalloc(&b, bytes) || hipMalloc(&o, 4)) return 1;
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0;
            for (int it = 0; it < 3; ++it) {
                (void)hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) {
                    if (k == 0) rd<<<blocks, 256>>>(a, n, o);
                    else if (k == 1) wr<<<blocks, 256>>>(b, n);
                    else cp<<<blocks, 256>>>(a, b, n);
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double gb = (k == 2 ? 2.0 : 1.0) * bytes * 5 / 1e9;
            printf("%s blocks=%5d: %.0f GB/s\n", k == 0 ? "read " : k == 1 ? "write" : "copy ", blocks, gb / (ms / 1e3));
        }
    }
    return 0;
}
