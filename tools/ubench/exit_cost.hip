// tools/ubench/exit_cost.hip -- what does a launch cost whose workgroups read one word and exit?
// (the marker launches behind the sparse-row SpaMat kernel when every row is sparse)
// hipcc --offload-arch=gfx950 -O3 -w exit_cost.hip -o exit_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *m, float *o, int stride) {
    if (m[(size_t)blockIdx.x * stride] != -1.0f) return;
    o[blockIdx.x * blockDim.x + threadIdx.x] = 1.f;
}
int main() {
    float *m, *o;
    hipMalloc(&m, 128 << 20); hipMalloc(&o, 128 << 20);
    hipMemset(m, 0, 128 << 20);
    const int cfg[][2] = {{4320, 512}, {8640, 512}, {8640, 256}, {17280, 256}, {4320, 256}, {4320, 1024}, {2160, 1024}, {1080, 1024}};
    for (auto &c : cfg) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 5; ++i) k<<<c[0], c[1]>>>(m, o, 972);
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) k<<<c[0], c[1]>>>(m, o, 972);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%6d workgroups x %4d threads: %.2f us per launch\n", c[0], c[1], 1e3 * ms / 50);
    }
    return 0;
}
