// What does a bounds-checked 16-byte buffer load return when only SOME of its dwords are in range,
// on the left (negative byte offset = huge unsigned) and on the right of the buffer?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, int n, float *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    const int x = lane - 6;                       // first element index: -6 .. 57 for n = 56
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, x * 4, 0, 0);
    out[lane * 4 + 0] = __int_as_float(v.x); out[lane * 4 + 1] = __int_as_float(v.y);
    out[lane * 4 + 2] = __int_as_float(v.z); out[lane * 4 + 3] = __int_as_float(v.w);
}
int main() {
    const int n = 56;
    float h[64], *d, *o, ho[256];
    for (int i = 0; i < 64; ++i) h[i] = 100.f + i;   // elements >= n exist in memory but are out of range
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, n, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int i = l - 6 + e;
            const float want = (i >= 0 && i < n) ? 100.f + i : 0.f;
            if (ho[l * 4 + e] != want) { if (bad < 12) printf("lane %d e %d (index %d): got %g want %g\n", l, e, i, ho[l * 4 + e], want); ++bad; }
        }
    printf("partial-range b128 buffer load: %s (%d mismatches)\n", bad ? "NOT per-dword zero fill" : "OK: per-dword zero fill on both sides", bad);
    return 0;
}
