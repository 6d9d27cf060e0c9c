// Does a bounds-checked buffer_load ... lds (LDS-DMA) write ZEROS for out-of-range lanes?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* src, int bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    int lane = threadIdx.x;
    int voff = (lane % 3 == 0) ? 0x7fffffff : lane * 16;          // every third lane out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    // second instruction into the next 1 KB, all in range, with an instruction offset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float h[512], *d, *o, ho[512];
    for (int i = 0; i < 512; ++i) h[i] = i + 1;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 1024, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        float exp1 = (l % 3 == 0) ? 0.f : (float)(l * 4 + e + 1);
        if (ho[l * 4 + e] != exp1) { if (bad < 8) printf("lane %d e %d got %f want %f\n", l, e, ho[l*4+e], exp1); ++bad; }
        if (ho[256 + l * 4 + e] != (float)(l * 4 + e + 1)) { if (bad < 8) printf("2nd: lane %d got %f\n", l, ho[256+l*4+e]); ++bad; }
    }
    printf("lds-dma oob test: %s (%d mismatches)\n", bad ? "FAIL" : "OK: out-of-range lanes wrote zeros", bad);
    return 0;
}
