// decnet_amd/csrc/capi.hip -- extern "C" entry points for SpaMat / SpaVar (include/decnet_hip.h).
#include "common.h"

// kernels in spamat_rowtile.hip
int decnet_rowtile_forward(int mode, const float *ref, const float *tar, const float *rmask,
                           const float *tmask, const float *disparity, float *out, float *var_out,
                           float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                           hipStream_t stream);
int decnet_rowtile_backward(int var, const float *ref, const float *tar, const float *rmask,
                            const float *tmask, const float *disparity, const float *out,
                            const float *sum_sim, const float *max_cost, const float *grad_out,
                            float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                            int W, int max_disp, hipStream_t stream);
int decnet_check_spamat_args(const void *const *ptrs, int n, int B, int C, int H, int W,
                             int max_disp);
// kernels in spamat_mfma.hip (banded cost tiles on the matrix cores)
int decnet_mfma_forward(int mode, const float *ref, const float *tar, const float *rmask,
                        const float *tmask, const float *disparity, float *out, float *var_out,
                        float *sum_sim, float *max_cost, int B, int C, int H, int W, int max_disp,
                        int allow_compact, int mbits, hipStream_t stream);

// kernels in spamat_bwd_mfma.hip
int decnet_mfma_backward(int var, const float *ref, const float *tar, const float *rmask,
                         const float *tmask, const float *disparity, const float *out,
                         const float *sum_sim, const float *max_cost, const float *grad_out,
                         float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                         int W, int max_disp, hipStream_t stream);

// spamat_wide.hip: disparity ranges wider than 18 tiles (max_disp > 272) as several band-kernel calls + per-pixel merges
int decnet_wide_forward(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
                        const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost, int B, int C,
                        int H, int W, int D, int allow_compact, int mbits, hipStream_t stream);
int decnet_wide_backward(int var, const float *ref, const float *tar, const float *rmask, const float *tmask,
                         const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
                         const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                         int W, int D, hipStream_t stream);

#include <stdlib.h>
#include <string.h>

static int spamat_pinned() {       // DECNET_SPAMAT_KERNEL, read once
    static const int pinned = [] {
        const char *e = getenv("DECNET_SPAMAT_KERNEL");
        if (!e) return 0;
        return !strcmp(e, "rowtile") ? 1 : !strcmp(e, "mfma") ? 2 : !strcmp(e, "mfma_dense") ? 3 : 0;
    }();
    return pinned;
}

// ---- DECNET_CHECK_FINITE=1: the NaN contract of include/decnet_hip.h made checkable --------------------------------
// The reference propagates a NaN / Inf feature into every output whose candidate set touches it (fmaxf / expf,
// SM_kernel.cu:46-58); the forward kernels here are built with -fno-honor-nans and give an unspecified value there.
// With the knob set, both feature maps are swept by one reduction kernel before the launch and a non-finite element is
// an error (DECNET_ERR_NONFINITE, nothing else launched) instead of a silently different result.  The check waits for
// the stream (one 4-byte read-back), so it is a debugging aid: skipped while the stream is being captured into a graph.
namespace {
__global__ __launch_bounds__(256) void count_nonfinite(const float *__restrict__ x, size_t n4, size_t n,
                                                       unsigned *__restrict__ count) {
    unsigned bad = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        // exponent all ones <=> Inf or NaN (integer test: immune to -fno-honor-nans style folding)
        bad += ((__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u) + ((__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u) +
               ((__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u) + ((__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u);
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)   // tail (everything when
        bad += (__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u;                                // the map is not 16-byte aligned)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
}  // namespace

static bool check_finite_on() {
    static const bool on = [] { const char *e = getenv("DECNET_CHECK_FINITE"); return e && atoi(e) != 0; }();
    return on;
}

// 0: all finite (or the check is off / the stream is capturing); DECNET_ERR_NONFINITE; positive hipError_t
static int check_finite(const float *ref, const float *tar, int B, int C, int H, int W, hipStream_t stream) {
    if (!check_finite_on()) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return 0;
    }
    const bool aligned = ((((uintptr_t)ref) | ((uintptr_t)tar)) & 15) == 0;   // (torch allocations are; a view may not be)
    unsigned *d = nullptr, h = 0;
    hipError_t e = hipMallocAsync((void **)&d, sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d, 0, sizeof(unsigned), stream);
    const size_t n = (size_t)B * C * H * W, n4 = aligned ? n >> 2 : 0;
    const size_t nwork = aligned ? n4 : n;
    const unsigned grid = (unsigned)((nwork + 255) / 256 < 2048 ? (nwork + 255) / 256 + 1 : 2048);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(count_nonfinite, dim3(grid), dim3(256), 0, stream, ref, n4, n, d);
        hipLaunchKernelGGL(count_nonfinite, dim3(grid), dim3(256), 0, stream, tar, n4, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFreeAsync(d, stream);
    if (e != hipSuccess) return (int)e;
    return h ? DECNET_ERR_NONFINITE : 0;
}

// Backward dispatch: matrix-core kernels, row-tile kernels for what they do not cover.
static int backward_dispatch(int var, const float *ref, const float *tar, const float *rmask,
                             const float *tmask, const float *disparity, const float *out,
                             const float *sum_sim, const float *max_cost, const float *grad_out,
                             float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                             int W, int max_disp, hipStream_t stream) {
    const int pinned = spamat_pinned();
    if (pinned != 1) {
        int rc = decnet_mfma_backward(var, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost,
                                      grad_out, grad_ref, grad_tar, grad_disp, B, C, H, W, max_disp,
                                      stream);
        if (rc == DECNET_ERR_UNSUPPORTED && max_disp > 272)       // wider than 18 tiles: the same kernels band by band
            rc = decnet_wide_backward(var, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref,
                                      grad_tar, grad_disp, B, C, H, W, max_disp, stream);
        if (rc != DECNET_ERR_UNSUPPORTED || pinned >= 2) return rc;
    }
    return decnet_rowtile_backward(var, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost,
                                   grad_out, grad_ref, grad_tar, grad_disp, B, C, H, W, max_disp,
                                   stream);
}

// Forward dispatch: the MFMA band kernel; the row-tile kernel covers what it cannot
// (band wider than 18 tiles, LDS overflow).  DECNET_SPAMAT_KERNEL=rowtile|mfma|mfma_dense pins
// one variant (read once; used by the A/B benchmarks and the parity tests of the variants);
// mfma_dense = MFMA kernel with the sparse-row compaction path switched off.
static int forward_dispatch(int mode, const float *ref, const float *tar, const float *rmask,
                            const float *tmask, const float *disparity, float *out, float *var_out,
                            float *sum_sim, float *max_cost, int B, int C, int H, int W,
                            int max_disp, hipStream_t stream) {
    const int pinned = spamat_pinned();
    if (int rc = check_finite(ref, tar, B, C, H, W, stream)) return rc;
    if (pinned != 1) {
        int rc = decnet_mfma_forward(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim,
                                     max_cost, B, C, H, W, max_disp, pinned != 3, 0, stream);
        if (rc == DECNET_ERR_UNSUPPORTED && max_disp > 272)       // wider than 18 tiles: the same kernels band by band
            rc = decnet_wide_forward(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim, max_cost, B, C, H, W,
                                     max_disp, pinned != 3, 0, stream);
        if (rc != DECNET_ERR_UNSUPPORTED || pinned >= 2) return rc;
    }
    return decnet_rowtile_forward(mode, ref, tar, rmask, tmask, disparity, out, var_out, sum_sim,
                                  max_cost, B, C, H, W, max_disp, stream);
}

extern "C" {

const char *decnet_version(void) { return "decnet_hip 0.1.0 gfx950"; }

int decnet_spamat_forward(const float *ref, const float *tar, const float *ref_mask,
                          const float *tar_mask, float *output, float *sum_similarities,
                          float *max_cost, int B, int C, int H, int W, int max_disp, void *stream) {
    const void *p[] = {ref, tar, ref_mask, tar_mask, output, sum_similarities, max_cost};
    int rc = decnet_check_spamat_args(p, 7, B, C, H, W, max_disp);
    if (rc) return rc;
    return forward_dispatch(0, ref, tar, ref_mask, tar_mask, nullptr, output, nullptr,
                                  sum_similarities, max_cost, B, C, H, W, max_disp,
                                  (hipStream_t)stream);
}

int decnet_spavar_forward(const float *ref, const float *tar, const float *ref_mask,
                          const float *tar_mask, const float *disparity, float *output,
                          float *sum_similarities, float *max_cost, int B, int C, int H, int W,
                          int max_disp, void *stream) {
    const void *p[] = {ref, tar, ref_mask, tar_mask, disparity, output, sum_similarities, max_cost};
    int rc = decnet_check_spamat_args(p, 8, B, C, H, W, max_disp);
    if (rc) return rc;
    return forward_dispatch(1, ref, tar, ref_mask, tar_mask, disparity, nullptr, output,
                                  sum_similarities, max_cost, B, C, H, W, max_disp,
                                  (hipStream_t)stream);
}

int decnet_spamatvar_forward(const float *ref, const float *tar, const float *ref_mask,
                             const float *tar_mask, float *output, float *variance,
                             float *sum_similarities, float *max_cost, int B, int C, int H, int W,
                             int max_disp, void *stream) {
    const void *p[] = {ref, tar, ref_mask, tar_mask, output, variance, sum_similarities, max_cost};
    int rc = decnet_check_spamat_args(p, 8, B, C, H, W, max_disp);
    if (rc) return rc;
    return forward_dispatch(2, ref, tar, ref_mask, tar_mask, nullptr, output, variance,
                                  sum_similarities, max_cost, B, C, H, W, max_disp,
                                  (hipStream_t)stream);
}

int decnet_spamatvar_forward_bits(const float *ref, const float *tar, const unsigned long long *ref_bits,
                                  const unsigned long long *tar_bits, float *output, float *variance,
                                  float *sum_similarities, float *max_cost, int B, int C, int H, int W,
                                  int max_disp, void *stream) {
    const void *p[] = {ref, tar, ref_bits, tar_bits, output, variance, sum_similarities, max_cost};
    int rc = decnet_check_spamat_args(p, 8, B, C, H, W, max_disp);
    if (rc) return rc;
    // the matrix-core kernels only (the row-tile fallback reads float planes); above max_disp 272 (18 tiles) band by
    // band on masks unpacked into scratch planes (spamat_wide.hip; UNSUPPORTED while the stream is being captured).
    // DECNET_SPAMAT_KERNEL=rowtile pins a kernel this entry does not have -> UNSUPPORTED, the caller falls back to the
    // float-mask entry (decnet_amd.model does)
    if (spamat_pinned() == 1) return DECNET_ERR_UNSUPPORTED;
    if (int rc2 = check_finite(ref, tar, B, C, H, W, (hipStream_t)stream)) return rc2;
    rc = decnet_mfma_forward(2, ref, tar, reinterpret_cast<const float *>(ref_bits),
                             reinterpret_cast<const float *>(tar_bits), nullptr, output, variance,
                             sum_similarities, max_cost, B, C, H, W, max_disp, spamat_pinned() != 3, 1,
                             (hipStream_t)stream);
    if (rc == DECNET_ERR_UNSUPPORTED && max_disp > 272)
        rc = decnet_wide_forward(2, ref, tar, reinterpret_cast<const float *>(ref_bits),
                                 reinterpret_cast<const float *>(tar_bits), nullptr, output, variance, sum_similarities,
                                 max_cost, B, C, H, W, max_disp, spamat_pinned() != 3, 1, (hipStream_t)stream);
    return rc;
}

int decnet_spamat_backward(const float *ref, const float *tar, const float *ref_mask,
                           const float *tar_mask, const float *output,
                           const float *sum_similarities, const float *max_cost,
                           const float *grad_output, float *grad_ref, float *grad_tar, int B, int C,
                           int H, int W, int max_disp, void *stream) {
    const void *p[] = {ref, tar, ref_mask, tar_mask, output, sum_similarities, max_cost,
                       grad_output, grad_ref, grad_tar};
    int rc = decnet_check_spamat_args(p, 10, B, C, H, W, max_disp);
    if (rc) return rc;
    return backward_dispatch(0, ref, tar, ref_mask, tar_mask, nullptr, output,
                                   sum_similarities, max_cost, grad_output, grad_ref, grad_tar,
                                   nullptr, B, C, H, W, max_disp, (hipStream_t)stream);
}

int decnet_spavar_backward(const float *ref, const float *tar, const float *ref_mask,
                           const float *tar_mask, const float *disparity, const float *output,
                           const float *sum_similarities, const float *max_cost,
                           const float *grad_output, float *grad_ref, float *grad_tar,
                           float *grad_disparity, int B, int C, int H, int W, int max_disp,
                           void *stream) {
    const void *p[] = {ref, tar, ref_mask, tar_mask, disparity, output, sum_similarities, max_cost,
                       grad_output, grad_ref, grad_tar, grad_disparity};
    int rc = decnet_check_spamat_args(p, 12, B, C, H, W, max_disp);
    if (rc) return rc;
    return backward_dispatch(1, ref, tar, ref_mask, tar_mask, disparity, output,
                                   sum_similarities, max_cost, grad_output, grad_ref, grad_tar,
                                   grad_disparity, B, C, H, W, max_disp, (hipStream_t)stream);
}

}  // extern "C"
