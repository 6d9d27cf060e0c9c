// decnet_amd/csrc/spamat_wide.hip -- SpaMat / SpaVar for disparity ranges wider than the band kernels' 18 tiles
// (max_disp > 272: the reference's own demo data carries ndisp 400 / 610 -> max_disp 405 / 621 at stage 3,
// demo.py:149-155), forward and backward, on the matrix-core kernels instead of the VALU row-tile fallback.
//
// The candidate range [0, D) is cut into nb = ceil(D / 272) bands [d0, d0 + Db).  Band b is an ordinary call of the band
// kernels on a SHIFTED right view:  R_b[c][x] = R[c][x - d0],  tmask_b[x] = tmask[x - d0]  (both 0 for x < d0), so that
// its candidate d' of left pixel x is the original candidate d = d' + d0 and "right pixel left of the row" (SM_kernel.cu:42)
// is a masked-off pixel of the shifted view.  Bands are merged per pixel like chunks of an online softmax:
//     m = max(m_a, m_b),   S - 1e-6 = (S_a - 1e-6) e^(m_a - m) + (S_b - 1e-6) e^(m_b - m),
//     T = sum e d:  T_b = out_b S_b - 1e-6 + d0 (S_b - 1e-6),   out = (1e-6 + T) / S          (SM_kernel.cu:110-122)
// and the variance the same way with the band's disparity input shifted by d0 (SV_kernel.cu:112-121); the FUSED call
// (variance around its own disparity) is one sweep of fused band calls merged by the parallel-variance rule
// (merge_band_fused).  A band without a
// valid candidate comes back as m_b = 1e-6, S_b = 1e-6, out_b = 1 and contributes exactly nothing.  The backward kernels
// take the GLOBAL max / sum / output (minus d0 where it is a disparity), write the band's gradients into scratch planes and
// those are accumulated (the right gradient shifted back by d0).  Scratch comes from the stream-ordered allocator
// (hipMallocAsync: (C + 8) planes forward, (3 C + 4) backward); while the stream is being captured the entry reports
// DECNET_ERR_UNSUPPORTED and capi.hip falls back to the row-tile kernels.
#include "common.h"

int decnet_mfma_forward(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
                        const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost, int B, int C,
                        int H, int W, int max_disp, int allow_compact, int mbits, hipStream_t stream);
int decnet_mfma_backward(int var, const float *ref, const float *tar, const float *rmask, const float *tmask,
                         const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
                         const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                         int W, int max_disp, hipStream_t stream);

namespace {

constexpr int WIDE_BAND = 272;          // widest band of the matrix-core kernels (18 tiles)
constexpr int EW_THREADS = 256;

// dst[p][y][x] = x >= d0 ? src[p][y][x - d0] : 0   over `planes` planes of H x W (features: B*C planes, masks: B)
__global__ __launch_bounds__(EW_THREADS) void shift_planes(const float *__restrict__ src, float *__restrict__ dst,
                                                           size_t rows, int W, int d0) {
    const size_t n = rows * (size_t)W;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) {
        const int x = (int)(i % W);
        dst[i] = x >= d0 ? src[i - d0] : 0.f;
    }
}
// dst = src - d0 (the disparity-like input of a band)
__global__ __launch_bounds__(EW_THREADS) void offset_plane(const float *__restrict__ src, float *__restrict__ dst, size_t n,
                                                           float d0) {
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS)
        dst[i] = src[i] - d0;
}
// running (q, S, m) <- merge with band (q_b, S_b, m_b); q is a quotient (1e-6 + Q) / S; d0f shifts the band's first moment
// (disparity: d0; variance: 0).  Pixels with the left mask off keep their zeros.
__global__ __launch_bounds__(EW_THREADS) void merge_band(const float *__restrict__ rmask, float *__restrict__ q,
                                                         float *__restrict__ S, float *__restrict__ m,
                                                         const float *__restrict__ qb, const float *__restrict__ Sb,
                                                         const float *__restrict__ mb, size_t n, float d0f) {
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) {
        if (rmask[i] == 0.f) continue;
        const float ma = m[i], mbv = mb[i], mn = fmaxf(ma, mbv);
        const float ea = expf(ma - mn), eb = expf(mbv - mn);
        const float Ea = S[i] - 1e-6f, Eb = Sb[i] - 1e-6f;                   // sum of exponentials of each part
        const float Qa = q[i] * S[i] - 1e-6f, Qb = qb[i] * Sb[i] - 1e-6f + d0f * Eb;
        const float Sn = 1e-6f + (Ea * ea + Eb * eb);
        q[i] = (1e-6f + (Qa * ea + Qb * eb)) / Sn;
        S[i] = Sn;
        m[i] = mn;
    }
}
// The fused call in ONE sweep: every band returns its own disparity q_b AND the variance v_b around it; two parts A, B with
// sums of exponentials E, first moments T and second moments V around their own means o combine around the joint mean mu as
//     V(mu) = sum_X e^(m_X - m) (V_X + 2 (o_X - mu) c_X + (o_X - mu)^2 E_X),   c_X = T_X - o_X E_X = 1e-6 (o_X - 1)
// (c_X from the definitions out = (1e-6 + T) / (1e-6 + E): exact, no difference of two rounded products).  Every term
// but the tiny c_X one is non-negative: no cancellation.  Running (q, v, S, m) <- merge with band (q_b, v_b, S_b, m_b) at d0.
__global__ __launch_bounds__(EW_THREADS) void merge_band_fused(const float *__restrict__ rmask, float *__restrict__ q,
                                                               float *__restrict__ v, float *__restrict__ S,
                                                               float *__restrict__ m, const float *__restrict__ qb,
                                                               const float *__restrict__ vb, const float *__restrict__ Sb,
                                                               const float *__restrict__ mb, size_t n, float d0f) {
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) {
        if (rmask[i] == 0.f) continue;
        const float ma = m[i], mbv = mb[i], mn = fmaxf(ma, mbv);
        const float ea = expf(ma - mn), eb = expf(mbv - mn);
        const float Sa = S[i], Sbv = Sb[i], Ea = Sa - 1e-6f, Eb = Sbv - 1e-6f;
        const float oa = q[i], ob = qb[i] + d0f;                             // the parts' own means, global coordinates
        const float Ta = oa * Sa - 1e-6f, Tb = qb[i] * Sbv - 1e-6f + d0f * Eb;
        const float Va = v[i] * Sa - 1e-6f, Vb = vb[i] * Sbv - 1e-6f;
        const float Sn = 1e-6f + (Ea * ea + Eb * eb);
        const float mu = (1e-6f + (Ta * ea + Tb * eb)) / Sn;
        const float da = oa - mu, db = ob - mu;
        const float ca = 1e-6f * (oa - 1.f), cb = 1e-6f * (qb[i] - 1.f) ;    // sum e (d - o) of each part (band coordinates for B: the same number)
        const float Vn = ea * (Va + da * (2.f * ca + da * Ea)) + eb * (Vb + db * (2.f * cb + db * Eb));
        q[i] = mu;
        v[i] = (1e-6f + Vn) / Sn;
        S[i] = Sn;
        m[i] = mn;
    }
}
// acc += add (planes of W-wide rows); add is read shifted right by d0: acc[x] += add[x + d0] for x + d0 < W
__global__ __launch_bounds__(EW_THREADS) void accumulate_planes(float *__restrict__ acc, const float *__restrict__ add,
                                                                size_t rows, int W, int d0) {
    const size_t n = rows * (size_t)W;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) {
        const int x = (int)(i % W);
        if (x + d0 < W) acc[i] += add[i + d0];
    }
}
// bit-packed masks [rows][ceil(W/64)] (bit i of word w = pixel 64 w + i) -> float planes
__global__ __launch_bounds__(EW_THREADS) void unpack_bits(const unsigned long long *__restrict__ bits, float *__restrict__ dst,
                                                          size_t rows, int W) {
    const size_t n = rows * (size_t)W;
    const int wpr = (W + 63) >> 6;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) {
        const size_t r = i / W;
        const int x = (int)(i - r * W);
        dst[i] = (float)((bits[r * wpr + (x >> 6)] >> (x & 63)) & 1ull);
    }
}

inline unsigned ew_grid(size_t n) {
    const size_t b = (n + EW_THREADS - 1) / EW_THREADS;
    return (unsigned)(b < 8192 ? (b ? b : 1) : 8192);
}
inline bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return cap != hipStreamCaptureStatusNone;
}
struct Scratch {                        // one stream-ordered allocation, freed (stream-ordered) when the call returns
    float *p = nullptr;
    hipStream_t s;
    explicit Scratch(hipStream_t st) : s(st) {}
    int get(size_t floats) { return (int)hipMallocAsync((void **)&p, floats * sizeof(float), s); }
    ~Scratch() {
        if (p) (void)hipFreeAsync(p, s);
    }
};
#define CK(expr)                              \
    do {                                      \
        int rc_ = (expr);                     \
        if (rc_) return rc_;                  \
    } while (0)
#define LAUNCH_OK() CK(decnet_launch_status())

}  // namespace

// mode 0 SpaMat (out, S, m), 1 SpaVar (var_out, S, m; `disparity` given), 2 fused (out, var_out, S, m).
// mbits: the masks are bit-packed (decnet_spamatvar_forward_bits): unpacked into scratch planes first.
int decnet_wide_forward(int mode, const float *ref, const float *tar, const float *rmask, const float *tmask,
                        const float *disparity, float *out, float *var_out, float *sum_sim, float *max_cost, int B, int C,
                        int H, int W, int D, int allow_compact, int mbits, hipStream_t stream) {
    if (D <= WIDE_BAND) return DECNET_ERR_UNSUPPORTED;
    if (capturing(stream)) return DECNET_ERR_UNSUPPORTED;
    const int nb = ceil_div(D, WIDE_BAND), Db = ceil_div(D, nb);
    const size_t np = (size_t)B * H * W, rowsF = (size_t)B * C * H, rowsM = (size_t)B * H;
    Scratch sc(stream);
    CK(sc.get((size_t)(C + 8) * np + 2 * np));
    float *Rb = sc.p, *tmb = Rb + (size_t)C * np, *qb = tmb + np, *Sb = qb + np, *mb = Sb + np, *db = mb + np,
          *S2 = db + np, *m2 = S2 + np, *rmf = m2 + np, *tmf = rmf + np;
    if (mbits) {
        hipLaunchKernelGGL(unpack_bits, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream,
                           reinterpret_cast<const unsigned long long *>(rmask), rmf, rowsM, W);
        hipLaunchKernelGGL(unpack_bits, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream,
                           reinterpret_cast<const unsigned long long *>(tmask), tmf, rowsM, W);
        LAUNCH_OK();
        rmask = rmf;
        tmask = tmf;
    }
    // one sweep over the bands for a quotient q in {disparity (SpaMat), variance (SpaVar)}: band 0 lands in (q, S, m)
    // itself, the others in (qb, Sb, mb) and are merged in
    auto sweep = [&](int var, const float *disp, float *q, float *S, float *m) -> int {
        for (int b = 0; b < nb; ++b) {
            const int d0 = b * Db, dw = (D - d0 < Db) ? D - d0 : Db;
            const float *Rv = tar, *tv = tmask, *dv = disp;
            if (b) {
                hipLaunchKernelGGL(shift_planes, dim3(ew_grid(rowsF * W)), dim3(EW_THREADS), 0, stream, tar, Rb, rowsF, W, d0);
                hipLaunchKernelGGL(shift_planes, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, tmask, tmb, rowsM, W, d0);
                if (var)
                    hipLaunchKernelGGL(offset_plane, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, disp, db, np, (float)d0);
                LAUNCH_OK();
                Rv = Rb; tv = tmb; dv = db;
            }
            float *qo = b ? qb : q, *So = b ? Sb : S, *mo = b ? mb : m;
            CK(decnet_mfma_forward(var ? 1 : 0, ref, Rv, rmask, tv, var ? dv : nullptr, var ? nullptr : qo, var ? qo : nullptr,
                                   So, mo, B, C, H, W, dw, allow_compact, 0, stream));
            if (b) {
                hipLaunchKernelGGL(merge_band, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, rmask, q, S, m, qb, Sb, mb, np,
                                   var ? 0.f : (float)d0);
                LAUNCH_OK();
            }
        }
        return 0;
    };
    if (mode == 1) return sweep(1, disparity, var_out, sum_sim, max_cost);
    if (mode == 0) return sweep(0, nullptr, out, sum_sim, max_cost);
    // fused: one sweep of fused band calls (each band's variance around its own disparity), merged around the joint mean
    (void)S2; (void)m2;
    for (int b = 0; b < nb; ++b) {
        const int d0 = b * Db, dw = (D - d0 < Db) ? D - d0 : Db;
        if (!b) {
            CK(decnet_mfma_forward(2, ref, tar, rmask, tmask, nullptr, out, var_out, sum_sim, max_cost, B, C, H, W, dw,
                                   allow_compact, 0, stream));
            continue;
        }
        hipLaunchKernelGGL(shift_planes, dim3(ew_grid(rowsF * W)), dim3(EW_THREADS), 0, stream, tar, Rb, rowsF, W, d0);
        hipLaunchKernelGGL(shift_planes, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, tmask, tmb, rowsM, W, d0);
        LAUNCH_OK();
        CK(decnet_mfma_forward(2, ref, Rb, rmask, tmb, nullptr, qb, db, Sb, mb, B, C, H, W, dw, allow_compact, 0, stream));
        hipLaunchKernelGGL(merge_band_fused, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, rmask, out, var_out, sum_sim,
                           max_cost, qb, db, Sb, mb, np, (float)d0);
        LAUNCH_OK();
    }
    return 0;
}

// var: 0 SpaMat, 1 SpaVar (also grad_disp).
int decnet_wide_backward(int var, const float *ref, const float *tar, const float *rmask, const float *tmask,
                         const float *disparity, const float *out, const float *sum_sim, const float *max_cost,
                         const float *grad_out, float *grad_ref, float *grad_tar, float *grad_disp, int B, int C, int H,
                         int W, int D, hipStream_t stream) {
    if (D <= WIDE_BAND) return DECNET_ERR_UNSUPPORTED;
    if (capturing(stream)) return DECNET_ERR_UNSUPPORTED;
    const int nb = ceil_div(D, WIDE_BAND), Db = ceil_div(D, nb);
    const size_t np = (size_t)B * H * W, nf = (size_t)C * np, rowsF = (size_t)B * C * H, rowsM = (size_t)B * H;
    Scratch sc(stream);
    CK(sc.get(3 * nf + 3 * np));
    float *Rb = sc.p, *glb = Rb + nf, *grb = glb + nf, *tmb = grb + nf, *sh = tmb + np, *gdb = sh + np;
    for (int b = 0; b < nb; ++b) {
        const int d0 = b * Db, dw = (D - d0 < Db) ? D - d0 : Db;
        if (!b) {
            CK(decnet_mfma_backward(var, ref, tar, rmask, tmask, disparity, out, sum_sim, max_cost, grad_out, grad_ref, grad_tar,
                                    grad_disp, B, C, H, W, dw, stream));
            continue;
        }
        hipLaunchKernelGGL(shift_planes, dim3(ew_grid(rowsF * W)), dim3(EW_THREADS), 0, stream, tar, Rb, rowsF, W, d0);
        hipLaunchKernelGGL(shift_planes, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, tmask, tmb, rowsM, W, d0);
        // the plane that is a disparity moves with the band: SpaMat's output (SM_kernel.cu:191), SpaVar's input (SV_kernel.cu:191)
        hipLaunchKernelGGL(offset_plane, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, var ? disparity : out, sh, np, (float)d0);
        LAUNCH_OK();
        CK(decnet_mfma_backward(var, ref, Rb, rmask, tmb, var ? sh : nullptr, var ? out : sh, sum_sim, max_cost, grad_out, glb, grb,
                                var ? gdb : nullptr, B, C, H, W, dw, stream));
        hipLaunchKernelGGL(accumulate_planes, dim3(ew_grid(nf)), dim3(EW_THREADS), 0, stream, grad_ref, glb, rowsF, W, 0);
        hipLaunchKernelGGL(accumulate_planes, dim3(ew_grid(nf)), dim3(EW_THREADS), 0, stream, grad_tar, grb, rowsF, W, d0);
        if (var) hipLaunchKernelGGL(accumulate_planes, dim3(ew_grid(np)), dim3(EW_THREADS), 0, stream, grad_disp, gdb, rowsM, W, 0);
        LAUNCH_OK();
    }
    return 0;
}
