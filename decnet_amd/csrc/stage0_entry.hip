// decnet_amd/csrc/stage0_entry.hip -- the whole stage-0 branch behind ONE C entry point.
//
// SparseDenseNetRefinementMask.forward :127-137 (get_disp_samples -> GetCostVolume -> CostRegNetNoDown ->
// disparity_regression; submodule.py:389-390, 479-522, 650-662, 766-777) = cost volume + 7 x Conv3dUnit
// (216 -> 216, residual around units 2-4) + the 216 -> 1 unit fused with the soft-argmax.  A C / C++ caller
// gets it with one call and one workspace instead of re-implementing the buffer ping-pong of
// decnet_amd/stage0.py (which itself goes through this entry).
#include <stdlib.h>

#include "common.h"

namespace {
// the algorithm the Python side picks too (stage0.py:conv_algo): F(4,3)^3 where depth tiles of 4 pay
inline int auto_variant(int D) { return 216 * ((D + 3) / 4) <= 0.92 * 144 * ((D + 1) / 2) ? 2 : 1; }
inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }
}  // namespace

// csrc/stage0.hip (library-internal)
extern "C" int decnet_conv3d_pointwise_pair(const float *x, const float *w, float *y, const float *x2, const float *w2,
                                            float *y2, int B, int Ci, int Co, int P, int ldw, void *stream);

extern "C" {

size_t decnet_stage0_cf_workspace_floats(int B, int C, int H, int W, int D, int variant, int cost_func) {
    if (cost_func != DECNET_COST_COR && cost_func != DECNET_COST_SSD && cost_func != DECNET_COST_CAT) return 0;
    const size_t n = decnet_stage0_workspace_floats(B, C, H, W, D, variant);
    // "cat": the two feature maps behind their halves of conv_pre
    return n && cost_func == DECNET_COST_CAT ? n + 2 * align64((size_t)B * C * H * W) : n;
}

size_t decnet_stage0_workspace_floats(int B, int C, int H, int W, int D, int variant) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || D < 1 || variant < -1 || variant > 3) return 0;
    if (variant < 0) variant = auto_variant(D);
    const size_t act = align64((size_t)B * D * H * W * C);
    size_t wino = 0;
    if (variant <= 2) {
        wino = decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, variant);
        if (!wino) return 0;
        // the fused stack (decnet_conv3d_wino_stack_bn_act) keeps its residual plane behind the V/M scratch
        const size_t stack = decnet_conv3d_wino_stack_workspace_floats(B, D, H, W, C, variant);
        if (stack > wino) wino = stack;
    }
    // cost volume + three activation buffers (+ the Winograd V/M scratch); the last layer's tap products
    // (decnet_conv3d_cout1_workspace_floats) reuse an activation buffer that is free by then when they fit
    const size_t last = decnet_conv3d_cout1_workspace_floats(B, D, H, W);
    return 4 * act + align64(wino) + (last > act ? align64(last) : 0);
}

int decnet_stage0_forward(const float *left, const float *right, const decnet_stage0_params *p,
                          float *workspace, float *reg, float *pred, int B, int C, int H, int W, int D,
                          int variant, void *stream) {
    return decnet_stage0_forward_cf(left, right, p, nullptr, workspace, reg, pred, B, C, H, W, D, variant, DECNET_COST_COR,
                                    stream);
}

int decnet_stage0_forward_cf(const float *left, const float *right, const decnet_stage0_params *p, const float *w_pre,
                             float *workspace, float *reg, float *pred, int B, int C, int H, int W, int D,
                             int variant, int cost_func, void *stream) {
    if (!left || !right || !p || !workspace || !pred || !p->w_last) return DECNET_ERR_NULL_POINTER;
    if (cost_func != DECNET_COST_COR && cost_func != DECNET_COST_SSD && cost_func != DECNET_COST_CAT)
        return DECNET_ERR_BAD_SHAPE;
    if (cost_func == DECNET_COST_CAT && !w_pre) return DECNET_ERR_NULL_POINTER;
    for (int i = 0; i < 7; ++i)
        if (!p->w[i] || !p->scale[i] || !p->shift[i]) return DECNET_ERR_NULL_POINTER;
    if (B < 1 || C < 1 || H < 1 || W < 1 || D < 1 || variant < -1 || variant > 3) return DECNET_ERR_BAD_SHAPE;
    if (C % 4) return DECNET_ERR_UNSUPPORTED;          // channel counts move in 16-byte groups (pad to x4)
    if (variant < 0) variant = auto_variant(D);
    if (!decnet_stage0_cf_workspace_floats(B, C, H, W, D, variant, cost_func)) return DECNET_ERR_UNSUPPORTED;
    const size_t act = align64((size_t)B * D * H * W * C);
    float *cv = workspace, *a = cv + act, *b = a + act, *c = b + act, *ws = c + act;
    const size_t stack = variant <= 2 ? decnet_conv3d_wino_stack_workspace_floats(B, D, H, W, C, variant) : 0;
    size_t wino = variant <= 2 ? decnet_conv3d_wino_workspace_floats(B, D, H, W, C, C, variant) : 0;
    wino = align64(stack > wino ? stack : wino);
    const size_t last = decnet_conv3d_cout1_workspace_floats(B, D, H, W);
    float *t_last = last > act ? ws + wino : a;

    int rc;
    int cf = cost_func;
    if (cost_func == DECNET_COST_CAT) {
        // conv_pre(cat(left_vol, right_vol)) (submodule.py:514, 651-652) = SUM volume of (W_l left, W_r right): the 1x1x1
        // convolution is per voxel and linear, the warp (bilinear, zero padded) and the x >= d mask are per channel
        if (H < 1 || W < 1 || (double)H * W >= 2147483648.0) return DECNET_ERR_BAD_SHAPE;
        const size_t fm = align64((size_t)B * C * H * W);
        float *pl = ws + wino + (last > act ? align64(last) : 0), *pr = pl + fm;
        if ((rc = decnet_conv3d_pointwise_pair(left, w_pre, pl, right, w_pre + C, pr, B, C, C, H * W, 2 * C, stream))) return rc;
        left = pl;
        right = pr;
        cf = DECNET_COST_SUM;
    }
    rc = DECNET_ERR_UNSUPPORTED;
    if (stack)      // cost volume + all seven layers, neither the volume nor the activations between the layers in HBM
        rc = decnet_costvol_wino_stack_bn_act_cf(left, right, p->w, p->scale, p->shift, 7, 1, 4, c, ws, B, C, H, W, D,
                                                 variant, cf, stream);
    if (rc == DECNET_OK) {
        if (C <= 256 && D <= 256)
            return decnet_conv3d_cout1_softargmax_ws(c, p->w_last, p->scale_last, p->shift_last, reg, pred, t_last, B,
                                                     D, H, W, C, stream);
        return decnet_conv3d_cout1_softargmax(c, p->w_last, p->scale_last, p->shift_last, reg, pred, B, D, H, W, C,
                                              stream);
    }
    if (rc != DECNET_ERR_UNSUPPORTED) return rc;
    if ((rc = decnet_costvol_forward_cf(left, right, cv, B, C, H, W, D, cf, stream))) return rc;
    auto conv = [&](int i, const float *src, float *dst, const float *res) {
        if (variant <= 2)
            return decnet_conv3d_wino_bn_act(src, p->w[i], p->scale[i], p->shift[i], res, dst, ws, B, D, H, W, C, C,
                                             1, variant, stream);
        return decnet_conv3d_bn_act(src, p->w[i], p->scale[i], p->shift[i], res, dst, B, D, H, W, C, C, 1, stream);
    };
    // CostRegNetNoDown.forward submodule.py:650-662
    rc = DECNET_ERR_UNSUPPORTED;
    if (stack)                                          // all seven layers with the activations between them on chip
        rc = decnet_conv3d_wino_stack_bn_act(cv, p->w, p->scale, p->shift, 7, 1, 4, c, ws, B, D, H, W, C, variant, stream);
    if (rc == DECNET_OK) {
        if (C <= 256 && D <= 256)
            return decnet_conv3d_cout1_softargmax_ws(c, p->w_last, p->scale_last, p->shift_last, reg, pred, t_last, B,
                                                     D, H, W, C, stream);
        return decnet_conv3d_cout1_softargmax(c, p->w_last, p->scale_last, p->shift_last, reg, pred, B, D, H, W, C,
                                              stream);
    }
    if (rc != DECNET_ERR_UNSUPPORTED) return rc;
    if ((rc = conv(0, cv, a, nullptr))) return rc;
    if ((rc = conv(1, a, c, nullptr))) return rc;       // c = output0
    if ((rc = conv(2, c, a, nullptr))) return rc;
    if ((rc = conv(3, a, b, nullptr))) return rc;
    if ((rc = conv(4, b, a, c))) return rc;             // conv1(output0) + output0
    if ((rc = conv(5, a, b, nullptr))) return rc;
    if ((rc = conv(6, b, c, nullptr))) return rc;
    if (C <= 256 && D <= 256)
        return decnet_conv3d_cout1_softargmax_ws(c, p->w_last, p->scale_last, p->shift_last, reg, pred, t_last, B,
                                                 D, H, W, C, stream);
    return decnet_conv3d_cout1_softargmax(c, p->w_last, p->scale_last, p->shift_last, reg, pred, B, D, H, W, C,
                                          stream);
}

}  // extern "C"
