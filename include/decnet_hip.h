/*
 * include/decnet_hip.h -- C ABI of libdecnet_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for DecNet's data-parallel hot path.  Every entry point takes
 * plain device pointers, sizes and a hipStream_t passed as void*; no torch types.
 * Each one names the reference interface (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - All tensors are fp32, dense, row-major in the layout stated per argument.
 *     Feature maps are NCHW, masks / per-pixel planes are N,H,W  (functions/SpaMat.py:13-16).
 *   - Device pointers must be valid on the device that is current when the call is made;
 *     work is enqueued on `stream` (NULL = the null stream) and the call returns
 *     immediately (asynchronous, like the reference's launches SM_kernel.cu:384-386).
 *   - Re-entrant: no global mutable state; safe to call from one host thread per GPU
 *     (the reference is driven that way by DataParallel, eval.py:146).
 *   - Outputs are fully written by the callee (entries the reference leaves to the
 *     caller's zero fill, functions/SpaMat.py:25-27,42-43, are written as 0), so the
 *     caller does not have to clear them.  Inputs are never modified.
 *   - Inputs must be finite.  The reference propagates a NaN feature into every output it
 *     touches (fmaxf/expf on NaN, SM_kernel.cu:46-58); the SpaMat/SpaVar forward kernels here
 *     are built with -fno-honor-nans (decnet_amd/build.py), so a NaN input gives an unspecified
 *     value at the pixels whose candidate set contains it -- never a fault, never an effect on
 *     other pixels.  DECNET_CHECK_FINITE=1 (environment, read once) makes every SpaMat/SpaVar forward entry
 *     sweep both feature maps first (one reduction kernel each, then a 4-byte read-back: the call
 *     waits for the stream) and return DECNET_ERR_NONFINITE with nothing else launched; the check is
 *     skipped while the stream is being captured into a graph.
 *   - Return value: 0 (DECNET_OK) on success, a negative DECNET_ERR_* for rejected
 *     arguments (nothing is enqueued), or a positive hipError_t from the launch.
 *     (The reference's pybind functions always return 1 and check nothing,
 *     SM_cuda.cpp:7-27; the Python shim in decnet_amd/ext.py restores that `1`.)
 */
#ifndef DECNET_HIP_H
#define DECNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DECNET_OK 0
#define DECNET_ERR_NULL_POINTER (-1)
#define DECNET_ERR_BAD_SHAPE (-2)     /* non-positive dim, max_disp < 1, index space > 2^31 */
#define DECNET_ERR_UNSUPPORTED (-3)   /* shape does not fit the kernels' LDS tiling */
#define DECNET_ERR_NONFINITE (-4)     /* DECNET_CHECK_FINITE=1 and a feature map holds a NaN / Inf (nothing launched) */

/* Library / build identification: "decnet_hip <version> gfx950". */
const char *decnet_version(void);

/* ---------------------------------------------------------------------------------------
 * SpaMat forward.  Replaces sparse_matching_cuda_forward (SM_cuda.cpp:7-15), i.e. the two
 * launches get_max_cost + sparse_matching_forward (SM_kernel.cu:22-60, 76-125, 359-387).
 *   ref, tar            [B,C,H,W]  left / right feature maps
 *   ref_mask, tar_mask  [B,H,W]    0 = off, anything else = on
 *   output              [B,H,W]    soft-argmax disparity   (0 where ref_mask == 0)
 *   sum_similarities    [B,H,W]    1e-6 + sum_d exp(cost_d - max_cost)
 *   max_cost            [B,H,W]    max(1e-6, max_d cost_d)
 *   max_disp            candidates d in [0, min(max_disp, x+1))
 * ------------------------------------------------------------------------------------- */
int decnet_spamat_forward(const float *ref, const float *tar, const float *ref_mask,
                          const float *tar_mask, float *output, float *sum_similarities,
                          float *max_cost, int B, int C, int H, int W, int max_disp,
                          void *stream);

/* SpaMat backward.  Replaces sparse_matching_cuda_backward (SM_cuda.cpp:17-27): the launches
 * sparse_matching_ref_backward + sparse_matching_tar_backward (SM_kernel.cu:143-195, 300-355,
 * 389-429).  grad_ref / grad_tar [B,C,H,W] are fully written.                              */
int decnet_spamat_backward(const float *ref, const float *tar, const float *ref_mask,
                           const float *tar_mask, const float *output,
                           const float *sum_similarities, const float *max_cost,
                           const float *grad_output, float *grad_ref, float *grad_tar,
                           int B, int C, int H, int W, int max_disp, void *stream);

/* SpaVar forward.  Replaces sparse_var_cuda_forward (SV_cuda.cpp:7-17; kernels
 * SV_kernel.cu:22-60, 76-124, 329-359): output = (1e-6 + sum_d e_d (d - disparity)^2) / S. */
int decnet_spavar_forward(const float *ref, const float *tar, const float *ref_mask,
                          const float *tar_mask, const float *disparity, float *output,
                          float *sum_similarities, float *max_cost, int B, int C, int H, int W,
                          int max_disp, void *stream);

/* SpaVar backward.  Replaces sparse_var_cuda_backward (SV_cuda.cpp:19-32; kernels
 * SV_kernel.cu:142-325, 361-410).  grad_disparity [B,H,W].                                */
int decnet_spavar_backward(const float *ref, const float *tar, const float *ref_mask,
                           const float *tar_mask, const float *disparity, const float *output,
                           const float *sum_similarities, const float *max_cost,
                           const float *grad_output, float *grad_ref, float *grad_tar,
                           float *grad_disparity, int B, int C, int H, int W, int max_disp,
                           void *stream);

/* Fused SpaMat + SpaVar forward for the only way the model uses SpaVar
 * (SparseDenseNetRefinementMask.py:183-192: same features and masks, disparity = SpaMat's
 * output, under no_grad).  One read of ref/tar instead of four.
 *   output, variance, sum_similarities, max_cost   [B,H,W]                                 */
int decnet_spamatvar_forward(const float *ref, const float *tar, const float *ref_mask,
                             const float *tar_mask, float *output, float *variance,
                             float *sum_similarities, float *max_cost, int B, int C, int H,
                             int W, int max_disp, void *stream);
/* decnet_spamatvar_forward with bit-packed masks: ref_bits / tar_bits [B,H,ceil(W/64)] 64-bit words, bit i of word w
 * of a row = pixel 64 w + i, zero past W (what decnet_detail_mask writes beside the float plane of the reference's
 * contract).  Same results as the float-mask call; 8 of the pass's 88 bytes per pixel (C = 8) are not read.
 * Above max_disp 272 the range is done in bands of <= 272 with the masks unpacked into scratch planes (stream-ordered
 * allocation); DECNET_ERR_UNSUPPORTED there only while `stream` is being captured into a graph. */
int decnet_spamatvar_forward_bits(const float *ref, const float *tar, const unsigned long long *ref_bits,
                                  const unsigned long long *tar_bits, float *output, float *variance,
                                  float *sum_similarities, float *max_cost, int B, int C, int H, int W,
                                  int max_disp, void *stream);

/* ---------------------------------------------------------------------------------------
 * Stage 0 (coarsest level): dense cost volume -> 3-D conv aggregation -> soft-argmax.
 * Internal activation layout is channels-last  [B, D, H, W, C]  ("NDHWC").
 * ------------------------------------------------------------------------------------- */

/* Cost volume, cost_func="cor", warp_ope="homgrp" with disp_samples = arange(D)
 * (get_disp_samples submodule.py:389-390; GetCostVolume submodule.py:479-522, 532-562):
 *   cost[b,d,y,x,c] = (x >= d ? left[b,c,y,x] : 0) * bilinear(right[b,c]; xs, ys)
 *   xs = (x-d)*W/(W-1) - 0.5,  ys = y*H/(H-1) - 0.5, zero padding   (grid_sample with
 *   align_corners=False on align_corners=True-normalised coordinates).
 *   left,right [B,C,H,W];  cost_ndhwc [B,D,H,W,C].                                         */
int decnet_costvol_forward(const float *left, const float *right, float *cost_ndhwc, int B,
                           int C, int H, int W, int D, void *stream);
/* The other cost functions of GetCostVolume (submodule.py:511-530, chosen by --cost_func, demo.py:31 / eval.py:43), with
 * l = (x >= d ? left[b,c,y,x] : 0) and r = bilinear(right[b,c]; ...) as above:
 *   DECNET_COST_COR  l * r                               (:518-522; what decnet_costvol_forward computes)
 *   DECNET_COST_SSD  (l^2 + r^2) / 2 - ((l + r) / 2)^2   (:524-530, the same fp32 operation sequence)
 *   DECNET_COST_CAT  cost[..., c] = l, cost[..., C + c] = r: cost_ndhwc is [B,D,H,W,2C]   (:512-516)
 *   DECNET_COST_SUM  l + r -- not a reference function: CostRegNetNoDown.conv_pre (the 1x1x1 convolution behind the
 *                    "cat" volume, :618-619) is linear, so conv_pre(cat(l_vol, r_vol)) is the SUM volume of the two
 *                    feature maps after each went through its half of conv_pre (decnet_stage0_forward_cf does that). */
#define DECNET_COST_COR 0
#define DECNET_COST_SSD 1
#define DECNET_COST_CAT 2
#define DECNET_COST_SUM 3
int decnet_costvol_forward_cf(const float *left, const float *right, float *cost_ndhwc, int B,
                              int C, int H, int W, int D, int cost_func, void *stream);
/* Conv3d with kernel 1, stride 1, no bias (CostRegNetNoDown.conv_pre, submodule.py:618-619, 651-652):
 *   y[b,co,p] = sum_ci w[co*ldw + ci] * x[b,ci,p],  p < P positions per sample, one fp32 fma chain in channel order;
 *   channels_last = 0: x [B,Ci,P], y [B,Co,P] (NCHW / NCDHW);  1: x [B,P,Ci], y [B,P,Co] (the NDHWC volumes above).
 *   ldw >= Ci: row pitch of w (2C when one half of conv_pre.weight [C,2C,1,1,1] is applied to one feature map). */
int decnet_conv3d_pointwise(const float *x, const float *w, float *y, int B, int Ci, int Co, int P, int ldw,
                            int channels_last, void *stream);

/* Repack one Conv3d weight  [Co,Ci,3,3,3] (torch layout, submodule.py:109) into the kernels'
 * [27, Ci, CoP] layout, CoP = decnet_conv3d_packed_cout(Co), zero padded.                  */
int decnet_conv3d_packed_cout(int Co);
int decnet_conv3d_pack_weight(const float *w_oidhw, float *w_packed, int Co, int Ci,
                              void *stream);

/* One Conv3dUnit in eval mode (submodule.py:115-123: Conv3d k3 s1 p1 no bias ->
 * BatchNorm3d(running stats) -> ReLU), channels-last in and out:
 *   y = act(conv(x) * scale[co] + shift[co]) (+ residual)
 *   scale = gamma / sqrt(var + eps),  shift = beta - mean * scale   (caller folds BN)
 *   x [B,D,H,W,Ci];  w_packed from decnet_conv3d_pack_weight;  y [B,D,H,W,Co];
 *   residual: NULL or [B,D,H,W,Co], added after the activation (CostRegNetNoDown.forward
 *   submodule.py:656: conv1(o0) + o0);  relu: 0/1.                                         */
int decnet_conv3d_bn_act(const float *x, const float *w_packed, const float *scale,
                         const float *shift, const float *residual, float *y, int B, int D,
                         int H, int W, int Ci, int Co, int relu, void *stream);

/* The same Conv3dUnit by Winograd minimal filtering in fp32 (differs from decnet_conv3d_bn_act by
 * fp32 rounding only).  variant 0: F(2,3) on D,H,W -- 64 transform points, 8 multiplies per output
 * instead of 27, ~1e-6 relative;  variant 1: F(2,3) on D, F(4,3) on H and W -- 144 points, 4.5
 * multiplies per output, ~2e-6 relative;  variant 2: F(4,3) on D, H and W -- 216 points, 3.4 multiplies
 * per output, ~4e-6 relative (F(4,3) on the points {0, +-3/4, +-3/2, inf}).
 *   u          weights transformed once by decnet_conv3d_wino_pack_weight:
 *              [Co,Ci,3,3,3] -> U^T [points][ceil(Ci/16)][224][16]
 *              (decnet_conv3d_wino_weight_floats floats)
 *   workspace  decnet_conv3d_wino_workspace_floats(...) floats of device scratch
 *   everything else as decnet_conv3d_bn_act.                                                */
size_t decnet_conv3d_wino_weight_floats(int Ci, int variant);
int decnet_conv3d_wino_pack_weight(const float *w_oidhw, float *u, int Co, int Ci, int variant,
                                   void *stream);
size_t decnet_conv3d_wino_workspace_floats(int B, int D, int H, int W, int Ci, int Co, int variant);
/* its GEMM stage alone: M[xi] = V[xi] * U[xi] for every transform point xi, 16 channels (64 bytes)
 * being the unit of all three layouts:
 *   V [points][ceil(Ci/16)][nt][16] (transformed input tiles), M [points][ceil(Co/16)][nt][16]. */
int decnet_conv3d_wino_gemm(const float *V, const float *u, float *M, int nt, int Ci, int Co,
                            int variant, void *stream);
int decnet_conv3d_wino_bn_act(const float *x, const float *u, const float *scale,
                              const float *shift, const float *residual, float *y,
                              float *workspace, int B, int D, int H, int W, int Ci, int Co,
                              int relu, int variant, void *stream);

/* A stack of n_layers >= 2 consecutive C -> C Conv3dUnits (submodule.py:115-123) with the activations between the
 * layers kept on chip: one kernel does the output transform of layer i (BN, ReLU, residual) into LDS and the input
 * transform of layer i + 1 out of it.  u / scale / shift: per-layer pointers as for decnet_conv3d_wino_bn_act.
 * res_src, res_dst: the output of layer res_src is added to the output of layer res_dst after its ReLU
 * (CostRegNetNoDown.forward submodule.py:653-658: 1 and 4), or -1, -1; needs res_src < res_dst < n_layers - 1.
 * DECNET_ERR_UNSUPPORTED (nothing launched; decnet_conv3d_wino_stack_workspace_floats returns 0) when the fused
 * kernels do not cover the shape: variant 2, C = 216, one sample x 4 channels of the volume <= 160 KB of LDS. */
size_t decnet_conv3d_wino_stack_workspace_floats(int B, int D, int H, int W, int C, int variant);
int decnet_conv3d_wino_stack_bn_act(const float *x, const float *const *u, const float *const *scale,
                                    const float *const *shift, int n_layers, int res_src, int res_dst, float *y,
                                    float *workspace, int B, int D, int H, int W, int C, int variant, void *stream);
/* The same stack fed by the stage-0 cost volume of (left, right) [B,C,H,W] -- the values decnet_costvol_forward writes
 * (GetCostVolume, submodule.py:479-522), D disparity planes -- without that volume ever reaching HBM: the first layer's
 * input transform is formed on chip from the two feature maps.  Workspace as above.  DECNET_ERR_UNSUPPORTED (nothing
 * launched) where decnet_conv3d_wino_stack_bn_act is, or when the feature planes do not fit in LDS beside the volume. */
int decnet_costvol_wino_stack_bn_act(const float *left, const float *right, const float *const *u,
                                     const float *const *scale, const float *const *shift, int n_layers, int res_src,
                                     int res_dst, float *y, float *workspace, int B, int C, int H, int W, int D,
                                     int variant, void *stream);
/* ... with the cost function DECNET_COST_COR / _SSD / _SUM (DECNET_ERR_BAD_SHAPE for _CAT: see decnet_stage0_forward_cf). */
int decnet_costvol_wino_stack_bn_act_cf(const float *left, const float *right, const float *const *u,
                                        const float *const *scale, const float *const *shift, int n_layers, int res_src,
                                        int res_dst, float *y, float *workspace, int B, int C, int H, int W, int D,
                                        int variant, int cost_func, void *stream);
/* Last Conv3dUnit (Ci -> 1, BN, no ReLU; submodule.py:641) fused with disparity_regression
 * over disp_samples = arange(D) (submodule.py:766-777):
 *   reg[b,d,y,x]  = conv(x)[b,d,y,x] * scale + shift        (optional output, may be NULL)
 *   pred[b,y,x]   = sum_d softmax_d(reg)[d] * d
 *   x [B,D,H,W,Ci];  w [1,Ci,3,3,3] torch layout (no repack needed).                       */
int decnet_conv3d_cout1_softargmax(const float *x, const float *w_oidhw, float scale,
                                   float shift, float *reg, float *pred, int B, int D, int H,
                                   int W, int Ci, void *stream);

/* The same operator in two passes (a [positions x Ci] x [Ci x 27] product on the matrix cores, then
 * a 27-tap gather + soft-argmax) that read x once instead of 27 times.  workspace:
 * decnet_conv3d_cout1_workspace_floats(B,D,H,W) floats of device scratch.  Ci <= 256, D <= 256. */
size_t decnet_conv3d_cout1_workspace_floats(int B, int D, int H, int W);
int decnet_conv3d_cout1_softargmax_ws(const float *x, const float *w_oidhw, float scale,
                                      float shift, float *reg, float *pred, float *workspace,
                                      int B, int D, int H, int W, int Ci, void *stream);

/* ---------------------------------------------------------------------------------------
 * The whole stage-0 branch in one call: replaces SparseDenseNetRefinementMask.forward :127-137, i.e.
 * get_disp_samples (submodule.py:389-390) -> GetCostVolume.forward (:532-562) ->
 * CostRegNetNoDown.forward (:650-662) -> disparity_regression (:766-777) for the shipped configuration
 * (cost_func="cor", warp_ope="homgrp", 8 Conv3dUnits C -> C ... C -> 1).
 *   left, right [B,C,H,W] feature maps of the coarsest level;  D = max_disp / 27 hypotheses 0 .. D-1
 *   params      the seven C -> C layers: w[i] = decnet_conv3d_wino_pack_weight(...) of the layer for
 *               variant 0..2 (decnet_conv3d_pack_weight for variant 3 = direct 27-tap GEMM), scale / shift =
 *               folded BatchNorm3d [C]; the last layer: w_last [1,C,3,3,3] (torch layout), scale_last, shift_last
 *   variant     -1 = automatic (F(4,3)^3 where depth tiles of 4 pay, else F(2,3)xF(4,3)^2), else as above
 *   workspace   decnet_stage0_workspace_floats(B,C,H,W,D,variant) floats of device scratch (cost volume,
 *               three ping-pong activation buffers, the Winograd intermediates); contents are scratch
 *   reg         NULL or [B,D,H,W]: the regularised volume (what CostRegNetNoDown returns)
 *   pred        [B,H,W]: the stage-0 disparity
 * C must be a multiple of 4 (pad features and weights with zero channels otherwise).             */
typedef struct decnet_stage0_params {
    const float *w[7];
    const float *scale[7];
    const float *shift[7];
    const float *w_last;
    float scale_last, shift_last;
} decnet_stage0_params;
size_t decnet_stage0_workspace_floats(int B, int C, int H, int W, int D, int variant);
int decnet_stage0_forward(const float *left, const float *right, const decnet_stage0_params *params,
                          float *workspace, float *reg, float *pred, int B, int C, int H, int W, int D,
                          int variant, void *stream);
/* The same branch for every --cost_func of the reference (submodule.py:552-560): cost_func = DECNET_COST_COR, _SSD or
 * _CAT.  w_pre: CostRegNetNoDown.conv_pre.weight [C,2C,1,1,1] (torch layout) for _CAT, ignored (may be NULL) otherwise.
 * _CAT never forms the 2C-channel volume: conv_pre is applied to the two feature maps (its left half to `left`, its right
 * half to `right`: two [C x C] products over B*H*W pixels instead of one [C x 2C] over B*D*H*W voxels -- bilinear warping
 * and the left mask commute with a per-pixel linear map) and the stack runs on their DECNET_COST_SUM volume.
 * Workspace: decnet_stage0_cf_workspace_floats floats. */
size_t decnet_stage0_cf_workspace_floats(int B, int C, int H, int W, int D, int variant, int cost_func);
int decnet_stage0_forward_cf(const float *left, const float *right, const decnet_stage0_params *params,
                             const float *w_pre, float *workspace, float *reg, float *pred, int B, int C, int H, int W,
                             int D, int variant, int cost_func, void *stream);

/* disparity_regression for arbitrary samples (submodule.py:766-777):
 *   cost, samples [B,S,H,W] -> pred [B,H,W]                                                */
int decnet_disparity_regression(const float *cost, const float *samples, float *pred, int B,
                                int S, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------
 * 2-D trunk: the full-resolution, few-channel Conv2dUnit / Deconv2dUnit layers
 * (modules/submodule.py:15-87) in eval mode, conv -> BatchNorm2d(running stats) -> ReLU fused:
 *   y = act(conv(x) * scale[co] + shift[co]);  x [B,Cin,H,W], y [B,Cout,H',W'] NCHW;
 *   scale/shift: folded BN (or 1 / bias when the unit has no BN).  Cout <= 24 (<= 8 for the
 *   transposed convolution).
 * Weights are repacked once by decnet_conv2d_pack_weight into decnet_conv2d_packed_floats(...)
 * floats: torch [Cout,Cin,k,k] (transposed = 0) or ConvTranspose2d [Cin,Cout,3,3] (transposed = 1)
 * -> [Cin][k][k][co padded].
 * decnet_conv2d_bn_act: k = 1 or 3, stride 1, padding dilation*(k/2) (output size = input size).
 * decnet_deconv2d_k3s3_bn_act: ConvTranspose2d k = 3, stride 3, padding 0 (output 3H x 3W;
 *   submodule.py:162-178 Deconv2dBlock, GenerateSparseMask).
 * ------------------------------------------------------------------------------------- */
size_t decnet_conv2d_packed_floats(int Cin, int Cout, int k, int transposed);   /* 0: unsupported */
int decnet_conv2d_pack_weight(const float *w, float *w_packed, int Cin, int Cout, int k,
                              int transposed, void *stream);
int decnet_conv2d_bn_act(const float *x, const float *w_packed, const float *scale,
                         const float *shift, float *y, int B, int Cin, int Cout, int H, int W, int k, int dilation,
                         int relu, void *stream);
/* decnet_conv2d_bn_act on the channel concatenation of nseg (<= 6) tensors xs[i] [B,cins[i],H,W]
 * (host arrays of device pointers / channel counts; weights packed for Cin = sum of cins): the
 * torch.cat in front of the Deconv2dBlock / Refinement / SoftAttention convolutions
 * (submodule.py:176, 755, SparseDenseNetRefinementMask.py:195-199) is never materialised. */
int decnet_conv2d_cat_bn_act(const float *const *xs, const int *cins, int nseg, const float *w_packed,
                             const float *scale, const float *shift, float *y, int B, int Cout, int H,
                             int W, int k, int dilation, int relu, void *stream);
/* decnet_conv2d_cat_bn_act for a layer with ONE output channel, with the elementwise tail of its caller fused
 * (ea, eb: [B,H,W] planes; y [B,1,H,W]):
 *   epilogue 1  SoftAttention's last layer + the fusion of the stage loop (submodule.py:593-604,
 *               SparseDenseNetRefinementMask.py:195-202): s = sigmoid(conv), y = ea (1 - s) + s eb   (ea dense, eb sparse)
 *   epilogue 2  Refinement's last layer + residual (submodule.py:716): y = ea + conv                             */
int decnet_conv2d_cat_epilogue(const float *const *xs, const int *cins, int nseg, const float *w_packed,
                               const float *scale, const float *shift, float *y, int B, int H, int W, int k,
                               int dilation, int relu, int epilogue, const float *ea, const float *eb, void *stream);
/* The many-channel stride-1 Conv2dUnit layers (FeatureExtraction conv1-conv3_2 and the Deconv2dBlock convolutions
 * submodule.py:245-343, 162-178; DynamicUpsampling.weight_learning :566-577; Refinement :690-717) on the bf16
 * matrix cores at fp32 accuracy (each fp32 operand split into three bf16 terms, six partial products): k = 1 or 3,
 * stride 1, padding dilation*(k/2), any Cin / Cout, input = channel concatenation of nseg (<= 6) tensors.
 * Weights: torch [Cout,Cin,k,k] packed once into decnet_conv2d_mfma_packed_bytes(...) bytes (0: unsupported). */
size_t decnet_conv2d_mfma_packed_bytes(int Cin, int Cout, int k);
int decnet_conv2d_mfma_pack_weight(const float *w, void *w_packed, int Cin, int Cout, int k, void *stream);
int decnet_conv2d_mfma_cat_bn_act(const float *const *xs, const int *cins, int nseg, const void *w_packed,
                                  const float *scale, const float *shift, float *y, int B, int Cout, int H,
                                  int W, int k, int dilation, int relu, void *stream);
/* Deconv2dUnit (ConvTranspose2d k = 3, stride 3, padding 0 + BN + ReLU, submodule.py:48-87; FeatureExtraction's
 * deconv2 / deconv3 at 24 / 72 output channels) on the same kernels: every input pixel owns its 3 x 3 output block, so
 * the layer is the 1 x 1 convolution to 9 Cout channels with a pixel-shuffle store.  w: torch [Cin,Cout,3,3];
 * x [B,Cin,H,W] -> y [B,Cout,3H,3W]. */
size_t decnet_deconv2d_mfma_packed_bytes(int Cin, int Cout);
int decnet_deconv2d_mfma_pack_weight(const float *w, void *w_packed, int Cin, int Cout, void *stream);
int decnet_deconv2d_mfma_k3s3_bn_act(const float *x, const void *w_packed, const float *scale, const float *shift,
                                     float *y, int B, int Cin, int Cout, int H, int W, int relu, void *stream);
/* Refinement.get_warped_feats_by_homgrp (submodule.py:719-745): out[b,c,y,x] = bilinear(right[b,c];
 * (x - disp[b,y,x]) * W/(W-1) - 0.5, y * H/(H-1) - 0.5), zero padding.  right,out [B,C,H,W], disp [B,H,W]. */
int decnet_warp_disparity(const float *right, const float *disp, float *out, int B, int C, int H, int W,
                          void *stream);
/* Tail of DynamicUpsampling.forward (submodule.py:581-589), down_scale 3: logits [B,81,h,w] (channel =
 * 9 * sub-position + neighbour), disp [B,h,w] -> out [B,3h,3w] =
 * 3 * pixel_shuffle(sum_k softmax_k(logits) * replicate-padded 3x3 neighbours of disp). */
int decnet_dynamic_upsample3(const float *logits, const float *disp, float *out, int B, int h, int w,
                             void *stream);
/* Tail of GenerateSparseMask.forward + the thresholding of the model loop in one pass (submodule.py:366-372:
 * res_info = (cur_fea - pre_fea)^2 -> Conv2dUnit(3,3,3x3,BN) -> Conv2dUnit(3,1,1x1,BN) = detail;
 * SparseDenseNetRefinementMask.py:158-170: mask = sigmoid(detail) > thold ? 1 : 0).
 *   cur3, pre3   [B,3,H,W] device: outputs of GenerateSparseMask.conv_sub / .deconv
 *   w3x3 [3,3,3,3] (torch [co,ci,ky,kx]), scale3/shift3 [3], w1x1 [3], scale1, shift1: HOST values, BatchNorm
 *                folded (scale = gamma / sqrt(var + eps), shift = beta - mean * scale); thold as float32
 *   mask         [B,H,W] float 0/1 (the reference's contract: what SpaMat / SoftAttention read)
 *   logits       NULL or [B,H,W]: `detail`
 *   bits         NULL or [B,H,ceil(W/64)] 64-bit words, bit i of word w = pixel 64 w + i of the row       */
int decnet_detail_mask(const float *cur3, const float *pre3, const float *w3x3, const float *scale3,
                       const float *shift3, const float *w1x1, float scale1, float shift1, float thold,
                       float *mask, float *logits, unsigned long long *bits, int B, int H, int W,
                       void *stream);
/* Head of DynamicUpsampling.forward (submodule.py:578-580): out = cat(disp, unfold(fea, 3, stride 3)):
 * fea [B,C,3h,3w], disp [B,h,w] -> out [B,9C+1,h,w], out[b,0] = disp, out[b,1+9c+3i+j,y,x] = fea[b,c,3y+i,3x+j]. */
int decnet_unfold3_cat(const float *fea, const float *disp, float *out, int B, int C, int h, int w,
                       void *stream);
/* Space-to-depth in front of the stride-3 convolutions of FeatExtNetChannelPlus (Conv2d k 3, stride 3, padding 1,
 * submodule.py:270-300): x [B,C,H,W] -> out [B,9C,Ho,Wo], Ho = (H-1)/3+1, Wo = (W-1)/3+1,
 * out[b,c*9+ky*3+kx,yo,xo] = x[b,c,3yo-1+ky,3xo-1+kx] (0 outside); the convolution is then the 1 x 1 convolution with the
 * weight [Cout,Cin,3,3] read as [Cout,9 Cin] (decnet_conv2d_mfma_cat_bn_act, k = 1). */
int decnet_s2d3_pad1(const float *x, float *out, int B, int C, int H, int W, void *stream);
/* y[b,c,:,:] = act(y[b,c,:,:] + shift[c]) in place, y [B,C,H,W]: the folded-BatchNorm bias and the ReLU
 * behind a library convolution, one pass.  B*C <= 65535. */
int decnet_bias_act_inplace(float *y, const float *shift, int B, int C, int H, int W, int relu,
                            void *stream);
/* Conv2d k = 3, stride 3, padding 1 (FeatExtNetChannelPlus down-sampling, submodule.py:245-343),
 * Cout <= 24; y [B,Cout,(H-1)/3+1,(W-1)/3+1]; weights packed with transposed = 0. */
int decnet_conv2d_k3s3_bn_act(const float *x, const float *w_packed, const float *scale,
                              const float *shift, float *y, int B, int Cin, int Cout, int H, int W,
                              int relu, void *stream);
int decnet_deconv2d_k3s3_bn_act(const float *x, const float *w_packed, const float *scale,
                                const float *shift, float *y, int B, int Cin, int Cout, int H,
                                int W, int relu, void *stream);

/* ---------------------------------------------------------------------------------------
 * Dilated 3x3 / 1x1 convolutions on small, many-channel images as a per-tap matrix product + gather:
 * the ASPP block of FeatExtNetChannelPlus (submodule.py:225-241).  All branches share the input:
 *   decnet_tapconv_to_chunks   x [B,Ci,H,W] -> V [ceil(Ci/16)][P=B*H*W][16]
 *   decnet_tapconv_pack_weight one branch's w [Co,Ci,k,k] -> taps tap0.. of u (decnet_tapconv_weight_floats)
 *   decnet_tapconv_split_weight after the last branch (optional): the bf16-term copy of u behind it
 *   decnet_tap_gemm            T[t] = V * u[t] for every tap t, T [ntaps][ceil(Co/16)][P][16].  split = 1 states that
 *                              decnet_tapconv_split_weight has run on this u since its last pack: the bf16x3 kernel
 *                              then reads that copy (Ci = 216); split = 0: fp32 MFMA on the packed matrices alone
 *   decnet_tapconv_gather      y[b, br*Co+co, y, x] = act(scale * sum_t T[t][co][p + offset] + shift),
 *                              taps outside the image skipped; y [B, nbranch*Co, H, W]
 * Ci % 4 == 0, Co <= 224, nbranch <= 4.
 * ------------------------------------------------------------------------------------- */
size_t decnet_tapconv_chunk_floats(int B, int Ci, int H, int W);
int decnet_tapconv_to_chunks(const float *x, float *V, int B, int Ci, int H, int W, void *stream);
size_t decnet_tapconv_weight_floats(int Ci, int ntaps);
int decnet_tapconv_pack_weight(const float *w, float *u, int Co, int Ci, int k, int tap0, void *stream);
int decnet_tapconv_split_weight(float *u, int Ci, int ntaps, void *stream);
int decnet_tap_gemm(const float *V, const float *u, float *M, int P, int Ci, int Co, int ntaps,
                    int split, void *stream);
int decnet_tapconv_gather(const float *M, const float *scale, const float *shift, float *y, int B,
                          int Co, int H, int W, int nbranch, const int *tap0, const int *k,
                          const int *dil, int relu, void *stream);

/* Layout helpers between the reference's [B,C,D,H,W] and the internal [B,D,H,W,C]. */
int decnet_ncdhw_to_ndhwc(const float *src, float *dst, int B, int C, int D, int H, int W,
                          void *stream);
int decnet_ndhwc_to_ncdhw(const float *src, float *dst, int B, int C, int D, int H, int W,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DECNET_HIP_H */
