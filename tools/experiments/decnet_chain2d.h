/* tools/experiments/decnet_chain2d.h -- NOT PART OF THE PRODUCT LIBRARY.
 * C ABI of the fused conv-chain experiment (tools/experiments/chain2d.hip): measured at parity with or slower than
 * the per-layer conv2d_small kernels, so the graph does not use it and
 * libdecnet_hip.so does not contain it.  tools/experiments/build.sh builds it into tools/experiments/libdecnet_chain2d.so. */
#ifndef DECNET_CHAIN2D_H
#define DECNET_CHAIN2D_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* ---------------------------------------------------------------------------------------
 * Fused chains of few-channel convolutions (csrc/chain2d.hip): source -> up to three Conv2dUnit layers with
 * <= 8 output channels each (3x3 with dilation d, padding d, stride 1; or 1x1) -> sink, ONE launch, intermediate
 * rows in LDS.  Covers, in eval mode (BatchNorm folded to scale / shift):
 *   FeatExtNetChannelPlus.conv0                      submodule.py:263-266   image -> 3x3 -> 3x3
 *   Deconv2dBlock of the finest level                submodule.py:162-178   cat(deconv(x), skip) -> 3x3 -> 3x3
 *   GenerateSparseMask (+ sigmoid > thold)           submodule.py:347-372, SparseDenseNetRefinementMask.py:158-170
 *   SoftAttention (+ the dense / sparse fusion)      submodule.py:593-604, SparseDenseNetRefinementMask.py:195-202
 *   Refinement: warp + first layers                  submodule.py:690-745
 * Source parts are concatenated along the channel axis in the order given:
 *   DECNET_PART_PLAIN   p [B,c,H,W] (p2: the tensor that holds samples b >= bsplit as its samples b - bsplit, or NULL)
 *   DECNET_PART_DECONV  8 channels = act(scale * ConvTranspose2d(k 3, stride 3)(p [B,cp,H/3,W/3]) + shift); aux = the
 *                       weights packed by decnet_conv2d_pack_weight(..., transposed = 1): [cp][3][3][8]
 *   DECNET_PART_WARP    c <= 8 channels = Refinement's disparity warp of p [B,c,H,W] by aux [B,H,W]
 *                       (= decnet_warp_disparity); generated parts must start at a multiple of 8 channels
 * Layer l: weights packed by decnet_chain2d_pack_weight into w_packed at the offset
 *   sum_{i<l} decnet_chain2d_packed_bytes(cin_i, k_i);  y = act(scale * conv(x) + shift);  epilogue
 *   DECNET_EPI_SUBSQ: y = (aux[b,co] - y)^2 (aux [B,aux_channels,H,W]).
 * Sinks: STORE out [B,cout,H,W]; BLEND (cout 1): out [B,H,W] = a (1 - s) + s b, s = sigmoid(y), a = sink_a, b = sink_b
 *   [B,H,W]; ADD (cout 1): out = sink_a + y; MASK (cout 3): z = mask_scale * sum_c mask_w[c] y_c + mask_shift,
 *   out [B,H,W] = sigmoid(z) > thold ? 1 : 0 and, if bits != NULL, the bit-packed copy [B,H,ceil(W/64)] words.
 * DECNET_ERR_UNSUPPORTED: more than 8 output channels, more than 96 source channels, H or W not a multiple of 3
 * with a DECONV part, LDS budget.                                                                               */
#define DECNET_CHAIN_MAX_LAYERS 3
#define DECNET_CHAIN_MAX_PARTS 6
#define DECNET_PART_PLAIN 0
#define DECNET_PART_DECONV 1
#define DECNET_PART_WARP 2
#define DECNET_EPI_AFFINE 0
#define DECNET_EPI_SUBSQ 1
#define DECNET_SINK_STORE 0
#define DECNET_SINK_BLEND 1
#define DECNET_SINK_ADD 2
#define DECNET_SINK_MASK 3
typedef struct decnet_chain_part {
    const float *p, *p2, *aux, *scale, *shift;
    int c, kind, cp, relu;
} decnet_chain_part;
typedef struct decnet_chain_layer {
    const float *scale, *shift;      /* HOST arrays [cout] */
    const float *aux;                /* device, DECNET_EPI_SUBSQ */
    int cin, cout, k, dilation, relu, epilogue, aux_channels;
} decnet_chain_layer;
typedef struct decnet_chain_desc {
    decnet_chain_part parts[DECNET_CHAIN_MAX_PARTS];
    decnet_chain_layer layers[DECNET_CHAIN_MAX_LAYERS];
    const void *w_packed;
    float *out;
    unsigned long long *bits;
    const float *sink_a, *sink_b;
    float mask_w[3], mask_scale, mask_shift, thold;
    int n_parts, n_layers, bsplit, B, H, W, sink;
    int force_tw, force_rows;        /* 0: planned; > 0: pin the strip width / rows per workgroup (tests, tuning) */
    int debug;                       /* 0.  Timing experiments only (results are wrong): 1 no MFMAs, 2 no source loads, 4 no
                                        source commit, 8 no output stores, 16 no level stores, 256 print the plan */
} decnet_chain_desc;
size_t decnet_chain2d_packed_bytes(int Cin, int k);
/* w [Cout,Cin,k,k] (torch Conv2d), sign NULL or [Cin] device (+-1: a negated input channel folded into the weights) */
int decnet_chain2d_pack_weight(const float *w, const float *sign, void *w_packed, int Cin, int Cout, int k,
                               void *stream);
int decnet_chain2d_forward(const decnet_chain_desc *desc, void *stream);

#ifdef __cplusplus
}
#endif
#endif
