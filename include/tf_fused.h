/*
 * include/tf_fused.h -- C ABI of the fused element-wise kernels of libtf_msda.so that sit between the
 * library GEMMs / convolutions of the per-frame path (hand-written HIP, gfx950).
 *
 * They replace chains of separate ATen element-wise kernels of the reference's eager PyTorch path:
 *
 *   tf_bias_act_f32        conv -> FrozenBatchNorm2d shift -> (+ identity) -> ReLU
 *                          (reference: models/backbone.py:45-55 FrozenBatchNorm2d.forward + torchvision
 *                          Bottleneck's `out += identity; out = relu(out)`): one in-place pass instead of 3-4
 *   tf_add_layernorm_f32   `src = src + dropout(src2); src = norm(src)` of every transformer layer
 *                          (reference: models/deformable_transformer.py:291-292, :285-286, :371-372, :378-379,
 *                          :360-361): residual add + LayerNorm in one pass
 *
 *   tf_linear_split_f32    nn.Linear (+ ReLU) of the encoder / decoder (ms_deform_attn.py:64-88,
 *                          deformable_transformer.py:282-297) as a split product on the matrix cores (fp16 pieces, three terms:
 *                          fp32-class, what trackformer_amd uses by default; or six bf16 terms -- THE SPLIT PRODUCT below)
 *   tf_linear_packed_f32   the same product with the weight packed once in fragment order (+ tf_linear_pack_weight_f32)
 *   tf_ffn_fused_f32       linear1 -> ReLU -> linear2 -> + residual -> LayerNorm of a transformer layer in one launch
 *   tf_linear_res_ln_f32   linear (256 -> 256) -> + residual -> LayerNorm in one launch (output projection + norm1)
 *   tf_mha_core_f32        softmax(q k^T * scale) v of the decoder's query self-attention
 *                          (deformable_transformer.py:364-383, nn.MultiheadAttention): one launch, fp32
 *
 * Conventions as in tf_msda.h: device pointers, caller-owned buffers, work enqueued on `stream`
 * (hipStream_t as void*), no synchronisation, returns 0 or a negative tf_msda_status.
 */
#ifndef TF_FUSED_H_
#define TF_FUSED_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * x[i] = act(x[i] + bias[i % C] + (residual ? residual[i] : 0)),  act = ReLU if relu != 0 else identity.
 * x / residual: `n` floats, channel-innermost (NHWC storage of a channels_last tensor), in place on x.
 * C % 4 == 0 and 16-byte aligned pointers are required.
 */
int tf_bias_act_f32(float *x, const float *bias, const float *residual, int64_t n, int C, int relu,
                    void *stream);

/*
 * out[r, :] = LayerNorm(x[r, :] + res[r, :]) * gamma + beta   for r in [0, rows), row length C
 * (C % 4 == 0, C <= 4096).  `res` may be NULL (plain LayerNorm).  out may alias x.
 * Statistics in fp32 with the biased variance and eps inside the square root, as torch.nn.LayerNorm.
 */
int tf_add_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta,
                         float *out, int64_t rows, int C, float eps, void *stream);

/*
 * Iterative box refinement of the decoder in one pass (reference: models/deformable_transformer.py:331-343 +
 * util/misc.py inverse_sigmoid): out[r, c] = sigmoid(delta[r, c] + (c < ref_dim ? logit_eps(ref[r, c]) : 0)),
 * delta / out [rows, 4], ref [rows, ref_dim], ref_dim in {2, 4}, logit_eps(x) = log(max(x', eps) / max(1 - x', eps)) with
 * x' = clamp(x, 0, 1).  out may alias delta.
 */
int tf_box_refine_f32(const float *delta, const float *ref, float *out, int64_t rows, int ref_dim, float eps, void *stream);

/*
 * The per-frame post-processing of the tracker in one pass (reference: models/deformable_detr.py DeformablePostProcess.forward
 * -- sigmoid, best class and its score, boxes cxcywh -> xyxy scaled to the image -- followed by models/tracker.py:300-303
 * clip_boxes_to_image and the stacking of what the association reads).  logits [Q, C], boxes [Q, 4] (cx, cy, w, h in [0, 1]),
 * out [Q, 6] = (x0, y0, x1, y1, score, label as float):
 *   score = max_c sigmoid(logits[q, c]), label = the first c that attains it;
 *   x0 = (cx - 0.5 w) img_w, y0 = (cy - 0.5 h) img_h, x1 = (cx + 0.5 w) img_w, y1 = (cy + 0.5 h) img_h, every operation rounded
 *   on its own (no fused multiply-add: the arithmetic of the separate ATen kernels), then, if clip != 0, clamped to [0, img_w] /
 *   [0, img_h].  Replaces ~17 element-wise launches on [Q, 4] tensors per frame.
 */
int tf_postprocess_pack_f32(const float *logits, const float *boxes, float *out, int64_t Q, int C, float img_h, float img_w,
                            int clip, void *stream);

/*
 * GroupNorm of a channels-innermost activation: x [N, HW, C] (row n starts at x + n * x_image_stride floats; the storage of
 * a channels_last NCHW tensor or a token-major projection output), G groups of C / G consecutive channels, statistics
 * per (image, group) with the biased variance and eps inside the square root (torch.nn.GroupNorm); out may alias x.
 * workspace: 2 * N * G doubles (zeroed by the call).  C <= 1024, C % 4 == 0, 16-byte aligned pointers.
 * For the reference's `input_proj` (models/deformable_detr.py:73-90: Conv2d -> GroupNorm(32, hidden_dim)).
 */
int tf_groupnorm_nhwc_f32(const float *x, const float *gamma, const float *beta, float *out, double *workspace, int N,
                          int HW, int C, int G, float eps, int64_t x_image_stride, int64_t out_image_stride, void *stream);
/* The statistics pass alone: workspace[n][g] = (sum, sum of squares) of x's group g of image n, as doubles -- for a consumer that applies
 * the normalisation in its own fetch (tf_conv3x3_merge_packed_f32). */
int tf_groupnorm_stats_nhwc_f32(const float *x, double *workspace, int N, int HW, int C, int G, int64_t x_image_stride, void *stream);
/* The same followed by ReLU in the same pass: `F.relu(gn(conv(x)))` of the mask head (reference: models/detr_segmentation.py:142-156). */
int tf_groupnorm_relu_nhwc_f32(const float *x, const float *gamma, const float *beta, float *out, double *workspace, int N,
                               int HW, int C, int G, float eps, int64_t x_image_stride, int64_t out_image_stride, void *stream);

/*
 * The FPN merge of the mask head in one pass (reference: models/detr_segmentation.py:142-156 `_expand(adapter(fpn), Q) +
 * F.interpolate(x, size=fpn.shape[-2:], mode="nearest")`): out[n, y, x, c] = low[n, ys, xs, c] + fpn[n / q_per_image, y, x, c] with
 * ys = min(floorf(y * (float)h / H), h - 1), xs likewise (torch's legacy "nearest" index).  low [N, h, w, C], fpn [N / q_per_image, H,
 * W, C], out [N, H, W, C] -- the storage of channels_last NCHW tensors; C % 4 == 0, 16-byte aligned.  The reference's two passes
 * write and re-read the up-sampled tensor (1.1 GB per 128 queries at the finest level of an 800 x 1333 frame).
 */
int tf_upsample_add_nhwc_f32(const float *low, const float *fpn, float *out, int N, int q_per_image, int h, int w, int H, int W, int C,
                             void *stream);

/*
 * A 3 x 3 / padding 1 convolution (split product, halo form) whose input is the mask head's FPN merge, computed in the convolution's
 * fetch and never written (reference: models/detr_segmentation.py:142-156 `x = adapter(fpn) + F.interpolate(x, size=fpn.shape[-2:])`,
 * `x = lay(x)`): input pixel (y, x), channel c of image n = act(low[n, ys, xs, c]) + fpn[n / q_per_image, y, x, c], (ys, xs) torch's
 * legacy nearest index.  act = identity when gn_workspace is NULL; else relu(GroupNorm(low)) from RAW statistics (the 2 * nimg * groups
 * doubles tf_groupnorm_*'s statistics pass leaves: sum | sum of squares per (image, group) over lh * lw * cin / groups values) -- the
 * previous layer's GroupNorm + ReLU folded in as well.  low [nimg, lh, lw, cin], fpn [nimg / q_per_image, H, W, cin], y [nimg, H, W,
 * cout]; cin % 32 == 0, cin <= 320; w_packed: tf_linear_pack_weight_f32 of the [cout, 9 * cin] tap-major weight.
 */
int tf_conv3x3_merge_packed_f32(const float *low, const float *fpn, const double *gn_workspace, const float *gamma, const float *beta,
                                int groups, float eps, const void *w_packed, const float *bias, float *y, int nimg, int q_per_image, int lh,
                                int lw, int H, int W, int cin, int cout, int relu, int terms, void *stream);

/*
 * The end of the mask head in one pass over the last hidden activation (reference: models/detr_segmentation.py:157-160
 * `out_lay(F.relu(gn5(x)))`, a 3 x 3 convolution to ONE channel, padding 1): statistics of x [N, H, W, C] per (image, group) as
 * tf_groupnorm_nhwc_f32 computes them (workspace: 2 * N * G doubles, zeroed by the call), then out[n, y, x] = bias + sum over the
 * 3 x 3 taps and C channels of relu(gn(x))[n, y + dy, x + dx, c] * weight[(dy, dx), c] in fp32 (zero outside the image); the
 * normalised activation is never written.  weight [9, C] tap-major (the [1, C, 3, 3] weight permuted to [3, 3, C]); C in {16, 32},
 * G divides C; 16-byte aligned pointers.
 */
int tf_groupnorm_relu_conv3x3_c1_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *weight, float bias,
                                          float *out, double *workspace, int N, int H, int W, int C, int G, float eps, void *stream);

/*
 * The tracker's mask post-processing in one pass (reference: models/detr_segmentation.py PostProcessSegm.forward -- bilinear resize
 * of the mask logits to the padded batch size, sigmoid, crop of the padding, nearest resize to the original image size -- followed by
 * models/tracker.py:521-532: a pixel belongs to the track with the largest probability there if that probability exceeds the
 * threshold).  logits [n, h, w] (mask-head outputs of n queries), order [n_tracks]: the row of `logits` that is track i's mask (-1: the
 * track has none); label [out_h, out_w] int16: the owning track's index or -1.  Per output pixel: its nearest source pixel in the
 * (img_h, img_w) crop of the (pad_h, pad_w) grid (torch's legacy "nearest" index), there the bilinear sample of every track's logits
 * (align_corners = False, torch's arithmetic operation by operation), its sigmoid; ties go to the first track, as torch.max.
 * The reference's chain writes and re-reads n full-size fp32 maps (~0.9 GB per frame at 100 tracks and 1080 x 1920).
 */
int tf_mask_label_map_f32(const float *logits, const int *order, int16_t *label, int n_tracks, int h, int w, int pad_h, int pad_w, int img_h,
                          int img_w, int out_h, int out_w, float threshold, void *stream);

/*
 * THE SPLIT PRODUCT (every matrix-core kernel below; trackformer_amd/csrc/split_product.h).  fp32 operands are cut into 16-bit
 * pieces (round to nearest even, residuals exact in fp32) and a product x . w is formed from v_mfma_f32_32x32x16_{bf16,f16}
 * terms with fp32 accumulation, smallest terms first.  The reference computes these layers in fp32 (nn.Linear / Conv2d;
 * models/ops/src/cuda/ms_deform_attn_cuda.cu:69 dispatches on fp32 tensors, no autocast anywhere).
 *   terms = 6   bf16 pieces hi, mid, lo of both operands:  x_lo.w_hi + x_hi.w_lo + x_mid.w_mid + x_mid.w_hi + x_hi.w_mid + x_hi.w_hi
 *               The three pieces carry all 24 significand bits; the dropped terms are below 2^-24 |x||w|: fp32 arithmetic in
 *               another summation order.
 *   terms = 16  fp16 pieces:  activation  xh = f16(x / 16),  xl = f16((x / 16 - xh) 2^11)
 *                             weight      wh = f16(w t_n),   wl = f16(w t_n - wh)
 *               three terms  xl.(wh 2^-11) + xh.wl + xh.wh,  result times r_n = 16 / t_n  (wh 2^-11 is formed in registers from the
 *               wh fragment: exact).  Two fp16 pieces carry 22 significand bits + the
 *               sign of the lower one (error of an operand <= 2^-23 of it); the only dropped product (lo.lo) is below 2^-22 |x||w|.
 *               t_n: the power of two that puts the largest |w| of output channel n into [2^13, 2^14) -- with the lower
 *               activation piece stored times 2^11 nothing falls into fp16's subnormals (|x| < 1.0e6; beyond that the row becomes
 *               NaN instead of saturating).  Half the matrix work of terms = 6 at fp32-class accuracy (against float64 on random
 *               operands: 3.7e-8 of sum |x||w| for the representation; the fp32 rounding of the sum itself is 2.4e-7).
 *   (terms = 3, two bf16 pieces and three terms with products good to 2^-16 -- the "fast mode" of rounds 2-4 -- was REMOVED in
 *   round 5: the fp16 product runs at the same speed with fp32-class accuracy, and three bf16 terms lost the reference's track ids
 *   at frame 14 of the 64-frame fixture, profiles/r04_id_parity_64.txt.  terms = 3 / (w_hi, w_mid) alone is TF_MSDA_ERR_BAD_DIMS.)
 * Entry points that take the weight as separate piece tensors (16-bit [N, K] each, made once by the caller) select by which are
 * given:  (w_hi, w_mid, w_lo)                 bf16 hi, mid, lo               -> six terms
 *         (w_hi, w_mid, NULL, w_scale)        fp16 wh, wl + r_n [N] fp32     -> the fp16 product
 * Entry points that take a PACKED weight select by `terms` (6 or 16), which must be the value the weight was packed with
 * (tf_linear_pack_weight_f32 computes t_n / r_n itself).  x is split inside the kernels.
 */

/*
 * y[M, N] = x[M, K] . w[N, K]^T + bias[N] (bias may be NULL), ReLU if relu != 0; fp32 in and out, row-major.
 * K % 32 == 0, 16-byte aligned x / w_hi / w_mid / w_lo; w_lo, w_scale: see THE SPLIT PRODUCT.  (trackformer_amd/csrc/linear_split.hip)
 */
int tf_linear_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                        float *y, int64_t M, int K, int N, int relu, void *stream);

/*
 * The same product with a residual in the epilogue: y = act(x . w^T + bias + residual), residual [M, N] fp32 (may alias y).
 * For the 1 x 1 convolutions that close a ResNet bottleneck (conv3 -> FrozenBatchNorm2d -> `out += identity` -> ReLU;
 * reference: models/backbone.py:45-55 + torchvision's Bottleneck.forward): on channels_last activations a stride-1 1 x 1
 * convolution IS this GEMM with M = N_img * H * W rows, the BN scale folded into w and its shift as bias.
 */
int tf_linear_split_res_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                            const float *residual, float *y, int64_t M, int K, int N, int relu, void *stream);

/*
 * 3 x 3 convolution (padding 1, stride 1 or 2, no groups / dilation) of a channels_last activation as the same split
 * product (an implicit GEMM over the output pixels, K = 9 * Cin): y[n, ho, wo, :] = act(sum_taps x[n, hi, wi, :] . w[:, tap, :]^T
 * + bias).  x [N, Hin, Win, Cin] and y [N, Hout, Wout, Cout] are NHWC (the storage of channels_last NCHW tensors); the weight
 * arrives as bf16 pieces of the [Cout, 3, 3, Cin] tensor (the storage of a channels_last OIHW weight), Cin % 32 == 0.
 * For torchvision's Bottleneck.conv2 + FrozenBatchNorm2d + ReLU (BN scale folded into w, shift as bias).  Input, output and
 * weight pieces below 3 GiB each (buffer-resource offsets), else TF_MSDA_ERR_BAD_DIMS.
 */
int tf_conv3x3_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                         float *y, int nimg, int hin, int win, int cin, int cout, int stride, int relu, void *stream);
/* The same convolution with the K loop (9 Cin) cut into `ksplit` pieces that run as separate workgroups -- for few output
 * pixels under a long K (the extra pyramid level of deformable_detr.py:55-79: 2048 -> 256 at 13 x 21; layer4's 3 x 3
 * convolutions).  The pieces write partial sums to `workspace` (ksplit * N*Hout*Wout * cout floats, 16-byte aligned), a second
 * launch adds them in a fixed order (deterministic, unlike atomics) and applies bias / ReLU.  ksplit in 1..64 (1: no workspace
 * needed, identical to tf_conv3x3_split_f32); cout % 4 == 0 and 16-byte aligned y / bias when ksplit > 1. */
int tf_conv3x3_splitk_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                          float *y, float *workspace, int ksplit, int nimg, int hin, int win, int cin, int cout, int stride, int relu,
                          void *stream);
/* The same kernel for a strided 1 x 1 convolution without padding (w [Cout, Cin]): the projections of the identity branch
 * (torchvision Bottleneck.downsample, stride 2) -- the rows of the GEMM are every stride-th pixel. */
int tf_conv1x1_strided_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                 const float *bias, float *y, int nimg, int hin, int win, int cin, int cout, int stride, int relu,
                                 void *stream);
/* The 1 x 1 convolution (stride 1 or 2, w [Cout, Cin]) with the K loop (Cin) cut into `ksplit` pieces, as tf_conv3x3_splitk_f32:
 * ResNet-50's reducing 1 x 1 convolutions of layer3 / layer4 (torchvision Bottleneck.conv1: 1024 -> 256 at 50 x 84, 2048 -> 512 at
 * 25 x 42 for an 800 x 1333 frame) are 132 / 68 workgroups of 32 / 64 K-slices -- fewer than the chip has CUs. */
int tf_conv1x1_splitk_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                          float *y, float *workspace, int ksplit, int nimg, int hin, int win, int cin, int cout, int stride, int relu,
                          void *stream);

/*
 * The backbone's first convolution (7 x 7, stride 2, padding 3, 3 -> 64 channels; reference: models/backbone.py:93-104 ->
 * torchvision resnet50.conv1) as a split product on the matrix cores (trackformer_amd/csrc/stem_conv.hip).
 *   x         [N, 3, H, W] fp32, planar (NCHW)
 *   w_packed  tf_linear_pack_weight_f32(K = 176, N = 64, terms) of the [64, 176] matrix w2[o][(c * 7 + ky) * 8 + kx] = w[o][c][ky][kx]
 *             (kx = 7 and k >= 168: zeros) -- with a following FrozenBatchNorm2d's scale folded in by the caller
 *   bias      [64] or NULL (the FrozenBatchNorm2d shift), relu != 0: ReLU
 *   y         [N, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 64] fp32, channels_last
 */
int tf_stem_conv7x7_f32(const float *x, const void *w_packed, const float *bias, float *y, int N, int H, int W, int relu,
                        int terms, void *stream);

/*
 * out[n, oy, ox, c] = max over the 3 x 3 window (stride 2, padding 1) of relu(x[n, iy, ix, c] + bias[c]) on channels_last
 * activations: FrozenBatchNorm2d shift + ReLU + MaxPool2d(3, 2, 1) after the backbone's first convolution (reference:
 * models/backbone.py:45-55, torchvision ResNet.relu / .maxpool) in one pass.  x [N, H, W, C], out [N, (H - 1) / 2 + 1,
 * (W - 1) / 2 + 1, C], C % 4 == 0, 16-byte aligned pointers.  Bit-identical to the separate passes (the shift and the ReLU
 * are monotone per channel).
 */
int tf_bias_relu_maxpool_f32(const float *x, const float *bias, float *out, int N, int H, int W, int C, void *stream);

/* y[M, N] = (x + x2)[M, K] . w^T + bias: tf_linear_split_f32 with an element-wise add in front, done as the activation tile is
 * staged -- `with_pos_embed(src, pos)` + a projection (models/deformable_transformer.py:279-283, ms_deform_attn.py:67-72)
 * without a separate pass over the tokens.  Bit-identical to adding first. */
int tf_linear_split_add_f32(const float *x, const float *x2, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                            const float *bias, float *y, int64_t M, int K, int N, void *stream);

/*
 * The same product with the weight in PACKED form (trackformer_amd/csrc/linear_stream.hip): the weight is split into
 * its bf16 pieces once and stored in matrix-core fragment order, so that the GEMM streams it from L2 into registers and
 * only the activations pass through LDS.  Results are bit-identical to tf_linear_split_f32 with the same number of terms.
 *   tf_linear_packed_bytes(K, N, terms)          size of the packed buffer (N padded to a multiple of 256; terms = 16: + a float
 *                                                per padded output channel behind the fragments), or -1; K % 16 == 0
 *   tf_linear_pack_weight_f32(w, packed, ...)    w [N, K] fp32 row-major -> packed (16-byte aligned pointers); one small kernel
 *   tf_linear_packed_f32                         y[M, N] = act(x[M, K] . w^T + bias + residual); K % 64 == 0, 16-byte aligned x;
 *                                                bias / residual [M, N] may be NULL, residual may alias y; y below 3 GiB
 * terms: 6 or 16 (see THE SPLIT PRODUCT above); a weight packed for 6 holds three pieces per fragment, for 16 two.
 * The residual form is the closing 1 x 1 convolution of a ResNet bottleneck (conv3 -> FrozenBatchNorm2d -> `out += identity` ->
 * ReLU; reference: models/backbone.py:45-55 + torchvision's Bottleneck.forward) on channels_last activations.
 */
int64_t tf_linear_packed_bytes(int K, int N, int terms);
int tf_linear_pack_weight_f32(const float *w, void *packed, int K, int N, int terms, void *stream);
int tf_linear_packed_f32(const float *x, const void *w_packed, const float *bias, const float *residual, float *y, int64_t M, int K,
                         int N, int relu, int terms, void *stream);

/*
 * Convolution of a channels_last activation through the same kernel (an implicit GEMM over the output pixels; the weight
 * fragments streamed from L2, only the shifted input pixels pass LDS): ks = 3 (padding 1) or 1 (no padding), stride 1 or 2.
 *   x [nimg, hin, win, cin] NHWC fp32, below 3 GiB;  y [nimg, hout, wout, cout] NHWC, below 3 GiB;  cin % 64 == 0
 *   w_packed   tf_linear_pack_weight_f32(K = ks * ks * cin, N = cout, terms) of the [cout, ks, ks, cin] weight (the storage of a
 *              channels_last OIHW tensor: K is tap-major) -- a following FrozenBatchNorm2d's scale folded in by the caller
 *   bias [cout] / residual [nimg, hout, wout, cout]: may be NULL
 *   ksplit     1..64 pieces of the K loop run as separate workgroups (few output pixels under a long K: layer3 / layer4, the
 *              extra pyramid level of deformable_detr.py:55-79); the pieces write partial sums to `workspace` (ksplit * M * cout
 *              floats), a second launch adds them in a fixed order (deterministic) and applies bias / residual / ReLU.
 *              ksplit > 1: cout % 4 == 0, 16-byte aligned y / bias / residual / workspace.
 * Same products in the same order as tf_conv3x3_split_f32 / tf_conv1x1_strided_split_f32: bit-identical for ksplit == 1.
 * (reference: torchvision Bottleneck.conv1 / conv2 / downsample under models/backbone.py:93-104)
 */
int tf_conv_packed_f32(const float *x, const void *w_packed, const float *bias, const float *residual, float *y, float *workspace,
                       int ksplit, int nimg, int hin, int win, int cin, int cout, int ks, int stride, int relu, int terms,
                       void *stream);

/*
 * The feed-forward block of a transformer layer in one launch (trackformer_amd/csrc/ffn_fused.hip):
 *
 *     y[M, d_model] = [LayerNorm]( residual + relu(x . w1^T + b1) . w2^T + b2 )
 *
 * (reference: models/deformable_transformer.py:282-297 forward_ffn + norm2 of the encoder layer, :371-379 of the decoder
 * layer; inference: the dropouts are identities).  The d_ffn-wide intermediate stays on the CU.  Same split
 * product as tf_linear_packed_f32 (same `terms`) for both GEMMs: without the LayerNorm the result is bit-identical to
 * tf_linear_packed_f32(relu) -> tf_linear_packed_f32 -> + residual.
 *   w1_packed, w2_packed   tf_linear_pack_weight_f32 of linear1.weight [d_ffn, d_model] (K = d_model, N = d_ffn) and of
 *                          linear2.weight [d_model, d_ffn] (K = d_ffn, N = d_model)
 *   b1 [d_ffn], b2 [d_model], residual [M, d_model]: each may be NULL;  ln_weight / ln_bias [d_model]: both or neither
 *   (neither: no LayerNorm);  y must not alias x or residual.
 * d_model == 256 or 288 (the reference's hidden sizes), d_ffn a multiple of 16 and >= 128, every pointer 16-byte aligned;
 * anything else: TF_MSDA_ERR_BAD_DIMS.
 */
/*
 * y[M, D] = [LayerNorm]( residual + x[M, D] . w^T + bias ), D = 256 or 288, in one launch: the attention's output projection with the
 * layer's residual add and norm1 (reference: models/ops/modules/ms_deform_attn.py:87 output_proj +
 * models/deformable_transformer.py:285-292).  Same split product as tf_linear_packed_f32 (bit-identical without the
 * LayerNorm); w_packed: tf_linear_pack_weight_f32 of the [D, D] weight.  K == N == D, pointers 16-byte aligned,
 * bias / residual may be NULL, ln_weight / ln_bias both or neither, y must not alias x or residual.
 */
int tf_linear_res_ln_f32(const float *x, const void *w_packed, const float *bias, const float *residual, const float *ln_weight,
                         const float *ln_bias, float ln_eps, float *y, int64_t M, int K, int N, int terms, void *stream);

int tf_ffn_fused_f32(const float *x, const void *w1_packed, const float *b1, const void *w2_packed, const float *b2,
                     const float *residual, const float *ln_weight, const float *ln_bias, float ln_eps, float *y, int64_t M,
                     int d_model, int d_ffn, int terms, void *stream);

/*
 * out[n, l, h, :] = sum_j softmax_j(scale * q[n, l, h, :] . k[n, j, h, :]) v[n, j, h, :]      (fp32)
 * Element (n, l, h, c) of q / k / v / out lives at base + (n * L + l) * ld + h * D + c with ld in floats (so q and
 * k may be the two halves of one projection output).  key_mask: [N, Lk] bytes, non-zero = the key is ignored
 * (nn.MultiheadAttention's key_padding_mask), or NULL.  D in {16, 32, 36, 64}, ld % 4 == 0, 16-byte aligned
 * pointers, Lk up to ~2400 (the 16 x Lk score tile lives in LDS).
 */
int tf_mha_core_f32(const float *q, const float *k, const float *v, float *out, const unsigned char *key_mask,
                    int N, int Lq, int Lk, int H, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                    void *stream);

/*
 * Greedy non-maximum suppression on HOST boxes (all pointers host; synchronous): torchvision.ops.nms semantics, which the
 * reference's tracker calls twice per frame (tracker.py:399,495).  boxes [n, 4] xyxy, scores [n]; boxes are visited in
 * stable descending-score order; keep [n] receives the indices of the kept boxes in that order, *n_keep their number.
 * IoU arithmetic: fp32, inter / (area_i + area_j - inter), as box_ops.box_iou.
 */
int tf_nms_host_f32(const float *boxes, const float *scores, int n, float iou_threshold, int64_t *keep, int *n_keep);

#ifdef __cplusplus
}
#endif
#endif /* TF_FUSED_H_ */
