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

#ifdef __cplusplus
}
#endif
#endif /* TF_FUSED_H_ */
