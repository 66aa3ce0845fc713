/*
 * include/tf_msda.h -- C ABI of libtf_msda.so, the MI355X (gfx950) implementation of TrackFormer's
 * multi-scale deformable attention operator.
 *
 * This is the drop-in boundary for the reference's native extension `MultiScaleDeformableAttention`:
 *
 *   reference interface (paths under /root/reference/src/trackformer/models/ops/)      replaced by
 *   ---------------------------------------------------------------------------------  -----------------------
 *   ms_deform_attn_forward(value, spatial_shapes, sampling_loc, attn_weight, step)      tf_msda_forward_{f32,f64}
 *     src/vision.cpp:5, src/ms_deform_attn.h:10-28, src/cuda/ms_deform_attn_cuda.cu:19-86
 *   ms_deform_attn_backward(value, spatial_shapes, sampling_loc, attn_weight,           tf_msda_backward_{f32,f64}
 *                           grad_output, step)
 *     src/vision.cpp:6, src/ms_deform_attn.h:30-49, src/cuda/ms_deform_attn_cuda.cu:89-168
 *
 * The binding that exposes these under the reference's Python names
 * (`MultiScaleDeformableAttention.ms_deform_attn_forward/backward`) lives in
 * trackformer_amd/dropin/MultiScaleDeformableAttention.py; see INTEGRATION.md.
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / ATen types.  All data pointers are DEVICE pointers
 *     on the current HIP device unless marked "host".  The caller owns every buffer.
 *   - Tensors are dense row-major:
 *       value        [N, S, M, D]          S = sum_l H_l*W_l; level l occupies rows start_l .. start_l+H_l*W_l
 *       shapes       [L, 2] int64 (H_l, W_l)
 *       loc          [N, Lq, M, L, P, 2]   (x, y) normalised to [0,1] over the level; pixel = loc*size - 0.5,
 *                                          zero padding outside, in range iff -1 < pixel < size
 *       attn         [N, Lq, M, L, P]
 *       out/grad_out [N, Lq, M*D]
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and the
 *     call returns without synchronising.  HIP-graph capturable.  Re-entrant: a call's RESULT depends on its arguments
 *     only.  What IS process-wide: (a) the kernel-selection knobs below (tf_msda_set_tiled / tf_msda_set_option and the
 *     environment variables they mirror) -- performance only, results identical up to fp32 summation order; they exist
 *     for A/B measurements, a deployment leaves them alone; (b) the per-thread last-HIP-error slot; (c) per-(function,
 *     device) one-time attribute calls (dynamic LDS above 64 KB).
 *   - Return value: TF_MSDA_OK (0) or a negative tf_msda_status.  Never throws.
 *   - im2col_step of the reference API only chunks the batch (cu:44-66) and does not change results;
 *     this ABI has no such parameter.
 */
#ifndef TF_MSDA_H_
#define TF_MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_MSDA_ABI_VERSION 3
#define TF_MSDA_MAX_LEVELS 16

typedef enum tf_msda_status {
    TF_MSDA_OK = 0,
    TF_MSDA_ERR_NULL_POINTER = -1,  /* a required pointer was NULL */
    TF_MSDA_ERR_BAD_DIMS = -2,      /* a dimension <= 0, L > TF_MSDA_MAX_LEVELS, or sizes overflow */
    TF_MSDA_ERR_SHAPE_SUM = -3,     /* sum_l H_l*W_l != S (host-shape entry points only) */
    TF_MSDA_ERR_LAUNCH = -4,        /* HIP reported an error enqueueing work (see tf_msda_last_hip_error) */
    TF_MSDA_ERR_NO_DEVICE = -5      /* no HIP device available */
} tf_msda_status;

/* ABI version of the loaded library (== TF_MSDA_ABI_VERSION it was built with). */
int tf_msda_abi_version(void);

/* Human-readable text for a tf_msda_status. Never NULL. */
const char *tf_msda_strerror(int status);

/* hipError_t (as int) of the most recent failing HIP call on this thread, 0 if none. */
int tf_msda_last_hip_error(void);

/*
 * Name of the device kernel the most recent forward / backward call of THIS thread enqueued (e.g.
 * "msda_fwd_f32_pquad2<fused,4w,2p>"), "" before the first call.  The string is static.  Measurement aid: bench.py labels
 * its roofline with what the library dispatched instead of inferring it from the options.
 */
const char *tf_msda_last_kernel(void);

/*
 * Kernel selection knob (process-wide, performance only -- results are identical up to fp32 summation
 * order) for encoder-shaped forward calls (Lq == S, fp32, D == 32 or 36, P == 4, L <= 4, host shapes):
 * 2 = the LDS-window kernels (msda_fwd_f32_pquad: persistent workgroups, 4 lanes per pair; msda_fwd_f32_quad where it
 * declines; the default), 0 = msda_fwd_f32_direct (row gathers by buffer loads, what every other shape uses),
 * -1 restores the default (environment variable TF_MSDA_TILED, 2 when unset).  Returns the previous setting.
 */
int tf_msda_set_tiled(int mode);

/*
 * Generic form of the knob above (process-wide, performance only).  Sets option `name` to `value` and
 * returns the previous value, or INT_MIN for an unknown name.  Names:
 *   "tiled"         0 / 2 / -1 as tf_msda_set_tiled
 *   "pquad"         1 / 0: the persistent encoder kernel on / off (off: msda_fwd_f32_quad)
 *   "pquad_npass" "pquad_lds_kb" "pquad_wg_per_cu" "pquad_wide" "pquad_prefetch" "pquad_skew" "pquad_halo_y"
 *   "pquad_halo_x" "pquad_tile_h" "pquad_tile_w"    its tile plan (TF_MSDA_PQUAD="npass=2,lds=52,wgs=3,...")
 *   "quad_ta_mask"  bit l set: level l is gathered by buffer loads instead of an LDS window (0, 8 or 12)
 *   "quad_waves"    wavefronts per workgroup (4 or 8);  "quad_npass"  passes of 16 pairs per wave (1..3)
 *   "quad_lds_kb"   LDS per workgroup (decides the workgroups per CU and the window capacity)
 *   "quad_halo_y" / "quad_halo_x"   clamp of the data-adaptive windows around the tile footprint
 *   "quad_tile_h" / "quad_tile_w"   tile size in level-0 pixels (0 = search);  "quad_split"  staging rounds
 *   "direct9"       1 / 0: msda_fwd_f32_direct9 for D == 36 decoder calls (off: msda_fwd_f32_buf)
 *   "ffn_ti" "linln_ti" "linear_stream_ti"   row tiles per block of tf_ffn_fused_f32 / tf_linear_res_ln_f32 /
 *                   tf_linear_packed_f32 (include/tf_fused.h; 0 = per shape)
 * Knobs of experiments that were measured and removed (linear_variant, linear_bufstore, linear_deep, linear_astat,
 * conv3_bufload, bwd_sorted2, tiled = 1) are unknown names now.
 */
int tf_msda_set_option(const char *name, int value);

/*
 * Debug aid of tools/msda_bench --trace (not part of the operator contract): while `device_buffer` is not
 * NULL, every workgroup of msda_fwd_f32_quad writes 16 uint64 phase timestamps (s_memrealtime, 100 MHz) to
 * device_buffer[16 * blockIdx + i].  The buffer must hold 16 * grid entries; pass NULL to switch it off.
 */
void tf_msda_debug_trace_buffer(void *device_buffer);

/*
 * Forward.  out[N,Lq,M*D] = sum_{l,p} attn * bilinear(value_l, loc)      (Appendix A of SURVEY.md)
 *
 * shapes_hw_host : HOST pointer to L*2 int64 (H_l, W_l).  Passed by value to the kernel; nothing is
 *                  read from it after the call returns.
 * replaces ms_deform_attn_cuda_forward (cu:19-86) + ms_deformable_im2col_gpu_kernel (cuh:165-237).
 */
int tf_msda_forward_f32(const float *value, const int64_t *shapes_hw_host, const float *loc,
                        const float *attn, float *out, int N, int S, int M, int D, int L, int Lq,
                        int P, void *stream);
int tf_msda_forward_f64(const double *value, const int64_t *shapes_hw_host, const double *loc,
                        const double *attn, double *out, int N, int S, int M, int D, int L, int Lq,
                        int P, void *stream);

/*
 * Same, but the level shapes are read from DEVICE memory inside the kernel (what the reference does,
 * cuh:194-196) -- for callers that only hold the reference's device-resident `spatial_shapes` tensor
 * and must not synchronise.  sum_l H_l*W_l == S is the caller's responsibility (not checkable without a
 * device->host copy).
 */
int tf_msda_forward_f32_dshapes(const float *value, const int64_t *shapes_hw_dev, const float *loc,
                                const float *attn, float *out, int N, int S, int M, int D, int L,
                                int Lq, int P, void *stream);
int tf_msda_forward_f64_dshapes(const double *value, const int64_t *shapes_hw_dev,
                                const double *loc, const double *attn, double *out, int N, int S,
                                int M, int D, int L, int Lq, int P, void *stream);

/*
 * Forward with the operator's prologue fused in (fp32, inference): takes the RAW outputs of the query
 * projections and the reference points and performs softmax + sampling-location arithmetic inside the
 * kernel.  Replaces ops/modules/ms_deform_attn.py:69-86 (view / softmax / location arithmetic /
 * MSDeformAttnFunction.apply) in one launch.
 *   ref_points [N, Lq, L, ref_dim]   ref_dim 2: loc = ref + off / (H_l, W_l)   (x with H_l, y with W_l,
 *                                               exactly as ms_deform_attn.py:78-79 is written)
 *                                    ref_dim 4: loc = ref[:2] + off / P * ref[2:] * 0.5   (:81-82)
 *   qproj      [N*Lq, ld] floats; row r holds the query's M*L*P*2 raw offsets (order m, l, p, xy) starting
 *              at column off_col and its M*L*P attention logits (order m, l, p) at column logit_col
 *              (e.g. one GEMM with the two Linear weights concatenated: ld = 3*M*L*P, off_col = 0,
 *              logit_col = 2*M*L*P).  off_col and ld must be even.
 * Requires D % 4 == 0, P in {1,2,4,8}, 16-byte aligned value/out, tensors < 4 GiB.
 */
int tf_msda_forward_fused_f32(const float *value, const int64_t *shapes_hw_host,
                              const float *ref_points, int ref_dim, const float *qproj, int ld,
                              int off_col, int logit_col, float *out, int N, int S, int M, int D,
                              int L, int Lq, int P, void *stream);

/*
 * Backward.  Writes all three gradients; grad_value is zero-filled on `stream` by the library before
 * accumulation (reference: at::zeros_like, cu:119-121), grad_loc / grad_attn are fully overwritten.
 * grad_value accumulation uses hardware floating-point atomics, so its summation order (and therefore
 * its last bits) is not deterministic -- as in the reference (cuh:301).
 * replaces ms_deform_attn_cuda_backward (cu:89-168) + ms_deformable_col2im_gpu_kernel (cuh:239-306) +
 * ms_deformable_col2im_coord_gpu_kernel (cuh:308-378).
 */
int tf_msda_backward_f32(const float *value, const int64_t *shapes_hw_host, const float *loc,
                         const float *attn, const float *grad_out, float *grad_value,
                         float *grad_loc, float *grad_attn, int N, int S, int M, int D, int L,
                         int Lq, int P, void *stream);
int tf_msda_backward_f64(const double *value, const int64_t *shapes_hw_host, const double *loc,
                         const double *attn, const double *grad_out, double *grad_value,
                         double *grad_loc, double *grad_attn, int N, int S, int M, int D, int L,
                         int Lq, int P, void *stream);
int tf_msda_backward_f32_dshapes(const float *value, const int64_t *shapes_hw_dev, const float *loc,
                                 const float *attn, const float *grad_out, float *grad_value,
                                 float *grad_loc, float *grad_attn, int N, int S, int M, int D,
                                 int L, int Lq, int P, void *stream);
int tf_msda_backward_f64_dshapes(const double *value, const int64_t *shapes_hw_dev,
                                 const double *loc, const double *attn, const double *grad_out,
                                 double *grad_value, double *grad_loc, double *grad_attn, int N,
                                 int S, int M, int D, int L, int Lq, int P, void *stream);

/*
 * The operator for HOST tensors: every pointer is a host pointer, the call computes synchronously on the calling thread
 * plus worker threads (split by batch x head: no atomics, deterministic gradients) and returns when done.
 * The reference has no CPU implementation -- ms_deform_attn.h:27,48 raise "Not implemented on the CPU",
 * cpu/ms_deform_attn_cpu.cpp:17-40 are stubs -- SURVEY.md section 8(b) asks for a real one in place of the error.  Reached
 * only for tensors that already live in host memory; the device entry points above never fall back to it.
 * grad_value is zero-filled by the call, grad_loc / grad_attn are fully overwritten.
 */
int tf_msda_forward_host_f32(const float *value, const int64_t *shapes_hw, const float *loc, const float *attn,
                             float *out, int N, int S, int M, int D, int L, int Lq, int P);
int tf_msda_forward_host_f64(const double *value, const int64_t *shapes_hw, const double *loc, const double *attn,
                             double *out, int N, int S, int M, int D, int L, int Lq, int P);
int tf_msda_backward_host_f32(const float *value, const int64_t *shapes_hw, const float *loc, const float *attn,
                              const float *grad_out, float *grad_value, float *grad_loc, float *grad_attn, int N,
                              int S, int M, int D, int L, int Lq, int P);
int tf_msda_backward_host_f64(const double *value, const int64_t *shapes_hw, const double *loc, const double *attn,
                              const double *grad_out, double *grad_value, double *grad_loc, double *grad_attn, int N,
                              int S, int M, int D, int L, int Lq, int P);

#ifdef __cplusplus
}
#endif
#endif /* TF_MSDA_H_ */
