// trackformer_amd/csrc/fused_ops.hip -- fused element-wise kernels (include/tf_fused.h), gfx950.
//
// Both kernels are HBM-bound streaming passes: 16 bytes per lane per access, grid-stride, one read of
// every operand and one write of the result.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "msda_common.h"
#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// x += bias[channel] (+ residual) (ReLU), in place.  The residual / ReLU tests are template parameters and two grid strides
// are processed per iteration with all of their loads issued first: the straightforward loop (one element per iteration, a
// run-time residual pointer test) compiled to load x, load bias, s_waitcnt vmcnt(0), [branch] load residual, s_waitcnt
// vmcnt(0), store -- two sequential memory round trips per 16 bytes and lane (tools/isa_audit.py --stream); bit-identical
// results (round 3: the one-element kernel was removed after both were validated against each other on MI355X).
template <bool RES, bool RELU>
__global__ void __launch_bounds__(256)
bias_act_kernel(float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ residual,
                        long long n4, int C4)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // channel quad of element i: one 64-bit remainder up front, 32-bit increments with a wrap afterwards
    const int step = (int)(stride % C4);
    int q0 = (int)(first % C4);
    for (long long i = first; i < n4; i += 2 * stride) {
        const long long i1 = i + stride;
        const bool two = i1 < n4;
        const long long j1 = two ? i1 : i;   // a harmless reload when the second element does not exist
        int q1 = q0 + step;
        q1 = q1 >= C4 ? q1 - C4 : q1;
        const f32x4_t b0 = reinterpret_cast<const f32x4_t *>(bias)[q0];
        const f32x4_t b1 = reinterpret_cast<const f32x4_t *>(bias)[two ? q1 : q0];
        f32x4_t v0 = reinterpret_cast<const f32x4_t *>(x)[i];
        f32x4_t v1 = reinterpret_cast<const f32x4_t *>(x)[j1];
        f32x4_t r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if constexpr (RES) {
            r0 = reinterpret_cast<const f32x4_t *>(residual)[i];
            r1 = reinterpret_cast<const f32x4_t *>(residual)[j1];
        }
        q0 = q1 + step;
        q0 = q0 >= C4 ? q0 - C4 : q0;
        v0 += b0;
        v1 += b1;
        if constexpr (RES) {
            v0 += r0;
            v1 += r1;
        }
        if constexpr (RELU) {
            v0.x = v0.x < 0.f ? 0.f : v0.x; v0.y = v0.y < 0.f ? 0.f : v0.y; v0.z = v0.z < 0.f ? 0.f : v0.z; v0.w = v0.w < 0.f ? 0.f : v0.w;
            v1.x = v1.x < 0.f ? 0.f : v1.x; v1.y = v1.y < 0.f ? 0.f : v1.y; v1.z = v1.z < 0.f ? 0.f : v1.z; v1.w = v1.w < 0.f ? 0.f : v1.w;
        }
        tfm::stream_store(reinterpret_cast<f32x4_t *>(x) + i, v0);
        if (two) tfm::stream_store(reinterpret_cast<f32x4_t *>(x) + i1, v1);
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// One wavefront per row; lane j owns float4 chunks j, j+64, ... of the row (kept in registers).
template <int MAXCH>
__global__ void __launch_bounds__(256)
add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ res,
                     const float *__restrict__ gamma, const float *__restrict__ beta,
                     float *__restrict__ out, long long rows, int C, float eps)
{
    const int lane = threadIdx.x & 63;
    const int C4 = C >> 2;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const f32x4_t *xr = reinterpret_cast<const f32x4_t *>(x + r * C);
        const f32x4_t *rr = res ? reinterpret_cast<const f32x4_t *>(res + r * C) : nullptr;
        f32x4_t v[MAXCH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int j = lane + k * 64;
            v[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (j < C4) {
                v[k] = xr[j];
                if (rr) v[k] += rr[j];
                s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int j = lane + k * 64;
            if (j < C4) {
                const f32x4_t d = v[k] - mean;
                sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        f32x4_t *orow = reinterpret_cast<f32x4_t *>(out + r * C);
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int j = lane + k * 64;
            if (j < C4) {
                const f32x4_t g = reinterpret_cast<const f32x4_t *>(gamma)[j];
                const f32x4_t b = reinterpret_cast<const f32x4_t *>(beta)[j];
                tfm::stream_store(orow + j, (v[k] - mean) * rstd * g + b);
            }
        }
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- GroupNorm over channels-innermost activations: x [N, HW, C] (the storage of a channels_last NCHW tensor, or a
// token-major projection output), statistics per (image, group) over HW x (C / G) elements.  Two passes:
//   groupnorm_stats_kernel   partial sums per workgroup (fp32 over <= 64 rows per thread, then double), one double
//                            atomic per group and workgroup into ws[n][g][sum | sum of squares]
//   groupnorm_apply_kernel   y = (x - mean) * rstd * gamma + beta, mean / rstd from ws (double arithmetic, once per
//                            workgroup into LDS), 16 bytes per lane
// Replaces the ATen path behind the reference's `input_proj` (models/deformable_detr.py:73-90: Conv2d + GroupNorm(32, 256)),
// which for channels_last inputs is a layout copy + RowwiseMoments over only N * 32 rows + a parameter kernel + an
// element-wise kernel (0.2 ms per frame, profiles/r02_e2e_eager_per_frame.txt).
__global__ void __launch_bounds__(256) groupnorm_zero_kernel(double *__restrict__ ws, unsigned n)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) ws[i] = 0.0;
}

constexpr int kGnRowsPerBlock = 64;   // 261 workgroups for the largest level at 800 x 1333 (16 700 pixels)

__global__ void __launch_bounds__(256)
groupnorm_stats_kernel(const float *__restrict__ x, double *__restrict__ ws, int HW, int C, int G, long long x_image_stride)
{
    __shared__ double s_stat[2 * 256];   // [group][sum | sumsq], G <= 256
    const int C4 = C >> 2, cpg = C / G;
    const int n = blockIdx.y;
    const int r0 = blockIdx.x * kGnRowsPerBlock, r1 = min(HW, r0 + kGnRowsPerBlock);
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) s_stat[i] = 0.0;
    __syncthreads();
    const int nslots = 256 / C4;   // >= 1 (host: C <= 1024)
    const int q = threadIdx.x % C4, rslot = threadIdx.x / C4;
    if (rslot < nslots) {
        const f32x4_t *xp = reinterpret_cast<const f32x4_t *>(x + (long long)n * x_image_stride);
        f32x4_t sum = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
        int r = r0 + rslot;
        for (; r + 3 * nslots < r1; r += 4 * nslots) {   // four rows in flight per thread, accumulated in row order
            const f32x4_t v0 = xp[(long long)r * C4 + q], v1 = xp[(long long)(r + nslots) * C4 + q];
            const f32x4_t v2 = xp[(long long)(r + 2 * nslots) * C4 + q], v3 = xp[(long long)(r + 3 * nslots) * C4 + q];
            sum += v0;
            sq += v0 * v0;
            sum += v1;
            sq += v1 * v1;
            sum += v2;
            sq += v2 * v2;
            sum += v3;
            sq += v3 * v3;
        }
        for (; r < r1; r += nslots) {
            const f32x4_t v = xp[(long long)r * C4 + q];
            sum += v;
            sq += v * v;
        }
        // the four channels of a quad may straddle two groups (hidden 288: 9 channels per group)
        const int g0 = (q * 4) / cpg, g1 = (q * 4 + 1) / cpg, g2 = (q * 4 + 2) / cpg, g3 = (q * 4 + 3) / cpg;
        if (g0 == g3) {
            atomicAdd(&s_stat[2 * g0], (double)((sum.x + sum.y) + (sum.z + sum.w)));
            atomicAdd(&s_stat[2 * g0 + 1], (double)((sq.x + sq.y) + (sq.z + sq.w)));
        } else {
            atomicAdd(&s_stat[2 * g0], (double)sum.x); atomicAdd(&s_stat[2 * g0 + 1], (double)sq.x);
            atomicAdd(&s_stat[2 * g1], (double)sum.y); atomicAdd(&s_stat[2 * g1 + 1], (double)sq.y);
            atomicAdd(&s_stat[2 * g2], (double)sum.z); atomicAdd(&s_stat[2 * g2 + 1], (double)sq.z);
            atomicAdd(&s_stat[2 * g3], (double)sum.w); atomicAdd(&s_stat[2 * g3 + 1], (double)sq.w);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) atomicAdd(&ws[(long long)n * 2 * G + i], s_stat[i]);
}

template <bool RELU>
__global__ void __launch_bounds__(256)
groupnorm_apply_kernel(const float *__restrict__ x, const double *__restrict__ ws, const float *__restrict__ gamma,
                       const float *__restrict__ beta, float *__restrict__ out, int HW, int C, int G, float eps,
                       long long x_image_stride, long long out_image_stride)
{
    __shared__ float s_mean[256], s_rstd[256];
    const int C4 = C >> 2, cpg = C / G;
    const int n = blockIdx.y;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const double cnt = (double)HW * (double)cpg;
        const double mean = ws[(long long)n * 2 * G + 2 * g] / cnt;
        double var = ws[(long long)n * 2 * G + 2 * g + 1] / cnt - mean * mean;   // biased, as torch.nn.GroupNorm
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const f32x4_t *xp = reinterpret_cast<const f32x4_t *>(x + (long long)n * x_image_stride);
    f32x4_t *op = reinterpret_cast<f32x4_t *>(out + (long long)n * out_image_stride);
    const long long n4 = (long long)HW * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int q = (int)(i % C4);
        const int g0 = (q * 4) / cpg, g1 = (q * 4 + 1) / cpg, g2 = (q * 4 + 2) / cpg, g3 = (q * 4 + 3) / cpg;
        const f32x4_t v = xp[i];
        const f32x4_t ga = reinterpret_cast<const f32x4_t *>(gamma)[q], be = reinterpret_cast<const f32x4_t *>(beta)[q];
        f32x4_t y;
        y.x = (v.x - s_mean[g0]) * s_rstd[g0] * ga.x + be.x;
        y.y = (v.y - s_mean[g1]) * s_rstd[g1] * ga.y + be.y;
        y.z = (v.z - s_mean[g2]) * s_rstd[g2] * ga.z + be.z;
        y.w = (v.w - s_mean[g3]) * s_rstd[g3] * ga.w + be.w;
        if (RELU) {
            y.x = y.x < 0.f ? 0.f : y.x;
            y.y = y.y < 0.f ? 0.f : y.y;
            y.z = y.z < 0.f ? 0.f : y.z;
            y.w = y.w < 0.f ? 0.f : y.w;
        }
        tfm::stream_store(op + i, y);
    }
}

// ---- the FPN merge of the mask head (include/tf_fused.h tf_upsample_add_nhwc_f32): one thread per (output pixel, 4 channels)
__global__ void __launch_bounds__(256)
upsample_add_nhwc_kernel(const float *__restrict__ low, const float *__restrict__ fpn, float *__restrict__ out, int q_per_image,
                         int h, int w, int H, int W, int C4, float sy, float sx, long long total)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C4);
        const long long p = i / C4;
        const int x = (int)(p % W);
        const long long q = p / W;
        const int y = (int)(q % H);
        const int n = (int)(q / H);
        const int ys = min((int)floorf((float)y * sy), h - 1), xs = min((int)floorf((float)x * sx), w - 1);
        const f32x4_t a = reinterpret_cast<const f32x4_t *>(low)[(((long long)n * h + ys) * w + xs) * C4 + c];
        const f32x4_t b = reinterpret_cast<const f32x4_t *>(fpn)[(((long long)(n / q_per_image) * H + y) * W + x) * C4 + c];
        reinterpret_cast<f32x4_t *>(out)[i] = a + b;
    }
}

// ---- GroupNorm + ReLU + 3 x 3 convolution to one channel (include/tf_fused.h tf_groupnorm_relu_conv3x3_c1_nhwc_f32): a workgroup
// owns 8 x 32 output pixels of one image; their 10 x 34 halo is normalised on its way into LDS (zeros outside the image: the
// convolution pads the NORMALISED activation), one thread per output pixel.  A pixel's channels sit 4 floats apart from the next
// pixel's (pitch C + 4): the 16 lanes of a ds_read_b128 cycle read neighbouring pixels, 20 (36) banks apart -- no two on a bank.
template <int C>
__global__ void __launch_bounds__(256)
gn_relu_conv3x3_c1_kernel(const float *__restrict__ x, const double *__restrict__ ws, const float *__restrict__ gamma,
                          const float *__restrict__ beta, const float *__restrict__ weight, float bias, float *__restrict__ out, int H,
                          int W, int G, float eps)
{
    constexpr int TY = 8, TX = 32, HY = TY + 2, HX = TX + 2, C4 = C / 4, PITCH = C + 4;
    __shared__ __attribute__((aligned(16))) float s_in[HY * HX * PITCH];
    __shared__ __attribute__((aligned(16))) float s_w[9 * C];
    __shared__ float s_mean[C], s_rstd[C];   // per CHANNEL (its group's statistics)
    const int n = blockIdx.z, y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const double cnt = (double)H * (double)W * (double)cpg;
        const double mean = ws[(long long)n * 2 * G + 2 * g] / cnt;
        double var = ws[(long long)n * 2 * G + 2 * g + 1] / cnt - mean * mean;   // biased, as torch.nn.GroupNorm
        if (var < 0.0) var = 0.0;
        s_mean[c] = (float)mean;
        s_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    }
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) s_w[i] = weight[i];
    __syncthreads();
    const f32x4_t *xp = reinterpret_cast<const f32x4_t *>(x + (long long)n * H * W * C);
    for (int i = threadIdx.x; i < HY * HX * C4; i += blockDim.x) {
        const int q = i % C4, pix = i / C4;
        const int py = y0 - 1 + pix / HX, px = x0 - 1 + pix % HX;
        f32x4_t y = {0.f, 0.f, 0.f, 0.f};
        if (py >= 0 && py < H && px >= 0 && px < W) {
            const f32x4_t v = xp[((long long)py * W + px) * C4 + q];
            const f32x4_t ga = reinterpret_cast<const f32x4_t *>(gamma)[q], be = reinterpret_cast<const f32x4_t *>(beta)[q];
            // the expression of groupnorm_apply_kernel, operation by operation
            y.x = (v.x - s_mean[4 * q]) * s_rstd[4 * q] * ga.x + be.x;
            y.y = (v.y - s_mean[4 * q + 1]) * s_rstd[4 * q + 1] * ga.y + be.y;
            y.z = (v.z - s_mean[4 * q + 2]) * s_rstd[4 * q + 2] * ga.z + be.z;
            y.w = (v.w - s_mean[4 * q + 3]) * s_rstd[4 * q + 3] * ga.w + be.w;
            y.x = y.x < 0.f ? 0.f : y.x;
            y.y = y.y < 0.f ? 0.f : y.y;
            y.z = y.z < 0.f ? 0.f : y.z;
            y.w = y.w < 0.f ? 0.f : y.w;
        }
        *reinterpret_cast<f32x4_t *>(&s_in[pix * PITCH + 4 * q]) = y;
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int oy = y0 + ty, ox = x0 + tx;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float *ip = &s_in[((ty + t / 3) * HX + tx + t % 3) * PITCH];
#pragma unroll
        for (int q = 0; q < C4; ++q)
            acc += *reinterpret_cast<const f32x4_t *>(ip + 4 * q) * *reinterpret_cast<const f32x4_t *>(&s_w[t * C + 4 * q]);
    }
    if (oy < H && ox < W) out[((long long)n * H + oy) * W + ox] = ((acc.x + acc.y) + (acc.z + acc.w)) + bias;
}

// ---- the tracker's mask post-processing (include/tf_fused.h tf_mask_label_map_f32): one thread per output pixel, a loop over the tracks
__global__ void __launch_bounds__(256)
mask_label_map_kernel(const float *__restrict__ logits, const int *__restrict__ order, short *__restrict__ label, int n_tracks, int h, int w,
                      int pad_h, int pad_w, int img_h, int img_w, int out_h, int out_w, float threshold)
{
#pragma clang fp contract(off)
    const int ox = blockIdx.x * 32 + (threadIdx.x & 31), oy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (ox >= out_w || oy >= out_h) return;
    // nearest: the source pixel in the cropped (img_h, img_w) grid (upsample_nearest2d: min(floorf(dst * scale), in - 1))
    const int py = min((int)floorf((float)oy * ((float)img_h / (float)out_h)), img_h - 1);
    const int px = min((int)floorf((float)ox * ((float)img_w / (float)out_w)), img_w - 1);
    // bilinear, align_corners = False (upsample_bilinear2d): source index = max(scale (dst + 0.5) - 0.5, 0)
    const float ry = (float)h / (float)pad_h, rx = (float)w / (float)pad_w;
    const float sy = fmaxf(ry * ((float)py + 0.5f) - 0.5f, 0.f), sx = fmaxf(rx * ((float)px + 0.5f) - 0.5f, 0.f);
    const int y1 = (int)sy, x1 = (int)sx;
    const int yp = (y1 < h - 1) ? 1 : 0, xp = (x1 < w - 1) ? 1 : 0;
    const float ly1 = sy - (float)y1, ly0 = 1.f - ly1, lx1 = sx - (float)x1, lx0 = 1.f - lx1;
    const int o00 = y1 * w + x1, o01 = o00 + xp, o10 = o00 + yp * w, o11 = o10 + xp;
    float best = -1.f;
    int owner = -1;
    for (int t = 0; t < n_tracks; ++t) {
        const int row = order[t];   // uniform
        if (row < 0) continue;
        const float *p = logits + (long long)row * h * w;
        const float v = ly0 * (lx0 * p[o00] + lx1 * p[o01]) + ly1 * (lx0 * p[o10] + lx1 * p[o11]);
        const float prob = 1.f / (1.f + expf(-v));
        if (prob > best) { best = prob; owner = t; }   // ties: the first track (torch.max)
    }
    label[(long long)oy * out_w + ox] = (short)((owner >= 0 && best > threshold) ? owner : -1);
}

// ---- iterative box refinement of the decoder (models/deformable_transformer.py:331-343 of the reference):
//   ref_dim 4: new = sigmoid(delta + inverse_sigmoid(ref))
//   ref_dim 2: new[:2] = sigmoid(delta[:2] + inverse_sigmoid(ref)), new[2:] = sigmoid(delta[2:])
// with inverse_sigmoid(x) = log(max(clamp(x, 0, 1), eps) / max(1 - clamp(x, 0, 1), eps)) (util/misc.py:inverse_sigmoid).
// One launch instead of the ~12 element-wise ATen launches per decoder layer over [queries, 4] tensors.
__global__ void __launch_bounds__(256)
box_refine_kernel(const float *__restrict__ delta, const float *__restrict__ ref, float *__restrict__ out, long long rows,
                  int ref_dim, float eps)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one component per thread
    if (i >= rows * 4) return;
    const long long r = i >> 2;
    const int c = (int)(i & 3);
    float v = delta[i];
    if (c < ref_dim) {
        float x = ref[r * ref_dim + c];
        x = fminf(fmaxf(x, 0.f), 1.f);
        const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
        v += logf(x1 / x2);
    }
    out[i] = 1.f / (1.f + expf(-v));
}

// ---- the tracker's per-frame post-processing (include/tf_fused.h tf_postprocess_pack_f32): one thread per query
__global__ void __launch_bounds__(256)
postprocess_pack_kernel(const float *__restrict__ logits, const float *__restrict__ boxes, float *__restrict__ out, long long Q,
                        int C, float img_h, float img_w, int clip)
{
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    float best = 1.f / (1.f + expf(-logits[q * C]));
    int label = 0;
    for (int c = 1; c < C; ++c) {
        const float s = 1.f / (1.f + expf(-logits[q * C + c]));
        if (s > best) { best = s; label = c; }   // (the first class that attains the maximum, as torch.max)
    }
    const float4 b = *reinterpret_cast<const float4 *>(boxes + q * 4);
    const float hw = 0.5f * b.z, hh = 0.5f * b.w;
    float x0 = (b.x - hw) * img_w, y0 = (b.y - hh) * img_h, x1 = (b.x + hw) * img_w, y1 = (b.y + hh) * img_h;
    if (clip) {
        x0 = fminf(fmaxf(x0, 0.f), img_w); x1 = fminf(fmaxf(x1, 0.f), img_w);
        y0 = fminf(fmaxf(y0, 0.f), img_h); y1 = fminf(fmaxf(y1, 0.f), img_h);
    }
    float *o = out + q * 6;
    o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1; o[4] = best; o[5] = (float)label;
}

// ---- out = maxpool3x3/s2/p1(relu(x + bias[c])) on channels_last activations: the stem of the backbone after its 7 x 7
// convolution (reference: models/backbone.py:45-55 FrozenBatchNorm2d shift, torchvision ResNet.relu + .maxpool).  x + bias and
// ReLU are monotone per channel, so max_i relu(x_i + b) == relu(max_i(x_i) + b) bit for bit: ONE pass reads the convolution's
// output and writes the 4x smaller pooled map, instead of an in-place bias pass and a pooling pass.  One thread per (output
// pixel, 4 channels); a wave covers whole pixels' channel vectors (coalesced 16-byte accesses).
__global__ void __launch_bounds__(256)
bias_relu_maxpool_kernel(const float *__restrict__ x, const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                         int C4, int Ho, int Wo, long long total)
{
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;   // total < 2^31 (host): 32-bit index arithmetic, no division loops
    if (idx >= (unsigned)total) return;
    const unsigned c4 = idx % (unsigned)C4;
    unsigned p = idx / (unsigned)C4;
    const int ox = (int)(p % (unsigned)Wo);
    p /= (unsigned)Wo;
    const int oy = (int)(p % (unsigned)Ho);
    const long long n = p / (unsigned)Ho;
    const f32x4_t *xi = reinterpret_cast<const f32x4_t *>(x) + n * (long long)H * W * C4 + c4;
    // the nine loads first (clamped addresses: a window position outside the image re-reads an inside one, which cannot
    // change a maximum), no branch between them
    f32x4_t v[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = min(max(2 * oy + dy, 0), H - 1);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = min(max(2 * ox + dx, 0), W - 1);
            v[(dy + 1) * 3 + dx + 1] = xi[((long long)iy * W + ix) * C4];
        }
    }
    f32x4_t m = v[4];   // the window's centre (2 oy, 2 ox) always lies inside
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        m.x = v[k].x > m.x ? v[k].x : m.x;
        m.y = v[k].y > m.y ? v[k].y : m.y;
        m.z = v[k].z > m.z ? v[k].z : m.z;
        m.w = v[k].w > m.w ? v[k].w : m.w;
    }
    const f32x4_t b = reinterpret_cast<const f32x4_t *>(bias)[c4];
    m += b;
    m.x = m.x < 0.f ? 0.f : m.x;
    m.y = m.y < 0.f ? 0.f : m.y;
    m.z = m.z < 0.f ? 0.f : m.z;
    m.w = m.w < 0.f ? 0.f : m.w;
    reinterpret_cast<f32x4_t *>(out)[idx] = m;
}

}  // namespace

extern "C" {

int tf_bias_relu_maxpool_f32(const float *x, const float *bias, float *out, int N, int H, int W, int C, void *stream)
{
    if (!x || !bias || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(x) || !aligned16(bias) || !aligned16(out)) return TF_MSDA_ERR_BAD_DIMS;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (total >= (1LL << 31)) return TF_MSDA_ERR_BAD_DIMS;
    hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias,
                       out, H, W, C / 4, Ho, Wo, total);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_box_refine_f32(const float *delta, const float *ref, float *out, int64_t rows, int ref_dim, float eps, void *stream)
{
    if (!delta || !ref || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (rows <= 0 || (ref_dim != 2 && ref_dim != 4) || rows > (1LL << 40)) return TF_MSDA_ERR_BAD_DIMS;
    const long long blocks = (rows * 4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    hipLaunchKernelGGL(box_refine_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), delta, ref,
                       out, (long long)rows, ref_dim, eps);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_postprocess_pack_f32(const float *logits, const float *boxes, float *out, int64_t Q, int C, float img_h, float img_w,
                            int clip, void *stream)
{
    if (!logits || !boxes || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (Q <= 0 || C <= 0 || Q > (1LL << 31)) return TF_MSDA_ERR_BAD_DIMS;
    if (reinterpret_cast<uintptr_t>(boxes) & 15) return TF_MSDA_ERR_BAD_DIMS;
    const long long blocks = (Q + 255) / 256;
    hipLaunchKernelGGL(postprocess_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), logits, boxes,
                       out, (long long)Q, C, img_h, img_w, clip);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_upsample_add_nhwc_f32(const float *low, const float *fpn, float *out, int N, int q_per_image, int h, int w, int H, int W, int C,
                             void *stream)
{
    if (!low || !fpn || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || q_per_image <= 0 || (N % q_per_image) || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3))
        return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(low) || !aligned16(fpn) || !aligned16(out)) return TF_MSDA_ERR_BAD_DIMS;
    const long long total = (long long)N * H * W * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(upsample_add_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), low, fpn, out,
                       q_per_image, h, w, H, W, C / 4, (float)h / (float)H, (float)w / (float)W, total);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_groupnorm_relu_conv3x3_c1_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *weight, float bias,
                                          float *out, double *workspace, int N, int H, int W, int C, int G, float eps, void *stream)
{
    if (!x || !gamma || !beta || !weight || !out || !workspace) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || N > 65535 || H <= 0 || W <= 0 || (C != 16 && C != 32) || G <= 0 || (C % G) != 0) return TF_MSDA_ERR_BAD_DIMS;
    if ((long long)H * W >= (1LL << 31)) return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int HW = H * W;
    const unsigned n_ws = 2u * (unsigned)N * (unsigned)G;
    hipLaunchKernelGGL(groupnorm_zero_kernel, dim3((n_ws + 255) / 256), dim3(256), 0, s, workspace, n_ws);   // (a kernel, not a memset node)
    if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    const unsigned sblocks = (unsigned)((HW + kGnRowsPerBlock - 1) / kGnRowsPerBlock);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(sblocks, (unsigned)N), dim3(256), 0, s, x, workspace, HW, C, G, (long long)HW * C);
    const dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 7) / 8), (unsigned)N);
    if (C == 16)
        hipLaunchKernelGGL(gn_relu_conv3x3_c1_kernel<16>, grid, dim3(256), 0, s, x, (const double *)workspace, gamma, beta, weight, bias,
                           out, H, W, G, eps);
    else
        hipLaunchKernelGGL(gn_relu_conv3x3_c1_kernel<32>, grid, dim3(256), 0, s, x, (const double *)workspace, gamma, beta, weight, bias,
                           out, H, W, G, eps);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_mask_label_map_f32(const float *logits, const int *order, int16_t *label, int n_tracks, int h, int w, int pad_h, int pad_w, int img_h,
                          int img_w, int out_h, int out_w, float threshold, void *stream)
{
    if (!logits || !order || !label) return TF_MSDA_ERR_NULL_POINTER;
    if (n_tracks <= 0 || n_tracks > 32767 || h <= 0 || w <= 0 || pad_h <= 0 || pad_w <= 0 || img_h <= 0 || img_w <= 0 || img_h > pad_h ||
        img_w > pad_w || out_h <= 0 || out_w <= 0)
        return TF_MSDA_ERR_BAD_DIMS;
    const dim3 grid((unsigned)((out_w + 31) / 32), (unsigned)((out_h + 7) / 8));
    hipLaunchKernelGGL(mask_label_map_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), logits, order, reinterpret_cast<short *>(label),
                       n_tracks, h, w, pad_h, pad_w, img_h, img_w, out_h, out_w, threshold);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

static int groupnorm_nhwc_impl(const float *x, const float *gamma, const float *beta, float *out, double *workspace, int N, int HW,
                               int C, int G, float eps, int64_t x_image_stride, int64_t out_image_stride, bool relu, void *stream)
{
    if (!x || !gamma || !beta || !out || !workspace) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 256 || C > 1024 || (C % G) != 0 || (C & 3) || N > 65535)
        return TF_MSDA_ERR_BAD_DIMS;
    if (x_image_stride < (int64_t)HW * C || out_image_stride < (int64_t)HW * C || (x_image_stride & 3) || (out_image_stride & 3))
        return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(out) || (reinterpret_cast<uintptr_t>(workspace) & 7))
        return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the statistics are accumulated by atomics: zeros first -- by a KERNEL, not hipMemsetAsync.  Inside a captured HIP graph
    // the memset node was seen (ROCm 7.2, MI355X; round 5, profiles/r05_graph_memset_groupnorm.txt) to land AFTER the
    // statistics kernel's atomics in about one replay out of a hundred when the graph was launched right behind device-to-
    // device copies: the 48-pixel level of a small frame then normalised with garbage statistics (inf) -- one NaN frame in a
    // 64-frame sequence.  Kernel -> kernel order inside a graph is an ordinary dependency edge.
    {
        const unsigned n_ws = 2u * (unsigned)N * (unsigned)G;
        hipLaunchKernelGGL(groupnorm_zero_kernel, dim3((n_ws + 255) / 256), dim3(256), 0, s, workspace, n_ws);
        if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    }
    const unsigned sblocks = (unsigned)((HW + kGnRowsPerBlock - 1) / kGnRowsPerBlock);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(sblocks, (unsigned)N), dim3(256), 0, s, x, workspace, HW, C, G,
                       (long long)x_image_stride);
    long long ablocks = ((long long)HW * (C / 4) + 255) / 256;
    if (ablocks > 256 * 8) ablocks = 256 * 8;
    if (relu)
        hipLaunchKernelGGL(groupnorm_apply_kernel<true>, dim3((unsigned)ablocks, (unsigned)N), dim3(256), 0, s, x,
                           (const double *)workspace, gamma, beta, out, HW, C, G, eps, (long long)x_image_stride,
                           (long long)out_image_stride);
    else
        hipLaunchKernelGGL(groupnorm_apply_kernel<false>, dim3((unsigned)ablocks, (unsigned)N), dim3(256), 0, s, x,
                           (const double *)workspace, gamma, beta, out, HW, C, G, eps, (long long)x_image_stride,
                           (long long)out_image_stride);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_groupnorm_stats_nhwc_f32(const float *x, double *workspace, int N, int HW, int C, int G, int64_t x_image_stride, void *stream)
{
    if (!x || !workspace) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 256 || C > 1024 || (C % G) != 0 || (C & 3) || N > 65535) return TF_MSDA_ERR_BAD_DIMS;
    if (x_image_stride < (int64_t)HW * C || (x_image_stride & 3) || !aligned16(x) || (reinterpret_cast<uintptr_t>(workspace) & 7))
        return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned n_ws = 2u * (unsigned)N * (unsigned)G;
    hipLaunchKernelGGL(groupnorm_zero_kernel, dim3((n_ws + 255) / 256), dim3(256), 0, s, workspace, n_ws);   // (a kernel, not a memset node)
    if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    const unsigned sblocks = (unsigned)((HW + kGnRowsPerBlock - 1) / kGnRowsPerBlock);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(sblocks, (unsigned)N), dim3(256), 0, s, x, workspace, HW, C, G, (long long)x_image_stride);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_groupnorm_nhwc_f32(const float *x, const float *gamma, const float *beta, float *out, double *workspace, int N,
                          int HW, int C, int G, float eps, int64_t x_image_stride, int64_t out_image_stride, void *stream)
{
    return groupnorm_nhwc_impl(x, gamma, beta, out, workspace, N, HW, C, G, eps, x_image_stride, out_image_stride, false, stream);
}

int tf_groupnorm_relu_nhwc_f32(const float *x, const float *gamma, const float *beta, float *out, double *workspace, int N,
                               int HW, int C, int G, float eps, int64_t x_image_stride, int64_t out_image_stride, void *stream)
{
    return groupnorm_nhwc_impl(x, gamma, beta, out, workspace, N, HW, C, G, eps, x_image_stride, out_image_stride, true, stream);
}

int tf_bias_act_f32(float *x, const float *bias, const float *residual, int64_t n, int C, int relu,
                    void *stream)
{
    if (!x || !bias) return TF_MSDA_ERR_NULL_POINTER;
    if (n <= 0 || C <= 0 || (C & 3) || (n % C) != 0) return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(x) || !aligned16(bias) || (residual && !aligned16(residual)))
        return TF_MSDA_ERR_BAD_DIMS;
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride beyond 16 workgroups per CU
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (residual && relu)
        hipLaunchKernelGGL((bias_act_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, bias, residual, n4, C / 4);
    else if (residual)
        hipLaunchKernelGGL((bias_act_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, bias, residual, n4, C / 4);
    else if (relu)
        hipLaunchKernelGGL((bias_act_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, bias, residual, n4, C / 4);
    else
        hipLaunchKernelGGL((bias_act_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, bias, residual, n4, C / 4);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int tf_add_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta,
                         float *out, int64_t rows, int C, float eps, void *stream)
{
    if (!x || !gamma || !beta || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (rows <= 0 || C <= 0 || (C & 3) || C > 4096) return TF_MSDA_ERR_BAD_DIMS;
    if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(out) ||
        (res && !aligned16(res)))
        return TF_MSDA_ERR_BAD_DIMS;
    long long blocks = (rows + 3) / 4;   // 4 waves (rows) per 256-thread workgroup
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = (C / 4 + 63) / 64;
    if (chunks <= 1)
        hipLaunchKernelGGL(add_layernorm_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, x, res,
                           gamma, beta, out, (long long)rows, C, eps);
    else if (chunks <= 2)
        hipLaunchKernelGGL(add_layernorm_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, x, res,
                           gamma, beta, out, (long long)rows, C, eps);
    else if (chunks <= 4)
        hipLaunchKernelGGL(add_layernorm_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, x, res,
                           gamma, beta, out, (long long)rows, C, eps);
    else
        hipLaunchKernelGGL(add_layernorm_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, x, res,
                           gamma, beta, out, (long long)rows, C, eps);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

}  // extern "C"
