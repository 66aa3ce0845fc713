// trackformer_amd/csrc/fused_ops.hip -- fused element-wise kernels (include/tf_fused.h), gfx950.
//
// Both kernels are HBM-bound streaming passes: 16 bytes per lane per access, grid-stride, one read of
// every operand and one write of the result.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
bias_act_kernel(float *__restrict__ x, const float *__restrict__ bias,
                const float *__restrict__ residual, long long n4, int C4, int relu)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4_t v = reinterpret_cast<const f32x4_t *>(x)[i];
        const f32x4_t b = reinterpret_cast<const f32x4_t *>(bias)[i % C4];
        v += b;
        if (residual != nullptr) v += reinterpret_cast<const f32x4_t *>(residual)[i];
        if (relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<f32x4_t *>(x)[i] = v;
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
                orow[j] = (v[k] - mean) * rstd * g + b;
            }
        }
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

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
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, bias, residual, n4, C / 4, relu);
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
