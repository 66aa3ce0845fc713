// trackformer_amd/csrc/mha_core.hip -- tf_mha_core_f32 (include/tf_fused.h), gfx950.
//
// Scaled-dot-product attention core of the decoder's self-attention among the (track + object) queries
// (reference: models/deformable_transformer.py:364-383, nn.MultiheadAttention with q = k = tgt + query_pos,
// v = tgt; eval mode, so no dropout):  out[n, l, h, :] = softmax_j(scale * q[n, l, h, :] . k[n, j, h, :]) v[n, j, h, :].
// The problem is tiny (cfg 2: 400 x 400 x 8 heads x 32, 0.08 GFLOP; cfg 4: 800 x 800 x 8 x 36): what matters is
// launch count and latency, not the matrix cores.  One launch, fp32 throughout (the reference's arithmetic type):
//   * grid (ceil(Lq / 16), heads, batch), 256 threads; a workgroup owns 16 queries of one head;
//   * phase 1: the 16 x Lk score tile goes to LDS: thread (query t >> 4, key t & 15 + 16 i) takes the dot product
//     of its query (LDS, broadcast inside the 16 lanes) with one key row (global, L2 resident: K and V of a head are
//     51 KB each and every workgroup of the head reads them);
//   * phase 2: row max / exp / row sum with the 16 lanes of a query (DPP row reductions), exp values stay in LDS;
//   * phase 3: P V: thread (query, 16-byte channel group) walks the keys, four at a time.
// Replaces the library SDPA kernel (an AOTriton-generated `attn_fwd`, 40 us per decoder layer at cfg 2).
#include <hip/hip_runtime.h>

#include <float.h>
#include <stdint.h>

#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int TQ = 16, THREADS = 256, MAXD4 = 16;

__device__ __forceinline__ float row16_max(float v)
{
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 16));
    return v;
}
__device__ __forceinline__ float row16_sum(float v)
{
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}

// q / k / v / out: element (n, l, h, c) at base + (n * L + l) * ld + h * D + c   (ld in floats: rows of a fused
// projection output may hold q | k side by side).  key_mask [N, Lk] bytes, non-zero = ignore that key (may be null).
__global__ void __launch_bounds__(THREADS)
mha_core_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                float *__restrict__ out, const unsigned char *__restrict__ key_mask, int Lq, int Lk, int D, int ldq,
                int ldk, int ldv, int ldo, float scale, int lk_pad)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_q = smem;                       // [TQ][D]
    float *s_s = smem + TQ * D;              // [TQ][lk_pad]
    float *s_inv = s_s + TQ * lk_pad;        // [TQ]
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, n = blockIdx.z;
    const int D4 = D >> 2;
    const int nq = min(TQ, Lq - q0);

    for (int i = tid; i < TQ * D4; i += THREADS) {
        const int qi = i / D4, c4 = i - qi * D4;
        const int ql = min(q0 + qi, Lq - 1);
        reinterpret_cast<f32x4_t *>(s_q)[i] =
            *reinterpret_cast<const f32x4_t *>(q + ((size_t)n * Lq + ql) * ldq + h * D + c4 * 4);
    }
    __syncthreads();

    const int qi = tid >> 4, kl = tid & 15;
    // ---- phase 1: scores
    {
        f32x4_t qv[MAXD4];
#pragma unroll
        for (int c = 0; c < MAXD4; ++c)
            if (c < D4) qv[c] = reinterpret_cast<const f32x4_t *>(s_q)[qi * D4 + c];
        for (int j = kl; j < Lk; j += 16) {
            const float *kr = k + ((size_t)n * Lk + j) * ldk + h * D;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < MAXD4; ++c)
                if (c < D4) {
                    const f32x4_t kv = *reinterpret_cast<const f32x4_t *>(kr + c * 4);
                    acc += qv[c].x * kv.x + qv[c].y * kv.y + qv[c].z * kv.z + qv[c].w * kv.w;
                }
            acc *= scale;
            if (key_mask != nullptr && key_mask[(size_t)n * Lk + j]) acc = -INFINITY;
            s_s[qi * lk_pad + j] = acc;
        }
    }
    __syncthreads();
    // ---- phase 2: softmax over the keys (16 lanes per query)
    {
        float m = -INFINITY;
        for (int j = kl; j < Lk; j += 16) m = fmaxf(m, s_s[qi * lk_pad + j]);
        m = row16_max(m);
        const float mm = m == -INFINITY ? 0.f : m;   // a fully masked row gives zeros, not NaN
        float sum = 0.f;
        for (int j = kl; j < Lk; j += 16) {
            const float e = expf(s_s[qi * lk_pad + j] - mm);
            s_s[qi * lk_pad + j] = e;
            sum += e;
        }
        sum = row16_sum(sum);
        if (kl == 0) s_inv[qi] = sum > 0.f ? 1.f / sum : 0.f;
    }
    __syncthreads();
    // ---- phase 3: out = P V
    if (kl < D4 && qi < nq) {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        const float *vb = v + (size_t)n * Lk * ldv + h * D + kl * 4;
        const float *p = s_s + qi * lk_pad;
        int j = 0;
        for (; j + 4 <= Lk; j += 4) {
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(vb + (size_t)(j + 0) * ldv);
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t *>(vb + (size_t)(j + 1) * ldv);
            const f32x4_t v2 = *reinterpret_cast<const f32x4_t *>(vb + (size_t)(j + 2) * ldv);
            const f32x4_t v3 = *reinterpret_cast<const f32x4_t *>(vb + (size_t)(j + 3) * ldv);
            const f32x4_t pp = *reinterpret_cast<const f32x4_t *>(p + j);
            a0 += v0 * pp.x;
            a1 += v1 * pp.y;
            a2 += v2 * pp.z;
            a3 += v3 * pp.w;
        }
        for (; j < Lk; ++j) a0 += *reinterpret_cast<const f32x4_t *>(vb + (size_t)j * ldv) * p[j];
        const f32x4_t r = ((a0 + a1) + (a2 + a3)) * s_inv[qi];
        *reinterpret_cast<f32x4_t *>(out + ((size_t)n * Lq + q0 + qi) * ldo + h * D + kl * 4) = r;
    }
}

}  // namespace

extern "C" int tf_mha_core_f32(const float *q, const float *k, const float *v, float *out, const unsigned char *key_mask,
                               int N, int Lq, int Lk, int H, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                               void *stream)
{
    if (!q || !k || !v || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || Lq <= 0 || Lk <= 0 || H <= 0 || D <= 0 || (D & 3) || D > 4 * MAXD4 || H > 65535 || N > 65535)
        return TF_MSDA_ERR_BAD_DIMS;
    if ((ldq | ldk | ldv | ldo) & 3) return TF_MSDA_ERR_BAD_DIMS;
    if (ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(out)) & 15)
        return TF_MSDA_ERR_BAD_DIMS;
    const int lk_pad = (Lk + 3) & ~3;
    const size_t lds = ((size_t)TQ * D + (size_t)TQ * lk_pad + TQ) * sizeof(float);
    if (lds > 160 * 1024) return TF_MSDA_ERR_BAD_DIMS;   // Lk <= ~2500
    if (lds > 64 * 1024) {
        static int raised_dev_mask = 0;   // a handful of devices at most; benign race (idempotent call)
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 31 || !(raised_dev_mask & (1 << dev))) {
            if (hipFuncSetAttribute((const void *)&mha_core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess)
                return TF_MSDA_ERR_LAUNCH;
            if (dev < 31) raised_dev_mask |= 1 << dev;
        }
    }
    const dim3 grid((unsigned)((Lq + TQ - 1) / TQ), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL(mha_core_kernel, grid, dim3(THREADS), lds, static_cast<hipStream_t>(stream), q, k, v, out, key_mask,
                       Lq, Lk, D, ldq, ldk, ldv, ldo, scale, lk_pad);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}
