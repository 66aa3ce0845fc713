// trackformer_amd/csrc/mha_core.hip -- tf_mha_core_f32 (include/tf_fused.h), gfx950.
//
// Scaled-dot-product attention core of the decoder's self-attention among the (track + object) queries
// (reference: models/deformable_transformer.py:364-383, nn.MultiheadAttention with q = k = tgt + query_pos,
// v = tgt; eval mode, so no dropout):  out[n, l, h, :] = softmax_j(scale * q[n, l, h, :] . k[n, j, h, :]) v[n, j, h, :].
// The problem is tiny (cfg 2: 400 x 400 x 8 heads x 32, 0.08 GFLOP; cfg 4: 800 x 800 x 8 x 36): what matters is
// launch count and latency, not the matrix cores.  One launch, fp32 throughout (the reference's arithmetic type):
//   * grid (ceil(Lq / 16), heads, batch), 256 threads; a workgroup owns 16 queries of one head;
//   * K and V of the head (51 KB each at cfg 2, L2 resident) pass through LDS in chunks of 64 keys, loaded with
//     coalesced 16-byte accesses (a first version read the key rows straight from global memory inside the dot
//     product loops: 53 us per call, slower than the library kernel it replaces -- latency-bound loops);
//   * phase 1: the 16 x Lk score tile goes to LDS: thread (query t >> 4, keys t & 15 + 16 i of the chunk);
//   * phase 2: row max / exp / row sum with the 16 lanes of a query, exp values stay in LDS;
//   * phase 3: P V: lane (query, 16-byte channel group, key half) walks its half of every chunk.
// Replaces the library SDPA kernel (an AOTriton-generated `attn_fwd`, 40 us per decoder layer at cfg 2).
#include <hip/hip_runtime.h>

#include <float.h>
#include <stdlib.h>

#include <atomic>
#include <stdint.h>

#include "msda_common.h"
#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int TQ = 16, THREADS = 256, MAXD4 = 16;   // 16 queries per workgroup; key chunks of `kc` keys (multiple of 64)

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
// K and V pass through LDS in chunks of 64 keys (coalesced 16-byte loads, rows padded by 4 floats so that the 16
// lanes of a query, which read 16 different key rows at the same column, hit different banks); the 16 x Lk score
// tile stays in LDS between the passes.
// (A staging loop with eight 16-byte loads in flight per thread was measured in round 3: 23.3 -> 22.7 us at 400 keys, 67.4 ->
// 76.7 us at 800 keys x 36 channels -- removed.)
template <int D4>
__global__ void __launch_bounds__(THREADS)
mha_core_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                float *__restrict__ out, const unsigned char *__restrict__ key_mask, int Lq, int Lk, int ldq,
                int ldk, int ldv, int ldo, float scale, int lk_pad, int kc)
{
    constexpr int D = D4 * 4, DP = D + 4;      // padded row length of the staged K / V chunk
    constexpr int HALVES = 2 * D4 <= 16 ? 2 : 1;   // phase 3: the 16 lanes of a query split the keys of a chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_kv = smem;                      // [kc][DP]  one chunk of K (phase 1) or V (phase 3)
    float *s_s = s_kv + kc * DP;             // [TQ][lk_pad] scores, then exp values
    float *s_inv = s_s + TQ * lk_pad;        // [TQ]
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, n = blockIdx.z;
    const int nq = min(TQ, Lq - q0);
    const int qi = tid >> 4, kl = tid & 15;

    // this thread's query row in registers (16 lanes share it; L1 serves the repeats)
    f32x4_t qv[D4];
    {
        const float *qr = q + ((size_t)n * Lq + min(q0 + qi, Lq - 1)) * ldq + h * D;
#pragma unroll
        for (int c = 0; c < D4; ++c) qv[c] = *reinterpret_cast<const f32x4_t *>(qr + c * 4);
    }
    auto stage = [&](const float *base, int ld, int j0) {   // rows j0 .. j0 + kc of one head -> s_kv
        for (int i = tid; i < kc * D4; i += THREADS) {
            const int r = i / D4, c = i - r * D4;
            const int j = min(j0 + r, Lk - 1);
            *reinterpret_cast<f32x4_t *>(s_kv + r * DP + c * 4) =
                *reinterpret_cast<const f32x4_t *>(base + ((size_t)n * Lk + j) * ld + h * D + c * 4);
        }
    };
    // ---- phase 1: scores = scale * q . k
    for (int j0 = 0; j0 < Lk; j0 += kc) {
        __syncthreads();
        stage(k, ldk, j0);
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < kc / 16; ++t) {
            const int r = kl + 16 * t, j = j0 + r;
            const float *kr = s_kv + r * DP;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < D4; ++c) {
                const f32x4_t kv = *reinterpret_cast<const f32x4_t *>(kr + c * 4);
                acc += qv[c].x * kv.x + qv[c].y * kv.y + qv[c].z * kv.z + qv[c].w * kv.w;
            }
            acc *= scale;
            if (j < Lk) {
                if (key_mask != nullptr && key_mask[(size_t)n * Lk + j]) acc = -INFINITY;
                s_s[qi * lk_pad + j] = acc;
            }
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
        for (int j = Lk + kl; j < lk_pad; j += 16) s_s[qi * lk_pad + j] = 0.f;   // padding keys weigh nothing
        sum = row16_sum(sum);
        if (kl == 0) s_inv[qi] = sum > 0.f ? 1.f / sum : 0.f;
    }
    // ---- phase 3: out = P V; lane (query, channel group cg, key half kh)
    const int cg = kl % D4, kh = kl / D4;
    const bool worker = kh < HALVES;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int j0 = 0; j0 < Lk; j0 += kc) {
        __syncthreads();
        stage(v, ldv, j0);
        __syncthreads();
        if (worker) {
            const int PER = kc / HALVES;
            const float *p = s_s + qi * lk_pad + j0 + kh * PER;
            const float *vr = s_kv + (kh * PER) * DP + cg * 4;
            const int lim = min(PER, lk_pad - (j0 + kh * PER));   // keys past lk_pad do not exist in s_s
#pragma unroll 4
            for (int r = 0; r < PER; r += 2) {
                if (r < lim) {
                    a0 += *reinterpret_cast<const f32x4_t *>(vr + r * DP) * p[r];
                    a1 += *reinterpret_cast<const f32x4_t *>(vr + (r + 1) * DP) * p[r + 1];
                }
            }
        }
    }
    f32x4_t r = a0 + a1;
    if (HALVES == 2) {   // combine the two key halves: lanes cg and cg + D4 of the same query
        f32x4_t o;
        o.x = __shfl(r.x, (tid & 63) + D4);
        o.y = __shfl(r.y, (tid & 63) + D4);
        o.z = __shfl(r.z, (tid & 63) + D4);
        o.w = __shfl(r.w, (tid & 63) + D4);
        r += o;
    }
    if (kl < D4 && qi < nq)
        *reinterpret_cast<f32x4_t *>(out + ((size_t)n * Lq + q0 + qi) * ldo + h * D + kl * 4) = r * s_inv[qi];
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
    const int lk_pad = (Lk + 1) & ~1;
    // key chunk: everything at once when K (or V) of a head fits next to the score tile (cfg 2: 400 keys), else the
    // largest multiple of 64 that does
    const size_t fixed = ((size_t)TQ * lk_pad + TQ) * sizeof(float), budget = 150 * 1024;
    if (fixed + 64 * (size_t)(D + 4) * sizeof(float) > budget) return TF_MSDA_ERR_BAD_DIMS;   // Lk <= ~2200
    int kc = (int)((budget - fixed) / ((size_t)(D + 4) * sizeof(float))) & ~63;
    if (kc > ((Lk + 63) & ~63)) kc = (Lk + 63) & ~63;
    // one workgroup per CU at most uses the big chunk; with more workgroups than CUs (cfg 4: 400) small chunks keep
    // several of them resident (measured: 66 vs 109 us at 800 x 800 x 8 x 36)
    if ((long long)((Lq + TQ - 1) / TQ) * H * N > 256) kc = 64;
    const size_t lds = (size_t)kc * (D + 4) * sizeof(float) + fixed;
    const void *fn = nullptr;
    switch (D / 4) {
    case 4: fn = (const void *)&mha_core_kernel<4>; break;
    case 8: fn = (const void *)&mha_core_kernel<8>; break;
    case 9: fn = (const void *)&mha_core_kernel<9>; break;
    case 16: fn = (const void *)&mha_core_kernel<16>; break;
    default: return TF_MSDA_ERR_BAD_DIMS;   // head dimensions 16, 32, 36, 64
    }
    if (lds > 64 * 1024) {
        static int raised_dev_mask[17] = {0};   // per kernel (index D / 4) and device; benign race
        int dev = 0;
        (void)hipGetDevice(&dev);
        int &mask = raised_dev_mask[D / 4];
        if (dev >= 31 || !(mask & (1 << dev))) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return TF_MSDA_ERR_LAUNCH;
            if (dev < 31) mask |= 1 << dev;
        }
    }
    const dim3 grid((unsigned)((Lq + TQ - 1) / TQ), (unsigned)H, (unsigned)N);
    void *argv[] = {(void *)&q, (void *)&k, (void *)&v, (void *)&out, (void *)&key_mask, (void *)&Lq, (void *)&Lk, (void *)&ldq,
                    (void *)&ldk, (void *)&ldv, (void *)&ldo, (void *)&scale, (void *)&lk_pad, (void *)&kc};
    return hipLaunchKernel(fn, grid, dim3(THREADS), argv, lds, static_cast<hipStream_t>(stream)) == hipSuccess
               ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}
