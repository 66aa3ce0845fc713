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
// Round 5: the default is mha_mfma_stream_kernel below -- both products as exact-fp32 matrix instructions
// (v_mfma_f32_16x16x4_f32), K / V operands fetched straight into registers, softmax on the accumulators: 9.5 us at 400 x 400 x
// 8 x 32 where the vector kernel above takes 23.1 (27.4 against 64.8 at 800 x 800 x 8 x 36; profiles/r05_mha_matrix_cores.txt).
// tf_msda_set_option("mha_mfma", 0 / 2) / TF_MHA_MFMA select the vector kernel / the matrix-core kernel with K and V staged in
// LDS (what key counts above 1024 use).
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


// ---- round 5: the same attention on the matrix cores, in exact fp32 ----------------------------------------------------------
// v_mfma_f32_16x16x4_f32 multiplies fp32 operands exactly and accumulates in fp32 (the fused multiply-add chain of the vector
// ALU at the same rate, MI355X_MICROARCH.md): no split product is needed for the reference's arithmetic type.  What it buys
// here is the instruction stream, not the rate: a 16 x 16 tile of scores is D / 4 instructions of one wave (8 at D = 32) instead
// of 16 x 16 x D / 4 LDS reads and four times as many fused multiply-adds spread over 256 threads.
//   * a workgroup still owns 16 queries of one head; K and V of the head pass through LDS in chunks of `kc` keys (a multiple of
//     16); when both fit next to the score tile at once (cfg 2: 400 keys x 32 channels) V is requested right behind K and
//     lands under the score and softmax phases;
//   * scores: wave w takes the 16-key tiles w, w + 4, ... of a chunk.  The reduction index of an MFMA step is (lane / 16): lane
//     (m = lane % 16, kq = lane / 16) holds channels 16 t + 4 kq .. + 3 of its row as ONE 16-byte read -- four steps per read;
//     both operands use the same channel order, so the sum is the plain dot product (channels beyond a multiple of 16 -- head
//     dimension 36 -- go through one 4-byte read per step);
//   * softmax: as in the vector kernel (16 lanes per query, exp values stay in LDS);
//   * P V: wave w takes the 16-key groups w, w + 4, ... of a chunk: A = P (one 16-byte read of four keys' weights per lane and
//     group), B = V (four 4-byte reads per 16-channel tile), D / 16 accumulator tiles per wave; the four waves' partial sums
//     meet in LDS at the end.
typedef float f32x4m_t __attribute__((ext_vector_type(4)));

constexpr int mha_row_floats(int D)   // padded K / V row in LDS: 16-byte aligned, and 4 rows apart must not share their banks
{
    int dp = D + 4;
    if ((4 * dp) % 32 == 0) dp += 4;
    return dp;
}

template <int D>
__global__ void __launch_bounds__(THREADS)
mha_mfma_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, float *__restrict__ out,
                const unsigned char *__restrict__ key_mask, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, float scale,
                int lk_pad, int kc, int two_buffers)
{
    constexpr int D4 = D / 4, DP = mha_row_floats(D);
    constexpr int G16 = D / 16, TAIL = (D % 16) / 4;   // 16-byte operand reads per row, 4-byte tail steps
    constexpr int CT = (D + 15) / 16;                   // 16-channel output tiles (the last one may be partly empty)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_k = smem;                                        // [kc][DP]: a chunk of K -- or of V when there is one buffer
    float *s_v = two_buffers ? s_k + (size_t)kc * DP : s_k;   // [kc][DP]
    float *s_s = s_k + (size_t)(two_buffers ? 2 : 1) * kc * DP;   // [TQ][lk_pad] scores, then exp values
    float *s_inv = s_s + TQ * lk_pad;                         // [TQ]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, n = blockIdx.z;
    const int nq = min(TQ, Lq - q0);
    const int m16 = lane & 15, kq = lane >> 4;

    auto stage = [&](float *dst, const float *base, int ld, int j0) {   // rows j0 .. j0 + kc of one head (clamped to the last key)
        for (int i = tid; i < kc * D4; i += THREADS) {
            const int r = i / D4, c = i - r * D4;
            const int j = min(j0 + r, Lk - 1);
            *reinterpret_cast<f32x4m_t *>(dst + r * DP + c * 4) =
                *reinterpret_cast<const f32x4m_t *>(base + ((size_t)n * Lk + j) * ld + h * D + c * 4);
        }
    };
    // this lane's part of its query row: the A operand of every score tile
    f32x4m_t qa[G16 > 0 ? G16 : 1];
    float qt[TAIL > 0 ? TAIL : 1];
    {
        const float *qr = q + ((size_t)n * Lq + min(q0 + m16, Lq - 1)) * ldq + h * D;
#pragma unroll
        for (int t = 0; t < G16; ++t) qa[t] = *reinterpret_cast<const f32x4m_t *>(qr + 16 * t + 4 * kq);
#pragma unroll
        for (int t = 0; t < TAIL; ++t) qt[t] = qr[16 * G16 + 4 * t + kq];
    }

    // ---- phase 1: scores = scale * q . k
    for (int j0 = 0; j0 < Lk; j0 += kc) {
        __syncthreads();
        stage(s_k, k, ldk, j0);
        if (two_buffers) stage(s_v, v, ldv, j0);   // (one chunk: Lk <= kc) lands under the phases below
        __syncthreads();
        const int ntiles = (min(kc, Lk - j0) + 15) >> 4;
        for (int t16 = wave; t16 < ntiles; t16 += THREADS / 64) {
            const float *kr = s_k + (size_t)(t16 * 16 + m16) * DP;   // this lane's key row of the tile
            f32x4m_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < G16; ++t) {
                const f32x4m_t kb = *reinterpret_cast<const f32x4m_t *>(kr + 16 * t + 4 * kq);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].x, kb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].y, kb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].z, kb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].w, kb.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < TAIL; ++t)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qt[t], kr[16 * G16 + 4 * t + kq], acc, 0, 0, 0);
            // C / D of the 16 x 16 tile: column (key) = lane % 16, rows (queries) 4 * (lane / 16) + 0 .. 3
            const int j = j0 + t16 * 16 + m16;
            if (j < Lk) {
                const bool masked = key_mask != nullptr && key_mask[(size_t)n * Lk + j];
#pragma unroll
                for (int r = 0; r < 4; ++r) s_s[(4 * kq + r) * lk_pad + j] = masked ? -INFINITY : acc[r] * scale;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: softmax over the keys (16 lanes per query), as in mha_core_kernel
    {
        const int qi = tid >> 4, kl = tid & 15;
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
    // ---- phase 3: out = P V
    f32x4m_t o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = f32x4m_t{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < Lk; j0 += kc) {
        __syncthreads();   // (the exp values are complete; with one buffer: every wave is done with the previous chunk)
        if (!two_buffers) {
            stage(s_v, v, ldv, j0);
            __syncthreads();
        }
        const int ngroups = (min(kc, lk_pad - j0) + 15) >> 4;
        for (int g = wave; g < ngroups; g += THREADS / 64) {
            // A: the weights of keys 16 g + 4 kq .. + 3 for query m16 (lk_pad is a multiple of 16: never past the row)
            const f32x4m_t pa = *reinterpret_cast<const f32x4m_t *>(s_s + (size_t)m16 * lk_pad + j0 + 16 * g + 4 * kq);
            const float *vr = s_v + (size_t)(16 * g + 4 * kq) * DP + m16;   // B: channel m16 of those four keys' rows
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const bool have = 16 * c + m16 < D;   // the last channel tile of head dimension 36 is a quarter full
                const float b0 = have ? vr[16 * c] : 0.f, b1 = have ? vr[16 * c + DP] : 0.f;
                const float b2 = have ? vr[16 * c + 2 * DP] : 0.f, b3 = have ? vr[16 * c + 3 * DP] : 0.f;
                o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, b0, o[c], 0, 0, 0);
                o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, b1, o[c], 0, 0, 0);
                o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, b2, o[c], 0, 0, 0);
                o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, b3, o[c], 0, 0, 0);
            }
        }
    }
    // the four waves' partial sums: [wave][query][channel] in the (finished) K buffer
    __syncthreads();
    float *s_p = s_k;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * c + m16 < D) s_p[((size_t)wave * TQ + 4 * kq + r) * D + 16 * c + m16] = o[c][r];
    __syncthreads();
    for (int i = tid; i < TQ * D4; i += THREADS) {
        const int qi = i / D4, c4 = i - qi * D4;
        if (qi < nq) {
            f32x4m_t r = *reinterpret_cast<const f32x4m_t *>(s_p + (size_t)qi * D + c4 * 4);
#pragma unroll
            for (int w = 1; w < THREADS / 64; ++w) r += *reinterpret_cast<const f32x4m_t *>(s_p + ((size_t)w * TQ + qi) * D + c4 * 4);
            *reinterpret_cast<f32x4m_t *>(out + ((size_t)n * Lq + q0 + qi) * ldo + h * D + c4 * 4) = r * s_inv[qi];
        }
    }
}

// ---- the matrix-core kernel without the LDS round trip of K and V ------------------------------------------------------------
// Inside one workgroup every element of K and of V is the B operand of exactly ONE matrix instruction, held by exactly one lane:
// there is nothing to share through LDS.  Each wave fetches the operands of its own key tiles straight into registers, a batch of
// PF tiles at a time with all of the batch's loads in flight at once (cfg 2: a wave owns 7 of the 25 tiles -- one batch), and
// (one batch) V is requested BEFORE the score phase, so it lands under the scores and the softmax.  The score tiles of a wave stay
// in its accumulators through the softmax: row maxima and sums are 16-lane reductions plus a 4 x 16 exchange between the waves,
// and the exponentials are taken from registers -- LDS only carries the weights P from the C layout (lane = key) to the A layout
// (lane = query) and the four waves' partial outputs: 34 KB at cfg 2 instead of 141 KB, three barriers in all.
//   * scores: B = K: lane (key m = lane % 16, kq = lane / 16) holds channels 16 t + 4 kq .. + 3 of its key (16-byte loads);
//   * softmax in base 2: p = 2^(c s - max(c s)) with c = scale * log2(e), one multiply and one v_exp_f32 per score;
//   * P V: the reduction index of step s is key 16 g + 4 kq + s; the COLUMN a lane owns in output tile c is a free choice, and
//     channel VW * m + c (VW = output tiles = D / 16) makes a lane's operands of all tiles one VW-float load per key: the 16 lanes of a
//     key read its 128-byte row of the head in one piece.  (Head dimension 36: the four channels behind 32 are a third tile
//     with a quarter of its columns in use.)
// NB = batches of PF tiles per wave whose accumulators stay live: Lk <= 16 * 4 * PF * NB (1: 512 keys at D <= 36, 2: 1024).
template <int D, int NB>
__global__ void __launch_bounds__(THREADS)
mha_mfma_stream_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, float *__restrict__ out,
                       const unsigned char *__restrict__ key_mask, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, float scale,
                       int lk_pad)
{
    constexpr int D4 = D / 4, G16 = D / 16, TAIL = (D % 16) / 4, VW = G16, XT = TAIL > 0 ? 1 : 0;
    constexpr int PF = D <= 36 ? 8 : 4, NW = THREADS / 64;
    typedef float vrow_t __attribute__((ext_vector_type(VW)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_s = smem;                     // [TQ][lk_pad] the weights P
    float *s_wmax = s_s + TQ * lk_pad;     // [NW][TQ] row maxima of the waves
    float *s_wsum = s_wmax + NW * TQ;      // [NW][TQ] row sums of the waves
    float *s_p = s_wsum + NW * TQ;         // [NW][TQ][D] partial outputs of the waves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.x * TQ, h = blockIdx.y, n = blockIdx.z;
    const int nq = min(TQ, Lq - q0);
    const int m16 = lane & 15, kq = lane >> 4;
    const int ntiles = (Lk + 15) >> 4;
    const float *kh = k + (size_t)n * Lk * ldk + h * D, *vh = v + (size_t)n * Lk * ldv + h * D;

    struct VBatch {
        vrow_t main[PF][4];
        float extra[PF][XT ? 4 : 1];
    };
    auto load_v = [&](int g0, VBatch &vb) {   // the wave's groups g0, g0 + NW, ...: rows 16 g + 4 kq + s; keys past the end count as zero
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int j = 16 * (g0 + NW * i) + 4 * kq + s4;
                const float *vr = vh + (size_t)min(j, Lk - 1) * ldv;
                vrow_t x = *reinterpret_cast<const vrow_t *>(vr + VW * m16);
                if (j >= Lk) x = vrow_t(0.f);
                vb.main[i][s4] = x;
                if constexpr (XT) vb.extra[i][s4] = (m16 < 4 * TAIL && j < Lk) ? vr[16 * G16 + m16] : 0.f;
            }
    };
    VBatch vb;
    if constexpr (NB == 1) load_v(wave, vb);   // in flight across the score and softmax phases

    f32x4m_t qa[G16 > 0 ? G16 : 1];
    float qt[TAIL > 0 ? TAIL : 1];
    {
        const float *qr = q + ((size_t)n * Lq + min(q0 + m16, Lq - 1)) * ldq + h * D;
#pragma unroll
        for (int t = 0; t < G16; ++t) qa[t] = *reinterpret_cast<const f32x4m_t *>(qr + 16 * t + 4 * kq);
#pragma unroll
        for (int t = 0; t < TAIL; ++t) qt[t] = qr[16 * G16 + 4 * t + kq];
    }
    // ---- phase 1: scores (in units of log2: c = scale * log2 e) of the wave's tiles, kept in the accumulators
    const float c2 = scale * 1.44269504088896340736f;
    f32x4m_t acc[NB][PF];
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int b0 = wave + b * NW * PF;
        f32x4m_t kb[PF][G16 > 0 ? G16 : 1];
        float kt[PF][TAIL > 0 ? TAIL : 1];
        if (b == 0 || b0 < ntiles) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const float *kr = kh + (size_t)min(16 * (b0 + NW * i) + m16, Lk - 1) * ldk;
#pragma unroll
                for (int t = 0; t < G16; ++t) kb[i][t] = *reinterpret_cast<const f32x4m_t *>(kr + 16 * t + 4 * kq);
#pragma unroll
                for (int t = 0; t < TAIL; ++t) kt[i][t] = kr[16 * G16 + 4 * t + kq];
            }
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int t16 = b0 + NW * i;
            f32x4m_t a = {0.f, 0.f, 0.f, 0.f};
            if (t16 < ntiles) {   // wave-uniform
#pragma unroll
                for (int t = 0; t < G16; ++t) {
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].x, kb[i][t].x, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].y, kb[i][t].y, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].z, kb[i][t].z, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t].w, kb[i][t].w, a, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TAIL; ++t) a = __builtin_amdgcn_mfma_f32_16x16x4f32(qt[t], kt[i][t], a, 0, 0, 0);
                const int j = 16 * t16 + m16;
                const bool dead = j >= Lk || (key_mask != nullptr && key_mask[(size_t)n * Lk + min(j, Lk - 1)]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = dead ? -INFINITY : a[r] * c2;
                    mx[r] = fmaxf(mx[r], a[r]);
                }
            }
            acc[b][i] = a;
        }
    }
    // ---- phase 2: softmax.  Row 4 kq + r: the 16 lanes of a DPP row hold its keys of this wave; the waves meet in LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mx[r] = row16_max(mx[r]);
        if (m16 == 0) s_wmax[wave * TQ + 4 * kq + r] = mx[r];
    }
    __syncthreads();
    float sum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m = s_wmax[4 * kq + r];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, s_wmax[w * TQ + 4 * kq + r]);
        mx[r] = m == -INFINITY ? 0.f : m;   // a fully masked row gives zeros, not NaN
        sum[r] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int t16 = wave + (b * PF + i) * NW;
            if (t16 < ntiles) {   // (a tile reaches to 16 ntiles = lk_pad' keys: the ones past Lk get weight 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(acc[b][i][r] - mx[r]);
                    sum[r] += e;
                    s_s[(4 * kq + r) * lk_pad + 16 * t16 + m16] = e;
                }
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sum[r] = row16_sum(sum[r]);
        if (m16 == 0) s_wsum[wave * TQ + 4 * kq + r] = sum[r];
    }
    __syncthreads();
    // ---- phase 3: out = P V
    f32x4m_t o[VW + XT];
#pragma unroll
    for (int c = 0; c < VW + XT; ++c) o[c] = f32x4m_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int g0 = wave + b * NW * PF;
        if (NB > 1 && g0 < ntiles) load_v(g0, vb);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int g = g0 + NW * i;
            if (g < ntiles) {   // wave-uniform
                const f32x4m_t pa = *reinterpret_cast<const f32x4m_t *>(s_s + (size_t)m16 * lk_pad + 16 * g + 4 * kq);
#pragma unroll
                for (int c = 0; c < VW; ++c) {
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vb.main[i][0][c], o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vb.main[i][1][c], o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vb.main[i][2][c], o[c], 0, 0, 0);
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vb.main[i][3][c], o[c], 0, 0, 0);
                }
                if constexpr (XT) {
                    o[VW] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vb.extra[i][0], o[VW], 0, 0, 0);
                    o[VW] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vb.extra[i][1], o[VW], 0, 0, 0);
                    o[VW] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vb.extra[i][2], o[VW], 0, 0, 0);
                    o[VW] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vb.extra[i][3], o[VW], 0, 0, 0);
                }
            }
        }
    }
    // the waves' partial sums: query 4 kq + r, channels VW m16 .. (+ 16 G16 + m16 of the extra tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *dst = s_p + ((size_t)wave * TQ + 4 * kq + r) * D;
        vrow_t x;
#pragma unroll
        for (int c = 0; c < VW; ++c) x[c] = o[c][r];
        *reinterpret_cast<vrow_t *>(dst + VW * m16) = x;
        if constexpr (XT)
            if (m16 < 4 * TAIL) dst[16 * G16 + m16] = o[VW][r];
    }
    __syncthreads();
    for (int i = tid; i < TQ * D4; i += THREADS) {
        const int qi = i / D4, c4 = i - qi * D4;
        if (qi < nq) {
            f32x4m_t r = *reinterpret_cast<const f32x4m_t *>(s_p + (size_t)qi * D + c4 * 4);
            float total = s_wsum[qi];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                r += *reinterpret_cast<const f32x4m_t *>(s_p + ((size_t)w * TQ + qi) * D + c4 * 4);
                total += s_wsum[w * TQ + qi];
            }
            *reinterpret_cast<f32x4m_t *>(out + ((size_t)n * Lq + q0 + qi) * ldo + h * D + c4 * 4) = r * (total > 0.f ? 1.f / total : 0.f);
        }
    }
}

std::atomic<int> g_mha_mfma{-1};   // -1: TF_MHA_MFMA from the environment on first use (default on); 0 / 1: set by tf_msda_set_option("mha_mfma")

}  // namespace

namespace tfm {
int mha_set_mfma(int v)   // 1: operands streamed into registers (default), 2: K / V staged in LDS, 0: the vector kernel
{
    return g_mha_mfma.exchange(v < 0 || v > 2 ? 1 : v);
}
}  // namespace tfm

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
    int use_mfma = g_mha_mfma.load(std::memory_order_relaxed);
    if (use_mfma < 0) {
        const char *e = getenv("TF_MHA_MFMA");
        use_mfma = e == nullptr ? 1 : atoi(e);
        g_mha_mfma.store(use_mfma);
    }
    const size_t budget = 150 * 1024;
    if (use_mfma == 1 && (D == 16 || D == 32 || D == 36 || D == 64)) {
        // K and V straight into the operand registers, the score tiles in the accumulators; LDS: the weights (rows 4 floats
        // longer than a multiple of 16 keys: the four row groups of a store land in different banks) and the waves' partial outputs
        const int lkp = ((Lk + 15) & ~15) + 4;
        const int per_batch = 16 * (THREADS / 64) * (D <= 36 ? 8 : 4);   // keys one batch of tiles covers
        const size_t lds = ((size_t)TQ * lkp + 2 * (THREADS / 64) * TQ + (size_t)(THREADS / 64) * TQ * D) * sizeof(float);
        if (lds <= budget && Lk <= 2 * per_batch) {
            const bool one = Lk <= per_batch;
            const void *fns = nullptr;
            switch (D) {
            case 16: fns = one ? (const void *)&mha_mfma_stream_kernel<16, 1> : (const void *)&mha_mfma_stream_kernel<16, 2>; break;
            case 32: fns = one ? (const void *)&mha_mfma_stream_kernel<32, 1> : (const void *)&mha_mfma_stream_kernel<32, 2>; break;
            case 36: fns = one ? (const void *)&mha_mfma_stream_kernel<36, 1> : (const void *)&mha_mfma_stream_kernel<36, 2>; break;
            default: fns = one ? (const void *)&mha_mfma_stream_kernel<64, 1> : (const void *)&mha_mfma_stream_kernel<64, 2>; break;
            }
            if (lds > 64 * 1024) {
                static int raised_stream[8] = {0};   // per kernel and device; benign race
                int dev = 0;
                (void)hipGetDevice(&dev);
                int &mask = raised_stream[(D == 16 ? 0 : D == 32 ? 1 : D == 36 ? 2 : 3) * 2 + (one ? 0 : 1)];
                if (dev >= 31 || !(mask & (1 << dev))) {
                    if (hipFuncSetAttribute(fns, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                        return TF_MSDA_ERR_LAUNCH;
                    if (dev < 31) mask |= 1 << dev;
                }
            }
            const dim3 grids((unsigned)((Lq + TQ - 1) / TQ), (unsigned)H, (unsigned)N);
            void *args[] = {(void *)&q, (void *)&k, (void *)&v, (void *)&out, (void *)&key_mask, (void *)&Lq, (void *)&Lk, (void *)&ldq,
                            (void *)&ldk, (void *)&ldv, (void *)&ldo, (void *)&scale, (void *)&lkp};
            return hipLaunchKernel(fns, grids, dim3(THREADS), args, lds, static_cast<hipStream_t>(stream)) == hipSuccess
                       ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
        }
    }
    if (use_mfma && (D == 16 || D == 32 || D == 36 || D == 64)) {
        // the matrix-core kernel: key chunks are multiples of 16; K and V together when both fit next to the score tile
        const int lkp = ((Lk + 15) & ~15) + 0;
        const size_t dp = (size_t)mha_row_floats(D), fixed = ((size_t)TQ * lkp + TQ) * sizeof(float);
        const size_t partial = (size_t)(THREADS / 64) * TQ * D * sizeof(float);   // the end-of-kernel partial sums reuse the K buffer
        if (fixed + 64 * dp * sizeof(float) <= budget) {
            const int lk16 = (Lk + 15) & ~15;
            int kc = (int)((budget - fixed) / (dp * sizeof(float))) & ~15, two = 0;
            if (kc > lk16) kc = lk16;
            if ((size_t)2 * lk16 * dp * sizeof(float) + fixed <= budget && (long long)((Lq + TQ - 1) / TQ) * H * N <= 256) {
                kc = lk16;   // one chunk, two buffers: V is requested right behind K (at most one workgroup per CU uses this much LDS)
                two = 1;
            } else if ((long long)((Lq + TQ - 1) / TQ) * H * N > 256) {
                kc = 64;     // more workgroups than CUs: small chunks keep several of them resident (as the vector kernel does)
            }
            if ((size_t)kc * dp * sizeof(float) < partial) kc = (int)((partial / (dp * sizeof(float)) + 16) & ~15);
            const size_t lds = (size_t)(two ? 2 : 1) * kc * dp * sizeof(float) + fixed;
            const void *fnm = D == 16 ? (const void *)&mha_mfma_kernel<16> : D == 32 ? (const void *)&mha_mfma_kernel<32>
                              : D == 36 ? (const void *)&mha_mfma_kernel<36> : (const void *)&mha_mfma_kernel<64>;
            if (lds > 64 * 1024) {
                static int raised_mfma[4] = {0, 0, 0, 0};   // per kernel and device; benign race
                int dev = 0;
                (void)hipGetDevice(&dev);
                int &mask = raised_mfma[D == 16 ? 0 : D == 32 ? 1 : D == 36 ? 2 : 3];
                if (dev >= 31 || !(mask & (1 << dev))) {
                    if (hipFuncSetAttribute(fnm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                        return TF_MSDA_ERR_LAUNCH;
                    if (dev < 31) mask |= 1 << dev;
                }
            }
            const dim3 gridm((unsigned)((Lq + TQ - 1) / TQ), (unsigned)H, (unsigned)N);
            void *argm[] = {(void *)&q, (void *)&k, (void *)&v, (void *)&out, (void *)&key_mask, (void *)&Lq, (void *)&Lk, (void *)&ldq,
                            (void *)&ldk, (void *)&ldv, (void *)&ldo, (void *)&scale, (void *)&lkp, (void *)&kc, (void *)&two};
            return hipLaunchKernel(fnm, gridm, dim3(THREADS), argm, lds, static_cast<hipStream_t>(stream)) == hipSuccess
                       ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
        }
    }
    const int lk_pad = (Lk + 1) & ~1;
    // key chunk: everything at once when K (or V) of a head fits next to the score tile (cfg 2: 400 keys), else the
    // largest multiple of 64 that does
    const size_t fixed = ((size_t)TQ * lk_pad + TQ) * sizeof(float);
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
