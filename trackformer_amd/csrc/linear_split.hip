// trackformer_amd/csrc/linear_split.hip
//
// tf_linear_split_f32 (include/tf_fused.h): Y[M, N] = X[M, K] . W[N, K]^T + bias, optionally ReLU -- the
// nn.Linear of the encoder / decoder (reference: models/ops/modules/ms_deform_attn.py:64-88 value_proj /
// sampling_offsets / attention_weights / output_proj, models/deformable_transformer.py:282-297 linear1 / linear2)
// with fp32 inputs and outputs, computed on the bf16 matrix cores as a SPLIT product:
//     x = x_hi + x_mid,  w = w_hi + w_mid  (bf16 pieces, round to nearest even),
//     x . w ~= x_hi . w_hi + x_hi . w_mid + x_mid . w_hi      (three v_mfma_f32_32x32x16_bf16 per K-step, fp32 accumulate)
// The dropped terms are below 2^-16 of the product: the model and the tracker stay inside the parity bar
// (tools/experiments/bf16_split_linear.py: all reference goldens, track ids exact), while three bf16 passes have
// 5x the throughput of the fp32 MFMA instructions the library GEMMs use.  First version (untuned, correct):
// 25.9 us for 22 223 x 256 -> 256 (hipBLASLt fp32, tuned: 32 us), 69.7 us for 256 -> 1024 (109 us),
// 68.2 us for 1024 -> 256 (100 us); profiles/r01_split_gemm_experiment.txt.  OPT-IN (TF_SPLIT_LINEAR=1) until the
// end-to-end numbers are in.
//
//   * 256 threads = 4 waves per 128 x 128 output block, each wave 64 x 64 (2 x 2 MFMA tiles of 32 x 32);
//   * per K-slice of 32: the X tile is loaded as fp32, split into (hi, mid) in registers (v_cvt_pk_bf16_f32)
//     and stored to LDS as bf16; the weight pieces are split once by the caller (constants in inference);
//   * operands are read from LDS with ds_read_b128 (8 consecutive k of one row per lane).  The A and B fragments
//     of v_mfma_f32_32x32x16_bf16 use the same (lane >> 5, element) -> k mapping, so loading the SAME k into the
//     same slot of both makes the result independent of what that mapping is; the C/D mapping is
//     col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BK = 32, THREADS = 256;
constexpr int LDS_STRIDE = BK + 8;   // bf16 elements per LDS row: 80 bytes, keeps 16-byte alignment, spreads banks

// BM x BN output block, 4 waves as 2 x 2, each wave (BM / 2) x (BN / 2) = TI x TJ MFMA tiles of 32 x 32.
// PREFETCH: the global loads of K-slice s + 1 are issued before the MFMAs of slice s and written to LDS after them
// (register double buffer), so a block's memory latency hides under its own matrix work instead of relying on a
// second block on the CU being in the other phase.
template <int BM, int BN, bool RELU, bool PREFETCH>
__global__ void __launch_bounds__(THREADS)
split_gemm_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                  const unsigned short *__restrict__ Wmid, const float *__restrict__ bias, float *__restrict__ Y,
                  int M, int K, int N)
{
    constexpr int TI = BM / 64, TJ = BN / 64;
    constexpr int XV = (BM * BK / 4) / THREADS;   // float4 of X per thread and slice
    constexpr int WV = (BN * BK / 8) / THREADS;   // 16-byte pieces of each weight tensor per thread and slice
    static_assert(XV >= 1 && WV >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * LDS_STRIDE];   // [hi | mid][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 xr[XV];
    u32x4 whr[WV], wmr[WV];
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;          // float4 index: 8 per row
            const int row = idx >> 3, c4 = idx & 7;
            const int grow = min(m0 + row, M - 1);       // rows past M read the last row, never stored
            xr[it] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * K + k0 + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;          // 16-byte index: 4 per row
            const int row = idx >> 2, c8 = idx & 3;
            const int grow = min(n0 + row, N - 1);
            const size_t g = (size_t)grow * K + k0 + c8 * 8;
            whr[it] = *reinterpret_cast<const u32x4 *>(Whi + g);
            wmr[it] = *reinterpret_cast<const u32x4 *>(Wmid + g);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 3, c4 = idx & 7;
            bf16x4 hi, mid;   // hardware conversion (v_cvt_pk_bf16_f32, round to nearest even)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (__bf16)xr[it][e];
                mid[e] = (__bf16)(xr[it][e] - (float)hi[e]);
            }
            *reinterpret_cast<bf16x4 *>(&sA[0][row * LDS_STRIDE + c4 * 4]) = hi;
            *reinterpret_cast<bf16x4 *>(&sA[1][row * LDS_STRIDE + c4 * 4]) = mid;
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 2, c8 = idx & 3;
            *reinterpret_cast<u32x4 *>(&sB[0][row * LDS_STRIDE + c8 * 8]) = whr[it];
            *reinterpret_cast<u32x4 *>(&sB[1][row * LDS_STRIDE + c8 * 8]) = wmr[it];
        }
    };

    if constexpr (PREFETCH) load_slice(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        if constexpr (!PREFETCH) load_slice(k0);
        store_slice();
        __syncthreads();
        if constexpr (PREFETCH)
            if (k0 + BK < K) load_slice(k0 + BK);   // in flight during the MFMAs below
        // ---- 2 K-steps of 16: lane -> row (lane & 31) of the 32-row tile, 8 consecutive k from (lane >> 5) * 8
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int koff = kk + (lane >> 5) * 8;
            bf16x8 a_hi[TI], a_mid[TI], b_hi[TJ], b_mid[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int r = (wm + i * 32 + (lane & 31)) * LDS_STRIDE + koff;
                a_hi[i] = *reinterpret_cast<const bf16x8 *>(&sA[0][r]);
                a_mid[i] = *reinterpret_cast<const bf16x8 *>(&sA[1][r]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int r = (wn + j * 32 + (lane & 31)) * LDS_STRIDE + koff;
                b_hi[j] = *reinterpret_cast<const bf16x8 *>(&sB[0][r]);
                b_mid[j] = *reinterpret_cast<const bf16x8 *>(&sB[1][r]);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid[i], b_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_mid[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // ---- epilogue: C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float b = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float v = acc[i][j][r] + b;
                    if (RELU) v = v > 0.f ? v : 0.f;
                    Y[(size_t)row * N + col] = v;
                }
            }
        }
}

std::atomic<int> g_variant{-1};   // -1: TF_LINEAR_VARIANT or the default

int variant()
{
    int v = g_variant.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_LINEAR_VARIANT");
        v = e ? atoi(e) : -2;   // -2: pick per shape
        g_variant.store(v);
    }
    return v;
}

template <int BM, int BN, bool PREFETCH>
int launch_variant(const float *x, const unsigned short *wh, const unsigned short *wm, const float *bias, float *y, int M, int K,
                   int N, int relu, hipStream_t s)
{
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
    if (relu)
        hipLaunchKernelGGL((split_gemm_kernel<BM, BN, true, PREFETCH>), grid, dim3(THREADS), 0, s, x, wh, wm, bias, y, M, K, N);
    else
        hipLaunchKernelGGL((split_gemm_kernel<BM, BN, false, PREFETCH>), grid, dim3(THREADS), 0, s, x, wh, wm, bias, y, M, K, N);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

}  // namespace

namespace tfm {
int linear_set_variant(int v)
{
    const int prev = variant();
    g_variant.store(v);
    return prev;
}
}  // namespace tfm

extern "C" int tf_linear_split_f32(const float *x, const void *w_hi, const void *w_mid, const float *bias, float *y,
                                   int64_t M, int K, int N, int relu, void *stream)
{
    if (!x || !w_hi || !w_mid || !y) return TF_MSDA_ERR_NULL_POINTER;
    if (M <= 0 || K <= 0 || N <= 0 || (K % BK) != 0 || M > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_mid)) & 15)
        return TF_MSDA_ERR_BAD_DIMS;
    const unsigned short *wh = static_cast<const unsigned short *>(w_hi), *wm = static_cast<const unsigned short *>(w_mid);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // block shape / pipelining variants (TF_LINEAR_VARIANT / tf_msda_set_option("linear_variant", v)); measured in
    // profiles/r02_split_gemm_variants.txt
    int var = variant();
    if (var < 0) {
        // per-shape choice from profiles/r02_split_gemm_variants.txt (22 223 x {256 -> 256, 256 -> 384, 256 -> 1024,
        // 1024 -> 256}, 400 x 256 -> 256): few rows want many small blocks, a long K a narrow N block
        if (M <= 4096) var = 5;
        else if (K >= 512 && N <= 256) var = 4;
        else if (N > 256 && N < 512) var = 3;
        else var = 2;
    }
    switch (var) {
    case 0: return launch_variant<128, 128, false>(x, wh, wm, bias, y, (int)M, K, N, relu, s);   // round 1
    case 1: return launch_variant<128, 128, true>(x, wh, wm, bias, y, (int)M, K, N, relu, s);
    case 3: return launch_variant<64, 128, false>(x, wh, wm, bias, y, (int)M, K, N, relu, s);
    case 4: return launch_variant<128, 64, true>(x, wh, wm, bias, y, (int)M, K, N, relu, s);
    case 5: return launch_variant<64, 64, true>(x, wh, wm, bias, y, (int)M, K, N, relu, s);
    default: return launch_variant<64, 128, true>(x, wh, wm, bias, y, (int)M, K, N, relu, s);
    }
}
