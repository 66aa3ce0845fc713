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

#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BM = 128, BN = 128, BK = 32, THREADS = 256;
constexpr int LDS_STRIDE = BK + 8;   // bf16 elements per LDS row: 80 bytes, keeps 16-byte alignment, spreads banks

template <bool RELU>
__global__ void __launch_bounds__(THREADS)
split_gemm_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                  const unsigned short *__restrict__ Wmid, const float *__restrict__ bias, float *__restrict__ Y,
                  int M, int K, int N)
{
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][BM * LDS_STRIDE];   // [hi | mid][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][BN * LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;   // this wave's 64 x 64 corner inside the block

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // ---- X tile: 128 rows x 32 floats = 1024 float4; split into bf16 hi / mid on the way to LDS
#pragma unroll
        for (int it = 0; it < (BM * BK / 4) / THREADS; ++it) {
            const int idx = it * THREADS + tid;          // float4 index
            const int row = idx >> 3, c4 = idx & 7;      // 8 float4 per row
            const int grow = min(m0 + row, M - 1);       // rows past M read the last row, never stored
            const f32x4 v = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * K + k0 + c4 * 4);
            bf16x4 hi, mid;   // hardware conversion (v_cvt_pk_bf16_f32, round to nearest even)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (__bf16)v[e];
                mid[e] = (__bf16)(v[e] - (float)hi[e]);
            }
            *reinterpret_cast<bf16x4 *>(&sA[0][row * LDS_STRIDE + c4 * 4]) = hi;
            *reinterpret_cast<bf16x4 *>(&sA[1][row * LDS_STRIDE + c4 * 4]) = mid;
        }
        // ---- W tiles: 128 rows x 32 bf16 = 512 x 16 bytes per piece
#pragma unroll
        for (int it = 0; it < (BN * BK / 8) / THREADS; ++it) {
            const int idx = it * THREADS + tid;          // 16-byte index
            const int row = idx >> 2, c8 = idx & 3;      // 4 x 16 bytes per row
            const int grow = min(n0 + row, N - 1);
            const size_t g = (size_t)grow * K + k0 + c8 * 8;
            *reinterpret_cast<u32x4 *>(&sB[0][row * LDS_STRIDE + c8 * 8]) = *reinterpret_cast<const u32x4 *>(Whi + g);
            *reinterpret_cast<u32x4 *>(&sB[1][row * LDS_STRIDE + c8 * 8]) = *reinterpret_cast<const u32x4 *>(Wmid + g);
        }
        __syncthreads();
        // ---- 2 K-steps of 16: lane -> row (lane & 31) of the 32-row tile, 8 consecutive k from (lane >> 5) * 8
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int koff = kk + (lane >> 5) * 8;
            bf16x8 a_hi[2], a_mid[2], b_hi[2], b_mid[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (wm + i * 32 + (lane & 31)) * LDS_STRIDE + koff;
                a_hi[i] = *reinterpret_cast<const bf16x8 *>(&sA[0][r]);
                a_mid[i] = *reinterpret_cast<const bf16x8 *>(&sA[1][r]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (wn + j * 32 + (lane & 31)) * LDS_STRIDE + koff;
                b_hi[j] = *reinterpret_cast<const bf16x8 *>(&sB[0][r]);
                b_mid[j] = *reinterpret_cast<const bf16x8 *>(&sB[1][r]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
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
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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


}  // namespace

extern "C" int tf_linear_split_f32(const float *x, const void *w_hi, const void *w_mid, const float *bias, float *y,
                                   int64_t M, int K, int N, int relu, void *stream)
{
    if (!x || !w_hi || !w_mid || !y) return TF_MSDA_ERR_NULL_POINTER;
    if (M <= 0 || K <= 0 || N <= 0 || (K % BK) != 0 || M > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_mid)) & 15)
        return TF_MSDA_ERR_BAD_DIMS;
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
    const unsigned short *wh = static_cast<const unsigned short *>(w_hi), *wm = static_cast<const unsigned short *>(w_mid);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu)
        hipLaunchKernelGGL(split_gemm_kernel<true>, grid, dim3(THREADS), 0, s, x, wh, wm, bias, y, (int)M, K, N);
    else
        hipLaunchKernelGGL(split_gemm_kernel<false>, grid, dim3(THREADS), 0, s, x, wh, wm, bias, y, (int)M, K, N);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}
