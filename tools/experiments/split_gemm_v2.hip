// tools/experiments/split_gemm_v2.hip -- EXPERIMENT, NOT YET RUN ON HARDWARE (written at the end of round 1, when the
// GPU budget was spent).  Candidate second version of tf_linear_split_f32 (trackformer_amd/csrc/linear_split.hip,
// the verified first version: 25.9 / 69.7 / 68.2 us for 22 223 x 256 -> 256 / 256 -> 1024 / 1024 -> 256).
// What it changes, and why (v1 facts from its ISA and timings):
//   * 128 x 256 output block per 512-thread workgroup (8 waves of 64 x 64): for the 256-wide projections X is read and
//     split ONCE (v1: two 128-column blocks read and convert it twice);
//   * K-slices of 64 instead of 32: half the barriers;
//   * the global loads of slice s + 1 are issued into registers BEFORE the MFMAs of slice s and converted / stored to LDS
//     after them: HBM latency overlaps the matrix work inside the workgroup (v1 relies on other workgroups for that, at
//     3 workgroups per CU).
// Budget of the 256 -> 256 linear: 7.5 us of HBM, 3.8 us of bf16 MFMA (three passes): the target is ~10 us.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/experiments/split_gemm_v2.hip \
//         -Ltrackformer_amd/lib -ltf_msda -Wl,-rpath,$PWD/trackformer_amd/lib -o tools/bin/split_gemm_v2
//   tools/bin/split_gemm_v2 [M K N]   -> error of v2 vs float64, us per launch of v2 and of the library's v1
// If it is correct and faster: move the kernel into linear_split.hip behind tf_linear_split_f32.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "tf_fused.h"
#include "tf_msda.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BM = 128, BN = 256, BK = 64, THREADS = 512;
constexpr int LDS_STRIDE = BK + 8;                       // bf16 elements per LDS row: 144 bytes (16-byte aligned rows)
constexpr int A_ELEMS = BM * LDS_STRIDE, B_ELEMS = BN * LDS_STRIDE;
constexpr size_t LDS_BYTES = (size_t)(2 * A_ELEMS + 2 * B_ELEMS) * 2;   // 110 592 B: one workgroup per CU

template <bool RELU>
__global__ void __launch_bounds__(THREADS)
split_gemm_v2_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                     const unsigned short *__restrict__ Wmid, const float *__restrict__ bias, float *__restrict__ Y,
                     int M, int K, int N)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short *sA_hi = lds, *sA_mid = lds + A_ELEMS, *sB_hi = lds + 2 * A_ELEMS, *sB_mid = lds + 2 * A_ELEMS + B_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;   // this wave's 64 x 64 corner inside the block

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-thread pieces of a K-slice: 4 float4 of X (128 rows x 16 float4), 4 x 16 bytes of each weight piece
    // (256 rows x 8 x 16 bytes)
    f32x4 px[4];
    u32x4 pwh[4], pwm[4];
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 4, c4 = idx & 15;
            const int grow = min(m0 + row, M - 1);       // rows past M read the last row, never stored
            px[it] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * K + k0 + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 3, c8 = idx & 7;
            const int grow = min(n0 + row, N - 1);
            const size_t g = (size_t)grow * K + k0 + c8 * 8;
            pwh[it] = *reinterpret_cast<const u32x4 *>(Whi + g);
            pwm[it] = *reinterpret_cast<const u32x4 *>(Wmid + g);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 4, c4 = idx & 15;
            bf16x4 hi, mid;   // v_cvt_pk_bf16_f32, round to nearest even
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (__bf16)px[it][e];
                mid[e] = (__bf16)(px[it][e] - (float)hi[e]);
            }
            *reinterpret_cast<bf16x4 *>(&sA_hi[row * LDS_STRIDE + c4 * 4]) = hi;
            *reinterpret_cast<bf16x4 *>(&sA_mid[row * LDS_STRIDE + c4 * 4]) = mid;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * THREADS + tid;
            const int row = idx >> 3, c8 = idx & 7;
            *reinterpret_cast<u32x4 *>(&sB_hi[row * LDS_STRIDE + c8 * 8]) = pwh[it];
            *reinterpret_cast<u32x4 *>(&sB_mid[row * LDS_STRIDE + c8 * 8]) = pwm[it];
        }
    };

    load_slice(0);
    store_slice();
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;   // uniform
        if (more) load_slice(k0 + BK);   // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int koff = kk + (lane >> 5) * 8;   // the same k goes into the same slot of A and B (see v1)
            bf16x8 a_hi[2], a_mid[2], b_hi[2], b_mid[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (wm + i * 32 + (lane & 31)) * LDS_STRIDE + koff;
                a_hi[i] = *reinterpret_cast<const bf16x8 *>(&sA_hi[r]);
                a_mid[i] = *reinterpret_cast<const bf16x8 *>(&sA_mid[r]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (wn + j * 32 + (lane & 31)) * LDS_STRIDE + koff;
                b_hi[j] = *reinterpret_cast<const bf16x8 *>(&sB_hi[r]);
                b_mid[j] = *reinterpret_cast<const bf16x8 *>(&sB_mid[r]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid[i], b_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_mid[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                 // every wave is done reading this slice
        if (more) {
            store_slice();
            __syncthreads();
        }
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

// fp32 -> bf16 bits, round to nearest even (finite inputs): how the caller splits a weight
static unsigned short bf16_rne(float x)
{
    unsigned int u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_f32(unsigned short h)
{
    const unsigned int u = (unsigned int)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char **argv)
{
    const int M = argc > 3 ? atoi(argv[1]) : 22223, K = argc > 3 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    if (K % BK) {
        fprintf(stderr, "K must be a multiple of %d\n", BK);
        return 2;
    }
    std::mt19937 rng(7);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> X((size_t)M * K), W((size_t)N * K), bias(N), Y((size_t)M * N);
    for (auto &v : X) v = nrm(rng);
    for (auto &v : W) v = nrm(rng) * 0.0625f;   // ~ 1 / sqrt(K): activations stay O(1), as in the model
    for (auto &v : bias) v = nrm(rng);
    std::vector<unsigned short> Whi(W.size()), Wmid(W.size());
    for (size_t i = 0; i < W.size(); ++i) {
        Whi[i] = bf16_rne(W[i]);
        Wmid[i] = bf16_rne(W[i] - bf16_to_f32(Whi[i]));
    }
    float *dX, *dB, *dY;
    unsigned short *dWhi, *dWmid;
    CK(hipMalloc(&dX, X.size() * 4));
    CK(hipMalloc(&dB, bias.size() * 4));
    CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMalloc(&dWhi, W.size() * 2));
    CK(hipMalloc(&dWmid, W.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWhi, Whi.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWmid, Wmid.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0xFF, Y.size() * 4));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    CK(hipFuncSetAttribute((const void *)&split_gemm_v2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    auto run = [&]() {
        hipLaunchKernelGGL(split_gemm_v2_kernel<false>, grid, dim3(THREADS), LDS_BYTES, stream, dX, dWhi, dWmid, dB, dY, M, K, N);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    };
    int rc = run();
    if (rc != 0) {
        fprintf(stderr, "tf_linear_split_f32 failed: %s\n", tf_msda_strerror(rc));
        return 2;
    }
    CK(hipStreamSynchronize(stream));
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    // ---- check against a double-precision reference on a sample of rows (all columns), incl. the block edges
    double max_err = 0.0, max_ref = 0.0;
    long long bad = 0, checked = 0;
    for (int s = 0; s < 512; ++s) {
        const int row = s < 256 ? (int)(((long long)s * M) / 256) : M - 1 - (s - 256);
        if (row < 0 || row >= M) continue;
        for (int n = 0; n < N; ++n) {
            double ref = bias[n];
            for (int k = 0; k < K; ++k) ref += (double)X[(size_t)row * K + k] * (double)W[(size_t)n * K + k];
            const double err = std::fabs(ref - (double)Y[(size_t)row * N + n]);
            if (!(err <= 1e-3)) ++bad;   // catches NaN too
            max_err = std::max(max_err, err);
            max_ref = std::max(max_ref, std::fabs(ref));
            ++checked;
        }
    }
    printf("split_gemm_v2 M=%d K=%d N=%d: checked %lld outputs, max |err| %.3g (max |ref| %.3g), outside 1e-3: %lld\n", M, K, N,
           checked, max_err, max_ref, bad);
    // ---- timing: 20 launches in one graph
    hipGraph_t graph;
    hipGraphExec_t gexec;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20;
    CK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) run();
    CK(hipStreamEndCapture(stream, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(gexec, stream));
    CK(hipStreamSynchronize(stream));
    CK(hipEventRecord(e0, stream));
    CK(hipGraphLaunch(gexec, stream));
    CK(hipEventRecord(e1, stream));
    CK(hipStreamSynchronize(stream));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters, flop = 2.0 * M * K * N;
    printf("  %.2f us per launch = %.1f TFLOP/s fp32-equivalent (fp32 MFMA peak 157; hipBLASLt fp32 on this shape: see DESIGN.md), "
           "%.1f GB/s of X + Y\n", us, flop / us * 1e-6, ((double)M * K + (double)M * N) * 4 / us * 1e-3);
    // ---- the library's first version on the same data, for comparison
    CK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) tf_linear_split_f32(dX, dWhi, dWmid, dB, dY, M, K, N, 0, stream);
    CK(hipStreamEndCapture(stream, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(gexec, stream));
    CK(hipStreamSynchronize(stream));
    CK(hipEventRecord(e0, stream));
    CK(hipGraphLaunch(gexec, stream));
    CK(hipEventRecord(e1, stream));
    CK(hipStreamSynchronize(stream));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  library v1 (tf_linear_split_f32): %.2f us per launch\n", ms * 1000.0 / iters);
    return bad ? 1 : 0;
}
