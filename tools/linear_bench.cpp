// tools/linear_bench.cpp -- standalone (no Python, no torch) parity + timing harness for tf_linear_split_f32
// (include/tf_fused.h): Y[M, N] = X[M, K] . W[N, K]^T + bias as a bf16 split product on the matrix cores.
//
//   tools/bin/linear_bench [M K N [packed | packedTI]]   (built by trackformer_amd/build.py; default 22223 256 256)
//     packed / packed2 / packed3 / packed4: tf_linear_packed_f32 (weight packed once by tf_linear_pack_weight_f32; the digit
//     forces the row tiles per block), whose output is also compared BIT FOR BIT with tf_linear_split_f32's
//   TF_SPLIT_TERMS=6: six bf16 terms (default: 16, fp16 pieces; the three-term bf16 product of rounds 2-4 is gone)
//
// Checks a sample of output rows (all columns, incl. the block edges) against a double-precision reference and
// times 20 launches captured in one HIP graph.  Round-1 numbers: profiles/r01_split_gemm_experiment.txt.
#include <hip/hip_runtime.h>

#include <algorithm>
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
    const bool packed = argc > 4 && strncmp(argv[4], "packed", 6) == 0;
    if (packed && argv[4][6]) tf_msda_set_option("linear_stream_ti", atoi(argv[4] + 6));
    const int Tenv = getenv("TF_SPLIT_TERMS") ? atoi(getenv("TF_SPLIT_TERMS")) : 16;
    const int T = Tenv == 6 ? 6 : 16;   // the split product (include/tf_fused.h; default: fp16 pieces)
    if (K % 32) {
        fprintf(stderr, "K must be a multiple of 32\n");
        return 2;
    }
    std::mt19937 rng(7);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> X((size_t)M * K), W((size_t)N * K), bias(N), Y((size_t)M * N);
    for (auto &v : X) v = nrm(rng);
    for (auto &v : W) v = nrm(rng) * 0.0625f;   // ~ 1 / sqrt(K): activations stay O(1), as in the model
    for (auto &v : bias) v = nrm(rng);
    std::vector<unsigned short> Whi(W.size()), Wmid(W.size()), Wlo(W.size());
    std::vector<float> Wsc(N, 1.f);
    if (T == 16) {   // fp16 pieces wh, wl of w t_n + the channels' factors 16 / t_n
        for (int n = 0; n < N; ++n) {
            float amax = 0.f;
            for (int k = 0; k < K; ++k) amax = std::max(amax, std::fabs(W[(size_t)n * K + k]));
            int e = 0;
            (void)std::frexp(amax, &e);   // amax = m 2^e, m in [0.5, 1)
            const float tn = amax > 0.f ? std::ldexp(1.f, 14 - e) : 1.f;
            Wsc[n] = 16.f / tn;
            for (int k = 0; k < K; ++k) {
                const size_t i = (size_t)n * K + k;
                const float ws = W[i] * tn;
                const _Float16 h = (_Float16)ws, l = (_Float16)(ws - (float)h);
                memcpy(&Whi[i], &h, 2);
                memcpy(&Wmid[i], &l, 2);
            }
        }
    } else {
        for (size_t i = 0; i < W.size(); ++i) {
            Whi[i] = bf16_rne(W[i]);
            const float r = W[i] - bf16_to_f32(Whi[i]);
            Wmid[i] = bf16_rne(r);
            Wlo[i] = bf16_rne(r - bf16_to_f32(Wmid[i]));
        }
    }
    float *dX, *dB, *dY, *dWsc = nullptr;
    unsigned short *dWhi, *dWmid, *dWlo = nullptr;
    CK(hipMalloc(&dX, X.size() * 4));
    CK(hipMalloc(&dB, bias.size() * 4));
    CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMalloc(&dWhi, W.size() * 2));
    CK(hipMalloc(&dWmid, W.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWhi, Whi.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWmid, Wmid.data(), W.size() * 2, hipMemcpyHostToDevice));
    if (T == 6) {
        CK(hipMalloc(&dWlo, W.size() * 2));
        CK(hipMemcpy(dWlo, Wlo.data(), W.size() * 2, hipMemcpyHostToDevice));
    }
    if (T == 16) {
        CK(hipMalloc(&dWsc, Wsc.size() * 4));
        CK(hipMemcpy(dWsc, Wsc.data(), Wsc.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(dY, 0xFF, Y.size() * 4));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    void *dWp = nullptr;
    float *dW = nullptr;
    long long not_identical = -1;
    if (packed) {
        const int64_t bytes = tf_linear_packed_bytes(K, N, T);
        if (bytes <= 0 || (K % 64)) {
            fprintf(stderr, "packed: K must be a multiple of 64\n");
            return 2;
        }
        CK(hipMalloc(&dWp, (size_t)bytes));
        CK(hipMalloc(&dW, W.size() * 4));
        CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
        int prc = tf_linear_pack_weight_f32(dW, dWp, K, N, T, stream);
        if (prc != 0) {
            fprintf(stderr, "tf_linear_pack_weight_f32 failed: %s\n", tf_msda_strerror(prc));
            return 2;
        }
        // the unpacked kernel's output first: the packed one must reproduce it bit for bit
        prc = tf_linear_split_f32(dX, dWhi, dWmid, dWlo, dWsc, dB, dY, M, K, N, 0, stream);
        if (prc != 0) return 2;
        CK(hipStreamSynchronize(stream));
        std::vector<float> Y0(Y.size());
        CK(hipMemcpy(Y0.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dY, 0xFF, Y.size() * 4));
        prc = tf_linear_packed_f32(dX, dWp, dB, nullptr, dY, M, K, N, 0, T, stream);
        if (prc != 0) {
            fprintf(stderr, "tf_linear_packed_f32 failed: %s\n", tf_msda_strerror(prc));
            return 2;
        }
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
        not_identical = 0;
        for (size_t i = 0; i < Y.size(); ++i) not_identical += memcmp(&Y[i], &Y0[i], 4) != 0;
        CK(hipMemset(dY, 0xFF, Y.size() * 4));
    }
    auto run = [&]() {
        return packed ? tf_linear_packed_f32(dX, dWp, dB, nullptr, dY, M, K, N, 0, T, stream)
                      : tf_linear_split_f32(dX, dWhi, dWmid, dWlo, dWsc, dB, dY, M, K, N, 0, stream);
    };
    int rc = run();
    if (rc != 0) {
        fprintf(stderr, "%s failed: %s\n", packed ? "tf_linear_packed_f32" : "tf_linear_split_f32", tf_msda_strerror(rc));
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
    printf("%s (%d terms) %s M=%d K=%d N=%d: checked %lld outputs, max |err| %.3g (max |ref| %.3g), outside 1e-3: %lld\n",
           packed ? "tf_linear_packed_f32" : "tf_linear_split_f32", T, argc > 4 ? argv[4] : "default", M, K, N, checked, max_err,
           max_ref, bad);
    if (packed) {
        printf("  outputs that differ from tf_linear_split_f32's bit pattern: %lld of %zu\n", not_identical, Y.size());
        if (not_identical) bad += not_identical;
    }
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
    return bad ? 1 : 0;
}
