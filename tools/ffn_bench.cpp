// tools/ffn_bench.cpp -- standalone (no Python, no torch) parity + timing harness for tf_ffn_fused_f32 (include/tf_fused.h):
// the feed-forward block of a transformer layer, y = LayerNorm(x + relu(x W1^T + b1) W2^T + b2), in one launch.
//
//   tools/bin/ffn_bench [M [d_ffn [TI [d_model]]]]   (built by trackformer_amd/build.py; default 22223 1024 3 256: the cfg-2
//   encoder; d_model 288: the multi-frame models, whose separate path is tf_linear_split_f32 -- K = 288 is not a multiple of 64)
//
// 1. without LayerNorm: compared BIT FOR BIT with tf_linear_packed_f32 (relu) -> tf_linear_packed_f32 -> + x
//    (the separate kernels of the default path), rows behind M checked untouched;
// 2. with LayerNorm: a sample of rows against a double-precision reference;
// 3. timing, 20 launches per HIP graph: the three launches of the default path (linear1 + ReLU, linear2, residual +
//    LayerNorm) against the one fused launch;
// 4. the same for tf_linear_res_ln_f32 (256 -> 256 projection + residual + LayerNorm; TF_LINLN_TI forces its rows per block).
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
#define TF(x)                                                                           \
    do {                                                                                \
        int rc_ = (x);                                                                  \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, tf_msda_strerror(rc_)); \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 22223, F = argc > 2 ? atoi(argv[2]) : 1024, D = argc > 4 ? atoi(argv[4]) : 256;
    if (argc > 3) tf_msda_set_option("ffn_ti", atoi(argv[3]));
    const int Tenv = getenv("TF_SPLIT_TERMS") ? atoi(getenv("TF_SPLIT_TERMS")) : 16;
    const int T = Tenv == 6 ? 6 : 16;   // the split product (include/tf_fused.h; default: fp16 pieces)
    printf("split product: %d terms\n", T);
    const int guard = 128;   // rows behind M that nothing may write
    std::mt19937 rng(11);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> X((size_t)M * D), W1((size_t)F * D), B1(F), W2((size_t)D * F), B2(D), G(D), Be(D);
    for (auto &v : X) v = nrm(rng);
    for (auto &v : W1) v = nrm(rng) * 0.0625f;
    for (auto &v : B1) v = nrm(rng) * 0.1f;
    for (auto &v : W2) v = nrm(rng) / std::sqrt((float)F);
    for (auto &v : B2) v = nrm(rng) * 0.1f;
    for (auto &v : G) v = 1.f + 0.1f * nrm(rng);
    for (auto &v : Be) v = 0.1f * nrm(rng);
    float *dX, *dW1, *dB1, *dW2, *dB2, *dG, *dBe, *dH, *dY0, *dY1, *dY;
    void *dP1, *dP2;
    auto up = [&](float **d, const std::vector<float> &h) {
        CK(hipMalloc(d, h.size() * 4));
        CK(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    };
    up(&dX, X); up(&dW1, W1); up(&dB1, B1); up(&dW2, W2); up(&dB2, B2); up(&dG, G); up(&dBe, Be);
    CK(hipMalloc(&dH, (size_t)M * F * 4));
    CK(hipMalloc(&dY0, (size_t)M * D * 4));
    CK(hipMalloc(&dY1, (size_t)M * D * 4));
    CK(hipMalloc(&dY, (size_t)(M + guard) * D * 4));
    CK(hipMalloc(&dP1, (size_t)tf_linear_packed_bytes(D, F, T)));
    CK(hipMalloc(&dP2, (size_t)tf_linear_packed_bytes(F, D, T)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    TF(tf_linear_pack_weight_f32(dW1, dP1, D, F, T, s));
    TF(tf_linear_pack_weight_f32(dW2, dP2, F, D, T, s));

    // the separate kernels: tf_linear_packed_f32 where it applies (K % 64 == 0), else tf_linear_split_f32 (same bits)
    auto split_of = [&](const std::vector<float> &W, int n_out, unsigned short **dhi, unsigned short **dmid, unsigned short **dlo, float **dsc) {
        std::vector<unsigned short> pc[3];
        for (auto &v : pc) v.resize(W.size());
        const size_t kk = W.size() / (size_t)n_out;
        std::vector<float> sc((size_t)n_out, 1.f);
        *dsc = nullptr;
        if (T == 16) {   // fp16 pieces wh, wl of w t_n + the channels' factors 16 / t_n
            for (int n = 0; n < n_out; ++n) {
                float amax = 0.f;
                for (size_t k = 0; k < kk; ++k) amax = std::max(amax, std::fabs(W[n * kk + k]));
                int e = 0;
                (void)std::frexp(amax, &e);
                const float tn = amax > 0.f ? std::ldexp(1.f, 14 - e) : 1.f;
                sc[n] = 16.f / tn;
                for (size_t k = 0; k < kk; ++k) {
                    const size_t i = n * kk + k;
                    const float ws = W[i] * tn;
                    const _Float16 h = (_Float16)ws, l = (_Float16)(ws - (float)h);
                    memcpy(&pc[0][i], &h, 2);
                    memcpy(&pc[1][i], &l, 2);
                }
            }
            CK(hipMalloc(dsc, sc.size() * 4));
            CK(hipMemcpy(*dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        } else {
            for (size_t i = 0; i < W.size(); ++i) {
                float r = W[i];
                for (int q = 0; q < 3; ++q) {   // round to nearest even, residual exact
                    unsigned u;
                    memcpy(&u, &r, 4);
                    u += 0x7FFFu + ((u >> 16) & 1u);
                    pc[q][i] = (unsigned short)(u >> 16);
                    unsigned hu = (unsigned)pc[q][i] << 16;
                    float hf;
                    memcpy(&hf, &hu, 4);
                    r -= hf;
                }
            }
        }
        unsigned short **d[3] = {dhi, dmid, dlo};
        for (int q = 0; q < 3; ++q) {
            CK(hipMalloc(d[q], W.size() * 2));
            CK(hipMemcpy(*d[q], pc[q].data(), W.size() * 2, hipMemcpyHostToDevice));
        }
        if (T != 6) *dlo = nullptr;   // fp16 pieces: no third piece
    };
    unsigned short *dW1hi, *dW1mid, *dW1lo, *dW2hi, *dW2mid, *dW2lo;
    float *dW1sc, *dW2sc, *dWosc;
    split_of(W1, F, &dW1hi, &dW1mid, &dW1lo, &dW1sc);
    split_of(W2, D, &dW2hi, &dW2mid, &dW2lo, &dW2sc);
    auto lin1 = [&]() { return (D % 64) ? tf_linear_split_f32(dX, dW1hi, dW1mid, dW1lo, dW1sc, dB1, dH, M, D, F, 1, s) : tf_linear_packed_f32(dX, dP1, dB1, nullptr, dH, M, D, F, 1, T, s); };
    auto lin2 = [&]() { return (F % 64) ? tf_linear_split_f32(dH, dW2hi, dW2mid, dW2lo, dW2sc, dB2, dY0, M, F, D, 0, s) : tf_linear_packed_f32(dH, dP2, dB2, nullptr, dY0, M, F, D, 0, T, s); };

    // ---- 1. bit identity without the LayerNorm
    TF(lin1());
    TF(lin2());
    CK(hipMemsetAsync(dY, 0xFF, (size_t)(M + guard) * D * 4, s));
    TF(tf_ffn_fused_f32(dX, dP1, dB1, dP2, dB2, dX, nullptr, nullptr, 0.f, dY, M, D, F, T, s));
    CK(hipStreamSynchronize(s));
    std::vector<float> Y0((size_t)M * D), Y((size_t)(M + guard) * D);
    CK(hipMemcpy(Y0.data(), dY0, Y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    long long differ = 0, touched = 0;
    double max_d = 0;
    for (size_t i = 0; i < Y0.size(); ++i) {
        const float ref = Y0[i] + X[i];
        differ += memcmp(&ref, &Y[i], 4) != 0;
        max_d = std::max(max_d, (double)std::fabs(ref - Y[i]));
    }
    for (size_t i = Y0.size(); i < Y.size(); ++i) {
        unsigned u;
        memcpy(&u, &Y[i], 4);
        touched += u != 0xFFFFFFFFu;
    }
    printf("tf_ffn_fused_f32 M=%d d_model=%d d_ffn=%d: outputs that differ from linear1 -> relu -> linear2 -> + x: %lld of %zu "
           "(max |d| %.3g), words written behind row M: %lld\n", M, D, F, differ, Y0.size(), max_d, touched);

    // ---- 2. with the LayerNorm, against double precision on a sample of rows
    TF(tf_ffn_fused_f32(dX, dP1, dB1, dP2, dB2, dX, dG, dBe, 1e-5f, dY, M, D, F, T, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(Y.data(), dY, (size_t)M * D * 4, hipMemcpyDeviceToHost));
    double max_err = 0;
    long long bad = 0;
    std::vector<double> h(F), o(D);
    for (int smp = 0; smp < 96; ++smp) {
        const int row = smp < 64 ? (int)(((long long)smp * M) / 64) : M - 1 - (smp - 64);
        if (row < 0 || row >= M) continue;
        for (int f = 0; f < F; ++f) {
            double a = B1[f];
            for (int k = 0; k < D; ++k) a += (double)X[(size_t)row * D + k] * W1[(size_t)f * D + k];
            h[f] = a > 0 ? a : 0;
        }
        double mean = 0, var = 0;
        for (int n = 0; n < D; ++n) {
            double a = B2[n] + X[(size_t)row * D + n];
            for (int f = 0; f < F; ++f) a += h[f] * W2[(size_t)n * F + f];
            o[n] = a;
            mean += a;
        }
        mean /= D;
        for (int n = 0; n < D; ++n) var += (o[n] - mean) * (o[n] - mean);
        var /= D;
        for (int n = 0; n < D; ++n) {
            const double ref = (o[n] - mean) / std::sqrt(var + 1e-5) * G[n] + Be[n];
            const double err = std::fabs(ref - (double)Y[(size_t)row * D + n]);
            if (!(err <= 1e-3)) ++bad;
            max_err = std::max(max_err, err);
        }
    }
    printf("  with LayerNorm: max |err| against double precision %.3g, outside 1e-3: %lld\n", max_err, bad);

    // ---- 3. timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20;
    auto time_graph = [&](auto &&body) {
        hipGraph_t graph;
        hipGraphExec_t gexec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) body();
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(gexec, s));
        CK(hipStreamSynchronize(s));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(gexec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        CK(hipGraphExecDestroy(gexec));
        CK(hipGraphDestroy(graph));
        return best * 1000.0 / iters;
    };
    const double us_sep = time_graph([&] {
        lin1();
        lin2();
        tf_add_layernorm_f32(dX, dY0, dG, dBe, dY1, M, D, 1e-5f, s);
    });
    const double us_fused = time_graph([&] { tf_ffn_fused_f32(dX, dP1, dB1, dP2, dB2, dX, dG, dBe, 1e-5f, dY, M, D, F, T, s); });
    const double us_fused_noln = time_graph([&] { tf_ffn_fused_f32(dX, dP1, dB1, dP2, dB2, dX, nullptr, nullptr, 0.f, dY, M, D, F, T, s); });
    // the output projection + residual + LayerNorm (tf_linear_res_ln_f32) against tf_linear_split_f32-class GEMM + tf_add_layernorm_f32
    void *dPo;
    float *dWo;
    std::vector<float> Wo((size_t)D * D);
    for (auto &v : Wo) v = nrm(rng) * 0.0625f;
    up(&dWo, Wo);
    CK(hipMalloc(&dPo, (size_t)tf_linear_packed_bytes(D, D, T)));
    TF(tf_linear_pack_weight_f32(dWo, dPo, D, D, T, s));
    unsigned short *dWohi, *dWomid, *dWolo;
    split_of(Wo, D, &dWohi, &dWomid, &dWolo, &dWosc);
    auto lino = [&]() { return (D % 64) ? tf_linear_split_f32(dX, dWohi, dWomid, dWolo, dWosc, dB2, dY0, M, D, D, 0, s) : tf_linear_packed_f32(dX, dPo, dB2, nullptr, dY0, M, D, D, 0, T, s); };
    TF(lino());
    CK(hipMemsetAsync(dY, 0xFF, (size_t)(M + guard) * D * 4, s));
    TF(tf_linear_res_ln_f32(dX, dPo, dB2, dX, nullptr, nullptr, 0.f, dY, M, D, D, T, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(Y0.data(), dY0, Y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    long long differ2 = 0, touched2 = 0;
    for (size_t i = 0; i < Y0.size(); ++i) {
        const float ref = Y0[i] + X[i];
        differ2 += memcmp(&ref, &Y[i], 4) != 0;
    }
    for (size_t i = Y0.size(); i < Y.size(); ++i) {
        unsigned u;
        memcpy(&u, &Y[i], 4);
        touched2 += u != 0xFFFFFFFFu;
    }
    const double us_lin_sep = time_graph([&] {
        lino();
        tf_add_layernorm_f32(dX, dY0, dG, dBe, dY1, M, D, 1e-5f, s);
    });
    const double us_lin_fused = time_graph([&] { tf_linear_res_ln_f32(dX, dPo, dB2, dX, dG, dBe, 1e-5f, dY, M, D, D, T, s); });
    printf("tf_linear_res_ln_f32 M=%d: outputs that differ from tf_linear_packed_f32 + x: %lld, words written behind row M: %lld\n"
           "  separate (packed linear, residual + LayerNorm): %.2f us;  fused: %.2f us = %.0f GB/s of x + residual + y\n",
           M, differ2, touched2, us_lin_sep, us_lin_fused, 3.0 * M * D * 4 / us_lin_fused * 1e-3);
    differ += differ2;
    touched += touched2;
    const double flop = 2.0 * 2.0 * M * D * F;   // fp32-equivalent; the three-term product issues 3x that in bf16
    printf("  separate (linear1 + ReLU, linear2, residual + LayerNorm): %.2f us;  fused: %.2f us (without LayerNorm %.2f us)\n"
           "  fused: %.1f TFLOP/s fp32-equivalent = %.1f TFLOP/s bf16 issued (dense bf16 MFMA peak ~2500)\n",
           us_sep, us_fused, us_fused_noln, flop / us_fused * 1e-6, 3.0 * flop / us_fused * 1e-6);
    return (differ || touched || bad) ? 1 : 0;
}
