// tools/msda_bench.cpp -- standalone (no Python, no torch) parity + timing harness for the forward kernels
// of libtf_msda.so at the BASELINE encoder call shape.  Starts in milliseconds on a fresh GPU box, which
// matters when GPU minutes are scarce.
//
//   hipcc -O2 -std=c++17 -Iinclude tools/msda_bench.cpp -Ltrackformer_amd/lib -ltf_msda \
//         -Wl,-rpath,'$ORIGIN/../../trackformer_amd/lib' -o tools/bin/msda_bench        (trackformer_amd/build.py does this)
//   tools/bin/msda_bench [--iters 20] [--patterns init,local,uniform] [--fused 1] [--n 1] \
//                        direct win quad:ta=12,waves=8,npass=1,lds=53 quad:ta=0,...
//
// For every sampling pattern the first configuration ("direct", the parity-tested default kernel) is the
// reference; every other configuration is compared with it element-wise (max |diff|, number of
// (query, head) pairs off by more than 1e-4, where they are) and timed: `iters` launches captured in one
// HIP graph, HIP events around the replay.  Patterns as in tools/bench_msda.py: init = what a
// default-initialised MSDeformAttn produces, local = reference point + N(0, 2 px), uniform = rand; pert = the bias
// grid + N(0, 0.8) raw offsets (what the perturbed-weight model of the parity tests produces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "tf_msda.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

static const int kH[4] = {100, 50, 25, 13}, kW[4] = {167, 84, 42, 21};
static const int M = 8, D = 32, L = 4, P = 4;

struct Inputs {
    int N, S, Lq;
    std::vector<float> value, loc, attn, qproj, ref;   // qproj [N*Lq, 3*M*L*P]: raw offsets | logits
};

static Inputs make_inputs(const std::string &mode, int N, unsigned seed)
{
    Inputs in;
    in.N = N;
    in.S = 0;
    for (int l = 0; l < L; ++l) in.S += kH[l] * kW[l];
    in.Lq = in.S;
    const int S = in.S, LP = L * P;
    std::mt19937 rng(seed);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    in.value.resize((size_t)N * S * M * D);
    for (auto &v : in.value) v = nrm(rng);
    in.loc.resize((size_t)N * S * M * LP * 2);
    in.attn.resize((size_t)N * S * M * LP);
    in.qproj.resize((size_t)N * S * 3 * M * LP);
    in.ref.resize((size_t)N * S * L * 2);
    static const int dirs[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
    for (int n = 0; n < N; ++n) {
        int q = 0;
        for (int lq = 0; lq < L; ++lq)
            for (int y = 0; y < kH[lq]; ++y)
                for (int x = 0; x < kW[lq]; ++x, ++q) {
                    const float rx = (x + 0.5f) / kW[lq], ry = (y + 0.5f) / kH[lq];
                    const size_t bq = (size_t)n * S + q;
                    for (int l = 0; l < L; ++l) {
                        in.ref[(bq * L + l) * 2 + 0] = rx;
                        in.ref[(bq * L + l) * 2 + 1] = ry;
                    }
                    float *qrow = &in.qproj[bq * 3 * M * LP];
                    for (int m = 0; m < M; ++m) {
                        float logits[16], mx = -1e30f, sum = 0.f;
                        for (int s = 0; s < LP; ++s) {
                            logits[s] = nrm(rng);
                            mx = std::max(mx, logits[s]);
                        }
                        for (int s = 0; s < LP; ++s) sum += std::exp(logits[s] - mx);
                        for (int l = 0; l < L; ++l)
                            for (int p = 0; p < P; ++p) {
                                const int s = l * P + p;
                                float ox, oy;   // raw offsets (what the sampling_offsets Linear outputs)
                                if (mode == "init") {
                                    ox = (float)dirs[m][0] * (p + 1);
                                    oy = (float)dirs[m][1] * (p + 1);
                                } else if (mode == "pert") {   // what tests/util_weights.perturb_state_dict makes of the
                                    // model: the bias grid + a data-dependent part, N(0, 0.05 * sqrt(256)) = N(0, 0.8)
                                    ox = (float)dirs[m][0] * (p + 1) + 0.8f * nrm(rng);
                                    oy = (float)dirs[m][1] * (p + 1) + 0.8f * nrm(rng);
                                } else if (mode == "local") {   // N(0, 2 px) in pixels of the sampled level
                                    ox = nrm(rng) * 2.f * kH[l] / kW[l];   // undo the (H, W) divisor quirk
                                    oy = nrm(rng) * 2.f * kW[l] / kH[l];
                                } else {                         // uniform over the level
                                    ox = (uni(rng) - rx) * kH[l];
                                    oy = (uni(rng) - ry) * kW[l];
                                }
                                qrow[(m * LP + s) * 2 + 0] = ox;
                                qrow[(m * LP + s) * 2 + 1] = oy;
                                qrow[2 * M * LP + m * LP + s] = logits[s];
                                const size_t pi = (bq * M + m) * LP + s;
                                // ms_deform_attn.py:78-79: x offset / H_l, y offset / W_l (as written)
                                in.loc[pi * 2 + 0] = rx + ox / (float)kH[l];
                                in.loc[pi * 2 + 1] = ry + oy / (float)kW[l];
                                in.attn[pi] = std::exp(logits[s] - mx) / sum;
                            }
                    }
                }
    }
    return in;
}

struct Config {
    std::string name;
    int tiled;                                            // 0 direct, 1 win, 2 quad, 3 pquad (tiled 2 + pquad on)
    std::vector<std::pair<std::string, int>> opts;        // quad_* options
};

static Config parse_config(const std::string &arg)
{
    Config c;
    c.name = arg;
    const std::string kind = arg.substr(0, arg.find(':'));
    c.tiled = kind == "direct" ? 0 : kind == "quad" ? 2 : kind == "pquad" ? 3 : -9;
    if (c.tiled == -9) {
        fprintf(stderr, "unknown configuration '%s'\n", arg.c_str());
        exit(2);
    }
    static const char *qkeys[][2] = {{"ta", "quad_ta_mask"}, {"waves", "quad_waves"}, {"npass", "quad_npass"},
                                     {"lds", "quad_lds_kb"}, {"hy", "quad_halo_y"}, {"hx", "quad_halo_x"},
                                     {"th", "quad_tile_h"},  {"tw", "quad_tile_w"},  {"split", "quad_split"}};
    static const char *pkeys[][2] = {{"wide", "pquad_wide"}, {"npass", "pquad_npass"}, {"lds", "pquad_lds_kb"},
                                     {"hy", "pquad_halo_y"}, {"hx", "pquad_halo_x"}, {"th", "pquad_tile_h"},
                                     {"tw", "pquad_tile_w"}, {"wgs", "pquad_wg_per_cu"}, {"pf", "pquad_prefetch"},
                                     {"skew", "pquad_skew"}, {"v2", "pquad_v2"}, {"waves", "pquad_waves"},
                                     {"hm", "pquad_headmix"}, {"prio", "pquad_prio"}, {"st", "pquad_store"}, {"ldnt", "pquad_ldnt"}, {"ti", "pquad_trace_iter"}, {"cf", "pquad_cf"}};
    const size_t nkeys = c.tiled == 3 ? 18 : 9;
    const char *(*keys)[2] = c.tiled == 3 ? pkeys : qkeys;
    size_t pos = arg.find(':');
    while (pos != std::string::npos && pos + 1 < arg.size()) {
        const size_t next = arg.find(',', pos + 1);
        const std::string kv = arg.substr(pos + 1, next == std::string::npos ? std::string::npos : next - pos - 1);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos)
            for (size_t ki = 0; ki < nkeys; ++ki)
                if (kv.substr(0, eq) == keys[ki][0]) c.opts.push_back({keys[ki][1], atoi(kv.c_str() + eq + 1)});
        pos = next;
    }
    return c;
}

static void apply(const Config &c)
{
    static const char *names[] = {"quad_ta_mask", "quad_waves",  "quad_npass",  "quad_lds_kb", "quad_halo_y",
                                  "quad_halo_x",  "quad_tile_h", "quad_tile_w", "quad_split"};
    static const int defaults[] = {0, 4, 3, 40, 6, 10, 0, 0, 1};   // = kQuadOptDefaults of the library
    static const char *pnames[] = {"pquad_wide", "pquad_npass", "pquad_lds_kb", "pquad_halo_y", "pquad_halo_x",
                                   "pquad_tile_h",  "pquad_tile_w", "pquad_wg_per_cu", "pquad_prefetch", "pquad_skew", "pquad_v2", "pquad_waves",
                                   "pquad_headmix", "pquad_prio", "pquad_store", "pquad_ldnt", "pquad_trace_iter", "pquad_cf"};
    static const int pdefaults[] = {1, 2, 52, 6, 10, 0, 0, 3, 0, 0, 1, 4, 0, 0, 1, 0, 0, 0};   // = kPqOptDefaults of the library
    for (int i = 0; i < 9; ++i) tf_msda_set_option(names[i], defaults[i]);
    for (int i = 0; i < 18; ++i) tf_msda_set_option(pnames[i], pdefaults[i]);
    for (auto &o : c.opts) tf_msda_set_option(o.first.c_str(), o.second);
    tf_msda_set_option("pquad", c.tiled == 3 ? 1 : 0);
    tf_msda_set_option("tiled", c.tiled == 3 ? 2 : c.tiled);
}

int main(int argc, char **argv)
{
    int iters = 20, N = 1, fused = 1, trace = 0, sets = 1;
    std::string patterns = "init,local,uniform", trace_dump;
    std::vector<Config> cfgs;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fused") && i + 1 < argc) fused = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--trace")) trace = 1;
        else if (!strcmp(argv[i], "--trace-dump") && i + 1 < argc) {   // raw stamps of every workgroup, one CSV row each
            trace = 1;
            trace_dump = argv[++i];
        }
        else if (!strcmp(argv[i], "--sets") && i + 1 < argc) sets = std::max(1, atoi(argv[++i]));   // rotate over K copies of
        // the tensors (K x 80..114 MB): with K >= 4 the working set exceeds the 256 MiB Infinity Cache, every launch reads HBM
        else if (!strcmp(argv[i], "--patterns") && i + 1 < argc) patterns = argv[++i];
        else cfgs.push_back(parse_config(argv[i]));
    }
    if (cfgs.empty() || cfgs[0].tiled != 0) cfgs.insert(cfgs.begin(), parse_config("direct"));

    int64_t shapes[8];
    for (int l = 0; l < L; ++l) {
        shapes[2 * l] = kH[l];
        shapes[2 * l + 1] = kW[l];
    }
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    size_t start = 0;
    while (start <= patterns.size()) {
        const size_t comma = patterns.find(',', start);
        const std::string mode = patterns.substr(start, comma == std::string::npos ? std::string::npos : comma - start);
        start = comma == std::string::npos ? patterns.size() + 1 : comma + 1;
        if (mode.empty()) continue;
        const Inputs in = make_inputs(mode, N, 1234);
        const int S = in.S, Lq = in.Lq, LP = L * P;
        const size_t n_out = (size_t)N * Lq * M * D;
        const double alg_bytes = 4.0 * ((double)N * S * M * D + 3.0 * N * Lq * M * LP + (double)n_out);
        float *d_value, *d_loc, *d_attn, *d_qproj, *d_ref, *d_out;
        // `sets` copies of every tensor, back to back; set 0 is the one checked, the timed graph rotates over all
        const size_t n_value = in.value.size(), n_loc = in.loc.size(), n_attn = in.attn.size(), n_qproj = in.qproj.size();
        CK(hipMalloc(&d_value, n_value * 4 * sets));
        CK(hipMalloc(&d_loc, n_loc * 4 * sets));
        CK(hipMalloc(&d_attn, n_attn * 4 * sets));
        CK(hipMalloc(&d_qproj, n_qproj * 4 * sets));
        CK(hipMalloc(&d_ref, in.ref.size() * 4));
        CK(hipMalloc(&d_out, n_out * 4 * sets));
        for (int k = 0; k < sets; ++k) {
            CK(hipMemcpy(d_value + k * n_value, in.value.data(), n_value * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_loc + k * n_loc, in.loc.data(), n_loc * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_attn + k * n_attn, in.attn.data(), n_attn * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_qproj + k * n_qproj, in.qproj.data(), n_qproj * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(d_ref, in.ref.data(), in.ref.size() * 4, hipMemcpyHostToDevice));

        for (int fz = 0; fz <= (fused ? 1 : 0); ++fz) {
            std::vector<float> ref_out, out(n_out);
            for (size_t ci = 0; ci < cfgs.size(); ++ci) {
                const Config &c = cfgs[ci];
                apply(c);
                auto run = [&](int k = 0) {
                    return fz ? tf_msda_forward_fused_f32(d_value + k * n_value, shapes, d_ref, 2, d_qproj + k * n_qproj,
                                                          3 * M * LP, 0, 2 * M * LP, d_out + k * n_out, N, S, M, D, L, Lq, P,
                                                          stream)
                              : tf_msda_forward_f32(d_value + k * n_value, shapes, d_loc + k * n_loc, d_attn + k * n_attn,
                                                    d_out + k * n_out, N, S, M, D, L, Lq, P, stream);
                };
                CK(hipMemsetAsync(d_out, 0xFF, n_out * 4, stream));   // NaN pattern: unwritten outputs show up
                int rc = run();
                if (rc != 0) {
                    printf("%-8s %-5s %-40s  launch failed: %s (hip %d)\n", mode.c_str(), fz ? "fused" : "plain",
                           c.name.c_str(), tf_msda_strerror(rc), tf_msda_last_hip_error());
                    continue;
                }
                const hipError_t se = hipStreamSynchronize(stream);
                if (se != hipSuccess) {
                    printf("%-8s %-5s %-40s  kernel failed: %s\n", mode.c_str(), fz ? "fused" : "plain", c.name.c_str(),
                           hipGetErrorString(se));
                    return 3;
                }
                CK(hipMemcpy(out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
                double maxd = 0.0;
                long long bad = 0, nanc = 0;
                long long bad_lvl[4] = {0, 0, 0, 0}, bad_head[8] = {0};
                std::vector<long long> first_bad;
                if (ci == 0) {
                    ref_out = out;
                } else {
                    int lstart[5] = {0};
                    for (int l = 0; l < L; ++l) lstart[l + 1] = lstart[l] + kH[l] * kW[l];
                    for (size_t pr = 0; pr < (size_t)N * Lq * M; ++pr) {
                        double d = 0.0;
                        bool isn = false;
                        for (int ch = 0; ch < D; ++ch) {
                            const float a = out[pr * D + ch], b = ref_out[pr * D + ch];
                            if (std::isnan(a)) isn = true;
                            d = std::max(d, (double)std::fabs(a - b));
                        }
                        if (isn) ++nanc;
                        if (!isn) maxd = std::max(maxd, d);
                        if (isn || d > 1e-4) {
                            ++bad;
                            const int m = (int)(pr % M);
                            const int q = (int)((pr / M) % Lq);
                            int l = 0;
                            while (l < L - 1 && q >= lstart[l + 1]) ++l;
                            ++bad_lvl[l];
                            ++bad_head[m];
                            if (first_bad.size() < 12) first_bad.push_back((long long)pr);
                        }
                    }
                }
                if (trace && c.tiled >= 2) {
                    // phase timestamps of every workgroup (wave 0): offsets from the earliest workgroup start
                    const size_t max_wg = 16384;
                    unsigned long long *d_tr;
                    CK(hipMalloc(&d_tr, max_wg * 16 * 8));
                    CK(hipMemset(d_tr, 0, max_wg * 16 * 8));
                    tf_msda_debug_trace_buffer(d_tr);
                    run();
                    CK(hipStreamSynchronize(stream));
                    tf_msda_debug_trace_buffer(nullptr);
                    std::vector<unsigned long long> tr(max_wg * 16);
                    CK(hipMemcpy(tr.data(), d_tr, max_wg * 16 * 8, hipMemcpyDeviceToHost));
                    CK(hipFree(d_tr));
                    unsigned long long t0 = ~0ull;
                    size_t nwg = 0;
                    for (size_t w = 0; w < max_wg; ++w)
                        if (tr[w * 16]) {
                            t0 = std::min(t0, tr[w * 16]);
                            nwg = w + 1;
                        }
                    static const char *qnames[14] = {"entry", "setup barrier", "points+bbox", "barrier A", "DMA r0 issued",
                                                     "load-gathers r0", "DMA r0 landed", "LDS gathers r0", "DMA r1 issued",
                                                     "load-gathers r1", "DMA r1 landed", "LDS gathers r1 + stores",
                                                     "(point loads issued)", "(level 0 bbox filed)"};
                    // msda_fwd_f32_pquad: first tile of every workgroup, then the end of its last tile
                    static const char *pnames[14] = {"entry", "tile 0 loads issued", "tile 0 points+bbox", "B0", "B1 (L0 landed)",
                                                     "(prefetch issued)", "L0 gathered", "B3 (L1-3 landed)", "stored",
                                                     "tile 1 points+bbox", "end of last tile", "B2 (L0 window free)", "DMA L1-3 issued",
                                                     "DMA L1-3 landed (own)"};
                    const char **names = c.tiled == 3 ? pnames : qnames;
                    printf("  trace of %zu workgroups (us after the first workgroup's entry; 100 MHz clock):\n", nwg);
                    if (!trace_dump.empty()) {
                        // block, then 16 stamps in 10 ns ticks after the earliest entry (0 = not stamped); appended per run
                        FILE *f = fopen(trace_dump.c_str(), "a");
                        if (f) {
                            fprintf(f, "# %s %s %s\n", mode.c_str(), fz ? "fused" : "plain", c.name.c_str());
                            for (size_t w = 0; w < nwg; ++w) {
                                fprintf(f, "%zu", w);
                                for (int i = 0; i < 16; ++i)
                                    fprintf(f, ",%lld", tr[w * 16 + i] ? (long long)(tr[w * 16 + i] - t0) : -1LL);
                                fprintf(f, "\n");
                            }
                            fclose(f);
                        }
                    }
                    static const int qorder[14] = {0, 1, 12, 13, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
                    static const int porder[14] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 7, 8, 9, 10};
                    const int *order = c.tiled == 3 ? porder : qorder;
                    for (int oi = 0; oi < 14; ++oi) {
                        const int i = order[oi];
                        std::vector<double> v, dur;
                        for (size_t w = 0; w < nwg; ++w)
                            if (tr[w * 16 + i]) {
                                v.push_back((double)(tr[w * 16 + i] - t0) * 0.01);
                                int oj = oi - 1;
                                while (oj >= 0 && !tr[w * 16 + order[oj]]) --oj;
                                if (oj >= 0) dur.push_back((double)(tr[w * 16 + i] - tr[w * 16 + order[oj]]) * 0.01);
                            }
                        if (v.empty()) continue;
                        std::sort(v.begin(), v.end());
                        double dm = 0;
                        for (double d : dur) dm += d;
                        if (!dur.empty()) dm /= dur.size();
                        printf("    %-26s at min %6.2f p10 %6.2f med %6.2f p90 %6.2f max %6.2f   phase mean %6.2f us (%zu)\n",
                               names[i], v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), dm,
                               v.size());
                    }
                }
                // timing: `iters` launches in one graph
                for (int w = 0; w < 3; ++w) run();
                CK(hipStreamSynchronize(stream));
                hipGraph_t graph;
                hipGraphExec_t gexec;
                CK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                for (int it = 0; it < iters; ++it) run(it % sets);
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
                CK(hipGraphExecDestroy(gexec));
                CK(hipGraphDestroy(graph));
                const double us = ms * 1000.0 / iters;
                printf("%-8s %-5s %-44s %8.2f us %7.1f GB/s frac %.3f", mode.c_str(), fz ? "fused" : "plain",
                       c.name.c_str(), us, alg_bytes / us * 1e-3, alg_bytes / us * 1e-3 / 8000.0);
                if (ci > 0) {
                    printf("  maxdiff %.3g bad %lld nan %lld", maxd, bad, nanc);
                    if (bad) {
                        printf("  by level [%lld %lld %lld %lld] by head [", bad_lvl[0], bad_lvl[1], bad_lvl[2], bad_lvl[3]);
                        for (int m = 0; m < M; ++m) printf("%lld ", bad_head[m]);
                        printf("] first:");
                        int lstart[5] = {0};
                        for (int l = 0; l < L; ++l) lstart[l + 1] = lstart[l] + kH[l] * kW[l];
                        for (long long pr : first_bad) {
                            const int m = (int)(pr % M), q = (int)((pr / M) % Lq);
                            int l = 0;
                            while (l < L - 1 && q >= lstart[l + 1]) ++l;
                            const int r = q - lstart[l];
                            printf(" (l%d y%d x%d m%d)", l, r / kW[l], r % kW[l], m);
                        }
                    }
                }
                printf("\n");
                fflush(stdout);
            }
        }
        CK(hipFree(d_value));
        CK(hipFree(d_loc));
        CK(hipFree(d_attn));
        CK(hipFree(d_qproj));
        CK(hipFree(d_ref));
        CK(hipFree(d_out));
    }
    return 0;
}
