// tests/emu/quad_emu.cpp -- host emulation of msda_fwd_f32_quad's data-dependent logic (test infrastructure).
//
// The HIP kernel (trackformer_amd/csrc/msda_fwd_quad.h) takes its tile partition, window geometry and
// staged / not-staged decisions from trackformer_amd/csrc/msda_quad_geom.h.  This file runs the SAME
// functions on the CPU, tile by tile: it partitions the queries, builds the bounding boxes, lays the
// windows out in an emulated LDS (rows 0, 1 zero, pixels outside the level staged as zeros), gathers
// staged points from those rows and everything else from `value` with per-tap validity -- exactly the
// kernel's decisions, minus the wave mechanics (DPP, LDS-DMA, barriers).  tests/test_quad_emulation.py
// compares the result with the oracle, so an off-by-one in the extended-coordinate windows, the two staging
// rounds or the all-or-nothing capacity rule shows up without a GPU.
//
// Built by the test with g++ (-O2 -shared -fPIC); never linked into the product.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../trackformer_amd/csrc/msda_quad_geom.h"

namespace {
struct Pt {
    float w[4];
    int x0, y0;
    bool in;
};
// the kernel's per-point arithmetic (cuh:227-237): single-rounding pixel mapping, in-range rule, weights
Pt make_point(float sx, float sy, float a, int H, int W)
{
    Pt p;
    const float Wf = (float)W, Hf = (float)H;
    const float xr = std::fmaf(sx, Wf, -0.5f), yr = std::fmaf(sy, Hf, -0.5f);
    p.in = (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
    const float x = p.in ? xr : 0.f, y = p.in ? yr : 0.f;
    const float xf = std::floor(x), yf = std::floor(y);
    const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
    p.x0 = (int)xf;
    p.y0 = (int)yf;
    const float aa = p.in ? a : 0.f;
    p.w[0] = gy * gx * aa;
    p.w[1] = gy * fx * aa;
    p.w[2] = fy * gx * aa;
    p.w[3] = fy * fx * aa;
    return p;
}
}  // namespace

extern "C" {

// stats[0]: points gathered from LDS windows, [1]: in-range points that left their window (buffer loads),
// [2]: (tile, head, level) windows that did not fit and went by buffer loads entirely, [3]: tiles,
// [4]: largest number of queries in a tile.  Returns 0, or -1 if a tile holds more than `tile_queries_cap`.
int quad_emu_forward(const float *value, const int64_t *shapes, const float *loc, const float *attn, float *out,
                     int N, int S, int M, int L, int TH, int TW, int HY, int HX, int cap_rows, int tile_queries_cap,
                     int round0_mask, int ta_mask, long long *stats)
{
    const int D = 32, P = 4, LP = L * P;
    int H[4], W[4], start[4];
    int acc = 0;
    for (int l = 0; l < L; ++l) {
        H[l] = (int)shapes[2 * l];
        W[l] = (int)shapes[2 * l + 1];
        start[l] = acc;
        acc += H[l] * W[l];
    }
    if (acc != S) return -2;
    for (int i = 0; i < 5; ++i) stats[i] = 0;
    const int tiles_y = (H[0] + TH - 1) / TH, tiles_x = (W[0] + TW - 1) / TW;
    std::vector<float> lds((size_t)(2 + cap_rows + 8) * D);
    std::vector<char> seen((size_t)N * S, 0);
    for (int b = 0; b < N; ++b)
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                // ---- the tile's queries (kernel: setup + query decode)
                const int y0t = ty * TH, y1t = tfq_min(H[0], y0t + TH), x0t = tx * TW, x1t = tfq_min(W[0], x0t + TW);
                std::vector<int> queries;
                for (int l = 0; l < L; ++l) {
                    const int ya = tfq_tile_bound(y0t, H[l], H[0]), yb = tfq_tile_bound(y1t, H[l], H[0]);
                    const int xa = tfq_tile_bound(x0t, W[l], W[0]), xb = tfq_tile_bound(x1t, W[l], W[0]);
                    for (int y = ya; y < yb; ++y)
                        for (int x = xa; x < xb; ++x) queries.push_back(start[l] + y * W[l] + x);
                }
                if ((int)queries.size() > tile_queries_cap) return -1;
                if ((long long)queries.size() > stats[4]) stats[4] = (long long)queries.size();
                ++stats[3];
                for (int q : queries) seen[(size_t)b * S + q] += 1;
                for (int m = 0; m < M; ++m) {
                    // ---- phase A: bounding boxes of the in-range points' floor coordinates, per LDS level
                    int bb[4][4];
                    for (int l = 0; l < L; ++l) {
                        bb[l][0] = INT_MAX, bb[l][1] = INT_MIN, bb[l][2] = INT_MAX, bb[l][3] = INT_MIN;
                        if ((ta_mask >> l) & 1) continue;
                        for (int q : queries) {
                            const size_t pair = ((size_t)b * S + q) * M + m;
                            for (int p = 0; p < P; ++p) {
                                const size_t s = pair * LP + l * P + p;
                                const Pt pt = make_point(loc[2 * s], loc[2 * s + 1], attn[s], H[l], W[l]);
                                if (!pt.in) continue;
                                bb[l][0] = tfq_min(bb[l][0], pt.x0);
                                bb[l][1] = tfq_max(bb[l][1], pt.x0);
                                bb[l][2] = tfq_min(bb[l][2], pt.y0);
                                bb[l][3] = tfq_max(bb[l][3], pt.y0);
                            }
                        }
                    }
                    std::vector<float> accv(queries.size() * D, 0.f);
                    // ---- the staging rounds
                    for (int round = 0; round < 2; ++round) {
                        const int rmask = round == 0 ? (round0_mask & 0xF) : (0xF & ~round0_mask);
                        const int loads_mask = round == 0 ? (rmask | ta_mask) : (rmask & ~ta_mask);
                        if (round == 1 && (rmask & ~ta_mask & ((1 << L) - 1)) == 0) break;
                        QuadWindow win[4];
                        bool by_loads[4] = {true, true, true, true};
                        int used = 0;
                        std::fill(lds.begin(), lds.end(), NAN);          // stale LDS contents must never be read
                        std::fill(lds.begin(), lds.begin() + 2 * D, 0.f);   // rows 0, 1
                        for (int l = 0; l < L; ++l) {
                            if (((ta_mask >> l) & 1) || !((rmask >> l) & 1)) continue;
                            int ny0, ny1, nx0, nx1;
                            tfq_nominal(y0t, y1t, H[l], 1.f / (float)H[0], HY, &ny0, &ny1);
                            tfq_nominal(x0t, x1t, W[l], 1.f / (float)W[0], HX, &nx0, &nx1);
                            bool fits;
                            win[l] = tfq_window(bb[l][0], bb[l][1], bb[l][2], bb[l][3], nx0, nx1, ny0, ny1, cap_rows - used,
                                                2 + used, &fits);
                            by_loads[l] = !fits;
                            if (!fits) ++stats[2];
                            const int nrows = win[l].wh * win[l].ww;
                            used += (nrows + 7) / 8 * 8;
                            for (int r = 0; r < (nrows + 7) / 8 * 8; ++r) {   // LDS-DMA: out-of-level pixels -> zeros
                                const int wy = r / tfq_max(win[l].ww, 1), wx = r - wy * win[l].ww;
                                const int py = win[l].wy0 + wy, px = win[l].wx0 + wx;
                                const bool ok = r < nrows && py >= 0 && py < H[l] && px >= 0 && px < W[l];
                                float *dst = &lds[(size_t)(win[l].roff + r) * D];
                                if (ok)
                                    memcpy(dst, value + (((size_t)b * S + start[l] + py * W[l] + px) * M + m) * D, D * 4);
                                else
                                    memset(dst, 0, D * 4);
                            }
                        }
                        // ---- gather: first the levels that go by buffer loads, then the LDS levels
                        for (int phase = 0; phase < 2; ++phase)
                            for (size_t qi = 0; qi < queries.size(); ++qi) {
                                const size_t pair = ((size_t)b * S + queries[qi]) * M + m;
                                for (int l = 0; l < L; ++l) {
                                    const bool ta = (ta_mask >> l) & 1;
                                    const int mask = phase == 1 ? (rmask & ~ta_mask) : loads_mask;
                                    if (!((mask >> l) & 1)) continue;
                                    if (ta && phase == 1) continue;
                                    if (!ta && by_loads[l] == (phase == 1)) continue;
                                    for (int p = 0; p < P; ++p) {
                                        const size_t s = pair * LP + l * P + p;
                                        const Pt pt = make_point(loc[2 * s], loc[2 * s + 1], attn[s], H[l], W[l]);
                                        bool need_global = pt.in;
                                        float *o = &accv[qi * D];
                                        if (phase == 1) {
                                            const bool staged = pt.in && tfq_staged(win[l], pt.x0, pt.y0);
                                            const int a0 = staged ? tfq_row(win[l], pt.x0, pt.y0) : 0;
                                            const int a1 = staged ? a0 + win[l].ww : 0;
                                            const float *r00 = &lds[(size_t)a0 * D], *r01 = r00 + D;
                                            const float *r10 = &lds[(size_t)a1 * D], *r11 = r10 + D;
                                            for (int c = 0; c < D; ++c)
                                                o[c] += r00[c] * pt.w[0] + r01[c] * pt.w[1] + r10[c] * pt.w[2] + r11[c] * pt.w[3];
                                            need_global = pt.in && !staged;
                                            if (staged) ++stats[0];
                                            if (need_global) ++stats[1];
                                        }
                                        if (!need_global) continue;
                                        for (int t = 0; t < 4; ++t) {
                                            const int ty_ = pt.y0 + (t >> 1), tx_ = pt.x0 + (t & 1);
                                            if (ty_ < 0 || ty_ > H[l] - 1 || tx_ < 0 || tx_ > W[l] - 1) continue;
                                            const float *row = value + (((size_t)b * S + start[l] + ty_ * W[l] + tx_) * M + m) * D;
                                            for (int c = 0; c < D; ++c) o[c] += row[c] * pt.w[t];
                                        }
                                    }
                                }
                            }
                    }
                    for (size_t qi = 0; qi < queries.size(); ++qi)
                        memcpy(out + (((size_t)b * S + queries[qi]) * M + m) * D, &accv[qi * D], D * 4);
                }
            }
    for (size_t i = 0; i < seen.size(); ++i)
        if (seen[i] != 1) return -3;   // the tiles must partition the queries
    return 0;
}
}
