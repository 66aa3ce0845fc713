// trackformer_amd/csrc/msda_pquad.hip
//
// msda_fwd_f32_pquad: MSDeformAttn forward for encoder-shaped calls (Lq == S), fp32, D == 32, P == 4,
// L <= 4 -- the PERSISTENT version of msda_fwd_f32_quad (msda_fwd_quad.h), built into libtf_msda.so as its
// own translation unit.
//
// Why.  msda_fwd_f32_quad launches one workgroup per (tile, head); at the cfg-2 encoder shape all 960 of
// them are resident at once and walk through their phases in lock-step (profiles/r01_msda_fwd_quad_phase_trace.txt):
// ~9.6 us in which every workgroup waits for its sampling points (34 MB of loc / attn or raw projections
// stream in, the LDS pipe and the vector ALUs idle) followed by ~14 us in which every workgroup gathers
// from LDS (HBM idle).  The phases ADD.  Here a workgroup owns several tiles (grid = 3 workgroups per CU,
// tile k, k + grid, ...) and the sampling points of its NEXT tile are requested before the gathers of the
// current one and consumed after them: the input stream of tile i + 1 runs under the LDS gathers of tile i.
// Per tile (parity p = i & 1 selects one of two copies of the small per-tile tables in LDS):
//
//     barrier B0      bounding boxes of tile i filed (LDS atomics), tables of tile i visible
//     stage level 0's window (LDS-DMA)   |  thread < 16: tables of tile i + 1 (query partition, nominal
//     levels that go by buffer loads     |  footprints, empty bounding boxes)
//     vmcnt(0), barrier B1               the window has landed
//     decode tile i + 1's queries, ISSUE its point loads (they land during the gathers below)
//     gather level 0 from LDS
//     barrier B2, stage the windows of levels 1..3 in the same rows, vmcnt(0), barrier B3
//     gather levels 1..3, store the tile's outputs
//     tile i + 1: softmax / location arithmetic (fused entry), bounding boxes -> LDS atomics
//
// Everything else is msda_fwd_f32_quad's design and is shared with it (msda_quad_dev.h, msda_quad_geom.h):
// 4 lanes per (query, head) pair, lane j owns point j of every level and computes its tap arithmetic once,
// the other lanes read it by DPP quad_perm broadcasts; data-adaptive windows in extended pixel coordinates
// staged by LDS-DMA (pixels outside the level arrive as zeros); a window that does not fit sends its level
// through buffer loads, a point that leaves its window takes buffer loads under a wave-uniform branch: any
// input is handled exactly.  Differences in the gather: the half-row order of a lane's two 16-byte pieces
// alternates with bit 2 of the quad index (not bit 1): neighbouring queries that read neighbouring rows AND
// neighbouring queries that share rows pairwise (a fine query level sampling a coarser value level) both
// spread over the four bank groups an LDS cycle serves.
//
// Arithmetic: SURVEY.md Appendix A; reference ms_deform_im2col_cuda.cuh:227-237 (pixel mapping, in-range
// rule), :24-67 (bilinear taps with zero padding); fused prologue ms_deform_attn.py:69-86.
#include <hip/hip_runtime.h>

#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "tf_msda.h"
#include "msda_common.h"
#include "msda_quad_geom.h"

namespace {
using namespace tfm;

#include "msda_quad_dev.h"

// Timing ablations for tools/msda_bench (tools/gpu_runs/: built into separate libraries, never into libtf_msda.so):
// TF_PQUAD_ABLATE is a bit mask of phases to leave out, so that T(full) - T(without X) gives X's marginal cost under
// the real overlap of the resident workgroups.  Results are WRONG by design with any bit set.
//   1 no LDS gathers   2 no LDS-DMA staging   4 no bounding boxes (windows = the clamped nominal footprints)
//   8 no output stores   16 no fused prologue arithmetic (softmax / locations)   32 no buffer-load fallback path
#ifndef TF_PQUAD_ABLATE
#define TF_PQUAD_ABLATE 0
#endif
constexpr int kPqAblate = TF_PQUAD_ABLATE;

constexpr int kPqLevels = 4;
constexpr int kPqThreads = 256;
constexpr int kPqPairs = kPqThreads / 4;   // (query, head) pairs per pass
constexpr int kML = TF_MSDA_MAX_LEVELS;
// LDS header: level table (3 x 16 ints) | query partition [2][4][16] | nominal footprints [2][4][16] |
// per-wave bounding boxes [2][4 waves][4 levels][4]; then rows 0, 1 (zeros) and the window rows
constexpr int kPqOffQ = 3 * kML;
constexpr int kPqOffNom = kPqOffQ + 2 * 4 * kML;
constexpr int kPqOffBb = kPqOffNom + 2 * 4 * kML;
constexpr int kPqOffGeo = kPqOffBb + 128;   // window table [2 parities][4 levels][8 ints]
constexpr int kPqHdrBytes = 2048;   // >= (kPqOffGeo + 64) * 4 = 1984, multiple of 128

struct PquadGeom {
    int TH, TW;        // tile size in level-0 pixels
    int HY, HX;        // windows are clamped to the tile footprint +- this many pixels
    int tiles_y, tiles_x;
    int n_items;       // N * tiles_y * tiles_x * M
    int cap_rows;      // LDS rows available for windows (multiple of 8)
    int skew;          // start-up skew between the workgroups of a CU, in 10 ns units (0: none)
    int cus;           // compute units of the device (workgroup b is the (b / cus)-th of its CU)
    int headmix;       // version 2: how item i maps to (tile, head) -- 0: head = i % M (workgroup b, XCD b % 8, always works on head
                       // b % 8); 1: rotated by the tile index; 2: the low bit of the head flips with every round of `cus` items
    int store;         // version 2: cache policy of the output stores -- 0: plain (the lines stay dirty in L2 until the end of the
                       // kernel writes them back), 1: nt, 2: sc1 (write-through), 3: sc0 sc1
    int cf;            // version 2: the conflict-free gather (msda_pquad2.h quad_taps_lds_cf; 4 waves x 2 passes and 8 waves x 1 pass)
    int ldnt;          // version 2: the point loads (read once) are non-temporal
    int trace_iter;    // debug: which tile of a workgroup (0 or 1) the phase stamps 3..8, 11..13 belong to
    int prio;          // version 2: static wave priority by the workgroup's slot on its CU (0: none, 1: first-dispatched highest,
                       // 2: last-dispatched highest)
    unsigned long long *trace;   // debug: 16 timestamps (s_memrealtime, 100 MHz) per workgroup, or null
};

template <int NPASS>
struct PqPoints {   // one tile's sampling points as the gathers need them + where its outputs go
    float sx[NPASS][kPqLevels], sy[NPASS][kPqLevels], sa[NPASS][kPqLevels];
    unsigned bq32[NPASS], pair32[NPASS];
    bool live[NPASS];
    int nq, b, m;
};
template <int NPASS>
struct PqRefs {     // fused entry only: reference points of the pair's query, per level
    float rx[NPASS][kPqLevels], ry[NPASS][kPqLevels];
};

// PF: always 0.  (Register prefetch of the next tile's points, PF = 2: +23 VGPRs per pass -> two workgroups per CU, 50 vs
//     43 us, removed in round 3.  An L2 prefetch by LDS-DMA into a sink: no effect, removed.  Round 3 also measured, and
//     did not keep: window hints from the previous call with the level-0 DMA issued next to the point loads (two barriers
//     fewer per tile, 47.0 vs 46.0 us), eight-wave workgroups (53 vs 47 us), four workgroups per CU at 128 VGPRs
//     (55 vs 44 us) -- profiles/r03_pquad_experiments.txt.  The kernel's time does not follow the latency chain of a
//     tile: it follows the ~3700 instructions a wave executes per tile.)
// WIDE: the points are loaded as 16-byte pieces (lane j of a quad reads level j's four points: 2 + 1 loads per pass
//     instead of 4 + 4, every cache line fetched once) and transposed inside the quad by DPP so that lane j ends
//     up with point j of every level.  Needs 16-byte aligned rows (the host checks).
template <int NPASS, int PF>
constexpr int pq_min_waves()
{
    return NPASS == 1 ? 4 : NPASS == 2 ? 3 : 2;
}

// 4 x 4 transpose across the lanes of a quad: in v[p] = element (row = this lane, column p), out v[l] = element
// (row l, column = this lane).  Two exchange steps (partner lane ^ 1, then ^ 2), 16 vector instructions.
__device__ __forceinline__ void quad_transpose4(float (&v)[4], bool b0, bool b1)
{
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const float send = b0 ? v[j] : v[j + 1];
        const float recv = dpp_f<kDppQuadXor1>(send);
        v[j] = b0 ? recv : v[j];
        v[j + 1] = b0 ? v[j + 1] : recv;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float send = b1 ? v[j] : v[j + 2];
        const float recv = dpp_f<kDppQuadXor2>(send);
        v[j] = b1 ? recv : v[j];
        v[j + 2] = b1 ? v[j + 2] : recv;
    }
}
__device__ __forceinline__ f32x4_t ldg_f4(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const f32x4_t *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}

// DH: head dimension, 32 (128-byte rows, 4 lanes x 8 channels) or 36 (hidden 288: 144-byte rows, 3 lanes x 12 channels,
//     the window rows packed without padding and staged in 16-byte pieces; see msda_quad_dev.h).
template <bool FUSED, int TA_MASK, int NPASS, int PF, bool WIDE, int DH>
__global__ void __launch_bounds__(kPqThreads, (pq_min_waves<NPASS, PF>()))
msda_fwd_f32_pquad(const DirectArgs da, const LevelTable lt, const PquadGeom pg)
{
    constexpr int PT = 4, D = DH, NL = kPqLevels, PAIRS = kPqPairs;
    static_assert(PF == 0, "the register-prefetch variant (PF = 2) measured slower (50 vs 43 us, round 2) and was removed in round 3");
    constexpr bool D36 = DH == 36;
    constexpr unsigned ROWB = D * 4;   // bytes of one (pixel, head) row
    static_assert(DH == 32 || DH == 36, "head dimension 32 or 36");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    int *s_q = s_tab + kPqOffQ;       // [parity][ya | yb | xa | xb][level]
    int *s_nom = s_tab + kPqOffNom;   // [parity][ny0 | ny1 | nx0 | nx1][level]
    int *s_bb = s_tab + kPqOffBb;     // [parity][wave][level][min x0, max x0, min y0, max y0]: no LDS atomics
    int *s_geo = s_tab + kPqOffGeo;   // [parity][level][wx0, wy0, ww, wh, limx, limy, fits on its own, -]
    unsigned char *s_rows = smem + kPqHdrBytes;   // rows 0, 1: zeros; the windows start at row 2

    const int L = da.L, M = da.M, S = da.S, LP = L * PT;
    const int G = (int)gridDim.x;
    int item = (int)blockIdx.x;
    if (item >= pg.n_items) return;

    if (pg.skew > 0) {
        // de-phase the workgroups that share a CU (observed placement: workgroup b is the (b / cus)-th of its CU)
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)((item / pg.cus) * pg.skew);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = threadIdx.x & 3;
    const int quad = threadIdx.x >> 2;
    const int hsel = (quad >> 2) & 1;
    const unsigned rbA = (unsigned)(hsel * 64 + sub * 16), rbB = (unsigned)((1 - hsel) * 64 + sub * 16);
    const unsigned lds_rows = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)s_rows);
    const unsigned ldsA = lds_rows + rbA, ldsB = lds_rows + rbB;
    // DH == 36: lanes 0..2 of the quad own 48 bytes of the row each; lane 3's global loads are sent out of range
    const unsigned rbL = (unsigned)sub * 48u, ldsL = lds_rows + rbL;
    const bool idle36 = sub == 3;

    auto stamp = [&](int i) {
        if (pg.trace != nullptr && threadIdx.x == 0 && i < 16)
            pg.trace[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memrealtime();
    };

    // ---- per-tile tables (threads 0..15): query partition, nominal footprints, empty bounding boxes --------
    auto decode_item = [&](int it, int &b, int &ty, int &tx, int &m) {
        m = it % M;
        int t = it / M;
        tx = t % pg.tiles_x;
        t /= pg.tiles_x;
        ty = t % pg.tiles_y;
        b = t / pg.tiles_y;
    };
    auto setup_tables = [&](int par, int ty, int tx) {
        if (threadIdx.x < 4 * NL) {
            const int l = threadIdx.x >> 2, k = threadIdx.x & 3;
            if (l < L) {
                const unsigned H0 = (unsigned)lt.H[0], W0 = (unsigned)lt.W[0];
                const unsigned Hl = (unsigned)lt.H[l], Wl = (unsigned)lt.W[l];
                const unsigned y0 = (unsigned)ty * pg.TH, y1 = min(H0, y0 + (unsigned)pg.TH);
                const unsigned x0 = (unsigned)tx * pg.TW, x1 = min(W0, x0 + (unsigned)pg.TW);
                s_q[(par * 4 + k) * kML + l] = k == 0   ? tfq_tile_bound(y0, Hl, H0)
                                               : k == 1 ? tfq_tile_bound(y1, Hl, H0)
                                               : k == 2 ? tfq_tile_bound(x0, Wl, W0)
                                                        : tfq_tile_bound(x1, Wl, W0);
                int lo, hi;
                if (k < 2)
                    tfq_nominal((int)y0, (int)y1, (int)Hl, 1.f / (float)H0, pg.HY, &lo, &hi);
                else
                    tfq_nominal((int)x0, (int)x1, (int)Wl, 1.f / (float)W0, pg.HX, &lo, &hi);
                s_nom[(par * 4 + k) * kML + l] = (k & 1) ? hi : lo;
            }
        }
    };

    if (threadIdx.x < NL) {
        const int l = threadIdx.x;
        s_tab[l] = l < L ? lt.H[l] : 1;
        s_tab[kML + l] = l < L ? lt.W[l] : 1;
        s_tab[2 * kML + l] = l < L ? lt.start[l] : 0;
    }
    if (threadIdx.x < 2 * D) reinterpret_cast<float *>(s_rows)[threadIdx.x] = 0.f;   // rows 0, 1
    int cb, cty, ctx, cm;
    decode_item(item, cb, cty, ctx, cm);
    setup_tables(0, cty, ctx);
    stamp(0);
    __syncthreads();

    int Hs[NL], Ws[NL], starts[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int lc = l < L ? l : 0;
        Hs[l] = __builtin_amdgcn_readfirstlane(s_tab[lc]);
        Ws[l] = __builtin_amdgcn_readfirstlane(s_tab[kML + lc]);
        starts[l] = __builtin_amdgcn_readfirstlane(s_tab[2 * kML + lc]);
    }

    float inv_h[NL], inv_w[NL];   // uniform
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        inv_h[l] = __builtin_amdgcn_rcpf((float)Hs[l]);
        inv_w[l] = __builtin_amdgcn_rcpf((float)Ws[l]);
    }

    // ---- decode the queries of a tile and issue the loads of its sampling points ---------------------------
    // (plain entry: finished locations / attention weights; fused entry: raw offsets / logits + reference points)
    auto decode_queries = [&](int par, int b, int m, PqPoints<NPASS> &p) {
        const int *q4 = s_q + par * 4 * kML;
        int qoff[NL + 1];
        qoff[0] = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l)
            qoff[l + 1] = qoff[l] + (l < L ? (q4[kML + l] - q4[l]) * (q4[3 * kML + l] - q4[2 * kML + l]) : 0);
        p.nq = qoff[NL];   // <= NPASS * PAIRS (the host checked the maximum)
        p.b = b;
        p.m = m;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int tq = ps * PAIRS + quad;
            int q = 0;
            p.live[ps] = tq < p.nq;
            if (p.live[ps]) {
                int l = 0, base = 0;
#pragma unroll
                for (int k = 1; k < NL; ++k)
                    if (tq >= qoff[k] && k < L) {
                        l = k;
                        base = qoff[k];
                    }
                const int rr = tq - base;
                const int nx = q4[3 * kML + l] - q4[2 * kML + l];
                // rr / nx: (rr + 0.5) / nx is at least 0.5 / nx away from an integer, far more than the float error
                const int yy = (int)(((float)rr + 0.5f) * __builtin_amdgcn_rcpf((float)nx));
                const int xx = rr - yy * nx;
                q = s_tab[2 * kML + l] + (q4[l] + yy) * s_tab[kML + l] + q4[2 * kML + l] + xx;
            }
            const unsigned bq = (unsigned)b * (unsigned)S + (unsigned)q;
            p.bq32[ps] = bq;
            p.pair32[ps] = bq * (unsigned)M + (unsigned)m;
        }
    };
    auto issue_loads = [&](PqPoints<NPASS> &p, PqRefs<NPASS> &r) {
        const int m = p.m;
        const bool b0 = (sub & 1) != 0, b1 = (sub & 2) != 0;
        const unsigned lsub = (unsigned)(sub < L ? sub : 0);   // WIDE: the level whose four points this lane loads
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const unsigned bq = p.bq32[ps];
            if constexpr (WIDE) {
                f32x4_t xy0, xy1, a4;
                if constexpr (!FUSED) {
                    const unsigned pr = p.pair32[ps] * (unsigned)LP + lsub * (unsigned)PT;
                    xy0 = ldg_f4(da.loc, pr * 8u);
                    xy1 = ldg_f4(da.loc, pr * 8u + 16u);
                    a4 = ldg_f4(da.attn, pr * 4u);
                } else {
                    const unsigned row = bq * (unsigned)da.fa.ld;
                    const unsigned s = (unsigned)(m * LP) + lsub * (unsigned)PT;
                    xy0 = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u);
                    xy1 = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u + 16u);
                    a4 = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.logit_col + s) * 4u);
                    const float2 rp = ldg_f2(da.fa.ref, (bq * (unsigned)L + lsub) * 8u);   // level `sub`'s reference point
                    r.rx[ps][0] = rp.x;   // [0]: this lane's own level until finish_points broadcasts them
                    r.ry[ps][0] = rp.y;
                }
                p.sx[ps][0] = xy0.x; p.sy[ps][0] = xy0.y; p.sx[ps][1] = xy0.z; p.sy[ps][1] = xy0.w;
                p.sx[ps][2] = xy1.x; p.sy[ps][2] = xy1.y; p.sx[ps][3] = xy1.z; p.sy[ps][3] = xy1.w;
                p.sa[ps][0] = a4.x; p.sa[ps][1] = a4.y; p.sa[ps][2] = a4.z; p.sa[ps][3] = a4.w;
                continue;
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int lc = l < L ? l : 0;
                const unsigned s = (unsigned)(lc * PT + sub);
                if constexpr (!FUSED) {
                    const float2 xy = ldg_f2(da.loc, (p.pair32[ps] * (unsigned)LP + s) * 8u);
                    p.sx[ps][l] = xy.x;
                    p.sy[ps][l] = xy.y;
                    p.sa[ps][l] = ldg_f(da.attn, (p.pair32[ps] * (unsigned)LP + s) * 4u);
                } else {
                    const unsigned row = bq * (unsigned)da.fa.ld;
                    const float2 off = ldg_f2(da.fa.qproj, (row + (unsigned)da.fa.off_col + ((unsigned)(m * LP) + s) * 2u) * 4u);
                    p.sx[ps][l] = off.x;
                    p.sy[ps][l] = off.y;
                    p.sa[ps][l] = l < L ? ldg_f(da.fa.qproj, (row + (unsigned)da.fa.logit_col + (unsigned)(m * LP) + s) * 4u)
                                        : -__builtin_inff();
                    const float2 rp = ldg_f2(da.fa.ref, (bq * (unsigned)L + (unsigned)lc) * 8u);   // ref_dim == 2
                    r.rx[ps][l] = rp.x;
                    r.ry[ps][l] = rp.y;
                }
            }
        }
    };
    // WIDE: lane j holds level j's four points -> point j of every level (and the reference points of every level)
    auto transpose_points = [&](PqPoints<NPASS> &p, PqRefs<NPASS> &r) {
        if constexpr (WIDE) {
            const bool b0 = (sub & 1) != 0, b1 = (sub & 2) != 0;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                quad_transpose4(p.sx[ps], b0, b1);
                quad_transpose4(p.sy[ps], b0, b1);
                quad_transpose4(p.sa[ps], b0, b1);
                if constexpr (FUSED) {
                    const float x = r.rx[ps][0], y = r.ry[ps][0];
                    r.rx[ps][0] = dpp_f<0x00>(x); r.ry[ps][0] = dpp_f<0x00>(y);
                    r.rx[ps][1] = dpp_f<0x55>(x); r.ry[ps][1] = dpp_f<0x55>(y);
                    r.rx[ps][2] = dpp_f<0xAA>(x); r.ry[ps][2] = dpp_f<0xAA>(y);
                    r.rx[ps][3] = dpp_f<0xFF>(x); r.ry[ps][3] = dpp_f<0xFF>(y);
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        if (l >= L) p.sa[ps][l] = -__builtin_inff();
                }
            }
        }
    };
    // ---- fused entry: softmax over the pair's L*P logits, sampling locations (ms_deform_attn.py:69-79) -----
    auto finish_points = [&](PqPoints<NPASS> &p, PqRefs<NPASS> &r) {
        transpose_points(p, r);
        if constexpr (FUSED && !(kPqAblate & 16)) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                float mx = p.sa[ps][0];
#pragma unroll
                for (int l = 1; l < NL; ++l) mx = fmaxf(mx, p.sa[ps][l]);
                mx = fmaxf(mx, dpp_f<kDppQuadXor1>(mx));
                mx = fmaxf(mx, dpp_f<kDppQuadXor2>(mx));
                float sum = 0.f;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    p.sa[ps][l] = l < L ? __expf(p.sa[ps][l] - mx) : 0.f;
                    sum += p.sa[ps][l];
                }
                sum += dpp_f<kDppQuadXor1>(sum);
                sum += dpp_f<kDppQuadXor2>(sum);
                // reciprocals (v_rcp_f32, 1 ulp) instead of 12 IEEE divisions per pass (~10 instructions each): the
                // normalised weights / locations move by <= 1 ulp, far below what the exp already differs by
                const float inv_sum = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    p.sa[ps][l] = p.sa[ps][l] * inv_sum;
                    p.sx[ps][l] = r.rx[ps][l] + p.sx[ps][l] * inv_h[l];   // x / H_l (as the reference writes it)
                    p.sy[ps][l] = r.ry[ps][l] + p.sy[ps][l] * inv_w[l];   // y / W_l
                }
            }
        }
    };

    // ---- phase A: bounding box of the floor coordinates of the tile's in-range points, per LDS level -------
    auto bbox_level = [&](auto lc, const PqPoints<NPASS> &p, int par) {
        constexpr int l = decltype(lc)::value;
        if constexpr ((kPqAblate & 4) != 0) {
            if (lane == 0) {   // the whole level: tfq_window clamps it to the nominal footprint
                int *slot = s_bb + ((par * 4 + wave) * 4 + l) * 4;
                slot[0] = -1;
                slot[1] = l < L ? Ws[l < NL ? l : 0] : INT_MIN;
                slot[2] = -1;
                slot[3] = l < L ? Hs[l < NL ? l : 0] : INT_MIN;
            }
        } else if constexpr (((TA_MASK >> l) & 1) == 0) {
            if (l < L) {
                int mnx = INT_MAX, mxx = INT_MIN, mny = INT_MAX, mxy = INT_MIN;
                const float Wf = (float)Ws[l], Hf = (float)Hs[l];
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const float xr = __builtin_fmaf(p.sx[ps][l], Wf, -0.5f);
                    const float yr = __builtin_fmaf(p.sy[ps][l], Hf, -0.5f);
                    const bool in = p.live[ps] && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
                    const int x0 = (int)__builtin_floorf(in ? xr : 0.f), y0 = (int)__builtin_floorf(in ? yr : 0.f);
                    mnx = min(mnx, in ? x0 : INT_MAX);
                    mxx = max(mxx, in ? x0 : INT_MIN);
                    mny = min(mny, in ? y0 : INT_MAX);
                    mxy = max(mxy, in ? y0 : INT_MIN);
                }
                mnx = min(mnx, dpp_i<kDppQuadXor1>(mnx));
                mxx = max(mxx, dpp_i<kDppQuadXor1>(mxx));
                mny = min(mny, dpp_i<kDppQuadXor1>(mny));
                mxy = max(mxy, dpp_i<kDppQuadXor1>(mxy));
                mnx = min(mnx, dpp_i<kDppQuadXor2>(mnx));
                mxx = max(mxx, dpp_i<kDppQuadXor2>(mxx));
                mny = min(mny, dpp_i<kDppQuadXor2>(mny));
                mxy = max(mxy, dpp_i<kDppQuadXor2>(mxy));
                mnx = min(mnx, dpp_i<kDppRowRor4>(mnx));
                mxx = max(mxx, dpp_i<kDppRowRor4>(mxx));
                mny = min(mny, dpp_i<kDppRowRor4>(mny));
                mxy = max(mxy, dpp_i<kDppRowRor4>(mxy));
                mnx = min(mnx, dpp_i<kDppRowRor8>(mnx));
                mxx = max(mxx, dpp_i<kDppRowRor8>(mxx));
                mny = min(mny, dpp_i<kDppRowRor8>(mny));
                mxy = max(mxy, dpp_i<kDppRowRor8>(mxy));
                // the four DPP rows of the wave -> one box per wave (scalar), filed in the wave's own slot: LDS atomics
                // (ds_min / ds_max on one address from every wave of the CU) serialise and cost microseconds per tile
                const int a0 = __builtin_amdgcn_readlane(mnx, 0), a1 = __builtin_amdgcn_readlane(mnx, 16);
                const int a2 = __builtin_amdgcn_readlane(mnx, 32), a3 = __builtin_amdgcn_readlane(mnx, 48);
                const int b0 = __builtin_amdgcn_readlane(mxx, 0), b1 = __builtin_amdgcn_readlane(mxx, 16);
                const int b2 = __builtin_amdgcn_readlane(mxx, 32), b3 = __builtin_amdgcn_readlane(mxx, 48);
                const int c0 = __builtin_amdgcn_readlane(mny, 0), c1 = __builtin_amdgcn_readlane(mny, 16);
                const int c2 = __builtin_amdgcn_readlane(mny, 32), c3 = __builtin_amdgcn_readlane(mny, 48);
                const int d0 = __builtin_amdgcn_readlane(mxy, 0), d1 = __builtin_amdgcn_readlane(mxy, 16);
                const int d2 = __builtin_amdgcn_readlane(mxy, 32), d3 = __builtin_amdgcn_readlane(mxy, 48);
                const int wmnx = min(min(a0, a1), min(a2, a3)), wmxx = max(max(b0, b1), max(b2, b3));
                const int wmny = min(min(c0, c1), min(c2, c3)), wmxy = max(max(d0, d1), max(d2, d3));
                if (lane == 0) {
                    int *slot = s_bb + ((par * 4 + wave) * 4 + l) * 4;
                    slot[0] = wmnx;
                    slot[1] = wmxx;
                    slot[2] = wmny;
                    slot[3] = wmxy;
                }
            } else if (l < NL && lane == 0) {   // levels the call does not have: an empty box
                int *slot = s_bb + ((par * 4 + wave) * 4 + l) * 4;
                slot[0] = INT_MAX;
                slot[1] = INT_MIN;
                slot[2] = INT_MAX;
                slot[3] = INT_MIN;
            }
        }
    };
    auto bbox = [&](const PqPoints<NPASS> &p, int par) {
        bbox_level(std::integral_constant<int, 0>{}, p, par);
        bbox_level(std::integral_constant<int, 1>{}, p, par);
        bbox_level(std::integral_constant<int, 2>{}, p, par);
        bbox_level(std::integral_constant<int, 3>{}, p, par);
    };

    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(da.value), 0, da.value_bytes, 0x00020000);

    // ---- the first tile's points: loaded and consumed right away (nothing to overlap with yet) ------------
    PqPoints<NPASS> cur, nxt;
    PqRefs<NPASS> nref;
    decode_queries(0, cb, cm, cur);
    issue_loads(cur, nref);
    stamp(1);
    finish_points(cur, nref);
    bbox(cur, 0);
    stamp(2);

    int par = 0;
    int iter = 0;
    while (true) {
        const int next_item = item + G;
        const bool has_next = next_item < pg.n_items;   // uniform
        int nb = 0, nty = 0, ntx = 0, nm = 0;
        if (has_next) decode_item(next_item, nb, nty, ntx, nm);

        __syncthreads();   // B0: the tile's bounding boxes are filed, its tables visible
        if (iter == 0) stamp(3);
        // window geometry: wave l works out level l's window (bounding box over the four waves' boxes, clamped to the
        // nominal footprint) and files it; every wave then only reads the table -- computing all four windows in
        // every wave cost ~240 vector instructions per wave and tile
        if (wave < NL) {
            const int l = wave;
            int *ge = s_geo + (par * 4 + l) * 8;
            QuadWindow w;
            w.wx0 = kQuadFar;
            w.wy0 = kQuadFar;
            w.ww = w.wh = w.limx = w.limy = 0;
            bool fits = false;
            if (l < L && ((TA_MASK >> l) & 1) == 0) {
                const int *bb = s_bb + par * 64 + 4 * l;   // + 16 * wave
                const int *nm4 = s_nom + par * 4 * kML;
                int bx0 = INT_MAX, bx1 = INT_MIN, by0 = INT_MAX, by1 = INT_MIN;
#pragma unroll
                for (int ww = 0; ww < kPqThreads / 64; ++ww) {
                    bx0 = min(bx0, __builtin_amdgcn_readfirstlane(bb[16 * ww + 0]));
                    bx1 = max(bx1, __builtin_amdgcn_readfirstlane(bb[16 * ww + 1]));
                    by0 = min(by0, __builtin_amdgcn_readfirstlane(bb[16 * ww + 2]));
                    by1 = max(by1, __builtin_amdgcn_readfirstlane(bb[16 * ww + 3]));
                }
                const int ny0 = __builtin_amdgcn_readfirstlane(nm4[l]);
                const int ny1 = __builtin_amdgcn_readfirstlane(nm4[kML + l]);
                const int nx0 = __builtin_amdgcn_readfirstlane(nm4[2 * kML + l]);
                const int nx1 = __builtin_amdgcn_readfirstlane(nm4[3 * kML + l]);
                w = tfq_window(bx0, bx1, by0, by1, nx0, nx1, ny0, ny1, pg.cap_rows, 2, &fits);
            }
            if (lane == 0) {
                ge[0] = w.wx0;
                ge[1] = w.wy0;
                ge[2] = w.ww;
                ge[3] = w.wh;
                ge[4] = w.limx;
                ge[5] = w.limy;
                ge[6] = fits ? 1 : 0;
            }
        }
        __syncthreads();   // B0': the window table is visible

        // window geometry (wave-uniform, scalar registers) of the current tile
        const unsigned head_base = (unsigned)((((long long)cur.b * S * M + cur.m) * D) * 4);
        int gwx0[NL], gwy0[NL], glimx[NL], glimy[NL], gww[NL], groff[NL];
        unsigned glvl[NL];
        bool by_loads[NL];   // the level is gathered by buffer loads (TA_MASK, or its window did not fit)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            glvl[l] = head_base + (unsigned)starts[l] * rowbytes;
            gwx0[l] = kQuadFar;
            gwy0[l] = kQuadFar;
            glimx[l] = 0;
            glimy[l] = 0;
            gww[l] = 0;
            groff[l] = 0;
            by_loads[l] = true;
        }
        int used = 0;   // LDS rows taken by the windows of the current round
        auto phase_b = [&](auto lc, auto rc) {
            constexpr int l = decltype(lc)::value;
            constexpr int RMASK = decltype(rc)::value;
            if constexpr (((TA_MASK >> l) & 1) == 0 && ((RMASK >> l) & 1) != 0) {
                if (l < L) {
                    const int H = Hs[l], W = Ws[l];
                    const int *ge = s_geo + (par * 4 + l) * 8;   // this level's window, computed by wave l after B0
                    const int ww = __builtin_amdgcn_readfirstlane(ge[2]), wh = __builtin_amdgcn_readfirstlane(ge[3]);
                    const bool fits = __builtin_amdgcn_readfirstlane(ge[6]) != 0 && wh * ww <= pg.cap_rows - used;
                    by_loads[l] = !fits;
                    if (!fits) return;   // all or nothing (msda_quad_geom.h): the level goes by buffer loads
                    const int wx0 = __builtin_amdgcn_readfirstlane(ge[0]), wy0 = __builtin_amdgcn_readfirstlane(ge[1]);
                    const int roff = 2 + used;
                    gwx0[l] = wx0;
                    gwy0[l] = wy0;
                    glimx[l] = __builtin_amdgcn_readfirstlane(ge[4]);
                    glimy[l] = __builtin_amdgcn_readfirstlane(ge[5]);
                    gww[l] = ww;
                    groff[l] = roff;
                    const int nrows = wh * ww;
                    if constexpr (!D36) {
                        const int nchunks = (nrows + 7) >> 3;   // one DMA wave-instruction = 8 rows of 128 B
                        used += nchunks * 8;
                        if (nchunks > 0) {
                            const unsigned lvl_base = glvl[l];
                            const float inv_ww = __builtin_amdgcn_rcpf((float)ww);
                            int r = wave * 8 + (lane >> 3);                       // < 64
                            int wy = (int)(((float)r + 0.5f) * inv_ww);
                            int wx = r - wy * ww;
                            constexpr int STEP = 8 * (kPqThreads / 64);
                            const int qstep = __builtin_amdgcn_readfirstlane((int)(((float)STEP + 0.5f) * inv_ww));
                            const int rstep = STEP - qstep * ww;
                            unsigned off = lvl_base + (unsigned)((wy0 + wy) * W + wx0 + wx) * rowbytes + (unsigned)(lane & 7) * 16u;
                            const unsigned step_a = (unsigned)(qstep * W + rstep) * rowbytes;
                            const unsigned step_b = (unsigned)(W - ww) * rowbytes;
                            for (int c = wave; c < nchunks; c += kPqThreads / 64) {
                                const int py = wy0 + wy, px = wx0 + wx;   // extended coordinates: may be -1 or size
                                const bool ok = r < nrows && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
                                if constexpr (!(kPqAblate & 2))
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                        rsrc, (__attribute__((address_space(3))) void *)(s_rows + (size_t)(roff + c * 8) * 128),
                                        16, ok ? off : kOobOffset /* hardware writes zeros */, 0, 0, 0);
                                r += STEP;
                                wy += qstep;
                                wx += rstep;
                                off += step_a;
                                if (wx >= ww) {
                                    wx -= ww;
                                    wy += 1;
                                    off += step_b;
                                }
                            }
                        }
                    } else {
                        // 144-byte rows, packed: the window is nrows * 9 pieces of 16 bytes, one DMA wave-instruction
                        // moves 64 consecutive pieces (7 rows and a piece); the next window starts on a row boundary
                        const int npieces = nrows * 9;
                        const int nchunks = (npieces + 63) >> 6;
                        used += (nchunks * 64 + 8) / 9;
                        const unsigned lvl_base = glvl[l];
                        const float inv_ww = __builtin_amdgcn_rcpf((float)ww);
                        for (int c = wave; c < nchunks; c += kPqThreads / 64) {
                            const int pc = c * 64 + lane;
                            const int r = (int)(((float)pc + 0.5f) * (1.f / 9.f));   // pc / 9 (pc < 2^16: exact)
                            const int piece = pc - r * 9;
                            const int wy = (int)(((float)r + 0.5f) * inv_ww);
                            const int wx = r - wy * ww;
                            const int py = wy0 + wy, px = wx0 + wx;   // extended coordinates: may be -1 or size
                            const bool ok = r < nrows && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
                            const unsigned off = lvl_base + (unsigned)(py * W + px) * rowbytes + (unsigned)piece * 16u;
                            if constexpr (!(kPqAblate & 2))
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                rsrc, (__attribute__((address_space(3))) void *)(s_rows + (size_t)roff * ROWB + (size_t)c * 1024),
                                16, ok ? off : kOobOffset /* hardware writes zeros */, 0, 0, 0);
                        }
                    }
                }
            }
        };

        f32x4_t accA[NPASS], accB[NPASS];
        f32x4_t acc36[NPASS][3];   // DH == 36: 12 channels per lane
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            accA[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accB[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) acc36[ps][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        auto level = [&](auto lc, auto psc, auto ldsc) {
            constexpr int l = decltype(lc)::value;
            constexpr int ps = decltype(psc)::value < NPASS ? decltype(psc)::value : 0;   // (never called out of range)
            constexpr bool LDS_PHASE = decltype(ldsc)::value;
            constexpr bool TA = ((TA_MASK >> l) & 1) != 0;
            if (l >= L) return;                          // uniform
            if constexpr (TA && LDS_PHASE) return;
            if (!TA && by_loads[l] == LDS_PHASE) return;   // uniform: a level runs in exactly one of the two phases
            if (ps * PAIRS >= cur.nq) return;            // uniform
            const int H = Hs[l], W = Ws[l];
            const float Wf = (float)W, Hf = (float)H;
            const float xr = __builtin_fmaf(cur.sx[ps][l], Wf, -0.5f);   // cuh:227-228 with one rounding (see make_tap, msda_hip.hip)
            const float yr = __builtin_fmaf(cur.sy[ps][l], Hf, -0.5f);
            const bool in = cur.live[ps] && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
            const float x = in ? xr : 0.f, y = in ? yr : 0.f;
            const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
            const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
            const int x0 = (int)xf, y0 = (int)yf;
            const float a = in ? cur.sa[ps][l] : 0.f;
            const float w[4] = {gy * gx * a, gy * fx * a, fy * gx * a, fy * fx * a};
            bool need_global = in;
            if constexpr (LDS_PHASE) {
                const int dx = x0 - gwx0[l], dy = y0 - gwy0[l];
                const bool staged = in && (unsigned)dx <= (unsigned)glimx[l] && (unsigned)dy <= (unsigned)glimy[l];
                const unsigned lo = (unsigned)(groff[l] + dy * gww[l] + dx) * ROWB;
                const unsigned a0 = staged ? lo : 0u;                            // rows 0, 1 are zeros
                const unsigned a1 = staged ? lo + (unsigned)gww[l] * ROWB : 0u;
                if constexpr ((kPqAblate & 1) != 0) {
                    accA[ps].x += (float)(a0 + a1) * w[0];   // keeps the tap arithmetic alive
                } else if constexpr (!D36) {
                    quad_taps_lds<0>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
                    quad_taps_lds<1>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
                    quad_taps_lds<2>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
                    quad_taps_lds<3>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
                } else {
                    quad_taps_lds36<0>(a0, a1, w, ldsL, acc36[ps]);
                    quad_taps_lds36<1>(a0, a1, w, ldsL, acc36[ps]);
                    quad_taps_lds36<2>(a0, a1, w, ldsL, acc36[ps]);
                    quad_taps_lds36<3>(a0, a1, w, ldsL, acc36[ps]);
                }
                need_global = in && !staged;
                if constexpr ((kPqAblate & 32) != 0) return;
                if (!__any(need_global)) return;   // wave-uniform: no point of this wave left its window
            }
            const bool kx0 = need_global && (x0 >= 0), kx1 = need_global && (x0 + 1 <= W - 1);
            const bool ky0 = (y0 >= 0), ky1 = (y0 + 1 <= H - 1);
            const int r0 = y0 * W + x0;
            const unsigned lvl_base = glvl[l];
            // staged / invalid taps: kOobBase + (lane offset < 128) is still out of range -> hardware zero
            const unsigned g[4] = {(ky0 && kx0) ? lvl_base + (unsigned)r0 * rowbytes : kOobBase,
                                   (ky0 && kx1) ? lvl_base + (unsigned)(r0 + 1) * rowbytes : kOobBase,
                                   (ky1 && kx0) ? lvl_base + (unsigned)(r0 + W) * rowbytes : kOobBase,
                                   (ky1 && kx1) ? lvl_base + (unsigned)(r0 + W + 1) * rowbytes : kOobBase};
            if constexpr (!D36) {
                quad_taps_global<0>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
                quad_taps_global<1>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
                quad_taps_global<2>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
                quad_taps_global<3>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
            } else {
                quad_taps_global36<0>(rsrc, g, w, rbL, idle36, acc36[ps]);
                quad_taps_global36<1>(rsrc, g, w, rbL, idle36, acc36[ps]);
                quad_taps_global36<2>(rsrc, g, w, rbL, idle36, acc36[ps]);
                quad_taps_global36<3>(rsrc, g, w, rbL, idle36, acc36[ps]);
            }
        };
        auto levels = [&](auto maskc, auto psc, auto ldsc) {
            constexpr int MASK = decltype(maskc)::value;
            if constexpr (MASK & 1) level(std::integral_constant<int, 0>{}, psc, ldsc);
            if constexpr (MASK & 2) level(std::integral_constant<int, 1>{}, psc, ldsc);
            if constexpr (MASK & 4) level(std::integral_constant<int, 2>{}, psc, ldsc);
            if constexpr (MASK & 8) level(std::integral_constant<int, 3>{}, psc, ldsc);
        };
        auto all_passes = [&](auto maskc, auto ldsc) {
            levels(maskc, std::integral_constant<int, 0>{}, ldsc);
            if constexpr (NPASS > 1) levels(maskc, std::integral_constant<int, 1>{}, ldsc);
            if constexpr (NPASS > 2) levels(maskc, std::integral_constant<int, 2>{}, ldsc);
        };
        constexpr int R0 = 0x1 & ~TA_MASK, R1 = 0xE & ~TA_MASK;   // LDS levels of the two rounds

        // ---- round 0: level 0 ----
        used = 0;
        phase_b(std::integral_constant<int, 0>{}, std::integral_constant<int, R0>{});
        all_passes(std::integral_constant<int, (0x1 | TA_MASK)>{}, std::false_type{});   // by buffer loads
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA landed
        __syncthreads();                      // B1: ... everybody's; the next tile's tables are visible
        if (iter == 0) stamp(4);
        PqRefs<NPASS> rr;
        if (has_next) setup_tables(par ^ 1, nty, ntx);   // off the staging path; visible after the barriers below
        if (iter == 0) stamp(5);
        all_passes(std::integral_constant<int, R0>{}, std::true_type{});
        if (iter == 0) stamp(6);

        // ---- round 1: levels 1..3 in the same rows ----
        __syncthreads();   // B2: every wave is done reading level 0's window
        if (iter == 0) stamp(11);
        used = 0;
        phase_b(std::integral_constant<int, 1>{}, std::integral_constant<int, R1>{});
        phase_b(std::integral_constant<int, 2>{}, std::integral_constant<int, R1>{});
        phase_b(std::integral_constant<int, 3>{}, std::integral_constant<int, R1>{});
        if (iter == 0) stamp(12);
        all_passes(std::integral_constant<int, R1>{}, std::false_type{});
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (iter == 0) stamp(13);
        __syncthreads();   // B3
        if (iter == 0) stamp(7);
        all_passes(std::integral_constant<int, R1>{}, std::true_type{});
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            if (cur.live[ps] && (!(kPqAblate & 8) || accA[ps].x == 12345.678f)) {
                float *o = reinterpret_cast<float *>(reinterpret_cast<char *>(da.out) + (size_t)(cur.pair32[ps] * (unsigned)(D * 4)));
                if constexpr (!D36) {
                    *reinterpret_cast<f32x4_t *>(o + rbA / 4) = accA[ps];
                    *reinterpret_cast<f32x4_t *>(o + rbB / 4) = accB[ps];
                } else if (!idle36) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4_t *>(o + rbL / 4 + 4 * c) = acc36[ps][c];
                }
            }
        if (iter == 0) stamp(8);
        if (!has_next) break;

        // ---- the next tile: prologue arithmetic and bounding boxes (its points arrived during the gathers) ----
        decode_queries(par ^ 1, nb, nm, nxt);
        issue_loads(nxt, rr);
        finish_points(nxt, rr);
        bbox(nxt, par ^ 1);
        if (iter == 0) stamp(9);
        cur = nxt;
        item = next_item;
        par ^= 1;
        ++iter;
    }
    stamp(10);
}

#include "msda_pquad2.h"

// ---- options, tile plan, launch ------------------------------------------------------------------------------
enum PqOpt { kPoWide, kPoNpass, kPoLdsKb, kPoHaloY, kPoHaloX, kPoTileH, kPoTileW, kPoWgPerCu, kPoPrefetch, kPoSkew, kPoEnable, kPoV2, kPoWaves,
             kPoHeadMix, kPoPrio, kPoStore, kPoLdNt, kPoTraceIter, kPoCf, kPoCount };
const char *const kPqOptNames[kPoCount] = {"pquad_wide", "pquad_npass", "pquad_lds_kb", "pquad_halo_y", "pquad_halo_x",
                                           "pquad_tile_h",  "pquad_tile_w", "pquad_wg_per_cu", "pquad_prefetch", "pquad_skew",
                                           "pquad", "pquad_v2", "pquad_waves", "pquad_headmix", "pquad_prio", "pquad_store", "pquad_ldnt", "pquad_trace_iter", "pquad_cf"};
const char *const kPqEnvKeys[kPoCount] = {"wide", "npass", "lds", "hy", "hx", "th", "tw", "wgs", "pf", "skew", "on", "v2", "waves", "hm", "prio", "st", "ldnt", "ti", "cf"};
// v2: msda_fwd_f32_pquad2 (msda_pquad2.h) where it applies (D == 32, two passes, 16-byte aligned inputs)
// waves: 4, or 8 (version 2 only: one pass of 128 pairs, two workgroups per CU -- use with lds=78)
constexpr int kPqOptDefaults[kPoCount] = {1, 2, 52, 6, 10, 0, 0, 3, 0, 0, 1, 1, 4, 0, 0, 1, 0, 0, 0};   // 3 x 52 KB = 156 KB of the CU's 160
std::atomic<int> g_pq_opt[kPoCount];
std::atomic<int> g_pq_epoch{0};
std::atomic<unsigned long long *> g_pq_trace{nullptr};

int pq_max_wgs(int npass, int pf)   // = pq_min_waves<NPASS, PF>(): workgroups per CU the register budget admits
{
    (void)pf;
    return npass == 1 ? 4 : npass == 2 ? 3 : 2;
}

void pq_opts_init()
{
    static const bool once = [] {
        for (int i = 0; i < kPoCount; ++i) g_pq_opt[i].store(kPqOptDefaults[i]);
        if (const char *e = getenv("TF_MSDA_PQUAD")) {   // comma-separated key=value list, e.g. "on=0" or "npass=3,wgs=2"
            const char *p = e;
            while (*p) {
                const char *eq = strchr(p, '=');
                if (!eq) break;
                for (int i = 0; i < kPoCount; ++i)
                    if ((size_t)(eq - p) == strlen(kPqEnvKeys[i]) && strncmp(p, kPqEnvKeys[i], eq - p) == 0)
                        g_pq_opt[i].store(atoi(eq + 1));
                const char *c = strchr(eq, ',');
                if (!c) break;
                p = c + 1;
            }
        }
        return true;
    }();
    (void)once;
}

long long pq_tile_max_queries(const LevelTable &lt, int L, int th, int tw)
{
    const int H0 = lt.H[0], W0 = lt.W[0];
    long long max_nq = 0;   // exact, same integer partition as the kernel (tfq_tile_bound)
    for (int y0 = 0; y0 < H0; y0 += th)
        for (int x0 = 0; x0 < W0; x0 += tw) {
            const int y1 = (y0 + th < H0) ? y0 + th : H0, x1 = (x0 + tw < W0) ? x0 + tw : W0;
            long long nq = 0;
            for (int l = 0; l < L; ++l) {
                const long long Hl = lt.H[l], Wl = lt.W[l];
                const long long ny = (2 * y1 * Hl + H0 - 1) / (2LL * H0) - (2 * y0 * Hl + H0 - 1) / (2LL * H0);
                const long long nx = (2 * x1 * Wl + W0 - 1) / (2LL * W0) - (2 * x0 * Wl + W0 - 1) / (2LL * W0);
                nq += ny * nx;
            }
            if (nq > max_nq) max_nq = nq;
        }
    return max_nq;
}

struct PqPlan {
    PquadGeom geom;
    size_t lds;
    int ta_mask, npass, wgs, pf;
    bool wide;
    bool v2;
    int waves;
};

int pq_num_cus()
{
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    return n;
}

bool pq_plan(const LevelTable &lt, int L, int M, int N, int D, PqPlan *out)
{
    pq_opts_init();
    int o[kPoCount];
    for (int i = 0; i < kPoCount; ++i) o[i] = g_pq_opt[i].load(std::memory_order_relaxed);
    if (!o[kPoEnable]) return false;
    const int epoch = g_pq_epoch.load(std::memory_order_relaxed);
    const int ta = 0, npass = o[kPoNpass], pf = o[kPoPrefetch];
    if (npass < 1 || npass > 3 || pf != 0 || o[kPoWgPerCu] < 1 || o[kPoSkew] < 0) return false;   // pquad_prefetch = 2: removed
    // eight-wave workgroups (version 2 only, one pass of 128 pairs): two workgroups per CU (<= 128 registers)
    const int waves = (o[kPoV2] && D == 32 && o[kPoWaves] == 8 && npass == 1) ? 8 : 4;
    const int max_wgs = waves == 8 ? 2 : pq_max_wgs(npass, pf);
    const int wgs = o[kPoWgPerCu] < max_wgs ? o[kPoWgPerCu] : max_wgs;
    if (o[kPoLdsKb] < 8 || o[kPoLdsKb] > 160 || o[kPoHaloY] < 0 || o[kPoHaloX] < 0 || o[kPoTileH] < 0 || o[kPoTileW] < 0)
        return false;
    for (int l = 0; l < L; ++l)
        if (lt.H[l] >= 32768 || lt.W[l] >= 32768) return false;   // 32-bit tile arithmetic in the kernel
    const int rowb = D * 4;
    // D == 36: one KB of slack behind the rows, the last DMA chunk of a window may end past its last row
    int cap_rows = (int)(((size_t)o[kPoLdsKb] * 1024 - kPqHdrBytes - (D == 36 ? 1024 : 0)) / rowb - 2);
    if (D == 32) cap_rows &= ~7;
    if (cap_rows < 8) return false;
    struct Memo {
        bool valid = false, ok = false;
        int L = 0, M = 0, N = 0, D = 0, epoch = -1;
        LevelTable lt;
        PqPlan plan;
    };
    static thread_local Memo memo;
    if (memo.valid && memo.L == L && memo.M == M && memo.N == N && memo.D == D && memo.epoch == epoch &&
        memcmp(&memo.lt, &lt, sizeof(lt)) == 0) {
        *out = memo.plan;
        return memo.ok;
    }
    memo.valid = true;
    memo.ok = false;
    memo.L = L;
    memo.M = M;
    memo.N = N;
    memo.D = D;
    memo.epoch = epoch;
    memo.lt = lt;
    const long long cap_q = (long long)(16 * waves) * npass;
    const long long slots = (long long)pq_num_cus() * wgs;
    int bth = 0, btw = 0;
    if (o[kPoTileH] > 0 && o[kPoTileW] > 0) {
        const long long nq = pq_tile_max_queries(lt, L, o[kPoTileH], o[kPoTileW]);
        if (nq >= 1 && nq <= cap_q) {
            bth = o[kPoTileH];
            btw = o[kPoTileW];
        }
    } else {
        // score: queries per (estimated) window row, times how evenly the tiles fill the resident workgroups
        // (every workgroup runs ceil(items / slots) tiles one after the other)
        double best = 0.0;
        for (int th = 1; th <= 32; ++th)
            for (int tw = 2; tw <= 32; tw += 2) {
                if ((long long)th * tw > cap_q) continue;
                const long long nq = pq_tile_max_queries(lt, L, th, tw);
                if (nq < 1 || nq > cap_q) continue;
                const long long items = (long long)N * ((lt.H[0] + th - 1) / th) * ((lt.W[0] + tw - 1) / tw) * M;
                const long long rounds = (items + slots - 1) / slots;
                const double balance = (double)items / (double)(rounds * slots);
                const double fill = (double)lt.H[0] * lt.W[0] * N * M / ((double)items * th * tw);   // edge tiles
                const double eff = (double)(th * tw) / ((double)(th + 4) * (double)(tw + 8));
                const double score = eff * balance * fill;
                if (score > best) {
                    best = score;
                    bth = th;
                    btw = tw;
                }
            }
    }
    if (!bth) return false;
    PqPlan r{};
    r.geom.TH = bth;
    r.geom.TW = btw;
    r.geom.HY = o[kPoHaloY];
    r.geom.HX = o[kPoHaloX];
    r.geom.tiles_y = (lt.H[0] + bth - 1) / bth;
    r.geom.tiles_x = (lt.W[0] + btw - 1) / btw;
    const long long items = (long long)N * r.geom.tiles_y * r.geom.tiles_x * M;
    if (items > 0x7fffffffLL) return false;
    r.geom.n_items = (int)items;
    r.geom.cap_rows = cap_rows;
    r.lds = (size_t)kPqHdrBytes + (size_t)(2 + cap_rows) * rowb + (D == 36 ? 1024 : 0);
    r.geom.skew = o[kPoSkew];
    r.geom.cus = pq_num_cus();
    r.geom.headmix = (o[kPoHeadMix] == 1 || (o[kPoHeadMix] == 2 && M % 2 == 0)) ? o[kPoHeadMix] : 0;
    r.geom.prio = (o[kPoPrio] == 1 || o[kPoPrio] == 2) ? o[kPoPrio] : 0;
    r.geom.store = (o[kPoStore] >= 0 && o[kPoStore] <= 3) ? o[kPoStore] : 0;
    r.geom.ldnt = o[kPoLdNt] != 0;
    r.geom.cf = o[kPoCf] != 0 && !(waves == 4 && npass == 1);
    r.geom.trace_iter = o[kPoTraceIter] == 1 ? 1 : 0;
    r.wide = o[kPoWide] != 0;
    r.v2 = o[kPoV2] != 0;
    r.waves = waves;
    r.ta_mask = ta;
    r.npass = npass;
    r.wgs = wgs;
    r.pf = pf;
    memo.plan = r;
    memo.ok = true;
    *out = r;
    static const bool verbose = getenv("TF_MSDA_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[tf_msda] pquad plan: tile %dx%d (%lld queries max of %lld), %dx%d tiles x %d heads = %lld items on "
                        "%lld workgroups, ta_mask %d, %d passes, prefetch %d, %d window rows, %zu B LDS\n", bth, btw,
                pq_tile_max_queries(lt, L, bth, btw), cap_q, r.geom.tiles_y, r.geom.tiles_x, M, items, slots, ta, npass, pf,
                cap_rows, r.lds);
    return true;
}

template <bool FUSED, bool WIDE>
const void *pq_kernel_n(int npass)
{
    return npass == 1   ? (const void *)&msda_fwd_f32_pquad<FUSED, 0, 1, 0, WIDE, 32>
           : npass == 2 ? (const void *)&msda_fwd_f32_pquad<FUSED, 0, 2, 0, WIDE, 32>
                        : (const void *)&msda_fwd_f32_pquad<FUSED, 0, 3, 0, WIDE, 32>;
}
// head dimension 36: 2 passes (the only variant built)
template <bool FUSED>
const void *pq_kernel_d36(bool wide)
{
    return wide ? (const void *)&msda_fwd_f32_pquad<FUSED, 0, 2, 0, true, 36>
                : (const void *)&msda_fwd_f32_pquad<FUSED, 0, 2, 0, false, 36>;
}
template <bool FUSED>
const void *pq_kernel(int npass, bool wide)
{
    return wide ? pq_kernel_n<FUSED, true>(npass) : pq_kernel_n<FUSED, false>(npass);
}

}  // namespace

namespace tfm {

bool launch_pquad(bool fused, const DirectArgs &da, const LevelTable &lt, int N, int D, int P, hipStream_t stream,
                  hipError_t *err)
{
    if (da.Lq != da.S || (D != 32 && D != 36) || P != 4 || da.L > kPqLevels) return false;
    if (fused && da.fa.ref_dim != 2) return false;
    // the kernel addresses loc / attn / qproj / ref / out with 32-bit byte offsets
    if (fused && (long long)N * da.Lq * da.fa.ld * 4 >= (1LL << 32)) return false;
    if ((long long)N * da.Lq * da.M * da.L * 4 * 8 >= (1LL << 32)) return false;
    PqPlan pl;
    if (!pq_plan(lt, da.L, da.M, N, D, &pl)) return false;
    if (D == 36 && (pl.npass != 2 || pl.pf != 0)) return false;
    long long grid = (long long)pq_num_cus() * pl.wgs;
    if (grid > pl.geom.n_items) grid = pl.geom.n_items;
    // 16-byte loads of the points need 16-byte aligned rows
    bool wide = pl.wide;
    if (fused)
        wide = wide && ((uintptr_t)da.fa.qproj % 16 == 0) && da.fa.ld % 4 == 0 && da.fa.off_col % 4 == 0 &&
               da.fa.logit_col % 4 == 0;
    else
        wide = wide && ((uintptr_t)da.loc % 16 == 0) && ((uintptr_t)da.attn % 16 == 0);
    const bool v2 = pl.v2 && D == 32 && wide && pl.ta_mask == 0 && ((pl.waves == 4 && pl.npass <= 2) || (pl.waves == 8 && pl.npass == 1));
    if (pl.waves == 8 && !v2) return false;   // (the plan was made for eight-wave workgroups: only version 2 has them)
    const bool cf = v2 && pl.geom.cf != 0;
    const void *fn = cf ? (pl.waves == 8 ? (fused ? (const void *)&msda_fwd_f32_pquad2<true, 8, 1, true> : (const void *)&msda_fwd_f32_pquad2<false, 8, 1, true>)
                                         : (fused ? (const void *)&msda_fwd_f32_pquad2<true, 4, 2, true> : (const void *)&msda_fwd_f32_pquad2<false, 4, 2, true>))
                     : v2 ? (pl.waves == 8 ? (fused ? (const void *)&msda_fwd_f32_pquad2<true, 8, 1> : (const void *)&msda_fwd_f32_pquad2<false, 8, 1>)
                           : pl.npass == 1 ? (fused ? (const void *)&msda_fwd_f32_pquad2<true, 4, 1> : (const void *)&msda_fwd_f32_pquad2<false, 4, 1>)
                                           : (fused ? (const void *)&msda_fwd_f32_pquad2<true, 4, 2> : (const void *)&msda_fwd_f32_pquad2<false, 4, 2>))
                     : D == 36 ? (fused ? pq_kernel_d36<true>(wide) : pq_kernel_d36<false>(wide))
                               : (fused ? pq_kernel<true>(pl.npass, wide) : pq_kernel<false>(pl.npass, wide));
    // the dynamic-LDS limit is a per-function, per-device attribute: cheap, set on every first (function, device)
    struct Raised { const void *fn; int dev; };
    static std::atomic<int> n_raised{0};
    static Raised raised[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool known = false;
    const int n = n_raised.load(std::memory_order_acquire);
    for (int i = 0; i < n && i < 64; ++i)
        if (raised[i].fn == fn && raised[i].dev == dev) known = true;
    if (!known) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        static std::atomic_flag lock = ATOMIC_FLAG_INIT;
        while (lock.test_and_set(std::memory_order_acquire)) {}
        const int k = n_raised.load(std::memory_order_relaxed);
        if (k < 64) {
            raised[k] = Raised{fn, dev};
            n_raised.store(k + 1, std::memory_order_release);
        }
        lock.clear(std::memory_order_release);
    }
    pl.geom.trace = g_pq_trace.load(std::memory_order_relaxed);
    void *argv[] = {(void *)&da, (void *)&lt, (void *)&pl.geom};
    *err = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(64 * pl.waves), argv, pl.lds, stream);
    note_kernel(cf ? (pl.waves == 8 ? (fused ? "msda_fwd_f32_pquad2<fused,8w,1p,cf>" : "msda_fwd_f32_pquad2<plain,8w,1p,cf>")
                                    : (fused ? "msda_fwd_f32_pquad2<fused,4w,2p,cf>" : "msda_fwd_f32_pquad2<plain,4w,2p,cf>"))
                : v2 ? (pl.waves == 8 ? (fused ? "msda_fwd_f32_pquad2<fused,8w,1p>" : "msda_fwd_f32_pquad2<plain,8w,1p>")
                      : pl.npass == 1 ? (fused ? "msda_fwd_f32_pquad2<fused,4w,1p>" : "msda_fwd_f32_pquad2<plain,4w,1p>")
                                      : (fused ? "msda_fwd_f32_pquad2<fused,4w,2p>" : "msda_fwd_f32_pquad2<plain,4w,2p>"))
                   : D == 36 ? (fused ? "msda_fwd_f32_pquad<fused,D=36>" : "msda_fwd_f32_pquad<plain,D=36>")
                             : (fused ? "msda_fwd_f32_pquad<fused>" : "msda_fwd_f32_pquad<plain>"));
    return true;
}

int pquad_set_option(const char *name, int value)
{
    pq_opts_init();
    for (int i = 0; i < kPoCount; ++i)
        if (strcmp(name, kPqOptNames[i]) == 0) {
            const int prev = g_pq_opt[i].exchange(value);
            g_pq_epoch.fetch_add(1);
            return prev;
        }
    return -1;
}

void pquad_set_trace(unsigned long long *device_buffer) { g_pq_trace.store(device_buffer); }

}  // namespace tfm
