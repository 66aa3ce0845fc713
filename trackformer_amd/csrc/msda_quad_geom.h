// trackformer_amd/csrc/msda_quad_geom.h
//
// Tile / window geometry of msda_fwd_f32_quad (the LDS-window encoder kernel with 4 lanes per
// (query, head) pair).  Plain C++ so that the same functions are compiled into the HIP kernel and into
// the host-side emulation that tests/test_quad_emulation.py checks against the oracle
// (tests/emu/quad_emu.cpp): the data-dependent logic of the kernel is testable without a GPU.
//
// Coordinates.  A sampling point of level l (H x W pixels) that is in range (cuh:227-229:
// -1 < x < W, -1 < y < H) has floor coordinates x0 in [-1, W-1], y0 in [-1, H-1]; its four bilinear
// taps are the pixels (y0 + {0,1}, x0 + {0,1}), some of which may lie OUTSIDE the level (the reference
// skips those: zero padding).  Windows are therefore described in EXTENDED pixel coordinates
// -1 .. W (and -1 .. H): pixels outside the level are staged as zeros (the LDS-DMA source offset is out
// of range, the hardware writes 0), so a staged point needs no per-tap validity test at all.
#ifndef TF_MSDA_QUAD_GEOM_H_
#define TF_MSDA_QUAD_GEOM_H_

#include <limits.h>

#if defined(__HIPCC__)
#define TFQ_HD __host__ __device__ __forceinline__
#else
#define TFQ_HD inline
#endif

// One level's window.  An empty window has ww == wh == 0, limx == limy == 0 and wx0 == wy0 == kQuadFar,
// which makes the unsigned comparisons of tfq_staged() fail for every point.
constexpr int kQuadFar = 1 << 30;
struct QuadWindow {
    int wx0, wy0;     // first column / row, extended coordinates (>= -1)
    int ww, wh;       // width / height in pixels
    int limx, limy;   // ww - 2, wh - 2: largest x0 - wx0 / y0 - wy0 of a staged point
    int roff;         // LDS row (128 B) of the window's first pixel
};

TFQ_HD int tfq_min(int a, int b) { return a < b ? a : b; }
TFQ_HD int tfq_max(int a, int b) { return a > b ? a : b; }

// First pixel index of level l (size_l pixels along this axis) whose centre (i + 0.5) / size_l is
// >= edge0 / size_0, i.e. lies at or after the level-0 tile edge `edge0`.  Integer exact, so that the
// tiles partition every level.  All products must stay below 2^31 (host checks sizes < 32768).
TFQ_HD int tfq_tile_bound(unsigned edge0, unsigned size_l, unsigned size_0)
{
    return (int)((2u * edge0 * size_l + size_0 - 1u) / (2u * size_0));
}

// Nominal footprint of the level-0 interval [e0, e1) in a level of `size_l` pixels, widened by `halo`
// and clipped to the extended range.  Only a clamp for the data-adaptive box: precision is irrelevant
// (float arithmetic; the kernel evaluates this once per level and wave).
TFQ_HD void tfq_nominal(int e0, int e1, int size_l, float inv_size_0, int halo, int *lo, int *hi)
{
    const float s = (float)size_l * inv_size_0;
    const int fa = (int)__builtin_floorf((float)e0 * s - 0.5f);
    const int fb = (int)__builtin_floorf((float)e1 * s - 0.5f);
    *lo = tfq_max(fa - halo, -1);
    *hi = tfq_min(fb + 1 + halo, size_l);
}

// Window of one level: the bounding box [bx0, bx1] x [by0, by1] of the floor coordinates (x0, y0) of
// the tile's in-range points (bx0 > bx1: there are none), widened by the +1 taps and clamped to the nominal
// footprint [nx0, nx1] x [ny0, ny1].  All or nothing: a window that needs more than `avail_rows` LDS rows
// is not staged at all (*fits = false) and the caller gathers the whole level by buffer loads -- a partly
// staged level would make every wave with a point outside it execute both paths.
TFQ_HD QuadWindow tfq_window(int bx0, int bx1, int by0, int by1, int nx0, int nx1, int ny0, int ny1,
                             int avail_rows, int roff, bool *fits)
{
    QuadWindow q;
    q.wx0 = kQuadFar;
    q.wy0 = kQuadFar;
    q.ww = 0;
    q.wh = 0;
    q.limx = 0;
    q.limy = 0;
    q.roff = roff;
    *fits = true;
    if (bx0 > bx1 || by0 > by1) return q;
    // written so that no input, however wild, overflows: wx0 >= nx0, wx1 <= nx1, and the sizes are only
    // formed once wx0 <= wx1 is known (the nominal footprint is a few hundred pixels at most)
    const int wx0 = tfq_max(bx0, nx0), wx1 = tfq_min(bx1, nx1 - 1) + 1;
    const int wy0 = tfq_max(by0, ny0), wy1 = tfq_min(by1, ny1 - 1) + 1;
    if (wx0 >= wx1 || wy0 >= wy1) return q;
    const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
    if (wh * ww > avail_rows) {
        *fits = false;
        return q;
    }
    q.wx0 = wx0;
    q.wy0 = wy0;
    q.ww = ww;
    q.wh = wh;
    q.limx = ww - 2;
    q.limy = wh - 2;
    return q;
}

// All four taps of a point with floor coordinates (x0, y0) lie inside the window.
TFQ_HD bool tfq_staged(const QuadWindow &q, int x0, int y0)
{
    return (unsigned)(x0 - q.wx0) <= (unsigned)q.limx && (unsigned)(y0 - q.wy0) <= (unsigned)q.limy;
}

// LDS row of the tap (y0, x0) of a staged point; (y0, x0 + 1) is the next row, (y0 + 1, .) are ww further.
TFQ_HD int tfq_row(const QuadWindow &q, int x0, int y0)
{
    return q.roff + (y0 - q.wy0) * q.ww + (x0 - q.wx0);
}

#endif  // TF_MSDA_QUAD_GEOM_H_
