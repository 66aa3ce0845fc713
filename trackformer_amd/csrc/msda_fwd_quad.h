// trackformer_amd/csrc/msda_fwd_quad.h -- included by msda_hip.hip inside its anonymous namespace.
//
// msda_fwd_f32_quad: MSDeformAttn forward for encoder-shaped calls (Lq == S), fp32, D == 32, P == 4, L <= 4.
// THE DEFAULT for those calls (TF_MSDA_TILED unset): 36 us at the cfg-2 encoder shape against 52 us for
// msda_fwd_f32_direct, HBM traffic 1.15x the algorithmic bytes (profiles/r01_msda_fwd_quad_*).
//
// Why another kernel.  msda_fwd_f32_direct moves 64 taps x 128 B per (query, head) pair through the
// vector-memory path (64 B/clk/CU: the texture addresser is 85 % busy), msda_fwd_f32_win moves them
// through LDS (256 B/clk/CU) but pays for it in instructions: 8 lanes per pair need the tap offsets and
// weights of all 16 points of the pair, exchanged through LDS (1/3 of its LDS wave-instructions, 27 KB of
// LDS), and every window tap carries its own validity test (1313 VALU instructions per wave, PMC).
// This kernel removes both and fixes the occupancy:
//   * 4 lanes per pair (a DPP quad), 8 channels per lane.  Lane j of the quad owns point j of every level
//     and computes its tap arithmetic ONCE; the other three lanes read the result with DPP quad_perm
//     broadcasts that the compiler folds into the consuming v_add_u32 (addresses) or issues as one
//     v_mov_b32_dpp per tap weight -- no LDS exchange, no fences, no exchange buffers.
//   * Windows live in EXTENDED pixel coordinates (-1 .. size): pixels outside the level are staged as
//     zeros by the LDS-DMA itself (out-of-range source offset), so a staged point is one box test and two
//     row addresses (y0 and y0 + 1; the x0 + 1 taps are the +128 B immediates), not four guarded ones.
//   * ROUND0 splits the staging: level 0's window first, then the windows of levels 1..3 in the SAME LDS rows.
//     All taps go through LDS in 40 KB per workgroup (4 workgroups per CU) instead of 65 KB (2 per CU).
//   * A window that does not fit is not staged at all and its level is gathered by buffer loads (before the
//     barrier, while the other windows land); so is any level in TA_MASK (compile time: the texture path and
//     the LDS pipe are separate resources; measured, the all-LDS split is faster).  Points whose taps leave
//     their (clamped) window take buffer loads under a wave-uniform branch: any input is handled exactly.
//   * A lane reads its 8 channels as two 16-byte pieces from opposite 64-byte halves of the row, the
//     half order alternating with bit 1 of the quad index: the four quads an LDS cycle serves then hit
//     four disjoint bank groups when neighbouring queries read neighbouring rows (the encoder's pattern).
// The tile / window geometry is in msda_quad_geom.h, shared with the host emulation (tests/emu/quad_emu.cpp,
// tests/test_quad_emulation.py) that checks this logic against the oracle on the CPU.
//
// Arithmetic: SURVEY.md Appendix A; reference ms_deform_im2col_cuda.cuh:227-237 (pixel mapping, in-range
// rule), :24-67 (bilinear taps with zero padding).

constexpr int kQuadHdrBytes = 768;   // level table | query partition | bounding boxes | nominal footprints
constexpr int kQuadLevels = 4;

struct QuadGeom {
    int TH, TW;        // tile size in level-0 pixels
    int HY, HX;        // windows are clamped to the tile footprint +- this many pixels
    int tiles_y, tiles_x;
    int cap_rows;      // LDS rows available for windows, all levels together (multiple of 8)
    unsigned long long *trace;   // debug: 16 timestamps (s_memrealtime, 100 MHz) per workgroup, or null
};

// (DPP helpers, quad_taps_lds / quad_taps_global, ldg_f / ldg_f2: msda_quad_dev.h)

// ROUND0: the levels whose windows are staged (and gathered) first; the remaining LDS levels reuse the same
// LDS rows in a second round (0xF: one round, all windows resident at once).
template <bool FUSED, int TA_MASK, int WAVES, int NPASS, int ROUND0>
__global__ void __launch_bounds__(WAVES * 64, NPASS == 1 ? 6 : NPASS == 2 ? 5 : 4)
msda_fwd_f32_quad(const DirectArgs da, const LevelTable lt, const QuadGeom qg)
{
    constexpr int PT = 4, D = 32, NL = kQuadLevels, THREADS = WAVES * 64, PAIRS = THREADS / 4;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);       // H | W | start                          (3 x 16 ints)
    int *s_q = s_tab + 3 * TF_MSDA_MAX_LEVELS;        // ya | yb | xa | xb per level             (4 x 16 ints)
    int *s_bb = s_q + 4 * TF_MSDA_MAX_LEVELS;         // min x0, max x0, min y0, max y0 per level (16 ints)
    int *s_nom = s_bb + 16;                           // nominal ny0 | ny1 | nx0 | nx1 per level  (4 x 16 ints)
    unsigned char *s_rows = smem + kQuadHdrBytes;     // rows 0, 1: zeros; the windows start at row 2

    const int L = da.L, M = da.M, S = da.S, LP = L * PT;
    const int m = blockIdx.x % M;                     // one head per XCD when M == 8
    int t = blockIdx.x / M;
    const int tx = t % qg.tiles_x;
    t /= qg.tiles_x;
    const int ty = t % qg.tiles_y;
    const int b = t / qg.tiles_y;

    if (threadIdx.x < 4 * NL) {
        // thread 4l+k: bound k (ya, yb, xa, xb) of the pixels of level l whose centre lies in the tile's
        // normalised rectangle, and bound k of the tile's nominal footprint in level l (window clamp)
        const int l = threadIdx.x >> 2, k = threadIdx.x & 3;
        if (l < L) {
            const unsigned H0 = (unsigned)lt.H[0], W0 = (unsigned)lt.W[0];
            const unsigned Hl = (unsigned)lt.H[l], Wl = (unsigned)lt.W[l];
            const unsigned y0 = (unsigned)ty * qg.TH, y1 = min(H0, y0 + (unsigned)qg.TH);
            const unsigned x0 = (unsigned)tx * qg.TW, x1 = min(W0, x0 + (unsigned)qg.TW);
            s_q[k * TF_MSDA_MAX_LEVELS + l] = k == 0   ? tfq_tile_bound(y0, Hl, H0)
                                              : k == 1 ? tfq_tile_bound(y1, Hl, H0)
                                              : k == 2 ? tfq_tile_bound(x0, Wl, W0)
                                                       : tfq_tile_bound(x1, Wl, W0);
            int lo, hi;
            if (k < 2)
                tfq_nominal((int)y0, (int)y1, (int)Hl, 1.f / (float)H0, qg.HY, &lo, &hi);
            else
                tfq_nominal((int)x0, (int)x1, (int)Wl, 1.f / (float)W0, qg.HX, &lo, &hi);
            s_nom[k * TF_MSDA_MAX_LEVELS + l] = (k & 1) ? hi : lo;
            if (k == 0) {
                s_tab[l] = lt.H[l];
                s_tab[TF_MSDA_MAX_LEVELS + l] = lt.W[l];
                s_tab[2 * TF_MSDA_MAX_LEVELS + l] = lt.start[l];
            }
        }
        s_bb[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
    }
    if (threadIdx.x < 64) reinterpret_cast<float *>(s_rows)[threadIdx.x] = 0.f;
    // phase timestamps of wave 0 (tools/msda_bench --trace): where a workgroup's time goes
    auto stamp = [&](int i) {
        if (qg.trace != nullptr && threadIdx.x == 0)
            qg.trace[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    __syncthreads();
    stamp(1);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = threadIdx.x & 3;
    const int quad = threadIdx.x >> 2;
    const int hsel = (quad >> 1) & 1;
    const unsigned rbA = (unsigned)(hsel * 64 + sub * 16), rbB = (unsigned)((1 - hsel) * 64 + sub * 16);
    const unsigned lds_rows =
        (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)s_rows);
    const unsigned ldsA = lds_rows + rbA, ldsB = lds_rows + rbB;

    int Hs[NL], Ws[NL], starts[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int lc = l < L ? l : 0;
        Hs[l] = __builtin_amdgcn_readfirstlane(s_tab[lc]);
        Ws[l] = __builtin_amdgcn_readfirstlane(s_tab[TF_MSDA_MAX_LEVELS + lc]);
        starts[l] = __builtin_amdgcn_readfirstlane(s_tab[2 * TF_MSDA_MAX_LEVELS + lc]);
    }
    int qoff[NL + 1];   // first tile-local index of every level's queries
    qoff[0] = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l)
        qoff[l + 1] = qoff[l] + (l < L ? (s_q[TF_MSDA_MAX_LEVELS + l] - s_q[l]) *
                                             (s_q[3 * TF_MSDA_MAX_LEVELS + l] - s_q[2 * TF_MSDA_MAX_LEVELS + l])
                                       : 0);
    const int nq = qoff[NL];   // <= NPASS * PAIRS (the host checked the maximum)

    // ---- the query of this lane's quad in every pass; the sampling points (l, sub) it owns --------
    unsigned bq32[NPASS], pair32[NPASS];   // b * S + q, (b * S + q) * M + m: byte offsets fit 32 bits (host checked)
    bool live[NPASS];
    float sx[NPASS][NL], sy[NPASS][NL], sa[NPASS][NL];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int tq = ps * PAIRS + quad;
        int q = 0;
        live[ps] = tq < nq;
        if (live[ps]) {
            int l = 0, base = 0;
#pragma unroll
            for (int k = 1; k < NL; ++k)
                if (tq >= qoff[k] && k < L) {
                    l = k;
                    base = qoff[k];
                }
            const int r = tq - base;   // < NPASS * PAIRS <= 512
            const int nx = s_q[3 * TF_MSDA_MAX_LEVELS + l] - s_q[2 * TF_MSDA_MAX_LEVELS + l];
            // r / nx: (r + 0.5) / nx is at least 0.5 / nx away from an integer, far more than the float error
            const int yy = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nx));
            const int xx = r - yy * nx;
            q = s_tab[2 * TF_MSDA_MAX_LEVELS + l] + (s_q[l] + yy) * s_tab[TF_MSDA_MAX_LEVELS + l] +
                s_q[2 * TF_MSDA_MAX_LEVELS + l] + xx;
        }
        const unsigned bq = (unsigned)b * (unsigned)S + (unsigned)q;
        bq32[ps] = bq;
        pair32[ps] = bq * (unsigned)M + (unsigned)m;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const unsigned s = (unsigned)((l < L ? l : 0) * PT + sub);
            if constexpr (!FUSED) {
                const float2 xy = ldg_f2(da.loc, (pair32[ps] * (unsigned)LP + s) * 8u);
                sx[ps][l] = xy.x;
                sy[ps][l] = xy.y;
                sa[ps][l] = ldg_f(da.attn, (pair32[ps] * (unsigned)LP + s) * 4u);
            } else {
                const unsigned row = bq * (unsigned)da.fa.ld;
                const float2 off = ldg_f2(da.fa.qproj, (row + (unsigned)da.fa.off_col + ((unsigned)(m * LP) + s) * 2u) * 4u);
                sx[ps][l] = off.x;
                sy[ps][l] = off.y;
                sa[ps][l] = l < L ? ldg_f(da.fa.qproj, (row + (unsigned)da.fa.logit_col + (unsigned)(m * LP) + s) * 4u)
                                  : -__builtin_inff();
            }
        }
    }
    stamp(12);   // point loads issued
    if constexpr (FUSED) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            // softmax over the pair's L*P logits: this lane holds L of them, the quad the rest
            float mx = sa[ps][0];
#pragma unroll
            for (int l = 1; l < NL; ++l) mx = fmaxf(mx, sa[ps][l]);
            mx = fmaxf(mx, dpp_f<kDppQuadXor1>(mx));
            mx = fmaxf(mx, dpp_f<kDppQuadXor2>(mx));
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                sa[ps][l] = l < L ? __expf(sa[ps][l] - mx) : 0.f;
                sum += sa[ps][l];
            }
            sum += dpp_f<kDppQuadXor1>(sum);
            sum += dpp_f<kDppQuadXor2>(sum);
            const unsigned bq = bq32[ps];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                sa[ps][l] = sa[ps][l] / sum;
                const int lc = l < L ? l : 0;
                const float *rp = da.fa.ref + ((size_t)bq * L + lc) * da.fa.ref_dim;
                if (da.fa.ref_dim == 2) {
                    sx[ps][l] = rp[0] + sx[ps][l] / (float)Hs[l];   // x / H_l (as the reference writes it)
                    sy[ps][l] = rp[1] + sy[ps][l] / (float)Ws[l];   // y / W_l
                } else {
                    sx[ps][l] = rp[0] + sx[ps][l] / (float)PT * rp[2] * 0.5f;
                    sy[ps][l] = rp[1] + sy[ps][l] / (float)PT * rp[3] * 0.5f;
                }
            }
        }
    }

    // ---- phase A: bounding box of the floor coordinates of the in-range points, per LDS level ---------
    auto phase_a = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if constexpr (((TA_MASK >> l) & 1) == 0) {
            if (l < L) {
                int mnx = INT_MAX, mxx = INT_MIN, mny = INT_MAX, mxy = INT_MIN;
                const float Wf = (float)Ws[l], Hf = (float)Hs[l];
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const float xr = __builtin_fmaf(sx[ps][l], Wf, -0.5f);
                    const float yr = __builtin_fmaf(sy[ps][l], Hf, -0.5f);
                    const bool in = live[ps] && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
                    const int x0 = (int)__builtin_floorf(in ? xr : 0.f), y0 = (int)__builtin_floorf(in ? yr : 0.f);
                    mnx = min(mnx, in ? x0 : INT_MAX);
                    mxx = max(mxx, in ? x0 : INT_MIN);
                    mny = min(mny, in ? y0 : INT_MAX);
                    mxy = max(mxy, in ? y0 : INT_MIN);
                }
                // every lane of the wave holds points of level l: reduce over the 16 lanes of a DPP row, then
                // one LDS atomic per row and bound (the compiler folds the four rows of a wave into one)
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
                if ((lane & 15) == 0 && mnx != INT_MAX) {
                    atomicMin(&s_bb[4 * l + 0], mnx);
                    atomicMax(&s_bb[4 * l + 1], mxx);
                    atomicMin(&s_bb[4 * l + 2], mny);
                    atomicMax(&s_bb[4 * l + 3], mxy);
                }
            }
        }
    };
    phase_a(std::integral_constant<int, 0>{});
    stamp(13);   // level 0's points arrived, its bounding box filed
    phase_a(std::integral_constant<int, 1>{});
    phase_a(std::integral_constant<int, 2>{});
    phase_a(std::integral_constant<int, 3>{});
    stamp(2);   // the sampling points have arrived, bounding boxes filed
    __syncthreads();
    stamp(3);

    // ---- phase B: window geometry (wave-uniform, kept in scalar registers) and LDS-DMA staging ---------
    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * S * M + m) * D) * 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(da.value), 0, da.value_bytes, 0x00020000);
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
        {
            if constexpr (((TA_MASK >> l) & 1) == 0 && ((RMASK >> l) & 1) != 0) {
                if (l < L) {
                    const int H = Hs[l], W = Ws[l];
                    const int bx0 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 0]);
                    const int bx1 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 1]);
                    const int by0 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 2]);
                    const int by1 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 3]);
                    const int ny0 = __builtin_amdgcn_readfirstlane(s_nom[l]);
                    const int ny1 = __builtin_amdgcn_readfirstlane(s_nom[TF_MSDA_MAX_LEVELS + l]);
                    const int nx0 = __builtin_amdgcn_readfirstlane(s_nom[2 * TF_MSDA_MAX_LEVELS + l]);
                    const int nx1 = __builtin_amdgcn_readfirstlane(s_nom[3 * TF_MSDA_MAX_LEVELS + l]);
                    bool fits;
                    const QuadWindow w = tfq_window(bx0, bx1, by0, by1, nx0, nx1, ny0, ny1, qg.cap_rows - used, 2 + used, &fits);
                    by_loads[l] = !fits;
                    const int ww = w.ww, wh = w.wh, wx0 = w.wx0, wy0 = w.wy0;
                    const int roff = 2 + used;
                    gwx0[l] = wx0;
                    gwy0[l] = wy0;
                    glimx[l] = w.limx;
                    glimy[l] = w.limy;
                    gww[l] = ww;
                    groff[l] = roff;
                    const int nrows = wh * ww;
                    const int nchunks = (nrows + 7) >> 3;   // one DMA wave-instruction = 8 rows of 128 B
                    used += nchunks * 8;
                    if (nchunks > 0) {
                        // lane (row r = 8 * chunk + lane / 8, 16-byte column lane % 8): window pixel (wy, wx) of
                        // its row, advanced incrementally by 8 * WAVES rows per iteration
                        const unsigned lvl_base = glvl[l];
                        const float inv_ww = __builtin_amdgcn_rcpf((float)ww);
                        int r = wave * 8 + (lane >> 3);                       // < 64
                        int wy = (int)(((float)r + 0.5f) * inv_ww);
                        int wx = r - wy * ww;
                        constexpr int STEP = 8 * WAVES;
                        const int qstep = __builtin_amdgcn_readfirstlane((int)(((float)STEP + 0.5f) * inv_ww));
                        const int rstep = STEP - qstep * ww;
                        unsigned off = lvl_base + (unsigned)((wy0 + wy) * W + wx0 + wx) * rowbytes + (unsigned)(lane & 7) * 16u;
                        const unsigned step_a = (unsigned)(qstep * W + rstep) * rowbytes;
                        const unsigned step_b = (unsigned)(W - ww) * rowbytes;
                        for (int c = wave; c < nchunks; c += WAVES) {
                            const int py = wy0 + wy, px = wx0 + wx;   // extended coordinates: may be -1 or size
                            const bool ok = r < nrows && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
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
                }
            }
        }
    };

    // ---- phase C: gather.  Levels served by buffer loads first (they do not need the windows: the DMA is
    // still landing), then -- behind the barrier -- the levels served from LDS --------------------------
    f32x4_t accA[NPASS], accB[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        accA[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        accB[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    auto level = [&](auto lc, auto psc, auto ldsc) {
        constexpr int l = decltype(lc)::value;
        constexpr int ps = decltype(psc)::value < NPASS ? decltype(psc)::value : 0;   // (never called out of range)
        constexpr bool LDS_PHASE = decltype(ldsc)::value;
        constexpr bool TA = ((TA_MASK >> l) & 1) != 0;
        if (l >= L) return;                          // uniform
        if constexpr (TA && LDS_PHASE) return;
        if (!TA && by_loads[l] == LDS_PHASE) return;   // uniform: a level runs in exactly one of the two phases
        if (ps * PAIRS >= nq) return;                // uniform
        const int H = Hs[l], W = Ws[l];
        const float Wf = (float)W, Hf = (float)H;
        const float xr = __builtin_fmaf(sx[ps][l], Wf, -0.5f);   // cuh:227-228, single rounding
        const float yr = __builtin_fmaf(sy[ps][l], Hf, -0.5f);
        const bool in = live[ps] && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
        const float x = in ? xr : 0.f, y = in ? yr : 0.f;
        const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
        const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
        const int x0 = (int)xf, y0 = (int)yf;
        const float a = in ? sa[ps][l] : 0.f;
        const float w[4] = {gy * gx * a, gy * fx * a, fy * gx * a, fy * fx * a};
        bool need_global = in;
        if constexpr (LDS_PHASE) {
            const int dx = x0 - gwx0[l], dy = y0 - gwy0[l];
            const bool staged = in && (unsigned)dx <= (unsigned)glimx[l] && (unsigned)dy <= (unsigned)glimy[l];
            const unsigned lo = (unsigned)(groff[l] + dy * gww[l] + dx) * 128u;
            const unsigned a0 = staged ? lo : 0u;                            // rows 0, 1 are zeros
            const unsigned a1 = staged ? lo + (unsigned)gww[l] * 128u : 0u;
            quad_taps_lds<0>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
            quad_taps_lds<1>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
            quad_taps_lds<2>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
            quad_taps_lds<3>(a0, a1, w, ldsA, ldsB, accA[ps], accB[ps]);
            need_global = in && !staged;
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
        quad_taps_global<0>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
        quad_taps_global<1>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
        quad_taps_global<2>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
        quad_taps_global<3>(rsrc, g, w, rbA, rbB, accA[ps], accB[ps]);
    };
    // one round: stage the windows of the levels in RMASK, gather the levels of the round that go by buffer
    // loads while the DMA lands (round 0 also takes the TA_MASK levels), then the levels served from LDS
    auto store = [&](auto psc) {
        constexpr int ps = decltype(psc)::value < NPASS ? decltype(psc)::value : 0;   // (never called out of range)
        if (live[ps]) {
            float *o = reinterpret_cast<float *>(reinterpret_cast<char *>(da.out) + (size_t)(pair32[ps] * (unsigned)(D * 4)));
            *reinterpret_cast<f32x4_t *>(o + rbA / 4) = accA[ps];
            *reinterpret_cast<f32x4_t *>(o + rbB / 4) = accB[ps];
        }
    };
    auto round = [&](auto rc, auto firstc, auto lastc) {
        constexpr int RMASK = decltype(rc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        constexpr bool LAST = decltype(lastc)::value;
        constexpr int LOADS_MASK = FIRST ? (RMASK | TA_MASK) : (RMASK & ~TA_MASK);
        if constexpr (!FIRST) __syncthreads();   // every wave is done reading the previous round's windows
        used = 0;
        phase_b(std::integral_constant<int, 0>{}, rc);
        phase_b(std::integral_constant<int, 1>{}, rc);
        phase_b(std::integral_constant<int, 2>{}, rc);
        phase_b(std::integral_constant<int, 3>{}, rc);
        stamp(FIRST ? 4 : 8);   // DMA issued
        auto levels = [&](auto psc, auto ldsc) {
            constexpr int MASK = decltype(ldsc)::value ? (RMASK & ~TA_MASK) : LOADS_MASK;
            if constexpr (MASK & 1) level(std::integral_constant<int, 0>{}, psc, ldsc);
            if constexpr (MASK & 2) level(std::integral_constant<int, 1>{}, psc, ldsc);
            if constexpr (MASK & 4) level(std::integral_constant<int, 2>{}, psc, ldsc);
            if constexpr (MASK & 8) level(std::integral_constant<int, 3>{}, psc, ldsc);
        };
        levels(std::integral_constant<int, 0>{}, std::false_type{});
        if constexpr (NPASS > 1) levels(std::integral_constant<int, 1>{}, std::false_type{});
        if constexpr (NPASS > 2) levels(std::integral_constant<int, 2>{}, std::false_type{});
        stamp(FIRST ? 5 : 9);   // gathers by buffer loads done
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA landed
        __syncthreads();                      // ... everybody's
        stamp(FIRST ? 6 : 10);
        // a pass's outputs are stored as soon as its last level is summed: the stores drain under the
        // gathers of the next pass instead of all at the end
        levels(std::integral_constant<int, 0>{}, std::true_type{});
        if constexpr (LAST) store(std::integral_constant<int, 0>{});
        if constexpr (NPASS > 1) levels(std::integral_constant<int, 1>{}, std::true_type{});
        if constexpr (NPASS > 1 && LAST) store(std::integral_constant<int, 1>{});
        if constexpr (NPASS > 2) levels(std::integral_constant<int, 2>{}, std::true_type{});
        if constexpr (NPASS > 2 && LAST) store(std::integral_constant<int, 2>{});
        stamp(FIRST ? 7 : 11);   // gathers from LDS done
    };
    constexpr int R1 = (0xF & ~ROUND0) & ~TA_MASK;   // the second round's levels
    if constexpr (R1 == 0) {
        round(std::integral_constant<int, ROUND0 & 0xF>{}, std::true_type{}, std::true_type{});
    } else {
        bool any = false;   // uniform: does any of the second round's levels exist?
#pragma unroll
        for (int l = 0; l < NL; ++l)
            if ((R1 >> l) & 1) any = any || (l < L);
        if (any) {
            round(std::integral_constant<int, ROUND0 & 0xF>{}, std::true_type{}, std::false_type{});
            round(std::integral_constant<int, R1>{}, std::false_type{}, std::true_type{});
        } else {
            round(std::integral_constant<int, ROUND0 & 0xF>{}, std::true_type{}, std::true_type{});
        }
    }
}
