// trackformer_amd/csrc/msda_bwd_mm.h -- msda_bwd_f32_mm, included by msda_hip.hip (inside its anonymous namespace, behind
// msda_bwd_f32_sorted2, whose argument structs, tile geometry and plan helpers it shares).  gfx950.
//
// Backward of the encoder-shaped call (Lq == S, fp32, D == 32, P == 4, L <= 4; reference: ms_deform_im2col_cuda.cuh:239-378,
// the col2im kernels) as TWO DENSE PRODUCTS ON THE MATRIX CORES AROUND A SPARSE TAP PHASE.
//
// Why (round 5, profiles/r05_msda_bwd_sorted2_ablations.txt): msda_bwd_f32_sorted2 spends 238 us per cfg-2 image of which the
// global atomics are 14 - 30 and the value gathers ~40; ~100 us are the counting sort and the segmented row sums (LDS atomics with
// return, a one-wave prefix sum, five barriers per level) and ~65 the per-tap vector arithmetic -- instruction- and latency-bound
// bookkeeping around very little arithmetic.  Per tile of PAIRS (query, head) pairs and window of R value rows both halves of the
// backward are linear maps through one small matrix each:
//     S[r][p]  = < value row r , grad_out of pair p >            (R x 32) (32 x PAIRS)      -- everything grad_loc / grad_attn need
//     GV[r][c] = sum_p Wt[r][p] grad_out[p][c]                   (R x PAIRS) (PAIRS x 32)   -- grad_value of the window,
// with Wt[r][p] = sum of (bilinear weight x attention) over the taps of pair p that land on row r.  v_mfma_f32_16x16x4_f32
// multiplies fp32 exactly and accumulates in fp32 -- the reference's arithmetic type, no split product -- and a 16 x 16 x 32
// block of S is 8 instructions of one wave where the vector formulation is 64 row gathers and 2048 multiply-adds.  Dense is
// wasteful (a pair touches ~64 of the R rows) and still several times cheaper than the bookkeeping it replaces.
//
// One workgroup = one tile (a 2-D block of level-0 pixels and the queries of the coarser levels under it: <= PAIRS queries) x one
// head; 4 PAIRS threads: thread = (level, pair) -- the four points of one pair at one level.
//   1. tables, the tile's queries, the lane's four points; bounding box of the valid taps per level
//   2. windows: per level the bounding box clamped to the tile's footprint +- halo and to the LDS rows that are left (coarse
//      levels first); row r of the concatenated windows <-> byte offset of its head slice in value / grad_value (s_row)
//   3. S = V_win GO^T: A = value rows straight from global memory (16-byte loads; a row tile per wave at a time, four in flight),
//      B = grad_out of the tile's pairs in registers; S -> LDS [row][pair], row stride PAIRS + 4 floats
//   4. taps: s_t = S[row_t][pair] (one 4-byte LDS read per tap),  grad_attn = sum_t w_t s_t,
//      d/dx = gy (s2 - s1) + fy (s4 - s3),  d/dy = gx (s3 - s1) + fx (s4 - s2)   (cuh:139-160, 365-376) -> global, final
//   5. S := 0;  Wt[row_t][pair] += w_t a  with ds_add_f32 (column `pair` of a level's rows belongs to ONE lane: no two lanes
//      ever add to the same word)
//   6. GV = Wt GO: A = Wt from LDS, B = grad_out in registers; one fp32 global atomic per non-zero element (a lane's 16
//      channels of a row are contiguous: 64-byte pieces)
// Taps outside the windows (rows the LDS has no room for, offsets beyond the halo) take a per-lane path: the dot product and
// the scatter of the tap's row directly in global memory.
constexpr int kMmOffQi = 512;                        // int[64]: query index of the tile's pairs (-1: none)
constexpr int kMmOffRow = 1024;                      // u32[kMmMaxRows]: byte offset of the window rows' head slices
constexpr int kMmMaxRows = 512;
constexpr int kMmOffS = kMmOffRow + 4 * kMmMaxRows;  // float[rows][PAIRS + 4]

constexpr size_t mm_lds_bytes(int pairs, int rows) { return (size_t)kMmOffS + (size_t)rows * (pairs + 4) * 4; }

template <int PAIRS>
__global__ void __launch_bounds__(PAIRS * 4, 2)
msda_bwd_f32_mm(const BwdSortArgs ba, const LevelTable lt, const WinGeom wg)
{
    constexpr int PT = 4, D = 32, NT = PAIRS * 4, NW = NT / 64, NPT = PAIRS / 16, SW = PAIRS + 4;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);   // H[4] | W[4] | start[4]
    int *s_q = s_tab + 12;                        // ya[4] | yb[4] | xa[4] | xb[4]: the tile's queries of every level
    int *s_bb = s_q + 16;                         // [level]: xmin xmax ymin ymax of the valid taps
    int *s_win = s_bb + 16;                       // [level]: wx0 wy0 ww wh base - - (level 0: [7] = rows of all windows)
    int *s_qi = reinterpret_cast<int *>(smem + kMmOffQi);
    unsigned *s_row = reinterpret_cast<unsigned *>(smem + kMmOffRow);
    float *s_S = reinterpret_cast<float *>(smem + kMmOffS);

    const int L = ba.L, M = ba.M, S = ba.S, LP = L * PT;
    const int m = blockIdx.x % M;
    int t = blockIdx.x / M;
    const int tx = t % wg.tiles_x;
    t /= wg.tiles_x;
    const int ty = t % wg.tiles_y;
    const int b = t / wg.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kq = lane >> 4;

    if (tid < 16) {   // exact partition of every level among the tiles (as msda_bwd_f32_sorted2)
        const int l = tid >> 2, k = tid & 3;
        if (l < L) {
            const unsigned H0 = (unsigned)lt.H[0], W0 = (unsigned)lt.W[0];
            const unsigned Hl = (unsigned)lt.H[l], Wl = (unsigned)lt.W[l];
            const unsigned y0 = (unsigned)ty * wg.TH, y1 = min(H0, y0 + (unsigned)wg.TH);
            const unsigned x0 = (unsigned)tx * wg.TW, x1 = min(W0, x0 + (unsigned)wg.TW);
            const unsigned num = k == 0 ? 2u * y0 * Hl + H0 - 1u : k == 1 ? 2u * y1 * Hl + H0 - 1u
                                 : k == 2 ? 2u * x0 * Wl + W0 - 1u : 2u * x1 * Wl + W0 - 1u;
            s_q[k * 4 + l] = (int)(num / (k < 2 ? 2u * H0 : 2u * W0));
            if (k == 0) {
                s_tab[l] = lt.H[l];
                s_tab[4 + l] = lt.W[l];
                s_tab[8 + l] = lt.start[l];
            }
        } else {
            s_q[k * 4 + l] = 0;
            if (k == 0) s_tab[l] = s_tab[4 + l] = 1, s_tab[8 + l] = 0;
        }
    }
    __syncthreads();

    // ---- 1. the tile's queries; this lane = (level lvl, pair): four points
    int qoff[5];
    qoff[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) qoff[l + 1] = qoff[l] + (l < L ? (s_q[4 + l] - s_q[l]) * (s_q[12 + l] - s_q[8 + l]) : 0);
    const int nq = qoff[4];
    const int pair = tid & (PAIRS - 1), lvl = tid / PAIRS;
    const bool live = pair < nq;
    int q = 0;
    if (live) {
        int l = 0, base = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (pair >= qoff[k] && k < L) {
                l = k;
                base = qoff[k];
            }
        const int r = pair - base, nx = s_q[12 + l] - s_q[8 + l];
        const int yy = r / nx, xx = r - yy * nx;
        q = s_tab[8 + l] + (s_q[l] + yy) * s_tab[4 + l] + s_q[8 + l] + xx;
    }
    if (lvl == 0) s_qi[pair] = live ? q : -1;
    const long long pidx = ((long long)b * S + q) * M + m;   // the pair's index in [N, Lq, M]
    const float *gop = ba.grad_out + pidx * D;               // its grad_out row
    const bool have = live && lvl < L;
    const int ml = lvl < L ? lvl : 0;
    const int H = s_tab[ml], W = s_tab[4 + ml];
    const float Wf = (float)W, Hf = (float)H;
    float px[PT], py[PT], pa[PT];
    {
        const float2 *lp = reinterpret_cast<const float2 *>(ba.loc + (pidx * LP + ml * PT) * 2);
        const float *ap = ba.attn + pidx * LP + ml * PT;
        int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float2 xy = have ? lp[p] : float2{0.f, 0.f};
            px[p] = xy.x;
            py[p] = xy.y;
            pa[p] = have ? ap[p] : 0.f;
            const float xr = __builtin_fmaf(xy.x, Wf, -0.5f), yr = __builtin_fmaf(xy.y, Hf, -0.5f);
            const bool in = have && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
            const int x0 = (int)__builtin_floorf(in ? xr : 0.f), y0 = (int)__builtin_floorf(in ? yr : 0.f);
            if (in) {
                mnx = min(mnx, x0 >= 0 ? x0 : x0 + 1);
                mxx = max(mxx, (x0 + 1 <= W - 1) ? x0 + 1 : x0);
                mny = min(mny, y0 >= 0 ? y0 : y0 + 1);
                mxy = max(mxy, (y0 + 1 <= H - 1) ? y0 + 1 : y0);
            }
        }
#pragma unroll
        for (int off = 1; off < PAIRS; off <<= 1) {   // the PAIRS lanes of a level are consecutive lanes of one wave
            mnx = min(mnx, __shfl_xor(mnx, off));
            mxx = max(mxx, __shfl_xor(mxx, off));
            mny = min(mny, __shfl_xor(mny, off));
            mxy = max(mxy, __shfl_xor(mxy, off));
        }
        if (pair == 0) {
            s_bb[4 * lvl + 0] = mnx;
            s_bb[4 * lvl + 1] = mxx;
            s_bb[4 * lvl + 2] = mny;
            s_bb[4 * lvl + 3] = mxy;
        }
    }
    __syncthreads();

    // ---- 2. windows: lanes 0 .. 3 of wave 0 take a level each; LDS rows go to the coarse levels first
    if (wave == 0) {
        const int l = lane;
        int wx0 = 0, wy0 = 0, ww = 1, wh = 0;
        if (l < L) {
            const int Hl = s_tab[l], Wl = s_tab[4 + l];
            const int H0 = s_tab[0], W0 = s_tab[4];
            const float rH0 = 1.f / (float)H0, rW0 = 1.f / (float)W0;
            const int y0t = ty * wg.TH, y1t = min(H0, y0t + wg.TH), x0t = tx * wg.TW, x1t = min(W0, x0t + wg.TW);
            const int bx0 = s_bb[4 * l + 0], bx1 = s_bb[4 * l + 1], by0 = s_bb[4 * l + 2], by1 = s_bb[4 * l + 3];
            const int ny0 = (int)__builtin_floorf((float)y0t * (float)Hl * rH0 - 0.5f) - wg.HY;
            const int ny1 = (int)__builtin_floorf((float)y1t * (float)Hl * rH0 - 0.5f) + 1 + wg.HY;
            const int nx0 = (int)__builtin_floorf((float)x0t * (float)Wl * rW0 - 0.5f) - wg.HX;
            const int nx1 = (int)__builtin_floorf((float)x1t * (float)Wl * rW0 - 0.5f) + 1 + wg.HX;
            wx0 = max(max(bx0, nx0), 0);
            wy0 = max(max(by0, ny0), 0);
            ww = min(min(bx1, nx1), Wl - 1) - wx0 + 1;
            wh = min(min(by1, ny1), Hl - 1) - wy0 + 1;
            if (ww <= 0 || wh <= 0 || bx0 == INT_MAX) {
                ww = 1;
                wh = 0;
            }
        }
        int left = wg.cap_rows;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            if (l == k && wh * ww > left) wh = left / ww;
            left -= __shfl(wh * ww, k);
        }
        int base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rk = __shfl(wh * ww, k);
            if (l > k) base += rk;
            total += rk;
        }
        if (l < 4) {
            s_win[8 * l + 0] = wx0;
            s_win[8 * l + 1] = wy0;
            s_win[8 * l + 2] = ww;
            s_win[8 * l + 3] = wh;
            s_win[8 * l + 4] = base;
        }
        if (l == 0) s_win[7] = total;
    }
    __syncthreads();

    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * S * M + m) * D) * 4);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ba.value), 0, ba.value_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(ba.grad_value, 0, ba.value_bytes, 0x00020000);
    const int RT = s_win[7], RT16 = (RT + 15) & ~15, nrt = RT16 >> 4;
    {
        const int b1 = s_win[8 + 4], b2 = s_win[16 + 4], b3 = s_win[24 + 4];
        for (int r = tid; r < RT16; r += NT) {
            unsigned off = kOobBase;
            if (r < RT) {
                const int l = (r >= b1) + (r >= b2) + (r >= b3);
                const int ww = s_win[8 * l + 2], rr = r - s_win[8 * l + 4];
                const int y = rr / ww, x = rr - y * ww;
                off = head_base + (unsigned)(s_tab[8 + l] + (s_win[8 * l + 1] + y) * s_tab[4 + l] + s_win[8 * l + 0] + x) * rowbytes;
            }
            s_row[r] = off;
        }
    }
    // grad_out of the tile's pairs as the B operand of S = V GO^T: column m16 = pair 16 pt + m16, k = channel 16 tc + 4 kq + e
    f32x4_t gb[NPT][2];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int qq = s_qi[16 * pt + m16];
        const float *g = ba.grad_out + (((long long)b * S + max(qq, 0)) * M + m) * D + 4 * kq;
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            gb[pt][tc] = *reinterpret_cast<const f32x4_t *>(g + 16 * tc);
            if (qq < 0) gb[pt][tc] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    // ---- 3. S = V_win GO^T (row tiles of 16 window rows, wave w takes tiles w, w + NW, ...; four tiles' loads in flight)
    for (int i0 = 0; wave + NW * i0 < nrt; i0 += 4) {
        f32x4_t va[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = wave + NW * (i0 + i);
            const unsigned off = rt < nrt ? s_row[16 * rt + m16] : kOobBase;   // row m16 of the tile, k = channel 16 tc + 4 kq + e
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
                va[i][tc] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, off + (unsigned)(16 * tc + 4 * kq) * 4u, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = wave + NW * (i0 + i);
            if (rt < nrt) {   // wave-uniform
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int tc = 0; tc < 2; ++tc) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i][tc].x, gb[pt][tc].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i][tc].y, gb[pt][tc].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i][tc].z, gb[pt][tc].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i][tc].w, gb[pt][tc].w, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_S[(16 * rt + 4 * kq + r) * SW + 16 * pt + m16] = acc[r];   // C: row 4 kq + r, column m16
                }
            }
        }
    }
    // grad_out as the B operand of GV = Wt GO (k = pair 16 pt + 4 kq + e, column m16 = channel 16 c + m16): requested here, used in 6.
    float gb2[NPT][4][2];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qq = s_qi[16 * pt + 4 * kq + e];
            const float *g = ba.grad_out + (((long long)b * S + max(qq, 0)) * M + m) * D + m16;
            gb2[pt][e][0] = qq >= 0 ? g[0] : 0.f;
            gb2[pt][e][1] = qq >= 0 ? g[16] : 0.f;
        }
    __syncthreads();

    // ---- 4. taps of this lane's four points: grad_loc / grad_attn from S; the grad_value weights stay in registers
    int trow[PT * 4];
    float twt[PT * 4];
    {
        const int wx0 = s_win[8 * ml + 0], wy0 = s_win[8 * ml + 1], ww = s_win[8 * ml + 2], wh = s_win[8 * ml + 3];
        const int wbase = s_win[8 * ml + 4];
        const int wx1 = wx0 + ww - 1, wy1 = wy0 + wh - 1;
        const unsigned lvl_base = head_base + (unsigned)s_tab[8 + ml] * rowbytes;
        float gax[PT], gay[PT], gat[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float a = pa[p];
            const float xr = __builtin_fmaf(px[p], Wf, -0.5f), yr = __builtin_fmaf(py[p], Hf, -0.5f);   // cuh:350-351
            const bool in = have && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);                // cuh:359
            const float x = in ? xr : 0.f, y = in ? yr : 0.f;
            const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
            const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
            const int x0 = (int)xf, y0 = (int)yf;
            const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
            const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
            const bool kt[4] = {ky0 && kx0, ky0 && kx1, ky1 && kx0, ky1 && kx1};
            const float w[4] = {gy * gx, gy * fx, fy * gx, fy * fx};
            float s[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int tx_ = x0 + (tp & 1), ty_ = y0 + (tp >> 1);
                const bool inside = kt[tp] && tx_ >= wx0 && tx_ <= wx1 && ty_ >= wy0 && ty_ <= wy1;
                const int r = wbase + (ty_ - wy0) * ww + (tx_ - wx0);
                const float wt = w[tp] * a;
                s[tp] = inside ? s_S[r * SW + pair] : 0.f;
                trow[p * 4 + tp] = (inside && wt != 0.f) ? r : -1;
                twt[p * 4 + tp] = wt;
                if (kt[tp] && !inside) {   // a tap the windows do not hold: its row straight from / to global memory
                    const unsigned ro = lvl_base + (unsigned)(ty_ * W + tx_) * rowbytes;
                    float dsum = 0.f;
#pragma unroll
                    for (int c4 = 0; c4 < D / 4; ++c4) {
                        const f32x4_t g = *reinterpret_cast<const f32x4_t *>(gop + 4 * c4);
                        const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, ro + 16u * c4, 0, 0));
                        dsum += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
                        if (wt != 0.f) {
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wt * g.x, rsrc_g, ro + 16u * c4, 0, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wt * g.y, rsrc_g, ro + 16u * c4 + 4u, 0, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wt * g.z, rsrc_g, ro + 16u * c4 + 8u, 0, 0);
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wt * g.w, rsrc_g, ro + 16u * c4 + 12u, 0, 0);
                        }
                    }
                    s[tp] = dsum;
                }
            }
            const float dot = (w[0] * s[0] + w[1] * s[1]) + (w[2] * s[2] + w[3] * s[3]);   // cuh:365,376
            const float dx = (s[1] - s[0]) * gy + (s[3] - s[2]) * fy;                      // cuh:150-160
            const float dy = (s[2] - s[0]) * gx + (s[3] - s[1]) * fx;                      // cuh:139-149
            const float ain = in ? a : 0.f;
            gax[p] = dx * ain * Wf;   // cuh:371,373
            gay[p] = dy * ain * Hf;   // cuh:371,374
            gat[p] = dot;             // cuh:376
        }
        if (have) {
            float2 *gl = reinterpret_cast<float2 *>(ba.grad_loc + (pidx * LP + ml * PT) * 2);
            float *ga = ba.grad_attn + pidx * LP + ml * PT;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                gl[p] = float2{gax[p], gay[p]};
                ga[p] = gat[p];
            }
        }
    }
    __syncthreads();
    // ---- 5. S := 0, then the tap weights into Wt (same LDS rows)
    for (int i = tid; i < RT16 * SW / 4; i += NT) reinterpret_cast<f32x4_t *>(s_S)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PT * 4; ++k)
        if (trow[k] >= 0) atomicAdd(&s_S[trow[k] * SW + pair], twt[k]);
    __syncthreads();
    // ---- 6. GV = Wt GO: 16 window rows x 32 channels per row tile; one global atomic per non-zero element
    for (int rt = wave; rt < nrt; rt += NW) {
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const f32x4_t wa = *reinterpret_cast<const f32x4_t *>(s_S + (16 * rt + m16) * SW + 16 * pt + 4 * kq);   // row m16, k = pair 16 pt + 4 kq + e
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, gb2[pt][0][c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, gb2[pt][1][c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, gb2[pt][2][c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, gb2[pt][3][c], acc[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned ro = s_row[16 * rt + 4 * kq + r];   // (rows behind the last window: out of range, dropped by the hardware)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (acc[c][r] != 0.f)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[c][r], rsrc_g, ro + (unsigned)(16 * c + m16) * 4u, 0, 0);
        }
    }
}

// Tile plan of msda_bwd_f32_mm: the 2-D tile with the most queries that still fits `pairs`.
bool plan_mm(const LevelTable &lt, int L, int D, int P, int pairs, int rows, WinGeom *wg)
{
    if (D != 32 || P != 4 || L > kWinLevels || (pairs != 32 && pairs != 64) || rows < 16 || rows > kMmMaxRows) return false;
    for (int l = 0; l < L; ++l)
        if (lt.H[l] >= 32768 || lt.W[l] >= 32768) return false;
    struct Memo {
        bool valid = false, ok = false;
        int L = 0, pairs = 0, rows = 0;
        LevelTable lt;
        WinGeom wg;
    };
    static thread_local Memo memo;
    if (memo.valid && memo.L == L && memo.pairs == pairs && memo.rows == rows && memcmp(&memo.lt, &lt, sizeof(lt)) == 0) {
        *wg = memo.wg;
        return memo.ok;
    }
    memo.valid = true;
    memo.ok = false;
    memo.L = L;
    memo.pairs = pairs;
    memo.rows = rows;
    memo.lt = lt;
    int hy = 5, hx = 5, th = 0, tw = 0;
    if (const char *e = getenv("TF_MSDA_BWD_MM_HALO")) sscanf(e, "%d,%d", &hy, &hx);
    if (const char *e = getenv("TF_MSDA_BWD_MM_TILE")) sscanf(e, "%d,%d", &th, &tw);
    if (hy < 0 || hx < 0 || th < 0 || tw < 0) return false;
    long long best = 0;
    int bth = 0, btw = 0;
    for (int ctw = tw ? tw : 16; ctw >= (tw ? tw : 1); --ctw)
        for (int cth = th ? th : 16; cth >= (th ? th : 1); --cth) {
            const long long nq = tile_max_queries(lt, L, cth, ctw);
            // the most queries; among equals the squarer tile (smaller windows)
            if (nq >= 1 && nq <= pairs && (nq > best || (nq == best && abs(cth - ctw) < abs(bth - btw)))) {
                best = nq;
                bth = cth;
                btw = ctw;
            }
        }
    if (!best) return false;
    wg->TH = bth;
    wg->TW = btw;
    wg->HY = hy;
    wg->HX = hx;
    wg->tiles_y = (lt.H[0] + bth - 1) / bth;
    wg->tiles_x = (lt.W[0] + btw - 1) / btw;
    wg->cap_rows = rows & ~15;
    memo.wg = *wg;
    memo.ok = true;
    return true;
}

// 0: msda_bwd_f32_sorted2; 32 / 64: msda_bwd_f32_mm with that many pairs per tile.  -1: TF_MSDA_BWD_MM or the default.
std::atomic<int> g_bwd_mm{-1};
std::atomic<int> g_bwd_mm_rows{-1};

int bwd_mm_pairs()
{
    int v = g_bwd_mm.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_MSDA_BWD_MM");
        v = e ? atoi(e) : 32;
        if (v != 0 && v != 32 && v != 64) v = 32;
        g_bwd_mm.store(v);
    }
    return v;
}

int bwd_mm_rows()
{
    int v = g_bwd_mm_rows.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_MSDA_BWD_MM_ROWS");
        v = e ? atoi(e) : 0;
        g_bwd_mm_rows.store(v);
    }
    return v;   // 0: by the number of pairs (256)
}
