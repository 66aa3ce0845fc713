// trackformer_amd/csrc/msda_pquad2.h -- msda_fwd_f32_pquad2: the persistent encoder forward kernel, second version.
// Included by msda_pquad.hip inside its anonymous namespace (shares PquadGeom, the tile plan, the launch code and the
// gather helpers of msda_quad_dev.h).  D == 32, P == 4, L <= 4, 16-byte aligned inputs, two passes of 64 (query, head)
// pairs per tile -- the cfg-2 / cfg-3 / cfg-5 encoder; everything else stays on msda_fwd_f32_pquad.
//
// Why a second version.  Four rounds of measurements say the first version's time is the ~3700
// instructions a wave executes per tile (without ANY gather it still takes 35 of its 45 us), of which the gathers' own are
// ~1000.  Same tiles, same windows, same LDS layout, same gather -- but the ~2700 other instructions are cut to ~700:
//
//   * LANE j OF A QUAD OWNS LEVEL j (not point j of every level).  The 16-byte loads already deliver a level's four points
//     to one lane: the three 4 x 4 DPP transposes per pass (and the reference-point broadcasts) disappear, every lane does the
//     tap arithmetic of its own four points with ITS level's constants held in registers, and the gather of level l
//     broadcasts from lane l of the quad (quad_perm [l,l,l,l]) instead of from lane `point`.
//   * the tap arithmetic of a point is done ONCE (the first version did it for the bounding boxes and again in every
//     level pass): floor coordinates as one packed int16 pair, the four weights, and -- after the window geometry -- the two
//     LDS row addresses are kept in registers (7 per point).
//   * bounding boxes as packed int16 min / max (v_pk_min_i16 / v_pk_max_i16): all four levels are reduced at once (each
//     lane carries its own level), two DPP row rotations + two ds_bpermute steps, one 8-byte LDS store per level and wave.
//   * window geometry per LANE (every lane works out its own level's window from the four waves' boxes; the few
//     wave-uniform numbers the LDS-DMA loops need are read back with v_readlane): no geometry table, one barrier fewer.
//   * LDS-DMA source offsets: 64 lanes compute the offsets of 64 window rows (8 DMA instructions' worth) at once and hand
//     them to the issuing lanes by ds_bpermute, instead of ~14 vector instructions in front of every DMA instruction.
//   * the tile's query list (tile-local index -> query) is worked out by 128 threads once per tile and read back from LDS,
//     not decoded with two divisions per lane and pass.
//   * ONE fallback path: a point outside its window and a level whose window does not fit are the same case (no staged
//     address -> buffer loads for exactly those points, under a wave-uniform branch).
//
// Arithmetic: SURVEY.md Appendix A; reference ms_deform_im2col_cuda.cuh:227-237 (pixel mapping, in-range rule), :24-67
// (bilinear taps with zero padding); fused prologue ms_deform_attn.py:69-86.  Results equal the first version's up to the
// order of two multiplications per weight and of the softmax sum (1 ulp each).
#ifndef TF_MSDA_PQUAD2_H_
#define TF_MSDA_PQUAD2_H_

typedef short s16x2_t __attribute__((ext_vector_type(2)));

// LDS header of version 2 (ints; <= kPqHdrBytes / 4 = 512)
constexpr int kP2OffQ = 16;               // [2 parities][4 levels][ya, yb, xa, xb]: the tile's queries of each level
constexpr int kP2OffNom = kP2OffQ + 32;   // [2][4 levels][ny0, ny1, nx0, nx1]: nominal footprints (the window clamp)
constexpr int kP2OffBb = kP2OffNom + 48;  // [2][4 waves][4 levels][min, max]: packed int16 pairs (x | y << 16)
constexpr int kP2OffNq = kP2OffBb + 64;   // [2]: queries of the tile
constexpr int kP2OffQi = kP2OffNq + 2;    // [2][128]: b * S + q of the tile's k-th query, -1 behind the last
static_assert((kP2OffQi + 256) * 4 <= kPqHdrBytes, "LDS header of msda_fwd_f32_pquad2");

// staging of the windows: LDS-DMA (buffer_load ... lds), or buffer loads into registers + ds_write_b128 (TF_P2_STAGE_LDS=0)
#ifndef TF_P2_STAGE_LDS
#define TF_P2_STAGE_LDS 1
#endif
constexpr bool kP2StageByLds = TF_P2_STAGE_LDS != 0;
// experiment: level 0 (the largest window) is not staged at all -- its taps go by buffer loads (the texture path)
#ifndef TF_P2_L0_BY_LOADS
#define TF_P2_L0_BY_LOADS 0
#endif
constexpr bool kP2Level0ByLoads = TF_P2_L0_BY_LOADS != 0;
constexpr int kP2Sentinel = (int)0x80008000u;   // packed floor coordinates of a point that is not in range: (-32768, -32768)

__device__ __forceinline__ int p2_pack16(int x, int y)   // -> x | y << 16 (both in int16 range)
{
    return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(x, y));
}
__device__ __forceinline__ int p2_pkmin(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
__device__ __forceinline__ int p2_pkmax(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
__device__ __forceinline__ int p2_lo16(int p) { return (int)(short)(p & 0xFFFF); }
__device__ __forceinline__ int p2_hi16(int p) { return p >> 16; }
// lane i <- lane i ^ MASK (MASK = 16 / 32: across the DPP rows of the wave); bp = 4 * (lane ^ MASK)
__device__ __forceinline__ int p2_bperm(int bp, int v) { return __builtin_amdgcn_ds_bpermute(bp, v); }

// One point's four taps from LDS windows, one window row (two taps, four 16-byte reads) at a time: half the registers in
// flight of quad_taps_lds (msda_quad_dev.h), same sums in the same order.
template <int K>
__device__ __forceinline__ void quad_taps_lds_rows(unsigned a0, unsigned a1, const float (&w)[4], unsigned ldsA, unsigned ldsB,
                                                   f32x4_t &accA, f32x4_t &accB)
{
    constexpr int C = K * 0x55;   // quad_perm [K,K,K,K]
    {
        const unsigned p0a = dpp_u<C>(a0) + ldsA, p0b = dpp_u<C>(a0) + ldsB;
        const float W0 = dpp_f<C>(w[0]), W1 = dpp_f<C>(w[1]);
        const f32x4_t v00a = lds_read16(p0a), v01a = lds_read16(p0a + 128u);
        const f32x4_t v00b = lds_read16(p0b), v01b = lds_read16(p0b + 128u);
        accA += v00a * W0;
        accB += v00b * W0;
        accA += v01a * W1;
        accB += v01b * W1;
    }
    {
        const unsigned p1a = dpp_u<C>(a1) + ldsA, p1b = dpp_u<C>(a1) + ldsB;
        const float W2 = dpp_f<C>(w[2]), W3 = dpp_f<C>(w[3]);
        const f32x4_t v10a = lds_read16(p1a), v11a = lds_read16(p1a + 128u);
        const f32x4_t v10b = lds_read16(p1b), v11b = lds_read16(p1b + 128u);
        accA += v10a * W2;
        accB += v10b * W2;
        accA += v11a * W3;
        accB += v11b * W3;
    }
}

// ... and by buffer loads (the points that are not staged), two taps at a time.
template <int K>
__device__ __forceinline__ void quad_taps_global_rows(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&g)[4], const float (&w)[4],
                                                      unsigned rbA, unsigned rbB, f32x4_t &accA, f32x4_t &accB)
{
    constexpr int C = K * 0x55;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned G0 = dpp_u<C>(g[2 * h]), G1 = dpp_u<C>(g[2 * h + 1]);
        const float W0 = dpp_f<C>(w[2 * h]), W1 = dpp_f<C>(w[2 * h + 1]);
        const u32x4_t v0a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G0 + rbA, 0, 0);
        const u32x4_t v0b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G0 + rbB, 0, 0);
        const u32x4_t v1a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G1 + rbA, 0, 0);
        const u32x4_t v1b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G1 + rbB, 0, 0);
        accA += __builtin_bit_cast(f32x4_t, v0a) * W0;
        accB += __builtin_bit_cast(f32x4_t, v0b) * W0;
        accA += __builtin_bit_cast(f32x4_t, v1a) * W1;
        accB += __builtin_bit_cast(f32x4_t, v1b) * W1;
    }
}

__device__ __forceinline__ f32x4_t ldg_f4_nt(const float *base, unsigned byte_off)
{
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(reinterpret_cast<const char *>(base) + (size_t)byte_off));
}

// ---- the conflict-free gather (TF_P2_CF) ---------------------------------------------------------------------------------------
// A ds_read_b128 is served in four LDS cycles of 16 lanes -- the quads {0, 3, 5, 6}, {1, 2, 4, 7} of each half wave
// (MI355X_MICROARCH.md, LDS) -- and is conflict-free when those 16 lanes hit 16 different 16-byte slots (address / 16 mod 16).
// A (pixel, head) row is 128 bytes = 8 slots; which 8 is decided by the parity of its LDS row, i.e. by the data: with a quad
// reading a half row of ONE tap per instruction, two quads of a cycle collide whenever their rows have the same parity (1.75
// cycles per cycle for random parities; measured share of conflict cycles 35 %).  But the two taps (y, x0) and (y, x0 + 1) of a
// sampling point are ALWAYS neighbouring LDS rows: opposite parities.  So the lanes of a quad split by tap COLUMN instead of by
// half row -- lane s of the quad reads column s & 1, half s >> 1, one 16-byte piece per instruction -- and a quad then covers
// the slots {c, c + 4, c + 8, c + 12} whatever the parity, c the piece index of that instruction.  The four quads of an LDS cycle
// take the pieces in four different orders (piece = instruction ^ g, g the quad's place in its cycle): every slot exactly once.
// Price: a lane accumulates 16 channels (4 pieces of ONE column) instead of 8, the two columns are added across lanes (sub ^ 1)
// once per tile, and the gather spends 8 instead of 4 address additions and a select per weight pair.
// The weights of a point travel as (gy a, fy a, fx) -- three registers instead of four products: the lane forms the two it needs,
// (gy a) c and (fy a) c with c = fx or 1 - fx by its column (the same two roundings as the products formed in the prologue).
template <int K>
__device__ __forceinline__ void quad_taps_lds_cf(unsigned a0, unsigned a1, const float (&w)[3], const unsigned (&C)[4], bool col,
                                                 f32x4_t (&acc)[4])
{
    constexpr int Q = K * 0x55;   // quad_perm [K,K,K,K]
    const float ga = dpp_f<Q>(w[0]), fa = dpp_f<Q>(w[1]), fx = dpp_f<Q>(w[2]);
    const float cx = col ? fx : 1.f - fx;
    const float W0 = ga * cx, W1 = fa * cx;   // rows y0 / y0 + 1 of this lane's column
    f32x4_t v0[4], v1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v0[t] = lds_read16(dpp_u<Q>(a0) + C[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) v1[t] = lds_read16(dpp_u<Q>(a1) + C[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += v0[t] * W0;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += v1[t] * W1;
}
// ... and by buffer loads (points that are not staged).  g[t]: byte offset of tap t's row (y0 x0, y0 x0+1, y1 x0, y1 x0+1) held by
// lane K; R[t]: this lane's byte offset of instruction t's piece inside a row (without the column).
template <int K>
__device__ __forceinline__ void quad_taps_global_cf(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&g)[4], const float (&w)[3],
                                                    const unsigned (&R)[4], bool col, f32x4_t (&acc)[4])
{
    constexpr int Q = K * 0x55;
    const unsigned g0 = dpp_u<Q>(g[0]), g1 = dpp_u<Q>(g[1]), g2 = dpp_u<Q>(g[2]), g3 = dpp_u<Q>(g[3]);
    const float ga = dpp_f<Q>(w[0]), fa = dpp_f<Q>(w[1]), fx = dpp_f<Q>(w[2]);
    const unsigned G0 = col ? g1 : g0, G1 = col ? g3 : g2;
    const float cx = col ? fx : 1.f - fx;
    const float W0 = ga * cx, W1 = fa * cx;
    u32x4_t v0[4], v1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v0[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G0 + R[t], 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) v1[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G1 + R[t], 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += __builtin_bit_cast(f32x4_t, v0[t]) * W0;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += __builtin_bit_cast(f32x4_t, v1[t]) * W1;
}

// 16-byte output store through a buffer descriptor with a cache policy (PquadGeom::store; gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1)
__device__ __forceinline__ void p2_store16(const __amdgpu_buffer_rsrc_t orsrc, unsigned byte_off, f32x4_t v, int mode)
{
    const u32x4_t u = __builtin_bit_cast(u32x4_t, v);
    if (mode == 1) __builtin_amdgcn_raw_buffer_store_b128(u, orsrc, byte_off, 0, 2);
    else if (mode == 2) __builtin_amdgcn_raw_buffer_store_b128(u, orsrc, byte_off, 0, 16);
    else if (mode == 3) __builtin_amdgcn_raw_buffer_store_b128(u, orsrc, byte_off, 0, 17);
    else __builtin_amdgcn_raw_buffer_store_b128(u, orsrc, byte_off, 0, 0);
}

// WAVES x NP: 4 waves x 2 passes of 64 pairs (three workgroups per CU, <= 168 registers), or 8 waves x 1 pass of 128 pairs (two
// workgroups per CU with twice the LDS each, <= 128 registers: 16 instead of 12 waves per CU, half the work per wave and tile),
// or 4 waves x 1 pass: tiles of 64 pairs, four workgroups per CU (<= 128 registers, 39 KB of LDS each) -- more tiles in flight per
// CU against a per-tile chain of latencies that does not shrink with the tile
// CF: the conflict-free gather (quad_taps_lds_cf above; option pquad_cf).  Measured on MI355X (profiles/r06_msda_pquad2_conflict_free.txt):
// SQ_LDS_BANK_CONFLICT 3.59 M -> 0, LDS-active cycles 10.27 M -> 6.68 M per launch, vector instructions 9.72 M -> 10.95 M -- and the
// launch takes 39.7 instead of 38.1 us: the LDS pipe was not what bounds the kernel; the vector ALUs' extra 12 % cost more than the
// conflicts did.  Kept as an option (and as the emulator's proof that the layout IS conflict-free), not the default.
template <bool FUSED, int WAVES, int NP, bool CF = false>
__global__ void __launch_bounds__(64 * WAVES, ((WAVES == 8 || NP == 1) ? 4 : 3))
msda_fwd_f32_pquad2(const DirectArgs da, const LevelTable lt, const PquadGeom pg)
{
    constexpr bool kP2ConflictFree = CF;
    static_assert((WAVES == 4 && NP == 2) || (WAVES == 8 && NP == 1) || (WAVES == 4 && NP == 1), "64 or 128 (query, head) pairs per tile");
    constexpr int PT = 4, D = 32, NL = 4, PAIRS = 16 * WAVES;
    constexpr unsigned ROWB = D * 4;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);   // [H0..3 | W0..3 | start0..3]
    int *s_q = s_tab + kP2OffQ;
    int *s_nom = s_tab + kP2OffNom;
    int *s_bb = s_tab + kP2OffBb;
    int *s_nq = s_tab + kP2OffNq;
    int *s_qi = s_tab + kP2OffQi;
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
    if (pg.prio != 0) {
        // static priorities by the workgroup's slot on its CU: workgroups that share a CU otherwise move through their phases
        // in lock-step (fair arbitration keeps them aligned), front ends and gathers never overlap
        const int slot = item / pg.cus;
        const int pr = pg.prio == 1 ? 2 - slot : slot;
        if (pr >= 2) __builtin_amdgcn_s_setprio(3);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = threadIdx.x & 3;            // the level this lane owns
    const int quad = threadIdx.x >> 2;
    const int hsel = (quad >> 2) & 1;
    const unsigned rbA = (unsigned)(hsel * 64 + sub * 16), rbB = (unsigned)((1 - hsel) * 64 + sub * 16);
    const unsigned lds_rows = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char *)s_rows);
    const unsigned ldsA = lds_rows + rbA, ldsB = lds_rows + rbB;
    // conflict-free gather: column (x0 / x0 + 1) and half row of this lane, the quad's place g in its LDS cycle ({0, 3, 5, 6} /
    // {1, 2, 4, 7}: (quad & 7) >> 1), this lane's piece of instruction t: t ^ g
    const bool cf_col = (sub & 1) != 0;
    const unsigned cf_g = (unsigned)((quad & 7) >> 1);
    unsigned cfC[4];
    const unsigned cf_base = lds_rows + (cf_col ? 128u : 0u);   // cfC[t] - cf_base: the piece's byte offset inside a (pixel, head) row
#pragma unroll
    for (int t = 0; t < 4; ++t) cfC[t] = cf_base + (unsigned)((sub >> 1) * 64) + 16u * ((unsigned)t ^ cf_g);
    const int bp16 = 4 * (lane ^ 16), bp32 = 4 * (lane ^ 32);
    const unsigned long long lanes_of_level0 = 0x1111111111111111ull;

    auto stamp = [&](int i) {
        if (pg.trace != nullptr && threadIdx.x == 0 && i < 16)
            pg.trace[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memrealtime();
    };
    auto decode_item = [&](int it, int &b, int &ty, int &tx, int &m) {
        m = it % M;
        int t = it / M;
        // which head: for every tile t the map i % M -> m stays a bijection.  0: workgroup b (XCD b % 8) always works on head
        // b % 8 -- the heads whose windows are larger (the diagonal directions of the default bias grid) then own whole XCDs
        const int per = pg.cus >= 8 ? pg.cus >> 3 : 1;   // tiles per round of `cus` items
        if (pg.headmix == 1) m = (m + t + t / per) % M;
        else if (pg.headmix == 2) m ^= (t / per) & 1;
        tx = t % pg.tiles_x;
        t /= pg.tiles_x;
        ty = t % pg.tiles_y;
        b = t / pg.tiles_y;
    };
    // ---- per-tile tables.  Step 1 (threads 0..15): query partition + nominal footprints of every level ------------------
    auto tables_1 = [&](int par, int ty, int tx) {
        if (threadIdx.x < 4 * NL) {
            const int l = threadIdx.x >> 2, k = threadIdx.x & 3;
            int qv = 0, nv = 0;
            if (l < L) {
                const unsigned H0 = (unsigned)lt.H[0], W0 = (unsigned)lt.W[0];
                const unsigned Hl = (unsigned)lt.H[l], Wl = (unsigned)lt.W[l];
                const unsigned y0 = (unsigned)ty * pg.TH, y1 = min(H0, y0 + (unsigned)pg.TH);
                const unsigned x0 = (unsigned)tx * pg.TW, x1 = min(W0, x0 + (unsigned)pg.TW);
                qv = k == 0 ? tfq_tile_bound(y0, Hl, H0) : k == 1 ? tfq_tile_bound(y1, Hl, H0)
                     : k == 2 ? tfq_tile_bound(x0, Wl, W0) : tfq_tile_bound(x1, Wl, W0);
                int lo, hi;
                if (k < 2)
                    tfq_nominal((int)y0, (int)y1, (int)Hl, 1.f / (float)H0, pg.HY, &lo, &hi);
                else
                    tfq_nominal((int)x0, (int)x1, (int)Wl, 1.f / (float)W0, pg.HX, &lo, &hi);
                nv = (k & 1) ? hi : lo;
            }
            s_q[(par * 4 + l) * 4 + k] = qv;       // a level the call does not have: an empty range
            s_nom[(par * 4 + l) * 4 + k] = nv;
        }
    };
    // ---- step 2 (threads 0..127, one barrier after step 1): the tile's query list -----------------------------------------
    auto tables_2 = [&](int par, int b) {
        if (threadIdx.x < NP * PAIRS) {
            const int tq = (int)threadIdx.x;
            const int *q4 = s_q + par * 16;
            int qoff = 0, q = -1;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int ya = q4[4 * l], yb = q4[4 * l + 1], xa = q4[4 * l + 2], xb = q4[4 * l + 3];
                const int nx = xb - xa, n = (yb - ya) * nx;
                const int rr = tq - qoff;
                if (rr >= 0 && rr < n) {
                    // rr / nx: (rr + 0.5) / nx is at least 0.5 / nx away from an integer, far more than the float error
                    const int yy = (int)(((float)rr + 0.5f) * __builtin_amdgcn_rcpf((float)nx));
                    const int xx = rr - yy * nx;
                    q = b * S + s_tab[8 + l] + (ya + yy) * s_tab[4 + l] + xa + xx;
                }
                qoff += n;
            }
            s_qi[par * (NP * PAIRS) + tq] = q;
            if (tq == 0) s_nq[par] = qoff;   // <= NP * PAIRS (the host checked the maximum)
        }
    };

    if (threadIdx.x < NL) {
        const int l = threadIdx.x;
        s_tab[l] = l < L ? lt.H[l] : 1;
        s_tab[4 + l] = l < L ? lt.W[l] : 1;
        s_tab[8 + l] = l < L ? lt.start[l] : 0;
    }
    if (threadIdx.x < 2 * D) reinterpret_cast<float *>(s_rows)[threadIdx.x] = 0.f;   // rows 0, 1
    int cb, cty, ctx, cm;
    decode_item(item, cb, cty, ctx, cm);
    tables_1(0, cty, ctx);
    stamp(0);
    __syncthreads();
    tables_2(0, cb);

    // wave-uniform level constants (LDS-DMA loops) and this lane's own level
    int Hs[NL], Ws[NL], starts[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        Hs[l] = __builtin_amdgcn_readfirstlane(s_tab[l]);
        Ws[l] = __builtin_amdgcn_readfirstlane(s_tab[4 + l]);
        starts[l] = __builtin_amdgcn_readfirstlane(s_tab[8 + l]);
    }
    const bool lvalid = sub < L;
    const int my_H = s_tab[sub], my_W = s_tab[4 + sub], my_start = s_tab[8 + sub];
    const float my_Hf = (float)my_H, my_Wf = (float)my_W;
    const float my_inv_h = __builtin_amdgcn_rcpf(my_Hf), my_inv_w = __builtin_amdgcn_rcpf(my_Wf);
    const unsigned lsub = (unsigned)(lvalid ? sub : 0);
    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(da.value), 0, da.value_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        da.out, 0, (unsigned)(da.nlq * M * D * 4), 0x00020000);
    __syncthreads();   // the first tile's query list is visible

    // ---- a tile's points: per pass the four points of this lane's level --------------------------------------------------
    int pk[NP][PT];               // packed floor coordinates (x0 | y0 << 16), kP2Sentinel: not in range
    float w[NP][PT][kP2ConflictFree ? 3 : 4];   // bilinear weights x attention weight; conflict-free gather: (gy a, fy a, fx)
    unsigned a0[NP][PT], a1[NP][PT];   // LDS byte offsets of the rows (y0, x0) / (y0 + 1, x0); 0: not staged (rows 0, 1 are zeros)
    unsigned pair32[NP];          // (b * S + q) * M + m
    bool live[NP];
    int nq = 0;

    // A tile's prologue in two halves: (1) the loads into xy0 / xy1 / a4 / rp and the tile's pair indices, (2) softmax / locations
    // (fused entry) -> tap arithmetic -> bounding boxes (filed in s_bb[par]).  (Measured in round 6 and removed: a pass-major tail that
    // issues the next tile's loads while the second pass still gathers -- the order alone changes nothing, 39.0 vs 38.7 us, and the
    // loads held across the gather cost 21 / 74 spilled registers: 43.7 / 70.9 us, profiles/r06_msda_pquad2_early_loads.txt.)
    f32x4_t xy0[NP], xy1[NP], a4[NP];
    float2 rp[NP];
    unsigned n_pair32[NP];
    bool n_live[NP];
    auto prologue_loads = [&](int par, int m) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int bqs = s_qi[par * (NP * PAIRS) + ps * PAIRS + quad];
            n_live[ps] = bqs >= 0;
            const unsigned bq = n_live[ps] ? (unsigned)bqs : 0u;
            n_pair32[ps] = bq * (unsigned)M + (unsigned)m;
            if constexpr (!FUSED) {
                const unsigned pr = n_pair32[ps] * (unsigned)LP + lsub * (unsigned)PT;
                if (pg.ldnt) {
                    xy0[ps] = ldg_f4_nt(da.loc, pr * 8u);
                    xy1[ps] = ldg_f4_nt(da.loc, pr * 8u + 16u);
                    a4[ps] = ldg_f4_nt(da.attn, pr * 4u);
                } else {
                    xy0[ps] = ldg_f4(da.loc, pr * 8u);
                    xy1[ps] = ldg_f4(da.loc, pr * 8u + 16u);
                    a4[ps] = ldg_f4(da.attn, pr * 4u);
                }
                rp[ps] = float2{0.f, 0.f};
            } else {
                const unsigned row = bq * (unsigned)da.fa.ld;
                const unsigned s = (unsigned)(m * LP) + lsub * (unsigned)PT;
                if (pg.ldnt) {   // read once: do not displace the value rows (re-read by the neighbouring tiles) from L2
                    xy0[ps] = ldg_f4_nt(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u);
                    xy1[ps] = ldg_f4_nt(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u + 16u);
                    a4[ps] = ldg_f4_nt(da.fa.qproj, (row + (unsigned)da.fa.logit_col + s) * 4u);
                } else {
                    xy0[ps] = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u);
                    xy1[ps] = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.off_col + s * 2u) * 4u + 16u);
                    a4[ps] = ldg_f4(da.fa.qproj, (row + (unsigned)da.fa.logit_col + s) * 4u);
                }
                rp[ps] = ldg_f2(da.fa.ref, (bq * (unsigned)L + lsub) * 8u);   // ref_dim == 2: this level's reference point
            }
        }
    };
    auto prologue_math = [&](int par) {
        nq = __builtin_amdgcn_readfirstlane(s_nq[par]);
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            live[ps] = n_live[ps];
            pair32[ps] = n_pair32[ps];
        }
        int mn = 0x7FFF7FFF, mx = kP2Sentinel;   // packed int16 (x, y) bounding box of this lane's in-range points
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            float sx[PT] = {xy0[ps].x, xy0[ps].z, xy1[ps].x, xy1[ps].z};
            float sy[PT] = {xy0[ps].y, xy0[ps].w, xy1[ps].y, xy1[ps].w};
            float sa[PT] = {a4[ps].x, a4[ps].y, a4[ps].z, a4[ps].w};
            if constexpr (FUSED && !(kPqAblate & 16)) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
                // softmax over the pair's L * P logits (ms_deform_attn.py:70-71): this lane's four, then across the quad
#pragma unroll
                for (int k = 0; k < PT; ++k) sa[k] = lvalid ? sa[k] : -__builtin_inff();
                float m4 = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]));
                m4 = fmaxf(m4, dpp_f<kDppQuadXor1>(m4));
                m4 = fmaxf(m4, dpp_f<kDppQuadXor2>(m4));
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    sa[k] = lvalid ? __expf(sa[k] - m4) : 0.f;
                    sum += sa[k];
                }
                sum += dpp_f<kDppQuadXor1>(sum);
                sum += dpp_f<kDppQuadXor2>(sum);
                const float inv_sum = __builtin_amdgcn_rcpf(sum);   // (v_rcp_f32, 1 ulp: as in the first version)
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    sa[k] = sa[k] * inv_sum;
                    sx[k] = rp[ps].x + sx[k] * my_inv_h;   // x / H_l (as the reference writes it: ms_deform_attn.py:78-79)
                    sy[k] = rp[ps].y + sy[k] * my_inv_w;   // y / W_l
                }
            }
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const float xr = __builtin_fmaf(sx[k], my_Wf, -0.5f);   // cuh:227-228 with one rounding (see make_tap, msda_hip.hip)
                const float yr = __builtin_fmaf(sy[k], my_Hf, -0.5f);
                const bool in = live[ps] && lvalid && (yr > -1.f) && (xr > -1.f) && (yr < my_Hf) && (xr < my_Wf);
                const float x = in ? xr : 0.f, y = in ? yr : 0.f;
                const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
                const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
                const float a = in ? sa[k] : 0.f;
                const float ga = gy * a, fa = fy * a;
                if constexpr (kP2ConflictFree) {
                    w[ps][k][0] = ga;
                    w[ps][k][1] = fa;
                    w[ps][k][2] = fx;
                } else {
                    w[ps][k][0] = ga * gx;
                    w[ps][k][1] = ga * fx;
                    w[ps][k][2] = fa * gx;
                    w[ps][k][3] = fa * fx;
                }
                const int p = p2_pack16((int)xf, (int)yf);
                pk[ps][k] = in ? p : kP2Sentinel;
                if constexpr ((kPqAblate & 4) == 0) {
                    mn = p2_pkmin(mn, in ? p : 0x7FFF7FFF);
                    mx = p2_pkmax(mx, in ? p : kP2Sentinel);
                }
            }
        }
        if constexpr ((kPqAblate & 4) != 0) {   // the whole level: the geometry clamps it to the nominal footprint
            mn = p2_pack16(-1, -1);
            mx = p2_pack16(my_W, my_H);
        }
        // lanes of one level: the 16 quads of the wave (two rotations inside a DPP row, then across the four rows)
        mn = p2_pkmin(mn, dpp_i<kDppRowRor4>(mn));
        mx = p2_pkmax(mx, dpp_i<kDppRowRor4>(mx));
        mn = p2_pkmin(mn, dpp_i<kDppRowRor8>(mn));
        mx = p2_pkmax(mx, dpp_i<kDppRowRor8>(mx));
        mn = p2_pkmin(mn, p2_bperm(bp16, mn));
        mx = p2_pkmax(mx, p2_bperm(bp16, mx));
        mn = p2_pkmin(mn, p2_bperm(bp32, mn));
        mx = p2_pkmax(mx, p2_bperm(bp32, mx));
        if (lane < NL) {
            int2 *slot = reinterpret_cast<int2 *>(s_bb + ((par * WAVES + wave) * 4 + lane) * 2);
            *slot = int2{mn, mx};
        }
    };

    prologue_loads(0, cm);
    prologue_math(0);
    stamp(2);

    int par = 0;
    int iter = 0;
    while (true) {
        const int next_item = item + G;
        const bool has_next = next_item < pg.n_items;   // uniform
        int nb = 0, nty = 0, ntx = 0, nm = 0;
        if (has_next) decode_item(next_item, nb, nty, ntx, nm);

        __syncthreads();   // B0: the tile's bounding boxes are filed; every wave is done with the previous tile's windows
        if (iter == pg.trace_iter) stamp(3);

        // ---- window geometry, per lane for its own level (msda_quad_geom.h tfq_window) ----------------------------------
        int g_wx0, g_wy0, g_limx, g_limy, g_ww, g_rows, g_roff;
        bool late3;
        {
            const int2 *bb = reinterpret_cast<const int2 *>(s_bb + par * WAVES * 8) + sub;   // + 4 * wave
            const int4 nom = *reinterpret_cast<const int4 *>(s_nom + (par * 4 + sub) * 4);   // ny0, ny1, nx0, nx1
            int mn = 0x7FFF7FFF, mx = kP2Sentinel;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) {
                const int2 b = bb[4 * wv];
                mn = p2_pkmin(mn, b.x);
                mx = p2_pkmax(mx, b.y);
            }
            const int bx0 = p2_lo16(mn), by0 = p2_hi16(mn), bx1 = p2_lo16(mx), by1 = p2_hi16(mx);
            const int wx0 = tfq_max(bx0, nom.z), wx1 = tfq_min(bx1, nom.w - 1) + 1;
            const int wy0 = tfq_max(by0, nom.x), wy1 = tfq_min(by1, nom.y - 1) + 1;
            const bool some = lvalid && bx0 <= bx1 && by0 <= by1 && wx0 < wx1 && wy0 < wy1;
            const int ww = some ? wx1 - wx0 + 1 : 0, wh = some ? wy1 - wy0 + 1 : 0;
            const int rows = ww * wh;
            // round 0: level 0 alone; round 1: levels 1..3 packed one behind the other, all or nothing per level
            const int cap = pg.cap_rows;
            const int rows8 = (rows + 7) & ~7;
            const int r1 = dpp_i<0x55>(rows), r2 = dpp_i<0xAA>(rows), r3 = dpp_i<0xFF>(rows);
            const int e1 = dpp_i<0x55>(rows8), e2 = dpp_i<0xAA>(rows8);
            const bool fit1 = r1 <= cap;
            const int u2 = fit1 ? e1 : 0;
            const bool fit2 = r2 <= cap - u2;
            const int u3 = u2 + (fit2 ? e2 : 0);
            const bool fit3 = r3 <= cap - u3;
            // level 3 does not fit behind levels 1 and 2 but would on its own (headline pattern: the four diagonal heads): a THIRD
            // round of its own instead of buffer loads for every one of its points
            late3 = !fit3 && r3 <= cap && r3 > 0;
            const bool fit = sub == 0 ? (rows <= cap && !kP2Level0ByLoads) : sub == 1 ? fit1 : sub == 2 ? fit2 : (fit3 || late3);
            const bool stage = some && fit;
            g_roff = 2 + (sub == 2 ? u2 : sub == 3 ? (late3 ? 0 : u3) : 0);
            g_wx0 = stage ? wx0 : kQuadFar;
            g_wy0 = stage ? wy0 : kQuadFar;
            g_ww = stage ? ww : 0;
            g_rows = stage ? rows : 0;
            g_limx = stage ? ww - 2 : 0;
            g_limy = stage ? wh - 2 : 0;
        }
        const unsigned g_pitchb = (unsigned)g_ww * ROWB;
        const unsigned g_roffb = (unsigned)g_roff * ROWB;   // (relative to s_rows: the gather adds ldsA / ldsB)
        // what the LDS-DMA loops need, wave-uniform: lane l of the wave owns level l
        int u_wx0[NL], u_wy0[NL], u_ww[NL], u_rows[NL], u_roff[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            u_wx0[l] = __builtin_amdgcn_readlane(g_wx0, l);
            u_wy0[l] = __builtin_amdgcn_readlane(g_wy0, l);
            u_ww[l] = __builtin_amdgcn_readlane(g_ww, l);
            u_rows[l] = __builtin_amdgcn_readlane(g_rows, l);
            u_roff[l] = __builtin_amdgcn_readlane(g_roff, l);
        }

        // ---- LDS row addresses of every point; which (pass, level) have points that are not staged --------------------------
        unsigned long long ngm[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            ngm[ps] = 0ull;
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const int p = pk[ps][k];
                const int dx = p2_lo16(p) - g_wx0, dy = p2_hi16(p) - g_wy0;
                const bool staged = (unsigned)dx <= (unsigned)g_limx && (unsigned)dy <= (unsigned)g_limy;
                const unsigned lo = g_roffb + (unsigned)dy * g_pitchb + ((unsigned)dx << 7);
                a0[ps][k] = staged ? lo : 0u;                  // rows 0, 1 are zeros
                a1[ps][k] = staged ? lo + g_pitchb : 0u;
                ngm[ps] |= __ballot(p != kP2Sentinel && !staged);
            }
        }
        const unsigned head_base = (unsigned)((((long long)cb * S * M + cm) * D) * 4);
        const unsigned my_lvl_base = head_base + (unsigned)my_start * rowbytes;

        // ---- staging: 64 lanes work out the source offsets of 64 window rows (8 DMA instructions), ds_bpermute hands each
        //      DMA lane its row's offset; a DMA wave-instruction moves 8 rows of 128 B.  Pixels outside the level (extended
        //      coordinates -1 / size) get an out-of-range offset: the hardware writes zeros.
        auto stage_level = [&](auto lc) {
            constexpr int l = decltype(lc)::value;
            const int nrows = u_rows[l];
            if (nrows <= 0) return;   // uniform
            const int nchunks = (nrows + 7) >> 3;
            const int H = Hs[l], W = Ws[l], ww = u_ww[l], wx0 = u_wx0[l], wy0 = u_wy0[l], roff = u_roff[l];
            const unsigned lvl_base = head_base + (unsigned)starts[l] * rowbytes;
            const float inv_ww = __builtin_amdgcn_rcpf((float)ww);
            constexpr int NW = WAVES;
            for (int c0 = wave; c0 < nchunks; c0 += 8 * NW) {   // this wave's chunks c0, c0 + NW, ...: 8 per group
                // lane j resolves row (j >> 3) of chunk c0 + NW * (j & 7): the DMA lanes of row a (lanes 8 a .. 8 a + 7) then find
                // chunk g's offset in lane 8 a + g -- a broadcast inside their group of 8 (ds_swizzle, no address register)
                const int r = (c0 + NW * (lane & 7)) * 8 + (lane >> 3);   // the window row this lane resolves
                const int wy = (int)(((float)r + 0.5f) * inv_ww);         // r / ww (r < 2^16: exact, see tables_2)
                const int wx = r - wy * ww;
                const int py = wy0 + wy, px = wx0 + wx;                    // extended coordinates: may be -1 or size
                const bool ok = r < nrows && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
                const unsigned off = ok ? lvl_base + (unsigned)(py * W + px) * rowbytes : kOobBase;
                if constexpr (kP2StageByLds) {
                    // all eight broadcasts first (one LDS round trip for the group, not one per DMA instruction)
                    unsigned src[8];
                    src[0] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (0 << 5));
                    src[1] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (1 << 5));
                    src[2] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (2 << 5));
                    src[3] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (3 << 5));
                    src[4] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (4 << 5));
                    src[5] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (5 << 5));
                    src[6] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (6 << 5));
                    src[7] = (unsigned)__builtin_amdgcn_ds_swizzle((int)off, 0x18 | (7 << 5));
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int c = c0 + NW * g;
                        if (c >= nchunks) break;   // uniform
                        if constexpr (!(kPqAblate & 2))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                rsrc, (__attribute__((address_space(3))) void *)(s_rows + (size_t)(roff + c * 8) * 128), 16,
                                src[g] + (unsigned)(lane & 7) * 16u, 0, 0, 0);
                    }
                } else {
                    // through registers: 8 buffer loads in flight, then 8 ds_write_b128 (an LDS-DMA instruction costs its wave
                    // 100-185 cycles of issue inside a busy phase, MI355X_MICROARCH.md; a load + a store far less)
                    u32x4_t v[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int c = c0 + NW * g;
                        const unsigned src = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((lane & 0x38) | g), (int)off) + (unsigned)(lane & 7) * 16u;
                        if (c < nchunks)   // uniform
                            v[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, src, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int c = c0 + NW * g;
                        if (c < nchunks)   // uniform
                            *reinterpret_cast<u32x4_t *>(s_rows + (size_t)(roff + c * 8) * 128 + (size_t)lane * 16) = v[g];
                    }
                }
            }
        };

        f32x4_t accA[NP], accB[NP];
        f32x4_t acc4[kP2ConflictFree ? NP : 1][4];   // conflict-free gather: instruction t's piece (t ^ g) of this lane's column
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            accA[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            accB[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (kP2ConflictFree) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc4[ps][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- one level: the staged points from LDS, the others (if the wave has any) by buffer loads --------------------------
        auto gather_lp = [&](auto lc, auto pc) {   // one level, one pass (both compile-time: the records are register arrays)
            constexpr int l = decltype(lc)::value;
            constexpr int ps = decltype(pc)::value;
            if (l >= L) return;   // uniform
            if constexpr (ps < NP) {
              do {
                if (ps * PAIRS >= nq) continue;   // uniform
                if constexpr ((kPqAblate & 1) != 0) {
#pragma unroll
                    for (int k = 0; k < PT; ++k) accA[ps].x += (float)(a0[ps][k] + a1[ps][k]) * w[ps][k][0];   // keeps the records alive
                } else {
#pragma unroll
                    for (int k = 0; k < PT; ++k) {
                        if constexpr (kP2ConflictFree) quad_taps_lds_cf<l>(a0[ps][k], a1[ps][k], w[ps][k], cfC, cf_col, acc4[ps]);
                        else quad_taps_lds<l>(a0[ps][k], a1[ps][k], w[ps][k], ldsA, ldsB, accA[ps], accB[ps]);
                    }
                }
                if constexpr ((kPqAblate & 32) != 0) continue;
                if ((ngm[ps] & (lanes_of_level0 << l)) == 0ull) continue;   // uniform: every point of this level and pass was staged
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    const int p = pk[ps][k];
                    const int x0 = p2_lo16(p), y0 = p2_hi16(p);
                    const bool need = p != kP2Sentinel && a0[ps][k] == 0u;   // in range, not staged (a staged row offset is >= 256)
                    const bool kx0 = need && (x0 >= 0), kx1 = need && (x0 + 1 <= my_W - 1);
                    const bool ky0 = (y0 >= 0), ky1 = (y0 + 1 <= my_H - 1);
                    const int r0 = y0 * my_W + x0;
                    // staged / invalid taps: kOobBase + (lane offset < 128) is still out of range -> hardware zero
                    const unsigned g4[4] = {(ky0 && kx0) ? my_lvl_base + (unsigned)r0 * rowbytes : kOobBase,
                                            (ky0 && kx1) ? my_lvl_base + (unsigned)(r0 + 1) * rowbytes : kOobBase,
                                            (ky1 && kx0) ? my_lvl_base + (unsigned)(r0 + my_W) * rowbytes : kOobBase,
                                            (ky1 && kx1) ? my_lvl_base + (unsigned)(r0 + my_W + 1) * rowbytes : kOobBase};
                    if constexpr (kP2ConflictFree) {
                        const unsigned cfR[4] = {cfC[0] - cf_base, cfC[1] - cf_base, cfC[2] - cf_base, cfC[3] - cf_base};
                        quad_taps_global_cf<l>(rsrc, g4, w[ps][k], cfR, cf_col, acc4[ps]);
                    }
                    else quad_taps_global<l>(rsrc, g4, w[ps][k], rbA, rbB, accA[ps], accB[ps]);
                }
              } while (false);
            }
        };
        auto gather_level = [&](auto lc) {   // one level, every pass
            gather_lp(lc, std::integral_constant<int, 0>{});
            gather_lp(lc, std::integral_constant<int, 1>{});
        };

        // ---- round 0: level 0 ----
        stage_level(std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA landed
        __syncthreads();                      // B1: ... everybody's
        if (iter == pg.trace_iter) stamp(4);
        if (has_next) tables_1(par ^ 1, nty, ntx);   // off the staging path; visible after B2
        if (iter == pg.trace_iter) stamp(5);
        gather_level(std::integral_constant<int, 0>{});
        if (iter == pg.trace_iter) stamp(6);

        // ---- round 1: levels 1..3 in the same rows ----
        __syncthreads();   // B2: every wave is done reading level 0's window
        if (iter == pg.trace_iter) stamp(11);
        if (has_next) tables_2(par ^ 1, nb);   // visible after B3
        const bool third_round = __builtin_amdgcn_readfirstlane((int)late3) != 0;   // (the same in every lane)
        stage_level(std::integral_constant<int, 1>{});
        stage_level(std::integral_constant<int, 2>{});
        if (!third_round) stage_level(std::integral_constant<int, 3>{});
        if (iter == pg.trace_iter) stamp(12);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (iter == pg.trace_iter) stamp(13);
        __syncthreads();   // B3
        if (iter == pg.trace_iter) stamp(7);
        gather_level(std::integral_constant<int, 1>{});
        gather_level(std::integral_constant<int, 2>{});
        if (third_round) {   // ---- round 2: level 3 on its own ----
            __syncthreads();
            stage_level(std::integral_constant<int, 3>{});
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
        gather_level(std::integral_constant<int, 3>{});
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            if constexpr (kP2ConflictFree) {
                // the two columns of a pair sit in the lanes sub and sub ^ 1 (same half row, same piece order): a lane sends the partner
                // the two pieces the partner stores and adds what it receives to the two it stores itself
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4_t mine = cf_col ? acc4[ps][2 + j] : acc4[ps][j];
                    const f32x4_t yours = cf_col ? acc4[ps][j] : acc4[ps][2 + j];
                    f32x4_t sum;
                    sum.x = mine.x + dpp_f<kDppQuadXor1>(yours.x);
                    sum.y = mine.y + dpp_f<kDppQuadXor1>(yours.y);
                    sum.z = mine.z + dpp_f<kDppQuadXor1>(yours.z);
                    sum.w = mine.w + dpp_f<kDppQuadXor1>(yours.w);
                    (j == 0 ? accA[ps] : accB[ps]) = sum;
                }
            }
            if (live[ps] && (!(kPqAblate & 8) || accA[ps].x == 12345.678f)) {
                const unsigned ob = pair32[ps] * (unsigned)(D * 4);
                if constexpr (kP2ConflictFree) {
                    p2_store16(orsrc, ob + (cf_col ? cfC[2] : cfC[0]) - cf_base, accA[ps], pg.store);
                    p2_store16(orsrc, ob + (cf_col ? cfC[3] : cfC[1]) - cf_base, accB[ps], pg.store);
                } else {
                    p2_store16(orsrc, ob + rbA, accA[ps], pg.store);
                    p2_store16(orsrc, ob + rbB, accB[ps], pg.store);
                }
            }
        }
        if (iter == pg.trace_iter) stamp(8);
        if (!has_next) break;

        // ---- the next tile: its points, prologue arithmetic and bounding boxes ----
        cb = nb;
        cm = nm;
        if (iter == 0) stamp(14);
        prologue_loads(par ^ 1, nm);
        prologue_math(par ^ 1);
        if (iter == 0) stamp(9);
        item = next_item;
        par ^= 1;
        ++iter;
    }
    stamp(10);
}

#endif  // TF_MSDA_PQUAD2_H_
