// trackformer_amd/csrc/msda_quad_dev.h -- device helpers of the 4-lanes-per-pair ("quad") forward kernels:
// DPP moves, one sampling point's four taps from LDS windows or by buffer loads.  Included inside a namespace
// that sees tfm::f32x4_t / u32x4_t / kOobBase.
#ifndef TF_MSDA_QUAD_DEV_H_
#define TF_MSDA_QUAD_DEV_H_

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppQuadXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int kDppRowRor4 = 0x124;
constexpr int kDppRowRor8 = 0x128;
constexpr int kDppRowHalfMirror = 0x141;   // lane i <- lane 7 - i of its group of 8

// One point's four taps from LDS windows.  a0 / a1: byte offsets (from the first LDS row) of the rows
// (y0, x0) and (y0 + 1, x0) held by lane K of the quad; the x0 + 1 taps are the next 128-byte rows.
// ldsA / ldsB: this lane's LDS byte addresses of its two 16-byte pieces of row 0 -- plain integers, so
// that the quad broadcast folds into the address add (v_add_u32_dpp) and the + 128 becomes an immediate.
typedef const __attribute__((address_space(3))) f32x4_t *lds_f32x4_ptr;
__device__ __forceinline__ f32x4_t lds_read16(unsigned addr)
{
    return *reinterpret_cast<lds_f32x4_ptr>((size_t)addr);
}
template <int K>
__device__ __forceinline__ void quad_taps_lds(unsigned a0, unsigned a1, const float (&w)[4], unsigned ldsA,
                                              unsigned ldsB, f32x4_t &accA, f32x4_t &accB)
{
    constexpr int C = K * 0x55;   // quad_perm [K,K,K,K]
    const unsigned p0a = dpp_u<C>(a0) + ldsA, p0b = dpp_u<C>(a0) + ldsB;
    const unsigned p1a = dpp_u<C>(a1) + ldsA, p1b = dpp_u<C>(a1) + ldsB;
    const float W0 = dpp_f<C>(w[0]), W1 = dpp_f<C>(w[1]), W2 = dpp_f<C>(w[2]), W3 = dpp_f<C>(w[3]);
    const f32x4_t v00a = lds_read16(p0a), v01a = lds_read16(p0a + 128u);
    const f32x4_t v00b = lds_read16(p0b), v01b = lds_read16(p0b + 128u);
    const f32x4_t v10a = lds_read16(p1a), v11a = lds_read16(p1a + 128u);
    const f32x4_t v10b = lds_read16(p1b), v11b = lds_read16(p1b + 128u);
    accA += v00a * W0;
    accB += v00b * W0;
    accA += v01a * W1;
    accB += v01b * W1;
    accA += v10a * W2;
    accB += v10b * W2;
    accA += v11a * W3;
    accB += v11b * W3;
}

// (A variant with a rolling set of reads in flight -- the y0 taps of point K + 1 requested as soon as point K's were consumed,
// bit-identical -- was measured on MI355X in round 3: 44.5 vs 44.4 us plain, 45.9 vs 47.2 us fused; removed.)

// One point's four taps by buffer loads.  g[t]: byte offset of tap t's row (this head's 128 bytes) held
// by lane K of the quad; invalid taps carry kOobBase, which stays out of range after + rbA / rbB.
template <int K>
__device__ __forceinline__ void quad_taps_global(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&g)[4],
                                                 const float (&w)[4], unsigned rbA, unsigned rbB,
                                                 f32x4_t &accA, f32x4_t &accB)
{
    constexpr int C = K * 0x55;
    u32x4_t va[4], vb[4];
    float W[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned G = dpp_u<C>(g[t]);
        W[t] = dpp_f<C>(w[t]);
        va[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G + rbA, 0, 0);
        vb[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G + rbB, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        accA += __builtin_bit_cast(f32x4_t, va[t]) * W[t];
        accB += __builtin_bit_cast(f32x4_t, vb[t]) * W[t];
    }
}

// ---- head dimension 36 (hidden 288: cfg 4).  A (pixel, head) row is 144 bytes = 9 pieces of 16 bytes; lanes 0..2 of
// a quad take 3 pieces (12 channels) each, lane 3 only contributes its sampling point's tap arithmetic (what it reads and
// sums is never stored).  ldsL: this lane's LDS byte address of its first piece of row 0; the x0 + 1 taps are + 144.
template <int K>
__device__ __forceinline__ void quad_taps_lds36(unsigned a0, unsigned a1, const float (&w)[4], unsigned ldsL,
                                                f32x4_t (&acc)[3])
{
    constexpr int C = K * 0x55;   // quad_perm [K,K,K,K]
    const unsigned p0 = dpp_u<C>(a0) + ldsL, p1 = dpp_u<C>(a1) + ldsL;
    const float W0 = dpp_f<C>(w[0]), W1 = dpp_f<C>(w[1]), W2 = dpp_f<C>(w[2]), W3 = dpp_f<C>(w[3]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f32x4_t v00 = lds_read16(p0 + 16u * c), v01 = lds_read16(p0 + 144u + 16u * c);
        const f32x4_t v10 = lds_read16(p1 + 16u * c), v11 = lds_read16(p1 + 144u + 16u * c);
        acc[c] += v00 * W0;
        acc[c] += v01 * W1;
        acc[c] += v10 * W2;
        acc[c] += v11 * W3;
    }
}
// g[t]: byte offset of tap t's 144-byte row; rbL = sub * 48; lane 3 (idle) loads from an out-of-range offset (zeros).
template <int K>
__device__ __forceinline__ void quad_taps_global36(const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&g)[4],
                                                   const float (&w)[4], unsigned rbL, bool idle, f32x4_t (&acc)[3])
{
    constexpr int C = K * 0x55;
    u32x4_t v[4][3];
    float W[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned Gd = dpp_u<C>(g[t]);
        const unsigned G = idle ? kOobBase : Gd;
        W[t] = dpp_f<C>(w[t]);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[t][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, G + rbL + 16u * c, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += __builtin_bit_cast(f32x4_t, v[t][c]) * W[t];
}

// 32-bit byte offsets from a kernel-uniform base: the compiler emits global_load ... v_off, s[base:base+1].
__device__ __forceinline__ float ldg_f(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}
__device__ __forceinline__ float2 ldg_f2(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}


#endif  // TF_MSDA_QUAD_DEV_H_
