// trackformer_amd/csrc/split_product.h -- the split product shared by the matrix-core kernels of this library
// (linear_split.hip, linear_stream.hip, ffn_fused.hip, stem_conv.hip): fp32 operands cut into 16-bit pieces, the product formed
// from v_mfma_f32_32x32x16_{bf16,f16} terms with fp32 accumulation.  Reference arithmetic: fp32 everywhere
// (models/ops/src/cuda/ms_deform_attn_cuda.cu:69 AT_DISPATCH_FLOATING_TYPES on fp32 tensors; nn.Linear / Conv2d of
// models/deformable_transformer.py, models/backbone.py; no autocast anywhere in the reference).
#ifndef TF_SPLIT_PRODUCT_H_
#define TF_SPLIT_PRODUCT_H_

#include <hip/hip_runtime.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 8 pieces of one kind: an MFMA operand fragment
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;   // 4 pieces of one kind

// ---- the split product, generic in the SCHEME SP (the template parameter of every kernel that uses it):
//   SP = 3   bf16 (hi, mid, lo) per operand: six terms    a_lo.b_hi + a_hi.b_lo + a_mid.b_mid + a_mid.b_hi + a_hi.b_mid + a_hi.b_hi
//   SP = 16  fp16 (hi, lo) of both operands, three terms  a_lo.(b_hi 2^-11) + a_hi.b_lo + a_hi.b_hi      (the default)
// Both are fp32-class; nothing cheaper exists in this library (the two-piece bf16 scheme -- three terms, product error < 2^-16 -- kept
// the reference's track ids for 14 of 64 frames where fp32 keeps 59, profiles/r04_id_parity_64.txt, and was deleted in round 5).
//
// bf16, three pieces (round to nearest even at every step) carry all 24 significand bits of an fp32 number (hi 8, mid 8, lo 8; the
// residuals x - hi and (x - hi) - mid are exact in fp32), and the six products kept are all those of weight >= 2^-16; the dropped
// ones (mid.lo, lo.mid, lo.lo) are below 2^-24 of |a||b| -- half an ulp of the fp32 product the reference rounds to.  With fp32
// accumulation on the matrix cores the six-term product is fp32 arithmetic in a different summation order.
//
// fp16 (round 4): an fp16 piece carries 11 significand bits, so TWO pieces (hi = rne16(x), lo = rne16(x - hi), the residual
// exact in fp32) carry 22 bits + the sign of lo: |x - hi - lo| <= 2^-23 |x| -- one bit short of fp32's own rounding -- and the
// only product dropped (lo.lo) is below 2^-22 of |a||b|.  Three MFMAs instead of six.  What fp16 lacks is exponent range (5 bits):
//   * the lo piece of a small number would fall into the subnormals.  It is therefore stored SCALED: lo' = rne16((x - hi) 2^11)
//     has the magnitude of hi's last place times 2^11, normal whenever hi is, and its partner in the product is the weight's hi
//     piece times 2^-11 -- made in registers from the hi fragment right before its MFMA (v_pk_mul_f16 by a power of two: exact; four
//     instructions per fragment, which serves every row tile of the wave), so a weight is stored as TWO pieces (the first version of this scheme stored the third operand: 1.5x the weight traffic and a
//     third more fragment registers -- 99.7 against 71.2 us in the fused feed-forward block, profiles/r04_f16_first_harness.txt);
//   * weights are scaled per OUTPUT CHANNEL by the power of two t_n that puts the channel's largest |w| into [2^13, 2^14) (exact;
//     a weight 2^-17 times the channel's largest still has all its bits, smaller ones an absolute error of 2^-39 of the largest);
//     the epilogue multiplies the accumulator by r_n = 1 / (kActScale t_n), again a power of two;
//   * activations are scaled by kActScale = 2^-4: |x| up to 1.0e6 is representable, |x| below 2^-10 has an absolute error of
//     2^-32 (hi subnormal, lo' picks up the residual).  An activation beyond 1.0e6 becomes inf -> (inf - inf) = NaN in lo': the
//     output row is NaN, loudly, instead of silently saturated.
// Measured against float64 on random operands (tools/experiments/f16_split.py): max |err| / sum |x||w| = 3.7e-8 for the
// representation alone (fp32 rounding of the SUM, which the reference's sgemm has as well: 2.4e-7), six-term bf16 3.3e-9.
// Terms are issued smallest first.
template <int SP> struct Split;
// NA / NB: pieces of an activation / of a weight as STORED (LDS, packed image, piece tensors); NBX: weight operands of the terms (B
// indexes them): the stored pieces [+ the one expand_weight() derives]
template <> struct Split<3> {
    static constexpr bool F16 = false;
    static constexpr int NA = 3, NB = 3, NBX = 3, N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0};
};
template <> struct Split<16> {
    static constexpr bool F16 = true;
    static constexpr int NA = 2, NB = 2, NBX = 3, N = 3;
    static constexpr int A[3] = {1, 0, 0}, B[3] = {2, 1, 0};
};

constexpr float kActScale = 0.0625f;        // fp16 scheme: activations are split as x 2^-4
constexpr float kLoScale = 2048.f;          // ... and the lo piece as (x 2^-4 - hi) 2^11
constexpr float kInvLoScale = 1.f / 2048.f;

// x[0..3] of an ACTIVATION -> NA pieces of 4 (bf16: v_cvt_pk_bf16_f32; fp16: v_cvt_f16_f32 -- both round to nearest even)
template <int SP>
__device__ __forceinline__ void split4(const f32x4 &x, u32x2 (&p)[Split<SP>::NA])
{
    if constexpr (Split<SP>::F16) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xs = x[e] * kActScale;
            h[e] = (_Float16)xs;
            l[e] = (_Float16)((xs - (float)h[e]) * kLoScale);
        }
        p[0] = __builtin_bit_cast(u32x2, h);
        p[1] = __builtin_bit_cast(u32x2, l);
    } else {
        bf16x4 q[Split<SP>::NA];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r = x[e];
#pragma unroll
            for (int i = 0; i < Split<SP>::NA; ++i) {
                q[i][e] = (__bf16)r;
                if (i + 1 < Split<SP>::NA) r -= (float)q[i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < Split<SP>::NA; ++i) p[i] = __builtin_bit_cast(u32x2, q[i]);
    }
}

// w[0..3] of a WEIGHT (fp16 scheme: already multiplied by its channel's t_n) -> NB pieces of 4
template <int SP>
__device__ __forceinline__ void split4_weight(const f32x4 &w, u32x2 (&p)[Split<SP>::NB])
{
    if constexpr (Split<SP>::F16) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (_Float16)w[e];
            l[e] = (_Float16)(w[e] - (float)h[e]);
        }
        p[0] = __builtin_bit_cast(u32x2, h);
        p[1] = __builtin_bit_cast(u32x2, l);
    } else {
        split4<SP>(w, p);
    }
}

// the weight operands of the terms from the stored pieces of a fragment: fp16 scheme: + hi 2^-11 (exact; v_pk_mul_f16 x 4)
template <int SP>
__device__ __forceinline__ void expand_weight(const u32x4 (&w)[Split<SP>::NB], u32x4 (&wx)[Split<SP>::NBX])
{
#pragma unroll
    for (int p = 0; p < Split<SP>::NB; ++p) wx[p] = w[p];
    if constexpr (Split<SP>::F16) {
        const f16x8 h = __builtin_bit_cast(f16x8, w[0]);
        wx[2] = __builtin_bit_cast(u32x4, h * (_Float16)kInvLoScale);
    }
}

// the power of two that puts `amax` (the largest |w| of an output channel) into [2^13, 2^14); 1 for an all-zero / non-finite channel
__device__ __forceinline__ float weight_scale_for(float amax)
{
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;   // floor(log2 amax) for normal numbers
    int s = 13 - e;
    s = s < -100 ? -100 : (s > 100 ? 100 : s);
    return __builtin_bit_cast(float, (unsigned)(s + 127) << 23);
}

// one term: acc += a . b (32 x 32 x 16) on the scheme's element type
template <int SP>
__device__ __forceinline__ f32x16 mfma16(const u32x4 &a, const u32x4 &b, const f32x16 &acc)
{
    if constexpr (Split<SP>::F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// acc += x . w over the pieces, smallest terms first (one 32 x 32 x 16 MFMA per term); XA: the activation fragment is the MFMA's
// A operand (rows of the accumulator tile = activation rows), else the weight fragment is (transposed tile)
template <int SP, bool XA = true>
__device__ __forceinline__ void mfma_terms(f32x16 &acc, const u32x4 (&x)[Split<SP>::NA], const u32x4 (&w)[Split<SP>::NB])
{
    using T = Split<SP>;
    u32x4 wx[T::NBX];
    expand_weight<SP>(w, wx);
#pragma unroll
    for (int t = 0; t < T::N; ++t) acc = XA ? mfma16<SP>(x[T::A[t]], wx[T::B[t]], acc) : mfma16<SP>(wx[T::B[t]], x[T::A[t]], acc);
}

// the same for a wave's TI x TJ tiles, term-major: consecutive MFMAs never share an accumulator; per accumulator the order is
// still smallest term first, k ascending
template <int SP, int TI, int TJ, bool XA = true>
__device__ __forceinline__ void mfma_tiles(f32x16 (&acc)[TI][TJ], const u32x4 (&x)[TI][Split<SP>::NA], const u32x4 (&w)[TJ][Split<SP>::NB])
{
    using T = Split<SP>;
    u32x4 wx[TJ][T::NBX];
#pragma unroll
    for (int j = 0; j < TJ; ++j) expand_weight<SP>(w[j], wx[j]);
#pragma unroll
    for (int t = 0; t < T::N; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = XA ? mfma16<SP>(x[i][T::A[t]], wx[j][T::B[t]], acc[i][j]) : mfma16<SP>(wx[j][T::B[t]], x[i][T::A[t]], acc[i][j]);
}

// `terms` argument of the C ABI -> scheme: 6 bf16 terms (three pieces per operand), 16 = fp16 pieces (three terms); 0: unknown.
// (The three-term bf16 product -- two pieces, products good to 2^-16 -- of rounds 2-4 was removed in round 5: terms == 3 is unknown.)
inline int split_scheme(int terms) { return terms == 6 ? 3 : terms == 16 ? 16 : 0; }
inline int scheme_pieces_b(int sp) { return sp == 16 ? 2 : sp; }   // 16-bit weight pieces stored per element

}  // namespace

#endif /* TF_SPLIT_PRODUCT_H_ */
