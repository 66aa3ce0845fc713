// trackformer_amd/csrc/split_product.h -- the bf16 split product shared by the matrix-core kernels of this library
// (linear_split.hip, linear_stream.hip, ffn_fused.hip, stem_conv.hip): fp32 operands cut into bf16 pieces, the product formed
// from v_mfma_f32_32x32x16_bf16 terms with fp32 accumulation.  Reference arithmetic: fp32 everywhere
// (models/ops/src/cuda/ms_deform_attn_cuda.cu:69 AT_DISPATCH_FLOATING_TYPES on fp32 tensors; nn.Linear / Conv2d of
// models/deformable_transformer.py, models/backbone.py; no autocast anywhere in the reference).
#ifndef TF_SPLIT_PRODUCT_H_
#define TF_SPLIT_PRODUCT_H_

#include <hip/hip_runtime.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---- the split product, generic in the number of bf16 PIECES per operand (NP):
//   NP = 2 (hi, mid):     three terms  a_mid.b_hi + a_hi.b_mid + a_hi.b_hi                 dropped terms < 2^-16 of the product
//   NP = 3 (hi, mid, lo): six terms    a_lo.b_hi + a_hi.b_lo + a_mid.b_mid + a_mid.b_hi + a_hi.b_mid + a_hi.b_hi
// Three bf16 pieces (round to nearest even at every step) carry all 24 significand bits of an fp32 number (hi 8, mid 8, lo 8; the
// residuals x - hi and (x - hi) - mid are exact in fp32), and the six products kept are all those of weight >= 2^-16; the dropped
// ones (mid.lo, lo.mid, lo.lo) are below 2^-24 of |a||b| -- half an ulp of the fp32 product the reference rounds to
// (ms_deform_attn_cuda.cu:69 and every nn.Linear / Conv2d of the path are fp32).  With fp32 accumulation on the matrix cores the
// six-term product is fp32 arithmetic in a different summation order; it is the DEFAULT (fused.set_split_terms(6)), the
// three-term product the opt-in fast mode: on the 64-frame reference-Tracker fixture fp32 keeps the reference's track ids for 59
// frames, the three-term product for 14 (profiles/r04_id_parity_64.txt).  Terms are issued smallest first.
template <int NP> struct SplitTerms;
template <> struct SplitTerms<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0};
};
template <> struct SplitTerms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0};
};

// x[0..3] -> NP bf16x4 pieces (v_cvt_pk_bf16_f32, round to nearest even)
template <int NP>
__device__ __forceinline__ void split4(const f32x4 &x, bf16x4 (&p)[NP])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = x[e];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            p[q][e] = (__bf16)r;
            if (q + 1 < NP) r -= (float)p[q][e];
        }
    }
}

// acc += a . b over the pieces, smallest terms first (one 32 x 32 x 16 MFMA per term)
template <int NP>
__device__ __forceinline__ void mfma_terms(f32x16 &acc, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP])
{
    using T = SplitTerms<NP>;
#pragma unroll
    for (int t = 0; t < T::N; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[T::A[t]], b[T::B[t]], acc, 0, 0, 0);
}

// the same for a wave's TI x TJ tiles, term-major: consecutive MFMAs never share an accumulator; per accumulator the order is
// still smallest term first, k ascending
template <int NP, int TI, int TJ>
__device__ __forceinline__ void mfma_tiles(f32x16 (&acc)[TI][TJ], const bf16x8 (&a)[TI][NP], const bf16x8 (&b)[TJ][NP])
{
    using T = SplitTerms<NP>;
#pragma unroll
    for (int t = 0; t < T::N; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][T::A[t]], b[j][T::B[t]], acc[i][j], 0, 0, 0);
}

// `terms` argument of the C ABI (3 or 6) -> pieces per operand, or 0
inline int split_pieces(int terms) { return terms == 3 ? 2 : terms == 6 ? 3 : 0; }

}  // namespace

#endif /* TF_SPLIT_PRODUCT_H_ */
