// trackformer_amd/csrc/ffn_fused.hip
//
// tf_ffn_fused_f32 (include/tf_fused.h): the feed-forward block of a transformer layer in ONE kernel,
//
//     y = [LayerNorm]( residual + linear2( relu( linear1(x) ) ) )
//
// (reference: models/deformable_transformer.py:282-297 DeformableTransformerEncoderLayer.forward_ffn -- `src2 =
// linear2(dropout2(activation(linear1(src)))); src = src + dropout3(src2); src = norm2(src)`, and :371-379 for the decoder
// layer).  OPT-IN (TF_FFN_FUSED=1 / fused.set_ffn_fused): written against the emulator, not yet run on hardware.
//
// Why: the two FFN GEMMs of an encoder layer (22 223 x 256 -> 1024 -> 256) take 59 + 51 us as separate launches, and the
// phase trace of the first one (profiles/r02_split_gemm_astat_trace.txt) shows its matrix work done long before its
// stores: 91 MB of fp32 intermediate go out at ~1.5 TB/s and come back in for linear2.  Here the 1024-wide intermediate
// never leaves the CU.
//
//   * Same arithmetic as the separate kernels (linear_split.hip / linear_stream.hip): every product is the bf16 split
//     product of split_product.h (scheme SP: 16 = fp16 pieces, three terms, fp32-class: what trackformer_amd uses by default; 3 = six
//     bf16 terms; 2 = three bf16 terms, the fast mode)
//     on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, per accumulator smallest terms first, k ascending (the numbers in
//     this header are for NP = 2); the intermediate is rounded to fp32 (bias, ReLU) and split again,
//     exactly what linear2 does with linear1's stored output.  Without the LayerNorm the result is bit-identical to
//     tf_linear_packed_f32 (relu) -> tf_linear_packed_f32 -> + residual.
//   * A block owns 32 TI rows (TI = 3: 96 rows, 232 blocks for 22 223 rows: one round on 256 CUs).  Its activation tile
//     is staged ONCE in LDS as bf16 hi / mid (96 x 264 x 2 x 2 B = 101 KB, all global loads of the tile in flight
//     together).  The hidden dimension is processed in chunks of 128: GEMM 1 computes the chunk (each of the 4 waves
//     32 hidden columns x 96 rows), bias + ReLU + split go to a second LDS tile (96 x 136 x 2 x 2 B = 52 KB), GEMM 2
//     accumulates the chunk's contribution to the 96 x 256 output (each wave 64 output columns).  Two barriers per chunk.
//   * Both weights are in the packed fragment form of tf_linear_pack_weight_f32 and are streamed L2 -> registers through
//     one ring of eight 4 KB units per wave, six units (24 KB per wave) ahead of their use, ACROSS the GEMM 1 / GEMM 2 /
//     chunk boundaries (the one-slice-ahead version of linear_stream.hip left the waves waiting on L2 for 2/3 of their
//     time with one block per CU: profiles/r02_split_gemm_packed_pmc.txt).  A block streams both weights once (2 MB):
//     232 x 2 MB = 464 MB from L2, what the two separate kernels read together.
//   * The weight fragment is the A operand and the activation fragment B, so the accumulators hold the TRANSPOSED tile
//     (see linear_stream.hip, BUFST = 2): a lane owns 4 consecutive columns of one row -- the intermediate goes to LDS
//     with 8-byte writes, the output to HBM with 16-byte buffer stores (rows >= M fall outside the resource), the
//     LayerNorm's row sums are 32 adds and one cross-half shuffle per lane, plus an LDS exchange between the 4 waves.
//
// Hidden 288 (the multi-frame models) runs the same kernel in a three-wave geometry (struct Geo below); the numbers here are
// for hidden 256.  Per block and wave: 8 chunks x (16 + 8) k-steps x 9 / 18 MFMAs = 2304 MFMAs of 32 cycles = 30.7 us at 2.4 GHz: the
// kernel's floor is the matrix pipe (the three-term product costs 3x the bf16 flops); LDS reads (144 b128 per chunk and
// wave) are half of that, the weight stream 28 B/clk/CU of the 64 the vector memory path has.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "msda_common.h"
#include "split_product.h"
#include "tf_fused.h"
#include "tf_msda.h"

namespace {

// Geometry per hidden size.  256: four waves, each two of the eight output tiles, hidden chunks of 128.  288 (the multi-frame
// models: 9 output tiles): THREE waves of three tiles, hidden chunks of 96 -- the last chunk of a hidden width that is not a
// multiple of the chunk runs past it: the W1 tile index and the W2 k-step are clamped to the last ones the packed weights
// hold, and the hidden values of columns >= d_ffn are set to zero before they are split (0 x finite weight = 0 in GEMM 2).
template <int D>
struct Geo {
    static_assert(D == 256 || D == 288, "hidden sizes of the reference's configurations");
    static constexpr int NW = D == 256 ? 4 : 3;        // waves per block
    static constexpr int NT = NW * 64;                 // threads per block
    static constexpr int TJ = D / 32 / NW;             // output tiles per wave
    static constexpr int XS = D + 8;                   // bf16 per LDS row of the activation tile (528 / 592 B: 16-byte aligned)
    static constexpr int CH = NW * 32;                 // hidden columns per chunk: one 32-wide MFMA tile per wave
    static constexpr int HS = CH + 8;                  // bf16 per LDS row of the hidden tile
    static constexpr int KQ1 = D / 16;                 // k-steps of GEMM 1
    static constexpr int U1 = KQ1 / 2;                 // weight units of GEMM 1 per chunk (two k-steps of one tile each)
    static constexpr int U2 = CH / 16;                 // weight units of GEMM 2 per chunk (one k-step of TJ tiles each)
    static constexpr int UPC = U1 + U2;                // 16 / 15
    static constexpr int us(int nb) { return nb * (TJ > 2 ? TJ : 2); }   // 16-byte pieces per lane and unit (nb weight pieces)
    static constexpr int RING = D == 256 ? 8 : 5;      // units in the ring (divides UPC: a unit's slot is a compile-time constant)
    static constexpr int AHEAD = RING - 2;             // prefetch distance
    static_assert(TJ * NW * 32 == D && KQ1 % 2 == 0 && UPC % RING == 0, "geometry");
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int D>
constexpr size_t ffn_lds_bytes(int ti, int na) { return (size_t)(32 * ti) * (Geo<D>::XS + Geo<D>::HS) * 2 * na; }   // na activation pieces

// ---- the block's activation tile (32 TI rows x D) -> LDS as its 16-bit pieces.  All of its global loads are in flight before the
// first conversion waits.  The caller issues the barrier.
template <int SP, int D, int TI>
__device__ __forceinline__ void stage_rows(const float *__restrict__ X, int M, int m0, unsigned short *sX, int tid)   // sX: [NA][32 TI][XS]
{
    using G = Geo<D>;
    constexpr int C4 = D / 4, NV = TI * 32 * C4 / G::NT;   // float4 per row / per thread
    static_assert(TI * 32 * C4 % G::NT == 0, "tile does not divide among the threads");
    f32x4 xr[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int idx = it * G::NT + tid;
        const int row = idx / C4, c4 = idx - row * C4;
        const int grow = min(m0 + row, M - 1);   // rows past M read the last row, never stored
        xr[it] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * D + c4 * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const int idx = it * G::NT + tid;
        const int row = idx / C4, c4 = idx - row * C4;
        u32x2 pc[Split<SP>::NA];   // round to nearest even, as torch's .to(bfloat16) / .to(float16)
        split4<SP>(xr[it], pc);
#pragma unroll
        for (int p = 0; p < Split<SP>::NA; ++p) *reinterpret_cast<u32x2 *>(&sX[(p * TI * 32 + row) * G::XS + c4 * 4]) = pc[p];
    }
}

// sum of the NW waves' partial row sums, pairwise as ((0 + 1) + (2 + 3)) / ((0 + 1) + 2)
template <int NW>
__device__ __forceinline__ float wave_partials(const float *s, int stride)
{
    if constexpr (NW == 4) return (s[0] + s[stride]) + (s[2 * stride] + s[3 * stride]);
    else return (s[0] + s[stride]) + s[2 * stride];
}

// ---- the epilogue both kernels share: + bias + residual [-> LayerNorm] -> Y.  accy: transposed 32 x 32 tiles (lane -> output
// row m0 + 32 i + (lane & 31); registers 4 g .. 4 g + 3 of tile j -> columns 32 TJ wave + 32 j + 8 g + 4 (lane >> 5) + 0..3);
// v: the residual values in the same layout on entry.  sRed: [2 passes][NW waves][BM] floats of LDS that alias a tile every
// wave has finished reading once it reaches the first barrier in here.  Rows >= M: stores are dropped by the buffer resource.
// F16 (the fp16 scheme of split_product.h): the accumulators are multiplied by their output channel's power of two (r: D floats).
template <int D, int TI, bool LN, bool F16>
__device__ __forceinline__ void rows_epilogue(const f32x16 (&accy)[TI][Geo<D>::TJ], f32x4 (&v)[TI][Geo<D>::TJ][4],
                                              const __amdgpu_buffer_rsrc_t b2rs, const float *__restrict__ r, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, float eps, const __amdgpu_buffer_rsrc_t yrs, float *sRed,
                                              int m0, int wave, int lane)
{
    using G = Geo<D>;
    constexpr int BM = TI * 32, TJ = G::TJ, NW = G::NW;
    const int frow = lane & 31, cbase = wave * (32 * TJ) + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b2rs, (unsigned)(cbase + 32 * j + 8 * g) * 4u, 0, 0));
            f32x4 rv = {1.f, 1.f, 1.f, 1.f};
            if constexpr (F16) rv = *reinterpret_cast<const f32x4 *>(r + cbase + 32 * j + 8 * g);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[i][j][g][e] = (F16 ? __builtin_fmaf(accy[i][j][4 * g + e], rv[e], b[e]) : accy[i][j][4 * g + e] + b[e]) + v[i][j][g][e];
        }
    if constexpr (LN) {
        // two-pass statistics over the D columns of a row: 16 TJ values in this lane, as many in lane ^ 32, the rest in the
        // other waves (through LDS: the caller's scratch tile is free after the barrier)
        float s[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) a += (v[i][j][g].x + v[i][j][g].y) + (v[i][j][g].z + v[i][j][g].w);
            s[i] = a + __shfl_xor(a, 32);
        }
        __syncthreads();   // all waves are past their last reads of the tile sRed aliases
        if (lane < 32)
#pragma unroll
            for (int i = 0; i < TI; ++i) sRed[wave * BM + i * 32 + lane] = s[i];
        __syncthreads();
        float mean[TI], rstd[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            mean[i] = wave_partials<NW>(sRed + i * 32 + frow, BM) * (1.f / D);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 d = v[i][j][g] - mean[i];
                    a += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            s[i] = a + __shfl_xor(a, 32);
        }
        if (lane < 32)
#pragma unroll
            for (int i = 0; i < TI; ++i) sRed[(NW + wave) * BM + i * 32 + lane] = s[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TI; ++i) rstd[i] = rsqrtf(wave_partials<NW>(sRed + NW * BM + i * 32 + frow, BM) * (1.f / D) + eps);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + cbase + 32 * j + 8 * g);
                const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + cbase + 32 * j + 8 * g);
#pragma unroll
                for (int i = 0; i < TI; ++i) v[i][j][g] = (v[i][j][g] - mean[i]) * rstd[i] * ga + be;
            }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i][j][g]), yrs,
                                                       (unsigned)((m0 + i * 32 + frow) * D + cbase + 32 * j + 8 * g) * 4u, 0, tfm::kStoreAux);
}

template <int SP, int D, int TI, bool LN>
__global__ void __launch_bounds__(Geo<D>::NT, 1)
ffn_fused_kernel(const float *__restrict__ X, const u32x4 *__restrict__ W1p, const float *__restrict__ b1,
                 const u32x4 *__restrict__ W2p, const float *__restrict__ b2, const float *R,
                 const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float *Y, int M, int F)
{
    using G = Geo<D>;
    constexpr int BM = TI * 32, TJ = G::TJ, NW = G::NW, XS = G::XS, HS = G::HS, RING = G::RING, UPC = G::UPC, U1 = G::U1, U2 = G::U2;
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB, US = G::us(NB);
    constexpr bool F16 = Split<SP>::F16;
    extern __shared__ __attribute__((aligned(16))) unsigned short s_f[];
    unsigned short *const sX = s_f;                      // [NA][BM][XS]
    unsigned short *const sH = s_f + NA * BM * XS;       // [NA][BM][HS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const int nchunks = (F + G::CH - 1) / G::CH, KQ2 = F >> 4;
    const int ntiles1 = (F + 255) / 256 * 8;   // hidden n-tiles the packed W1 holds (zero columns up to a multiple of 256)

    // ---- the weight stream.  Unit u of chunk c (UPC units per chunk, up to US x 16 bytes per lane each):
    //   u < U1   GEMM 1: k-steps 2u, 2u + 1 (NP pieces each) of hidden n-tile NW c + wave of W1   [contiguous 2 NP KB]
    //   u >= U1  GEMM 2: k-step U2 c + (u - U1) (NP pieces) of the wave's TJ output n-tiles of W2 (clamped to the last real one)
    // packed layout (linear_stream.hip pack_weight_kernel): piece (n-tile t, k-step q, part p) at ((t KQ + q) NP + p) 64 + lane
    u32x4 ring[RING][US];
    auto load_unit = [&](int c, auto uc, u32x4 (&dst)[US]) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u < U1) {
            const u32x4 *base = W1p + ((size_t)min(c * NW + wave, ntiles1 - 1) * G::KQ1 * NB + u * 2 * NB) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) dst[i] = base[i * 64];
        } else {
            const int q = min(c * U2 + (u - U1), KQ2 - 1);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int p = 0; p < NB; ++p) dst[j * NB + p] = W2p[(((size_t)(TJ * wave + j) * KQ2 + q) * NB + p) * 64 + lane];
        }
    };
    // the first units first (they have the longest way), then the whole activation tile
    static_for<G::AHEAD>([&](auto uc) { load_unit(0, uc, ring[decltype(uc)::value]); });
    __builtin_amdgcn_sched_barrier(0);
    stage_rows<SP, D, TI>(X, M, m0, sX, tid);
    __syncthreads();

    f32x16 accy[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accy[i][j][e] = 0.f;

    // fp16 scheme: the per-channel powers of two lie behind the fragments of each packed weight (linear_stream.hip)
    const float *const r1 = reinterpret_cast<const float *>(W1p + (size_t)ntiles1 * G::KQ1 * NB * 64);            // ntiles1 * 32 floats
    const float *const r2 = reinterpret_cast<const float *>(W2p + (size_t)((D + 255) / 256 * 8) * KQ2 * NB * 64);   // >= D floats
    const int frow = (lane & 31), fk = (lane >> 5) * 8;   // fragment: lane -> (row of the tile, first of 8 consecutive k)
    const int xoff = frow * XS + fk, hoff = frow * HS + fk;

    // every bias / LayerNorm vector through a buffer resource (NULL: zero records -> zeros, no branch around the load)
    const __amdgpu_buffer_rsrc_t b1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b1 ? b1 : X), 0, b1 ? (unsigned)F * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t b2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b2 ? b2 : X), 0, b2 ? (unsigned)D * 4u : 0u, 0x00020000);
    const unsigned bytes = (unsigned)((size_t)M * D * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : X), 0, R ? bytes : 0u, 0x00020000);
    const int cbase = wave * (32 * TJ) + 4 * (lane >> 5);
    f32x4 v[TI][TJ][4];   // the residual rows, then the output values

    for (int c = 0; c < nchunks; ++c) {
        const int cn = min(c + 1, nchunks - 1);   // after the last chunk: a harmless reload
        // this chunk's hidden bias, 4 consecutive columns per register group (transposed tile, see below)
        f32x4 b1v[4], r1v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            r1v[g] = f32x4{1.f, 1.f, 1.f, 1.f};
            b1v[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   b1rs, (unsigned)(c * G::CH + wave * 32 + 8 * g + 4 * (lane >> 5)) * 4u, 0, 0));
            if constexpr (F16)   // columns past the hidden width (last chunk of hidden 288): any in-range entry, the value is zeroed below
                r1v[g] = *reinterpret_cast<const f32x4 *>(r1 + min(c * G::CH + wave * 32 + 8 * g + 4 * (lane >> 5), ntiles1 * 32 - 4));
        }
        if (c == nchunks - 1) {
            // the residual rows: in flight during the last chunk.  Lane -> output row m0 + 32 i + (lane & 31); registers
            // 4 g .. 4 g + 3 of tile j -> columns 32 TJ wave + 32 j + 8 g + 4 (lane >> 5) + 0..3; rows >= M return zeros
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        v[i][j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   rrs, (unsigned)((m0 + i * 32 + frow) * D + cbase + 32 * j + 8 * g) * 4u, 0, 0));
        }
        f32x16 acch[TI][1];   // ([.][1]: the tile-array form mfma_tiles takes)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acch[i][0][e] = 0.f;

        auto prefetch = [&](auto uc) {
            constexpr int ahead = decltype(uc)::value + G::AHEAD;
            if constexpr (ahead < UPC) load_unit(c, std::integral_constant<int, ahead>{}, ring[ahead % RING]);
            else load_unit(cn, std::integral_constant<int, ahead - UPC>{}, ring[ahead % RING]);
        };
        // ---- GEMM 1: hidden[32 of this wave][BM rows] (transposed) = W1 tile . x^T.  One block per CU means one wave per
        // SIMD: nobody else covers the LDS latency, so the fragments of k-step st + 1 are read before the MFMAs of k-step st
        u32x4 xf[2][TI][NA];   // [k-step parity][row tile][piece]
        auto read_x = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int p = 0; p < NA; ++p)
                    xf[st & 1][i][p] = *reinterpret_cast<const u32x4 *>(&sX[(p * BM + i * 32) * XS + xoff + st * 16]);
        };
        read_x(std::integral_constant<int, 0>{});
        static_for<G::KQ1>([&](auto stc) {
            constexpr int st = decltype(stc)::value;   // k-step; weight unit st / 2
            if constexpr ((st & 1) == 0) prefetch(std::integral_constant<int, st / 2>{});
            if constexpr (st + 1 < G::KQ1) read_x(std::integral_constant<int, st + 1>{});
            __builtin_amdgcn_sched_barrier(0);   // loads and reads stay at the head of the step (see linear_stream.hip)
            const u32x4 (&cur)[US] = ring[(st / 2) % RING];
            u32x4 wf[1][NB];
#pragma unroll
            for (int p = 0; p < NB; ++p) wf[0][p] = cur[(st & 1) * NB + p];
            // the weight fragment is the A operand (transposed accumulators); x piece T::A[t] x weight operand T::B[t], smallest first
            mfma_tiles<SP, TI, 1, false>(acch, xf[st & 1], wf);
        });

        // ---- bias + ReLU + split -> the hidden tile in LDS.  C/D of the 32 x 32 MFMA with the weight as A: lane -> row
        // (lane & 31) of x, registers 4 g .. 4 g + 3 -> hidden columns 8 g + 4 (lane >> 5) + 0..3 of the wave's 32
        __syncthreads();   // every wave is past GEMM 2 of the previous chunk: the hidden tile is free
        const int hcol = c * G::CH + wave * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = F16 ? __builtin_fmaf(acch[i][0][4 * g + e], r1v[g][e], b1v[g][e]) : acch[i][0][4 * g + e] + b1v[g][e];
                    hv[e] = (t < 0.f || hcol + 8 * g >= F) ? 0.f : t;   // columns past the hidden width (last chunk only): zero
                }
                u32x2 pc[NA];
                split4<SP>(hv, pc);
                const int o = (i * 32 + frow) * HS + wave * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                for (int p = 0; p < NA; ++p) *reinterpret_cast<u32x2 *>(&sH[p * BM * HS + o]) = pc[p];
            }
        __syncthreads();

        // ---- GEMM 2: y[32 TJ columns of this wave][BM rows] (transposed) += W2 tiles . hidden^T, k = this chunk
        u32x4 hf[2][TI][NA];
        auto read_h = [&](auto vc) {
            constexpr int kv = decltype(vc)::value;   // k-step of the chunk
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int p = 0; p < NA; ++p)
                    hf[kv & 1][i][p] = *reinterpret_cast<const u32x4 *>(&sH[(p * BM + i * 32) * HS + hoff + kv * 16]);
        };
        read_h(std::integral_constant<int, 0>{});
        static_for<U2>([&](auto vc) {
            constexpr int kv = decltype(vc)::value;   // weight unit U1 + kv
            prefetch(std::integral_constant<int, U1 + kv>{});
            if constexpr (kv + 1 < U2) read_h(std::integral_constant<int, kv + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&cur)[US] = ring[(U1 + kv) % RING];
            u32x4 wf[TJ][NB];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int p = 0; p < NB; ++p) wf[j][p] = cur[j * NB + p];
            mfma_tiles<SP, TI, TJ, false>(accy, hf[kv & 1], wf);
        });
    }

    // ---- epilogue (the LayerNorm's exchange buffer aliases the hidden tile)
    rows_epilogue<D, TI, LN, F16>(accy, v, b2rs, r2, gamma, beta, eps, yrs, reinterpret_cast<float *>(sH), m0, wave, lane);
}

template <int SP, int D, int TI>
int launch_ffn(const float *x, const u32x4 *w1, const float *b1, const u32x4 *w2, const float *b2, const float *res,
               const float *gamma, const float *beta, float eps, float *y, int M, int F, hipStream_t s)
{
    const bool ln = gamma != nullptr;
    constexpr size_t lds = ffn_lds_bytes<D>(TI, Split<SP>::NA);
    static_assert(lds <= 160 * 1024, "the two tiles do not fit the LDS of a CU");
    const void *fn = ln ? (const void *)&ffn_fused_kernel<SP, D, TI, true> : (const void *)&ffn_fused_kernel<SP, D, TI, false>;
    static std::atomic<unsigned> raised[2];   // bit per device, per kernel
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 32 || !(raised[ln ? 1 : 0].load() & (1u << dev))) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TF_MSDA_ERR_LAUNCH;
        if (dev < 32) raised[ln ? 1 : 0].fetch_or(1u << dev);
    }
    const int blocks = (M + 32 * TI - 1) / (32 * TI);
    void *argv[] = {(void *)&x, (void *)&w1, (void *)&b1, (void *)&w2, (void *)&b2, (void *)&res,
                    (void *)&gamma, (void *)&beta, (void *)&eps, (void *)&y, (void *)&M, (void *)&F};
    return hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(Geo<D>::NT), argv, lds, s) == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

// ---- tf_linear_res_ln_f32: y = [LayerNorm](residual + x . w^T + bias) for a D x D weight -- the attention's output
// projection with the layer's residual add and norm1 (deformable_transformer.py:285-292 / ms_deform_attn.py:87).  The GEMM 2
// half of the kernel above with the activation tile as its operand: D / 16 k-steps, each wave 32 TJ output columns.
template <int D>
constexpr size_t linln_lds_bytes(int ti, int na) { return (size_t)(32 * ti) * Geo<D>::XS * 2 * na; }
constexpr int linln_min_blocks(int d, int ti) { return (d == 256 ? ti <= 2 : ti <= 1) ? 2 : 1; }   // resident blocks per CU the register budget is cut for
constexpr int linln_ring(int ti) { return ti <= 2 ? 4 : 8; }          // weight units in flight + 2

template <int SP, int D, int TI, bool LN>
__global__ void __launch_bounds__(Geo<D>::NT, (linln_min_blocks(D, TI)))
linear_res_ln_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias, const float *R,
                     const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float *Y, int M)
{
    using G = Geo<D>;
    constexpr int BM = TI * 32, TJ = G::TJ, XS = G::XS, KQ = G::KQ1, NA = Split<SP>::NA, NB = Split<SP>::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned short s_f[];
    unsigned short *const sX = s_f;   // [NA][BM][XS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    // weight unit u = k-step u (NP pieces) of the wave's TJ output n-tiles; a ring as above, RING - 2 units ahead
    constexpr int RING = linln_ring(TI), AHEAD = RING - 2;
    u32x4 ring[RING][NB * TJ];
    auto load_unit = [&](auto uc, u32x4 (&dst)[NB * TJ]) {
        constexpr int u = decltype(uc)::value;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int p = 0; p < NB; ++p) dst[j * NB + p] = Wp[(((size_t)(TJ * wave + j) * KQ + u) * NB + p) * 64 + lane];
    };
    static_for<AHEAD>([&](auto uc) { load_unit(uc, ring[decltype(uc)::value]); });
    __builtin_amdgcn_sched_barrier(0);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bias ? bias : X), 0, bias ? (unsigned)D * 4u : 0u, 0x00020000);
    const unsigned bytes = (unsigned)((size_t)M * D * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : X), 0, R ? bytes : 0u, 0x00020000);
    const int frow = lane & 31, cbase = wave * (32 * TJ) + 4 * (lane >> 5);
    f32x4 v[TI][TJ][4];   // the residual rows (rows >= M return zeros), then the output values
    stage_rows<SP, D, TI>(X, M, m0, sX, tid);
    __syncthreads();

    f32x16 accy[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accy[i][j][e] = 0.f;
    const int xoff = frow * XS + (lane >> 5) * 8;
    u32x4 xf[2][TI][NA];   // [k-step parity][row tile][piece]: the fragments of step st + 1 are read before the MFMAs of st
    auto read_x = [&](auto stc) {
        constexpr int st = decltype(stc)::value;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int p = 0; p < NA; ++p)
                xf[st & 1][i][p] = *reinterpret_cast<const u32x4 *>(&sX[(p * BM + i * 32) * XS + xoff + st * 16]);
    };
    read_x(std::integral_constant<int, 0>{});
    static_for<KQ>([&](auto stc) {
        constexpr int st = decltype(stc)::value;
        if constexpr (st + AHEAD < KQ) load_unit(std::integral_constant<int, st + AHEAD>{}, ring[(st + AHEAD) % RING]);
        if constexpr (st == KQ - AHEAD) {   // the weight stream has ended: the residual rows take its place in the queue
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        v[i][j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   rrs, (unsigned)((m0 + i * 32 + frow) * D + cbase + 32 * j + 8 * g) * 4u, 0, 0));
        }
        if constexpr (st + 1 < KQ) read_x(std::integral_constant<int, st + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 (&cur)[NB * TJ] = ring[st % RING];
        u32x4 wf[TJ][NB];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int p = 0; p < NB; ++p) wf[j][p] = cur[j * NB + p];
        mfma_tiles<SP, TI, TJ, false>(accy, xf[st & 1], wf);
    });
    const float *const r = reinterpret_cast<const float *>(Wp + (size_t)((D + 255) / 256 * 8) * KQ * NB * 64);   // fp16 scheme: see ffn_fused_kernel
    rows_epilogue<D, TI, LN, Split<SP>::F16>(accy, v, brs, r, gamma, beta, eps, yrs, reinterpret_cast<float *>(sX), m0, wave, lane);
}

template <int SP, int D, int TI>
int launch_linln(const float *x, const u32x4 *w, const float *b, const float *res, const float *gamma, const float *beta, float eps,
                 float *y, int M, hipStream_t s)
{
    const bool ln = gamma != nullptr;
    constexpr size_t lds = linln_lds_bytes<D>(TI, Split<SP>::NA);
    static_assert(lds <= 160 * 1024, "the activation tile does not fit the LDS of a CU");
    const void *fn = ln ? (const void *)&linear_res_ln_kernel<SP, D, TI, true> : (const void *)&linear_res_ln_kernel<SP, D, TI, false>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned> raised[2];   // bit per device, per kernel
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 32 || !(raised[ln ? 1 : 0].load() & (1u << dev))) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TF_MSDA_ERR_LAUNCH;
            if (dev < 32) raised[ln ? 1 : 0].fetch_or(1u << dev);
        }
    }
    const int blocks = (M + 32 * TI - 1) / (32 * TI);
    void *argv[] = {(void *)&x, (void *)&w, (void *)&b, (void *)&res, (void *)&gamma, (void *)&beta, (void *)&eps, (void *)&y, (void *)&M};
    return hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(Geo<D>::NT), argv, lds, s) == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

std::atomic<int> g_linln_ti{-1};   // -1: TF_LINLN_TI or automatic (0)
int linln_ti()
{
    int v = g_linln_ti.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_LINLN_TI");
        v = e ? atoi(e) : 0;
        if (v < 0 || v > 3) v = 0;
        g_linln_ti.store(v);
    }
    return v;
}

std::atomic<int> g_ffn_tail{-1};   // -1: TF_FFN_TAIL_SPLIT or the default (1): the rows behind the full rounds as 32-row blocks
std::atomic<int> g_ffn_ti{-1};   // -1: TF_FFN_TI or the default (3)
int ffn_ti()
{
    int v = g_ffn_ti.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_FFN_TI");
        v = e ? atoi(e) : 3;
        if (v < 1 || v > 3) v = 3;
        g_ffn_ti.store(v);
    }
    return v;
}

}  // namespace

namespace tfm {
int ffn_tail_split()
{
    int v = g_ffn_tail.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_FFN_TAIL_SPLIT");
        v = (e && e[0] == '0') ? 0 : 1;
        g_ffn_tail.store(v);
    }
    return v;
}
int ffn_set_tail_split(int v)
{
    const int prev = ffn_tail_split();
    g_ffn_tail.store(v ? 1 : 0);
    return prev;
}
int ffn_set_ti(int v)
{
    const int prev = ffn_ti();
    g_ffn_ti.store(v >= 1 && v <= 3 ? v : 3);
    return prev;
}
int linln_set_ti(int v)
{
    const int prev = linln_ti();
    g_linln_ti.store(v >= 1 && v <= 3 ? v : 0);
    return prev;
}
}  // namespace tfm

namespace {

int ffn_num_cus()
{
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    return n;
}

template <int SP, int D>
int dispatch_linln(const float *x, const u32x4 *w, const float *bias, const float *residual, const float *g, const float *b, float eps,
                   float *y, int M, hipStream_t s)
{
    // rows per block = 32 TI; automatic: few rows -> 32-row blocks (more of them), many -> 64 (two resident per CU with two
    // activation pieces; with three the 64-row tile is 99 KB of LDS: one block per CU)
    const int forced = linln_ti();
    // Three pieces: a 64-row tile is 99 KB of LDS -- one block per CU, rounds of `cus` blocks (348 blocks at the cfg-2 encoder: a
    // full round and a third of one).  32-row tiles (50 KB) keep three blocks resident per CU and the 695 blocks run as one
    // round: 26.2 us against 31.2 us with 64-row blocks (29.0 with the rows behind the full round split off as dispatch_ffn
    // does; profiles/r04_one_launch_blocks_tail_split.txt).  Two pieces: 64-row tiles, two resident per CU, as measured in round 3.
    // fp16 pieces, hidden 256: 96-row blocks (20.1 us against 21.1 / 22.6 with 32 / 64 rows; profiles/r04_f16_harness.txt)
    switch (forced ? forced : (M < 4096 || Split<SP>::NA == 3 ? 1 : (SP == 16 && D == 256 ? 3 : 2))) {
    case 1: return launch_linln<SP, D, 1>(x, w, bias, residual, g, b, eps, y, M, s);
    case 3:   // 96 rows per block: hidden 256 with two bf16 pieces only (at 288 the accumulators of three row tiles do not fit the
              // register file, with three pieces the tile does not fit the LDS / the weight ring does not fit the registers)
        if constexpr (D == 256 && Split<SP>::NB == 2) return launch_linln<SP, D, 3>(x, w, bias, residual, g, b, eps, y, M, s);
        else return launch_linln<SP, D, 2>(x, w, bias, residual, g, b, eps, y, M, s);
    default: return launch_linln<SP, D, 2>(x, w, bias, residual, g, b, eps, y, M, s);
    }
}

template <int SP, int D>
int dispatch_ffn(const float *x, const u32x4 *w1, const float *b1, const u32x4 *w2, const float *b2, const float *residual,
                 const float *g, const float *b, float eps, float *y, int M, int F, hipStream_t s)
{
    if (F < Geo<D>::CH || (F & 15)) return TF_MSDA_ERR_BAD_DIMS;   // at least one chunk; whole k-steps of GEMM 2
    // One block per CU (the two LDS tiles fill it), so a launch runs in ROUNDS of `cus` blocks.  With 64-row blocks (three
    // pieces, or hidden 288) the 22 223 rows of the cfg-2 encoder are 348 blocks = one full round + 92 blocks: the second round
    // takes as long as the first with a third of the chip (profiles/r04_pmc_dense_six_terms.txt: 63 % of a wave's cycles are
    // MFMA-busy, 38 % of the launch's).  When the last round would be less than half full, the rows behind the full rounds go
    // to a second launch of 32-row blocks: a short round instead of a long one.  Same arithmetic per row (a row's result does
    // not depend on the block it is in): bit-identical.
    constexpr bool three_tiles = D == 256 && Split<SP>::NB == 2;   // 96 rows per block: two stored weight pieces only (with three the
                                                                   // ring of weight units leaves no room for a third row tile's accumulators)
    const int want = ffn_ti();
    const bool ti2 = want == 2 || (want >= 3 && !three_tiles);
    if (ti2 && tfm::ffn_tail_split()) {
        const int cus = ffn_num_cus();
        const long long per_round = 64LL * cus;
        const int main_rows = (int)(M / per_round * per_round), rem = M - main_rows;
        if (main_rows > 0 && rem > 0 && rem <= 32LL * cus) {
            int rc = launch_ffn<SP, D, 2>(x, w1, b1, w2, b2, residual, g, b, eps, y, main_rows, F, s);
            if (rc != TF_MSDA_OK) return rc;
            const size_t off = (size_t)main_rows * D;
            return launch_ffn<SP, D, 1>(x + off, w1, b1, w2, b2, residual ? residual + off : nullptr, g, b, eps, y + off, rem, F, s);
        }
    }
    switch (want) {
    case 1: return launch_ffn<SP, D, 1>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
    case 2: return launch_ffn<SP, D, 2>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
    default:
        if constexpr (three_tiles) return launch_ffn<SP, D, 3>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
        else return launch_ffn<SP, D, 2>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
    }
}

template <int D>
int dispatch_linln_scheme(int sp, const float *x, const u32x4 *w, const float *bias, const float *residual, const float *g, const float *b,
                          float eps, float *y, int M, hipStream_t s)
{
    switch (sp) {
    case 3: return dispatch_linln<3, D>(x, w, bias, residual, g, b, eps, y, M, s);
    default: return dispatch_linln<16, D>(x, w, bias, residual, g, b, eps, y, M, s);
    }
}

template <int D>
int dispatch_ffn_scheme(int sp, const float *x, const u32x4 *w1, const float *b1, const u32x4 *w2, const float *b2, const float *residual,
                        const float *g, const float *b, float eps, float *y, int M, int F, hipStream_t s)
{
    switch (sp) {
    case 3: return dispatch_ffn<3, D>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
    default: return dispatch_ffn<16, D>(x, w1, b1, w2, b2, residual, g, b, eps, y, M, F, s);
    }
}

}  // namespace

extern "C" int tf_linear_res_ln_f32(const float *x, const void *w_packed, const float *bias, const float *residual,
                                    const float *ln_weight, const float *ln_bias, float ln_eps, float *y, int64_t M, int K, int N,
                                    int terms, void *stream)
{
    if (!x || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    if ((ln_weight == nullptr) != (ln_bias == nullptr)) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (M <= 0 || K != N || (K != 256 && K != 288) || (M + 128) * (int64_t)K * 4 > 0xFFFFFFFFLL || sp == 0) return TF_MSDA_ERR_BAD_DIMS;
    uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(y) |
                   reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(ln_weight) |
                   reinterpret_cast<uintptr_t>(ln_bias);
    if (al & 15) return TF_MSDA_ERR_BAD_DIMS;
    const u32x4 *w = static_cast<const u32x4 *>(w_packed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return K == 256 ? dispatch_linln_scheme<256>(sp, x, w, bias, residual, ln_weight, ln_bias, ln_eps, y, (int)M, s)
                    : dispatch_linln_scheme<288>(sp, x, w, bias, residual, ln_weight, ln_bias, ln_eps, y, (int)M, s);
}

extern "C" int tf_ffn_fused_f32(const float *x, const void *w1_packed, const float *b1, const void *w2_packed, const float *b2,
                                const float *residual, const float *ln_weight, const float *ln_bias, float ln_eps, float *y,
                                int64_t M, int d_model, int d_ffn, int terms, void *stream)
{
    if (!x || !w1_packed || !w2_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    if ((ln_weight == nullptr) != (ln_bias == nullptr)) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (M <= 0 || (d_model != 256 && d_model != 288) || d_ffn <= 0 || sp == 0 ||
        (M + 128) * (int64_t)d_model * 4 > 0xFFFFFFFFLL)   // 32-bit buffer offsets, incl. the rows of the last block past M
        return TF_MSDA_ERR_BAD_DIMS;
    uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1_packed) | reinterpret_cast<uintptr_t>(w2_packed) |
                   reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(ln_weight) | reinterpret_cast<uintptr_t>(ln_bias);
    if (al & 15) return TF_MSDA_ERR_BAD_DIMS;
    const u32x4 *w1 = static_cast<const u32x4 *>(w1_packed), *w2 = static_cast<const u32x4 *>(w2_packed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return d_model == 256 ? dispatch_ffn_scheme<256>(sp, x, w1, b1, w2, b2, residual, ln_weight, ln_bias, ln_eps, y, (int)M, d_ffn, s)
                          : dispatch_ffn_scheme<288>(sp, x, w1, b1, w2, b2, residual, ln_weight, ln_bias, ln_eps, y, (int)M, d_ffn, s);
}
