// trackformer_amd/csrc/linear_stream.hip
//
// tf_linear_packed_f32 / tf_linear_pack_weight_f32 (include/tf_fused.h): the same arithmetic as tf_linear_split_f32
// (linear_split.hip: Y = X . W^T + bias as a split product on v_mfma_f32_32x32x16_{f16,bf16} -- fp16 pieces, six bf16 terms or three
// in the fast mode -- split_product.h; the description below counts for the three-term form; reference
// modules: models/ops/modules/ms_deform_attn.py:64-88, models/deformable_transformer.py:282-297), restructured around
// what profiles/r02_split_gemm_mfma_*.json showed: the first kernel keeps the matrix pipes 23 % busy because both
// operands make an LDS round trip per 32-wide K-slice between two barriers and a wave only owns 1 x 2 MFMA tiles
// (12 ds_read_b128 per 12 MFMAs: the LDS pipe is as busy as the matrix pipe).
//
//   * The weight is a constant: tf_linear_pack_weight_f32 splits it ONCE into bf16 (hi, mid) and stores it in MFMA
//     FRAGMENT order -- for n-tile t (32 output features) and k-step q (16 inputs) the 64 lanes' 16-byte pieces are
//     contiguous (1 KB hi, then 1 KB mid).  The GEMM reads weight fragments straight from global memory / L2 into
//     registers with perfectly coalesced 1 KB wave loads, one K-slice ahead: the weight never touches LDS, and the
//     vector-memory path (64 B/clk/CU) works in parallel with the LDS path (128 B/clk/CU) instead of queueing behind it.
//   * Only the activations go through LDS (fp32 from HBM -> registers -> split -> bf16 hi / mid tiles), double
//     buffered: ONE barrier per K-slice, and the global loads run two slices ahead of their use.
//   * A block is (32 TI) rows x 256 columns, its 4 waves split the COLUMNS (each wave TI x 2 MFMA tiles): every wave
//     reads the whole activation tile from LDS but owns its weight fragments -- TI = 3: 12 ds_read_b128 and 8 global
//     fragment loads per 36 MFMAs per slice.  TI is chosen per shape (choose_ti below).
//   * Blocks that share activation rows (N > 256) are given consecutive slots on the SAME XCD (id & 7), so the rows
//     are read from HBM once and from that XCD's L2 afterwards.
//   * Accumulation order per output element is the one of linear_split.hip (per k-step: mid.hi, hi.mid, hi.hi), so
//     the two kernels give bit-identical results (tools/linear_bench checks that).
//
// Measured (profiles/r02_split_gemm_packed.txt, 22 223 rows): 256 -> 1024: 76.2 -> 59.3 us, 1024 -> 256: 63.6 -> 50.5 us;
// 256 -> 256 and 256 -> 384: 22.8 / 30.6 us against 21.4 / 28.9 us -- at K = 256 and N <= 384 a launch is ~230-350 blocks
// of 8 slices each, bound by memory latency rather than by either pipe, and the many small blocks of linear_split.hip
// hide that better.  trackformer_amd/fused.py therefore routes only the FFN shapes (N >= 512 or K >= 512) here.
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

constexpr int kThreads = 256, kSlice = 32;   // K per slice (two MFMA k-steps of 16)
constexpr int kStride = kSlice + 8;          // bf16 per LDS row: 80 bytes (16-byte aligned, 8 rows cover all banks)
#ifndef TF_STREAM_XDEPTH
#define TF_STREAM_XDEPTH 2   // register stages of activations ahead in the convolution form (4: measured in round 5, no gain; tools/build_variant.py for A/B)
#endif
#ifndef TF_HALO_ABLATE
#define TF_HALO_ABLATE 0     // timing ablations of conv3x3_halo_kernel (tools only): 1 weights from L1, 2 no re-staging, 4 no MFMAs
#endif
#ifndef TF_STREAM_ABLATE
#define TF_STREAM_ABLATE 0   // timing ablations of stream_gemm_kernel (tools only): 1 no activation split, 2 no MFMAs
#endif
constexpr int kTJ = 2;                       // MFMA column tiles per wave -> 4 waves x 2 x 32 = 256 columns per block
constexpr int kBN = 4 * kTJ * 32;

// ---- weight packing: one thread per (n-tile, k-step, lane) writes its NB pieces: piece p of (n-tile t, k-step q) at
// ((t KQ + q) NB + p) 64 + lane.  fp16 scheme: behind the fragments lie npad floats r_n = 1 / (kActScale t_n) (the epilogue's
// factor per output channel, split_product.h), written by pack_scale_kernel before this kernel runs.
__global__ void __launch_bounds__(256)
pack_scale_kernel(const float *__restrict__ W, float *__restrict__ rs, int K, int N, int npad)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per output channel
    if (n >= npad) return;
    float amax = 0.f;
    if (n < N)
        for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(W[(size_t)n * K + k]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    if (lane == 0) rs[n] = 1.f / (kActScale * weight_scale_for(amax));
}

template <int SP>
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float *__restrict__ W, u32x4 *__restrict__ out, const float *__restrict__ rs, int K, int N, long long total)
{
    constexpr int NB = Split<SP>::NB;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int KQ = K >> 4;
    const int lane = (int)(idx & 63);
    const long long tq = idx >> 6;
    const int q = (int)(tq % KQ), t = (int)(tq / KQ);
    const int n = t * 32 + (lane & 31), k = q * 16 + (lane >> 5) * 8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {   // rows past N (padding up to a whole block of columns): zeros
        a = *reinterpret_cast<const f32x4 *>(W + (size_t)n * K + k);
        b = *reinterpret_cast<const f32x4 *>(W + (size_t)n * K + k + 4);
        if constexpr (Split<SP>::F16) {
            const float tn = 1.f / (kActScale * rs[n]);   // powers of two: exact
            a *= tn;
            b *= tn;
        }
    }
    u32x2 pa[NB], pb[NB];   // round to nearest even, as torch's .to(bfloat16) / .to(float16)
    split4_weight<SP>(a, pa);
    split4_weight<SP>(b, pb);
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const u32x4 v = {pa[p].x, pa[p].y, pb[p].x, pb[p].y};
        out[(tq * NB + p) * 64 + lane] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// THE STREAM GEMM (round 4: one kernel for the linears, the 1 x 1 convolutions incl. the residual epilogue of a bottleneck, and
// -- as an implicit GEMM over the output pixels -- the 3 x 3 / strided convolutions; it replaces the LDS-staged block kernels
// of linear_split.hip wherever the weight is packed).  Why (profiles/r04_pmc_dense_six_terms.txt, six terms, MI355X): at equal
// matrix work the block kernel that stages BOTH operands in LDS takes 88.0 us (22 223 x 1024 -> 256) / 84.7 us (256 -> 1024) /
// 26.3 us (256 -> 256), this structure 65.6 / 67.5 / 23.9 us: a wave of the block kernel reads (TI + TJ) x NP KB of LDS per
// TI x TJ x 6 MFMAs, here the weight fragments come straight from L2 in fragment order and only the activations pass LDS.
//
//   block   256 threads = 4 waves as WR x WC (rows x columns), WC in {4, 2};  wave tile = TI x TJ MFMA tiles of 32 x 32
//           BM = WR TI 32 rows, BN = WC TJ 32 columns:  (WC, TJ) = (4, 2): 256 columns, (4, 1): 128, (2, 1): 64 (WR = 2)
//   A       fp32 rows (MODE 0) or shifted input pixels of one tap (MODE 1: buffer loads, taps outside the image read zeros from
//           beyond num_records) -> registers two slices ahead -> bf16 pieces -> LDS, double buffered: ONE barrier per 32-wide slice
//   B       packed fragments (pack_weight_kernel), L2 -> registers one slice ahead
//   split-K blockIdx.y walks `kslices` slices and writes partial sums to Y + z M N (host: no bias / ReLU / residual then; a
//           second launch adds the pieces in a fixed order)
//   epilogue  + bias, + residual (R may alias Y), ReLU; buffer stores (rows >= M / columns >= N fall outside the resource)
// Measured and NOT kept (round 4, fp16 pieces; profiles/r04_pmc_dense_fp16.txt has the counters that prompted them: 7 vector
// instructions per MFMA in the convolution form, waves issuing 37 % / issue-stalled 30 % / parked 33 % of their cycles):
//   * the weight fragments three slices ahead in a ring of four register stages for the convolution form (one or two blocks per
//     CU, each walking a long K): 1363 against 1326 us of backbone convolutions per frame, every 3 x 3 layer 1-2 us slower -- the
//     weight latency is not what those launches wait for (profiles/r04_stream_conv_weight_ring.txt);
//   * the split of slice s + 1 placed in front of slice s's MFMAs with sched_group_barrier asking for one MFMA, then a few
//     vector instructions, and so on: the compiler interleaves the LDS reads but leaves most of the split behind the MFMAs;
//     1340 us per frame, the linears 3-8 % slower (profiles/r04_stream_interleave.txt).
//   * 128-row blocks for every convolution (more independent accumulators per wave; TF_LINEAR_STREAM_TI=4): 1570 against 1309 us
//     per frame -- half the blocks, and the split-K policy does not make up for it (profiles/r04_stream_conv_row_tiles.txt).
// The obvious suspect after that, the split itself (every input pixel of a convolution is cut into pieces once per tap and column
// block), was then ABLATED before anything was built on it (raw bits to LDS instead of the pieces: wrong results, right amount of
// loads / LDS traffic / MFMAs; profiles/r04_stream_conv_ablate_split.txt): the 3 x 3 convolutions go from 33-34 us to 28-32, the
// backbone's convolutions from 1350 to 1272 us per frame, the 1024 -> 256 linear from 46.8 to 42.4 us -- the split is 6-17 % of
// these launches, not the two thirds its share of the instruction count suggests.  What a slice of this kernel waits for is not
// instruction issue; the next step is a phase trace (s_memrealtime stamps per slice, as for the encoder kernel), DESIGN.md section 9.
// Accumulation order per output element: k ascending, per k-step smallest terms first -- the order of linear_split.hip, so
// the results are bit-identical to tf_linear_split_f32 / tf_conv3x3_split_f32 with the same number of terms.
struct StreamConv {
    int nimg, hin, win, cin, hout, wout, ks, pad, stride;
};

// PHASE TRACE (tools/build_stream_trace.py builds a separate library with -DTF_STREAM_TRACE; libtf_msda.so never contains it).
// Version 2.  The first version (six stamps per slice, each written to memory at once: profiles/r04_stream_phase_trace.txt) doubled
// the launch time -- reading s_memtime waits for lgkmcnt(0), i.e. for the LDS reads in flight, and every stamp was a store with
// its address arithmetic.  Here wave 0 of the first kTraceBlocks blocks stamps THREE points per slice, each at a place where the
// LDS counter is drained anyway, keeps the stamps of slices kTraceFirst .. kTraceFirst + 3 in scalar registers and writes them once,
// behind the epilogue:
//   0 slice start (the barrier of the previous slice passed)   1 this slice's MFMAs issued (weight loads of the next slice issued
//   before them, LDS fragments read)   2 next slice split, written to LDS, its global loads issued: in front of the barrier
// so that 0 -> 1 is the matrix phase, 1 -> 2 the staging phase and 2 -> next 0 the wait at the barrier.
#ifdef TF_STREAM_TRACE
constexpr int kTraceBlocks = 64, kTraceSlices = 4, kTracePoints = 3, kTraceFirst = 4;
__device__ unsigned long long *g_stream_trace = nullptr;
struct TraceRegs {
    unsigned long long t[kTraceSlices][kTracePoints];
};
template <int PT>
__device__ __forceinline__ void trace_stamp(TraceRegs &r, int slot)
{
    const unsigned long long c = __builtin_readcyclecounter();
    if (slot == kTraceFirst) r.t[0][PT] = c;
    else if (slot == kTraceFirst + 1) r.t[1][PT] = c;
    else if (slot == kTraceFirst + 2) r.t[2][PT] = c;
    else if (slot == kTraceFirst + 3) r.t[3][PT] = c;
}
#define TF_TRACE(slot, pt) trace_stamp<pt>(trace_regs, (slot))
#else
#define TF_TRACE(slot, pt) \
    do {                   \
    } while (0)
#endif

template <int NB, int TJ>
struct WFrags {
    u32x4 v[TJ][2][NB];   // [column tile][k-step of the slice][weight piece]
};

constexpr int stream_min_waves(int ti, int tj, int wc) { return (4 / wc) * ti * tj <= 6 && ti <= 3 ? 2 : 1; }   // blocks per CU the register budget is cut for

template <int SP, int TI, int TJ, int WC, bool CONV>
__global__ void __launch_bounds__(kThreads, (stream_min_waves(TI, TJ, WC)))
stream_gemm_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias, const float *R,
                   float *Y, int M, int K, int N, int mblocks, int nblocks, int relu, int kslices, const StreamConv cv)
{
    constexpr int WR = 4 / WC, BM = WR * TI * 32, BN = WC * TJ * 32, NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr int XV = BM / 32;   // float4 of A per thread and slice: BM * 8 / 256
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][NA][BM * kStride];   // [buffer][activation piece][row][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - wr * WC;

    // ---- block -> (row block, column block): ids that are congruent mod 8 land on one XCD; the column blocks of a
    // row block sit 8 ids apart inside a group of 8 * nblocks ids
    const int per = 8 * nblocks;
    const int g = blockIdx.x / per, r = blockIdx.x - g * per;
    const int mb = g * 8 + (r & 7), nb = r >> 3;
    if (mb >= mblocks) return;   // whole block, before any barrier
    const int m0 = mb * BM, n0 = nb * BN;
    const int KQ = K >> 4;
    const int sbeg = kslices > 0 ? (int)blockIdx.y * kslices : 0;
    const int send = kslices > 0 ? min(K / kSlice, sbeg + kslices) : K / kSlice;   // this block's slices: [sbeg, send), an even number
    if (kslices > 0) Y += (size_t)blockIdx.y * M * N;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- activations: thread -> (row, 4 consecutive k) of the slice, XV rows 32 apart
    const int arow = tid >> 3, ac4 = tid & 7;
    const float *xp[CONV ? 1 : XV];
    unsigned xoff[CONV ? XV : 1];
    int ybase[CONV ? XV : 1], xbase[CONV ? XV : 1];
    bool rowok[CONV ? XV : 1];
    constexpr unsigned OOB = 0xC0000000u;   // >= num_records (host: the input lies below 3 GiB)
    __amdgpu_buffer_rsrc_t xrs;
    int nslice = sbeg, ntap_c0 = 0, ndx = 0, ndy = 0;   // the NEXT slice load_x fetches, as (slice, channel, tap column, tap row)
    if constexpr (!CONV) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int grow = min(m0 + it * 32 + arow, M - 1);   // rows past M read the last row, never stored
            xp[it] = X + (size_t)grow * K + ac4 * 4;
        }
    } else {
        xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (unsigned)((size_t)cv.nimg * cv.hin * cv.win * cv.cin * 4), 0x00020000);
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int row = m0 + it * 32 + arow;
            rowok[it] = row < M;
            const int rr = rowok[it] ? row : 0;
            const int img = rr / (cv.hout * cv.wout), rem = rr - img * (cv.hout * cv.wout);
            const int yo = rem / cv.wout, xo = rem - yo * cv.wout;
            ybase[it] = yo * cv.stride - cv.pad;
            xbase[it] = xo * cv.stride - cv.pad;
            // byte offset of the pixel's window origin, channel ac4 * 4 (wrap-around arithmetic: a border pixel's window starts
            // in front of the image, the sum with a valid tap's offset is back inside)
            xoff[it] = ((unsigned)((img * cv.hin + ybase[it]) * cv.win + xbase[it]) * (unsigned)cv.cin + (unsigned)(ac4 * 4)) * 4u;
        }
        const int k0 = sbeg * kSlice, tap = k0 / cv.cin;
        ntap_c0 = k0 - tap * cv.cin;
        ndy = tap / cv.ks;
        ndx = tap - ndy * cv.ks;
    }
    // the slices are fetched strictly in order (sbeg, sbeg + 1, ...); calls past the last slice re-fetch it (never used)
    auto load_x = [&](f32x4 (&dst)[XV]) {
        if constexpr (!CONV) {
            const int k0 = min(nslice, send - 1) * kSlice;
#pragma unroll
            for (int it = 0; it < XV; ++it) dst[it] = *reinterpret_cast<const f32x4 *>(xp[it] + k0);
            ++nslice;
        } else {
            const unsigned tapoff = (unsigned)((ndy * cv.win + ndx) * cv.cin + ntap_c0) * 4u;   // uniform
#pragma unroll
            for (int it = 0; it < XV; ++it) {
                const bool ok = rowok[it] && (unsigned)(ybase[it] + ndy) < (unsigned)cv.hin && (unsigned)(xbase[it] + ndx) < (unsigned)cv.win;
                dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xoff[it] + tapoff : OOB, 0, 0));
            }
            if (nslice + 1 < send) {   // step (channel, tap column, tap row) to the next slice
                ++nslice;
                ntap_c0 += kSlice;
                if (ntap_c0 == cv.cin) {
                    ntap_c0 = 0;
                    if (++ndx == cv.ks) {
                        ndx = 0;
                        ++ndy;
                    }
                }
            }
        }
    };
    auto store_x = [&](const f32x4 (&src)[XV], int buf) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            u32x2 pc[NA];
#if TF_STREAM_ABLATE & 1   // timing ablation (tools/build_variant.py --source linear_stream.hip): no fp32 -> pieces conversion, the bits as they are
#pragma unroll
            for (int p = 0; p < NA; ++p) pc[p] = u32x2{__builtin_bit_cast(unsigned, src[it].x) + (unsigned)p, __builtin_bit_cast(unsigned, src[it].z)};
#else
            split4<SP>(src[it], pc);
#endif
            const int o = (it * 32 + arow) * kStride + ac4 * 4;
#pragma unroll
            for (int p = 0; p < NA; ++p) *reinterpret_cast<u32x2 *>(&sA[buf][p][o]) = pc[p];
        }
    };
    // ---- weights: the wave's TJ column tiles, fragment order (see pack_weight_kernel)
    const u32x4 *wp[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) wp[j] = Wp + ((size_t)(nb * (WC * TJ) + wc * TJ + j) * KQ * NB) * 64 + lane;
    auto load_w = [&](int s, WFrags<NB, TJ> &w) {
        const int q0 = min(s, send - 1) * 2;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int p = 0; p < NB; ++p) w.v[j][kk][p] = wp[j][((q0 + kk) * NB + p) * 64];
    };

#ifdef TF_STREAM_TRACE
    TraceRegs trace_regs = {};
#endif
    // The activations of the slices AHEAD live in a ring of DX register stages: slice s + 1 in xr[(s + 1 - sbeg) % DX], ..., slice
    // s + DX in xr[(s - sbeg) % DX].  Two stages.  The phase trace of round 5 (profiles/r05_stream_phase_trace.txt) shows the
    // convolution form spending 806 of the 1660 cycles of a slice (128 -> 128 3 x 3) in its staging phase; FOUR stages
    // (-DTF_STREAM_XDEPTH=4) were measured against that and do not help -- 1360 against 1322 us per frame over ResNet-50's
    // convolutions, 37.9 against 31.6 us for layer1's 3 x 3 (fewer resident blocks), profiles/r05_stream_xdepth.txt: the phase is
    // not waiting for those loads.
    constexpr int DX = CONV ? TF_STREAM_XDEPTH : 2;
    static_assert(DX == 2 || DX == 4, "ring of 2 or 4 register stages");
    f32x4 xr[DX][XV];
    WFrags<NB, TJ> w0, w1;
    {
        f32x4 first[XV];
        load_x(first);
        load_w(sbeg, w0);
#pragma unroll
        for (int d = 1; d <= DX; ++d) load_x(xr[d % DX]);
        store_x(first, 0);
    }
    __syncthreads();

    // one K-slice; PAR = (s - sbeg) & 1 (the LDS buffer it reads) and XS = (s + 1 - sbeg) % DX (the register stage that holds
    // slice s + 1) as compile-time constants so that the register buffers need no copies
    auto slice = [&](int s, auto par, auto xs, const WFrags<NB, TJ> &cur, WFrags<NB, TJ> &nxt) {
        constexpr int PAR = decltype(par)::value, XS = decltype(xs)::value;
        TF_TRACE(s - sbeg, 0);
        load_w(s + 1, nxt);   // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler otherwise sinks them to the end of the
                                             // slice to shorten live ranges, and the next slice starts by waiting for L2
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = kk * 16 + (lane >> 5) * 8;
            u32x4 af[TI][NA], bfr[TJ][NB];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int o = ((wr * TI + i) * 32 + (lane & 31)) * kStride + koff;
#pragma unroll
                for (int p = 0; p < NA; ++p) af[i][p] = *reinterpret_cast<const u32x4 *>(&sA[PAR][p][o]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int p = 0; p < NB; ++p) bfr[j][p] = cur.v[j][kk][p];
            // term-major passes over the tiles: consecutive MFMAs never share an accumulator; per accumulator the order is
            // smallest terms first, as in linear_split.hip
#if TF_STREAM_ABLATE & 2   // timing ablation: no matrix instructions (the operands stay alive through one add)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j][kk] += __builtin_bit_cast(float, af[i][0].x ^ bfr[j][0].x);
#else
            mfma_tiles<SP, TI, TJ>(acc, af, bfr);
#endif
        }
#ifdef TF_STREAM_TRACE
        __builtin_amdgcn_sched_barrier(0);
#endif
        TF_TRACE(s - sbeg, 1);
        // slice s + 1 -> the LDS buffer nobody reads in this iteration (its readers passed the previous barrier),
        // then its registers take slice s + 1 + DX
        store_x(xr[XS], PAR ^ 1);
        load_x(xr[XS]);
#ifdef TF_STREAM_TRACE
        __builtin_amdgcn_sched_barrier(0);
#endif
        TF_TRACE(s - sbeg, 2);
        __syncthreads();
    };
    using std::integral_constant;
    if constexpr (DX == 2) {
        for (int s = sbeg; s < send; s += 2) {   // an even number of slices (host)
            slice(s, integral_constant<int, 0>{}, integral_constant<int, 1>{}, w0, w1);
            slice(s + 1, integral_constant<int, 1>{}, integral_constant<int, 0>{}, w1, w0);
        }
    } else {
        for (int s = sbeg; s < send; s += 4) {   // an even number of slices (host): the second pair may be absent (block-uniform)
            slice(s, integral_constant<int, 0>{}, integral_constant<int, 1>{}, w0, w1);
            slice(s + 1, integral_constant<int, 1>{}, integral_constant<int, 2>{}, w1, w0);
            if (s + 2 < send) {
                slice(s + 2, integral_constant<int, 0>{}, integral_constant<int, 3>{}, w0, w1);
                slice(s + 3, integral_constant<int, 1>{}, integral_constant<int, 0>{}, w1, w0);
            }
        }
    }

    // ---- epilogue: C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Buffer
    // resources over Y (and R): rows >= M land beyond num_records (dropped / read as zero by the hardware); columns >= N start
    // from 3 GiB, which stays out of range and does not wrap for any row delta (host: the tensor is < 3 GiB)
    const unsigned ybytes = (unsigned)((size_t)M * N * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : Y), 0, R ? ybytes : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = n0 + (wc * TJ + j) * 32 + (lane & 31);
        const bool colok = col < N;
        const float b = (bias && colok) ? bias[col] : 0.f;
        float rsc = 1.f;   // fp16 scheme: the output channel's power of two (behind the fragments of the packed weight)
        if constexpr (Split<SP>::F16) rsc = reinterpret_cast<const float *>(Wp + (size_t)((N + kBN - 1) / kBN * (kBN / 32)) * KQ * NB * 64)[colok ? col : 0];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int row0 = m0 + (wr * TI + i) * 32 + 4 * (lane >> 5);
            const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : 0xC0000000u;
            float rv[16];
            if (R != nullptr) {   // uniform
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, base + (unsigned)(((e & 3) + 8 * (e >> 2)) * N) * 4u, 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = Split<SP>::F16 ? __builtin_fmaf(acc[i][j][e], rsc, b) : acc[i][j][e] + b;
                if (R != nullptr) v += rv[e];
                if (relu) v = v < 0.f ? 0.f : v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, base + (unsigned)(((e & 3) + 8 * (e >> 2)) * N) * 4u, 0, tfm::kStoreAux);
            }
        }
    }
#ifdef TF_STREAM_TRACE
    if (g_stream_trace && threadIdx.x == 0 && blockIdx.x < kTraceBlocks && blockIdx.y == 0)
#pragma unroll
        for (int sl = 0; sl < kTraceSlices; ++sl)
#pragma unroll
            for (int pt = 0; pt < kTracePoints; ++pt)
                g_stream_trace[((size_t)blockIdx.x * kTraceSlices + sl) * kTracePoints + pt] = trace_regs.t[sl][pt];
#endif
}

// ---- split-K second pass: y = act(sum_z partial[z] + bias + residual), the partials added in the order z = 0, 1, ... (a fixed
// order: the result does not depend on scheduling, unlike atomic accumulation).  One thread per 4 consecutive outputs.
__global__ void __launch_bounds__(256)
stream_splitk_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias, const float *R, float *y, long long mn4,
                            int n4, int splits, int relu)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;   // mn4 < 2^31 (the output is below 3 GiB): 32-bit index arithmetic
    if (i >= (unsigned)mn4) return;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(part) + i;
    f32x4 acc = p[0];
    int z = 1;
    for (; z + 3 < splits; z += 4) {   // four loads in flight, added in the order z, z + 1, ...
        const f32x4 a = p[(long long)z * mn4], b = p[(long long)(z + 1) * mn4], c = p[(long long)(z + 2) * mn4], d = p[(long long)(z + 3) * mn4];
        acc += a;
        acc += b;
        acc += c;
        acc += d;
    }
    for (; z < splits; ++z) acc += p[(long long)z * mn4];
    if (bias != nullptr) acc += reinterpret_cast<const f32x4 *>(bias)[i % (unsigned)n4];
    if (R != nullptr) acc += reinterpret_cast<const f32x4 *>(R)[i];
    if (relu) {
        acc.x = acc.x < 0.f ? 0.f : acc.x;
        acc.y = acc.y < 0.f ? 0.f : acc.y;
        acc.z = acc.z < 0.f ? 0.f : acc.z;
        acc.w = acc.w < 0.f ? 0.f : acc.w;
    }
    tfm::stream_store(reinterpret_cast<f32x4 *>(y) + i, acc);
}

// ---------------------------------------------------------------------------------------------------------------------------
// THE LDS-DMA GEMM (round 6).  Every dense kernel of the frame moves its operands from L2 / HBM at 5-6 TB/s whatever its shape
// (256 -> 256 linear: 22.7 MB of rows + 128 KB of weights x 696 blocks in 18.6 us; 256 -> 384: 179 MB in 31 us; the 3 x 3 layers:
// ~160 MB in 32 us; the fused FFN: 464 MB of weight re-reads in 74 us): not a bandwidth limit -- the memory-level parallelism of the
// structure.  A wave of the stream form holds its fetches in REGISTERS (one slice of rows two slices ahead, one set of weight
// fragments one slice ahead: 8-16 KB in flight per wave, ~50 KB per CU), and 50 KB per CU over ~2 us of loaded latency is 6 TB/s.
// Here nothing is fetched into registers: both operands go global -> LDS by LDS-DMA (buffer_load ... lds) into a ring of STAGES
// slices per block, (STAGES - 1) slices in flight behind counted s_waitcnt vmcnt -- the bytes in flight are bounded by LDS, not by
// the register file.
//   block   256 threads = 4 waves splitting the ROWS: wave w owns rows 32 TI w .. + 32 TI - 1 and ALL BN = 32 TJ columns
//   A       raw fp32 rows of the slice (BM x 128 B), each wave fetches ITS rows (4 TI DMA instructions of 8 rows x 128 B) and is the
//           only one to read them: no barrier for A, no split pass through LDS -- the wave reads its MFMA A fragments as fp32 (two
//           ds_read_b128 per row tile and k-step) and cuts them into pieces in registers right in front of the MFMAs (every input
//           value is split exactly once).  LDS image: row r, 16-byte piece c at r * 8 + (c ^ ((r >> 1) & 7)) -- the XOR is applied to
//           the SOURCE offset of the DMA lane (the LDS side of an LDS-DMA is lane-linear); the 16 rows of an LDS cycle of
//           ds_read_b128 then hit 16 different slots
//   B       the packed weight fragments of the block's TJ column tiles (TJ x 2 k-steps x NB pieces x 1 KB per slice, in fragment order:
//           the B fragment read is base + 16 lane, conflict-free), fetched once per BLOCK (its four waves share them; each issues a
//           quarter of the DMA instructions) -- a quarter of the stream form's weight traffic per row
//   loop    wait for this wave's part of slice s (vmcnt counted so that the younger slices stay in flight) -> s_barrier (slice s is
//           complete for everybody, and everybody is done reading slice s - 1) -> refill the slot of slice s - 1 with slice
//           s + STAGES - 1 -> MFMAs of slice s.  One barrier per slice, raw (a __syncthreads() would drain the DMA queue).
// Accumulation order per output element: that of linear_split.hip / the stream form (k ascending, per k-step smallest terms first):
// bit-identical results.
template <int SP, int TI, int TJ, int STAGES>
__global__ void __launch_bounds__(kThreads, (TI * TJ <= 4 ? 2 : 1))
dma_gemm_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias, const float *R, float *Y,
                int M, int K, int N, int mblocks, int nblocks, int relu)
{
    constexpr int BM = 4 * TI * 32, BN = TJ * 32, NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr int A_BYTES = BM * 128, B_BYTES = TJ * 2 * NB * 1024, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = 4 * TI;                 // DMA instructions per wave and slice for its rows (8 rows each)
    constexpr int B_INSTR = TJ * 2 * NB / 4;        // ... and its quarter of the weight fragments
    static_assert((TJ * 2 * NB) % 4 == 0, "the weight fragments of a slice are dealt to four waves");
    constexpr int PER = A_INSTR + B_INSTR;          // DMA instructions per wave and slice
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * PER < 64, "ring of 2..4 slices; s_waitcnt vmcnt holds 6 bits");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // ONE shared array (a second one makes hipcc drain the DMAs)
    unsigned char *const lds0 = smem;   // (lambdas take the pointer: naming the __shared__ array itself inside one fails the host pass)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int per = 8 * nblocks;
    const int g = blockIdx.x / per, r = blockIdx.x - g * per;
    const int mb = g * 8 + (r & 7), nb = r >> 3;
    if (mb >= mblocks) return;   // whole block, before any barrier
    const int m0 = mb * BM, n0 = nb * BN;
    const int KQ = K >> 4, nsl = K / kSlice;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr unsigned OOB = 0xC0000000u;   // >= num_records (host: the tensors lie below 3 GiB)
    // ---- A: DMA instruction a of this wave covers its rows 8 a .. 8 a + 7 (tile-local row rl = 32 TI wave + 8 a + (lane >> 3)); the lane
    // fetches the piece that belongs at LDS position lane & 7 of that row: piece (lane & 7) ^ ((rl >> 1) & 7)
    unsigned a_src[A_INSTR];
#pragma unroll
    for (int a = 0; a < A_INSTR; ++a) {
        const int rl = 32 * TI * wave + 8 * a + (lane >> 3);
        const int row = m0 + rl;
        const int c = (lane & 7) ^ ((rl >> 1) & 7);
        a_src[a] = row < M ? ((unsigned)row * (unsigned)K + (unsigned)(c * 4)) * 4u : OOB;
    }
    // ---- B: fragment f = 4 b + wave of the slice's TJ * 2 * NB (b < B_INSTR), f = (j * 2 + kk) * NB + p
    unsigned b_src[B_INSTR];
#pragma unroll
    for (int b = 0; b < B_INSTR; ++b) {
        const int f = 4 * b + wave, p = f % NB, jk = f / NB, kk = jk & 1, j = jk >> 1;
        b_src[b] = (unsigned)((((size_t)(nb * TJ + j) * KQ + kk) * NB + p) * 1024) + (unsigned)lane * 16u;   // + slice * 2 * NB * 1024
    }
    // slice s of both operands -> ring slot `slot`: this wave's rows of A, its quarter of the weight fragments.  (The source offset goes
    // through a LOCAL: with an array element written directly into the builtin's argument -- a_src[a] + koff -- hipcc's host pass (ROCm 7.2)
    // drops the whole kernel template without a diagnostic: undefined stubs when the library is linked.  Found by bisection.)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4 *>(Wp), 0, (unsigned)((size_t)((N + kBN - 1) / kBN * (kBN / 32)) * KQ * NB * 1024), 0x00020000);
#define TF_DMA_ISSUE(S_, SLOT_)                                                                                                          \
    do {                                                                                                                                 \
        unsigned char *base_ = lds0 + (SLOT_) * STAGE;                                                                                    \
        const unsigned koff_ = (unsigned)(S_) * (kSlice * 4u), woff_ = (unsigned)(S_) * (2u * NB * 1024u);                                \
        _Pragma("unroll") for (int a = 0; a < A_INSTR; ++a) {                                                                            \
            const unsigned v_ = a_src[a] == OOB ? OOB : a_src[a] + koff_;   /* (a local: see above) */                                    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(base_ + (32 * TI * wave + 8 * a) * 128), \
                                                     16, v_, 0, 0, 0);                                                                    \
        }                                                                                                                                \
        _Pragma("unroll") for (int b = 0; b < B_INSTR; ++b) {                                                                            \
            const unsigned v_ = b_src[b] + woff_;                                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(base_ + A_BYTES + (4 * b + wave) * 1024), \
                                                     16, v_, 0, 0, 0);                                                                    \
        }                                                                                                                                \
    } while (0)
    // ---- A fragment addresses: lane -> (row lane & 31 of its tile, k group lane >> 5): fp32 k = 16 kk + 8 (lane >> 5) .. + 7 = the pieces
    // c0 = 4 kk + 2 (lane >> 5) and c0 + 1, at the positions c ^ ((row >> 1) & 7) of the row
    int a_off[TI][2];   // byte offsets (inside a stage) of the two pieces of k-step 0; k-step 1: ^ 64
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int rl = 32 * TI * wave + 32 * i + (lane & 31);
        const int sw = (rl >> 1) & 7, c0 = 2 * (lane >> 5);
        a_off[i][0] = rl * 128 + ((c0 ^ sw) << 4);
        a_off[i][1] = rl * 128 + (((c0 + 1) ^ sw) << 4);
    }

    // ---- prologue: STAGES - 1 slices in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsl) TF_DMA_ISSUE(s, s);

    for (int s = 0; s < nsl; ++s) {
        const int slot = s % STAGES;
        // this wave's part of slice s has landed: the slices issued after it (at most STAGES - 2, fewer at the end) may stay in flight
        const int younger = min(STAGES - 2, nsl - 1 - s);
        if (younger >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER) & 0xF) | (((2 * PER) >> 4) << 14));
        else if (younger == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER & 0xF) | ((PER >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();   // slice s is complete for every wave; every wave is done with slice s - 1
        if (s + STAGES - 1 < nsl) TF_DMA_ISSUE(s + STAGES - 1, (s + STAGES - 1) % STAGES);
        const unsigned char *st = smem + slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 af[TI][NA], bfr[TJ][NB];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(st + (a_off[i][0] ^ (kk * 64)));
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(st + (a_off[i][1] ^ (kk * 64)));
                u32x2 pl[NA], ph[NA];
                split4<SP>(lo, pl);
                split4<SP>(hi, ph);
#pragma unroll
                for (int p = 0; p < NA; ++p) af[i][p] = u32x4{pl[p].x, pl[p].y, ph[p].x, ph[p].y};
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int p = 0; p < NB; ++p)
                    bfr[j][p] = *reinterpret_cast<const u32x4 *>(st + A_BYTES + ((j * 2 + kk) * NB + p) * 1024 + lane * 16);
            mfma_tiles<SP, TI, TJ>(acc, af, bfr);
        }
    }

    // ---- epilogue (as the stream form): C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const unsigned ybytes = (unsigned)((size_t)M * N * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : Y), 0, R ? ybytes : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        const bool colok = col < N;
        const float b = (bias && colok) ? bias[col] : 0.f;
        float rsc = 1.f;
        if constexpr (Split<SP>::F16) rsc = reinterpret_cast<const float *>(Wp + (size_t)((N + kBN - 1) / kBN * (kBN / 32)) * KQ * NB * 64)[colok ? col : 0];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int row0 = m0 + (wave * TI + i) * 32 + 4 * (lane >> 5);
            const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : OOB;
            float rv[16];
            if (R != nullptr) {   // uniform
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, base + (unsigned)(((e & 3) + 8 * (e >> 2)) * N) * 4u, 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = Split<SP>::F16 ? __builtin_fmaf(acc[i][j][e], rsc, b) : acc[i][j][e] + b;
                if (R != nullptr) v += rv[e];
                if (relu) v = v < 0.f ? 0.f : v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, base + (unsigned)(((e & 3) + 8 * (e >> 2)) * N) * 4u, 0, tfm::kStoreAux);
            }
        }
    }
}

#undef TF_DMA_ISSUE

// ---------------------------------------------------------------------------------------------------------------------------
// THE HALO FORM OF THE 3 x 3 CONVOLUTION (round 6; stride 1, padding 1: sixteen of ResNet-50's convolutions).
// The stream form above treats a 3 x 3 convolution as a GEMM over K = 9 Cin in tap-major order: every (tap, 32-channel slice) fetches
// the block's shifted input pixels again, splits them into pieces again and writes them to LDS again -- nine times the work per input
// value, and per slice only TI x TJ x 2 x terms MFMAs stand against it: the phase trace of round 5 has a slice of the 128 -> 128
// layer spend 806 of its 1660 cycles staging and 704 in a "matrix phase" that holds 384 cycles of MFMAs
// (profiles/r05_stream_phase_trace.txt); the ablation without MFMAs runs 80 % as long as the kernel (r05_stream_gemm_ablations.txt).
// Here a block owns a PATCH of PH x 8 output pixels and walks the input channels in slices of 32: per slice it stages the patch's HALO
// ((PH + 2) x 10 input pixels, 1.4-1.6 pixels per output pixel instead of 9) once -- global -> registers -> pieces -> LDS, double
// buffered, one barrier per slice -- and runs all NINE taps from it: the A fragment of tap (dy, dx) is the same LDS tile read at a
// constant offset (an immediate of ds_read_b128), the B fragments are the packed weight's k-steps (tap Cin + c) / 16, prefetched one tap
// ahead.  18 x TI x TJ x terms MFMAs per barrier instead of 2 x TI x TJ x terms.
//   block   256 threads = 4 waves as WR x WC; BM = WR TI 32 output pixels = a patch of PH = BM / 8 rows x 8 columns, BN = WC TJ 32 channels
//   rows    tile row r of the block <-> patch pixel (r / 8, r % 8); a wave's tile i covers patch rows 4 (wr TI + i) .. + 3
//   split-K blockIdx.y walks `cslices` channel slices (all nine taps of each) and writes partial sums to Y + z M N (second pass: the
//           stream form's reduce kernel)
// Summation order per output element: channel slice, tap, k-step, smallest term first (the stream form: tap, channel) -- the same
// products, another fp32 order.
// MRG (round 6, the mask head's FPN levels -- reference: models/detr_segmentation.py:142-156 `x = adapter(fpn) + interpolate(x)`, then
// lay(x)): the convolution's input is never materialised.  X is the PREVIOUS layer's output at its own (lower) resolution [nimg, lh, lw,
// Cin]; the value of input pixel (iy, ix), channel c is  act(X[img, ys, xs, c]) + fpn[img / qpi, iy, ix, c]  with (ys, xs) torch's
// legacy nearest index (min(floorf(iy (float)lh / H), lh - 1)) and act = identity or, with `ws`, relu(GroupNorm(X)) from the raw
// statistics of tf_groupnorm's workspace (sum | sum of squares per (image, group)) in the expression of groupnorm_apply_kernel.
struct HaloMerge {
    const float *fpn;
    const double *ws;       // NULL: X is taken as it is
    const float *gamma, *beta;
    int qpi, lh, lw, G;
    float eps;
};

template <int SP, int TI, int TJ, int WC, bool MRG = false>
__global__ void __launch_bounds__(kThreads, (TI * TJ <= 2 ? 2 : 1))
conv3x3_halo_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias, const float *R, float *Y,
                    int N, int nblocks, int npatches, int tiles_x, int tiles_y, int relu, int cslices, const StreamConv cv,
                    const HaloMerge mg = HaloMerge{})
{
    constexpr int WR = 4 / WC, BM = WR * TI * 32, PW = 8, PH = BM / PW, HW = PW + 2, HH = PH + 2, HP = HH * HW;
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr int XV = (HP * 8 + kThreads - 1) / kThreads;   // float4 of the halo per thread and slice
    __shared__ __attribute__((aligned(16))) unsigned short sH[2][NA][HP * kStride];   // [buffer][activation piece][halo pixel][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - wr * WC;

    // block -> (patch, column block): the column blocks of a patch sit 8 ids apart (ids congruent mod 8 share an XCD and its L2)
    const int per = 8 * nblocks;
    const int g = blockIdx.x / per, r = blockIdx.x - g * per;
    const int pid = g * 8 + (r & 7), nb = r >> 3;
    if (pid >= npatches) return;   // whole block, before any barrier
    const int tx = pid % tiles_x, t2 = pid / tiles_x, ty = t2 % tiles_y, img = t2 / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW;
    const int H = cv.hin, W = cv.win, Cin = cv.cin;   // stride 1, padding 1: the output has the input's size
    const int nsl = Cin / kSlice;
    const int cbeg = cslices > 0 ? (int)blockIdx.y * cslices : 0;
    const int cend = cslices > 0 ? min(nsl, cbeg + cslices) : nsl;
    const long long Mtot = (long long)cv.nimg * H * W;
    if (cslices > 0) Y += (size_t)blockIdx.y * Mtot * N;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- the halo: thread -> float4 f = tid + 256 it of (halo pixel f >> 3, channels 4 (f & 7) ..); pixels outside the image (and
    // the slots behind the last halo pixel) read zeros from beyond num_records
    constexpr unsigned OOB = 0xC0000000u;
    constexpr int kMrgC = MRG ? 320 : 1;   // MRG: Cin <= 320 (host)
    __shared__ float s_gn[MRG ? 4 : 1][kMrgC];   // MRG with statistics: mean | rstd | gamma | beta per input channel of this block's image
    const size_t xbytes = MRG ? (size_t)cv.nimg * mg.lh * mg.lw * Cin * 4 : (size_t)cv.nimg * H * W * Cin * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (unsigned)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(MRG ? mg.fpn : X), 0, MRG ? (unsigned)((size_t)(cv.nimg / mg.qpi) * H * W * Cin * 4) : 0u, 0x00020000);
    unsigned xoff[XV], foff[MRG ? XV : 1];
    int lds_o[XV];
    if constexpr (MRG) {
        if (mg.ws != nullptr) {   // uniform
            const int cpg = Cin / mg.G;
            for (int c = tid; c < Cin; c += kThreads) {
                const int g = c / cpg;
                const double cnt = (double)mg.lh * (double)mg.lw * (double)cpg;
                const double mean = mg.ws[(long long)img * 2 * mg.G + 2 * g] / cnt;
                double var = mg.ws[(long long)img * 2 * mg.G + 2 * g + 1] / cnt - mean * mean;   // biased, as torch.nn.GroupNorm
                if (var < 0.0) var = 0.0;
                s_gn[0][c] = (float)mean;
                s_gn[1][c] = (float)(1.0 / sqrt(var + (double)mg.eps));
                s_gn[2][c] = mg.gamma[c];
                s_gn[3][c] = mg.beta[c];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < XV; ++it) {
        const int f = tid + kThreads * it;
        const int hp = f >> 3, c4 = f & 7;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        if constexpr (MRG) {
            const int ys = min((int)floorf((float)iy * ((float)mg.lh / (float)H)), mg.lh - 1);
            const int xs = min((int)floorf((float)ix * ((float)mg.lw / (float)W)), mg.lw - 1);
            xoff[it] = ok ? ((unsigned)((img * mg.lh + ys) * mg.lw + xs) * (unsigned)Cin + (unsigned)(c4 * 4)) * 4u : OOB;
            foff[it] = ok ? ((unsigned)(((img / mg.qpi) * H + iy) * W + ix) * (unsigned)Cin + (unsigned)(c4 * 4)) * 4u : OOB;
        } else {
            xoff[it] = ok ? ((unsigned)((img * H + iy) * W + ix) * (unsigned)Cin + (unsigned)(c4 * 4)) * 4u : OOB;
        }
        lds_o[it] = hp < HP ? hp * kStride + c4 * 4 : -1;
    }
    // (MRG: a slice's registers hold the low-resolution value and the fpn value: XV more f32x4; they meet in store_x)
    struct XRegs {
        f32x4 v[XV];
        f32x4 f[MRG ? XV : 1];
    };
    auto load_x = [&](int cs, XRegs &dst) {
        const unsigned coff = (unsigned)(min(cs, cend - 1) * kSlice) * 4u;   // (calls past the last slice re-fetch it: never used)
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            dst.v[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff[it] == OOB ? OOB : xoff[it] + coff, 0, 0));
            if constexpr (MRG)
                dst.f[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, foff[it] == OOB ? OOB : foff[it] + coff, 0, 0));
        }
    };
    auto store_x = [&](const XRegs &src, int buf, int cs) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            if (XV * kThreads > HP * 8 && lds_o[it] < 0) continue;
            f32x4 v = src.v[it];
            if constexpr (MRG) {
                if (mg.ws != nullptr && xoff[it] != OOB) {   // (outside the image the MERGED tensor is zero: the convolution's padding)
                    const int c = min(cs, cend - 1) * kSlice + ((tid + kThreads * it) & 7) * 4;
                    const f32x4 mean = *reinterpret_cast<const f32x4 *>(&s_gn[0][c]), rstd = *reinterpret_cast<const f32x4 *>(&s_gn[1][c]);
                    const f32x4 ga = *reinterpret_cast<const f32x4 *>(&s_gn[2][c]), be = *reinterpret_cast<const f32x4 *>(&s_gn[3][c]);
                    v.x = (v.x - mean.x) * rstd.x * ga.x + be.x;
                    v.y = (v.y - mean.y) * rstd.y * ga.y + be.y;
                    v.z = (v.z - mean.z) * rstd.z * ga.z + be.z;
                    v.w = (v.w - mean.w) * rstd.w * ga.w + be.w;
                    v.x = v.x < 0.f ? 0.f : v.x;
                    v.y = v.y < 0.f ? 0.f : v.y;
                    v.z = v.z < 0.f ? 0.f : v.z;
                    v.w = v.w < 0.f ? 0.f : v.w;
                }
                v += src.f[it];
            }
            u32x2 pc[NA];
            split4<SP>(v, pc);
#pragma unroll
            for (int p = 0; p < NA; ++p) *reinterpret_cast<u32x2 *>(&sH[buf][p][lds_o[it]]) = pc[p];
        }
    };
    // ---- weights: the wave's TJ column tiles in fragment order; k-step of (tap, channel slice cs, kk): tap Cin / 16 + 2 cs + kk
    const int KQ = (9 * Cin) >> 4, tapq = Cin >> 4;
    const u32x4 *wp[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) wp[j] = Wp + ((size_t)(nb * (WC * TJ) + wc * TJ + j) * KQ * NB) * 64 + lane;
    auto load_w = [&](int cs, int tap, WFrags<NB, TJ> &w) {
#if TF_HALO_ABLATE & 1   // timing ablation (tools/build_variant.py --source linear_stream.hip): the same 4 KB of weights every time (L1-resident)
        const int q0 = 0 * (tap + cs);
#else
        const int q0 = tap * tapq + 2 * min(cs, cend - 1);
#endif
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int p = 0; p < NB; ++p) w.v[j][kk][p] = wp[j][((q0 + kk) * NB + p) * 64];
    };
    // ---- A fragments: lane -> (tile row lane & 31 = patch pixel (row >> 3, row & 7), k group lane >> 5)
    int abase[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int prow = (wr * TI + i) * 4 + ((lane & 31) >> 3), pcol = lane & 7;
        abase[i] = (prow * HW + pcol) * kStride + (lane >> 5) * 8;   // halo pixel of tap (0, 0); tap (dy, dx): + (dy HW + dx) kStride
    }

    // The tap loop is software-pipelined by hand (one wave per SIMD at these grid sizes: nothing else hides a latency).  A slice is 18
    // steps (tap, k-step); in program order step s issues the LDS reads of step s + 1 (two register sets of A fragments) and, at the
    // first k-step of a tap, the weight fragments of tap + 2 (a ring of three sets: 9 taps = 3 turns, the next slice's taps 0 / 1 land
    // where this slice's did), THEN its own MFMAs -- the compiler's s_waitcnt counts let the younger requests stay in flight.
    XRegs xr;
    WFrags<NB, TJ> wr3[3];
    u32x4 af[2][TI][NA];
    auto read_a = [&](int buf, int step, u32x4 (&dst)[TI][NA]) {   // step = 2 tap + kk (compile-time at every call)
        const int tap = step >> 1, kk = step & 1;
        const int toff = ((tap / 3) * HW + (tap % 3)) * kStride + kk * 16;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int p = 0; p < NA; ++p) dst[i][p] = *reinterpret_cast<const u32x4 *>(&sH[buf][p][abase[i] + toff]);
    };
    {
        XRegs first;
        load_x(cbeg, first);
        load_w(cbeg, 0, wr3[0]);
        load_w(cbeg, 1, wr3[1]);
        load_x(cbeg + 1, xr);
        if constexpr (MRG) __syncthreads();   // s_gn
        store_x(first, 0, cbeg);
    }
    __syncthreads();

    for (int cs = cbeg; cs < cend; ++cs) {
        const int buf = (cs - cbeg) & 1;
        read_a(buf, 0, af[0]);
        auto step = [&](auto sc) {
            constexpr int st = decltype(sc)::value, tap = st >> 1, kk = st & 1;
            if constexpr (st < 17) read_a(buf, st + 1, af[(st + 1) & 1]);
            if constexpr (kk == 0) {
                if constexpr (tap + 2 < 9) load_w(cs, tap + 2, wr3[(tap + 2) % 3]);
                else load_w(cs + 1, tap + 2 - 9, wr3[(tap + 2) % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of this step's MFMAs
            u32x4 bfr[TJ][NB];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int p = 0; p < NB; ++p) bfr[j][p] = wr3[tap % 3].v[j][kk][p];
#if TF_HALO_ABLATE & 4   // timing ablation: no matrix instructions (the operands stay alive through one add)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j][kk] += __builtin_bit_cast(float, af[st & 1][i][0].x ^ bfr[j][0].x);
#else
            mfma_tiles<SP, TI, TJ>(acc, af[st & 1], bfr);
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        step(integral_constant<int, 0>{});
        step(integral_constant<int, 1>{});
        step(integral_constant<int, 2>{});
        step(integral_constant<int, 3>{});
        step(integral_constant<int, 4>{});
        step(integral_constant<int, 5>{});
        step(integral_constant<int, 6>{});
        step(integral_constant<int, 7>{});
        step(integral_constant<int, 8>{});
        step(integral_constant<int, 9>{});
        step(integral_constant<int, 10>{});
        step(integral_constant<int, 11>{});
        step(integral_constant<int, 12>{});
        step(integral_constant<int, 13>{});
        step(integral_constant<int, 14>{});
        step(integral_constant<int, 15>{});
        step(integral_constant<int, 16>{});
        step(integral_constant<int, 17>{});
        // the next slice -> the other LDS buffer (its readers passed the previous barrier), its registers take the slice after it
#if !(TF_HALO_ABLATE & 2)   // timing ablation 2: the halo is staged once (every slice reads the first one)
        if (cs + 1 < cend) {
            store_x(xr, buf ^ 1, cs + 1);
            load_x(cs + 2, xr);
        }
#endif
        __syncthreads();
    }

    // ---- epilogue: C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); row -> patch pixel ->
    // image pixel; pixels of the patch outside the image and columns >= N go beyond num_records (dropped by the hardware)
    const unsigned ybytes = (unsigned)((size_t)Mtot * N * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : Y), 0, R ? ybytes : 0u, 0x00020000);
    const int n0 = nb * (WC * TJ * 32);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = n0 + (wc * TJ + j) * 32 + (lane & 31);
        const bool colok = col < N;
        const float b = (bias && colok) ? bias[col] : 0.f;
        float rsc = 1.f;   // fp16 scheme: the output channel's power of two (behind the fragments of the packed weight)
        if constexpr (Split<SP>::F16) rsc = reinterpret_cast<const float *>(Wp + (size_t)((N + kBN - 1) / kBN * (kBN / 32)) * KQ * NB * 64)[colok ? col : 0];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            unsigned off[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wr * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int oy = y0 + (row >> 3), ox = x0 + (row & 7);
                off[e] = (colok && oy < H && ox < W) ? (unsigned)(((img * H + oy) * W + ox) * N + col) * 4u : OOB;
            }
            float rv[16];
            if (R != nullptr) {   // uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, off[e], 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = Split<SP>::F16 ? __builtin_fmaf(acc[i][j][e], rsc, b) : acc[i][j][e] + b;
                if (R != nullptr) v += rv[e];
                if (relu) v = v < 0.f ? 0.f : v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, off[e], 0, tfm::kStoreAux);
            }
        }
    }
}

int num_cus()
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

std::atomic<int> g_ti{-1};   // -1: TF_LINEAR_STREAM_TI or automatic (0)

int forced_ti()
{
    int v = g_ti.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_LINEAR_STREAM_TI");
        v = e ? atoi(e) : 0;
        if (v < 1 || v > 4) v = 0;
        g_ti.store(v);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------- host side
struct StreamCall {
    const float *x;
    const u32x4 *wp;
    const float *bias, *res;
    float *y;
    int M, K, N, relu;
    bool conv;
    StreamConv cv;
    float *workspace;   // split-K partial sums [pieces][M][N], or NULL
    int ksplit;         // pieces the K loop is cut into (1: none)
};

template <int SP, int TI, int TJ, int WC, bool CONV>
int launch_stream(const StreamCall &c, hipStream_t s)
{
    constexpr int BM = (4 / WC) * TI * 32, BN = WC * TJ * 32;
    const int mblocks = (c.M + BM - 1) / BM, nblocks = (c.N + BN - 1) / BN;
    const long long gx = (long long)((mblocks + 7) / 8) * 8 * nblocks;
    const int slices = c.K / kSlice;
    int kslices = 0, gz = 1;
    if (c.ksplit > 1) {
        kslices = ((slices + c.ksplit - 1) / c.ksplit + 1) & ~1;   // an even number of slices per piece
        gz = (slices + kslices - 1) / kslices;
        if (gz <= 1) kslices = 0, gz = 1;
    }
    if (gx > 0x7fffffffLL || gz > 65535) return TF_MSDA_ERR_BAD_DIMS;
    const bool partial = gz > 1;
    float *out = partial ? c.workspace : c.y;
    hipLaunchKernelGGL((stream_gemm_kernel<SP, TI, TJ, WC, CONV>), dim3((unsigned)gx, (unsigned)gz), dim3(kThreads), 0, s, c.x, c.wp,
                       partial ? nullptr : c.bias, partial ? nullptr : c.res, out, c.M, c.K, c.N, mblocks, nblocks, partial ? 0 : c.relu,
                       kslices, c.cv);
    if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    if (partial) {
        const long long mn4 = (long long)c.M * c.N / 4;
        hipLaunchKernelGGL(stream_splitk_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, s, c.workspace, c.bias, c.res,
                           c.y, mn4, c.N / 4, gz, c.relu);
        if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    }
    return TF_MSDA_OK;
}

// Block shape per call.  Columns: N <= 64 -> 64-column blocks (2 x 2 waves), N <= 128 -> 128 (4 waves side by side, one column
// tile each), else 256.  Rows: 32 TI per row group.  Measured at 22 223 rows with three terms (profiles/r02_split_gemm_packed.txt,
// 256-column blocks): two row tiles win at K = 256 (59.3 / 67.3 / 84.2 us for TI = 2 / 3 / 4 at N = 1024: a block's eight K-slices
// are too short a loop to hide its own memory latency and the co-resident blocks have to), three at K = 1024 (55.6 / 50.5 / 58.0).
// The narrow shapes take 128-row blocks while that leaves at least ~2 blocks per CU, else 64.
template <int SP, bool CONV>
int stream_dispatch(const StreamCall &c, hipStream_t s)
{
    const int f = forced_ti();
    const long long pieces = c.ksplit > 1 ? c.ksplit : 1;
    if (c.N <= 64) {
        const bool big = f ? f >= 2 : (long long)((c.M + 127) / 128) * pieces >= 2LL * num_cus();
        return big ? launch_stream<SP, 2, 1, 2, CONV>(c, s) : launch_stream<SP, 1, 1, 2, CONV>(c, s);
    }
    if (c.N <= 128) {
        // (round 5, measured and removed: 128 x 128 blocks as 2 x 2 waves of 64 x 64 -- half the LDS re-reads of the activations per
        // MFMA, twice the weight fragments per wave: 150.6 us against 146.9 at 131 072 rows, profiles/r05_conv3_tiles_2x2_waves.txt)
        const bool big = f ? f >= 4 : (long long)((c.M + 127) / 128) * pieces >= 2LL * num_cus();
        return big ? launch_stream<SP, 4, 1, 4, CONV>(c, s) : launch_stream<SP, 2, 1, 4, CONV>(c, s);
    }
    if constexpr (CONV) {   // (three row tiles of the convolution form do not fit the register file with three pieces)
        const bool big = f ? f >= 4 : (long long)((c.M + 127) / 128) * ((c.N + 255) / 256) * pieces >= 2LL * num_cus();
        return big ? launch_stream<SP, 4, 2, 4, CONV>(c, s) : launch_stream<SP, 2, 2, 4, CONV>(c, s);
    } else {
        // (fp16 pieces: three row tiles at every K -- 22 223 rows: 256 -> 1024 48.8 vs 51.3 us, 1024 -> 256 43.2 vs 50.5, 256 -> 256
        // 18.2 vs 18.6; profiles/r04_f16_harness.txt)
        switch (f ? f : (c.K >= 512 || SP == 16 ? 3 : 2)) {
        case 1:
        case 2: return launch_stream<SP, 2, 2, 4, CONV>(c, s);
        case 4: return launch_stream<SP, 4, 2, 4, CONV>(c, s);
        default: return launch_stream<SP, 3, 2, 4, CONV>(c, s);
        }
    }
}

// ---- the halo form: block shape per call (as stream_dispatch: 64 / 128 / 256 columns; 64-pixel patches, 128 where that still leaves
// about two blocks per CU)
std::atomic<int> g_halo{-1};   // -1: TF_CONV_HALO or the default (1)
bool halo_enabled()
{
    int v = g_halo.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_CONV_HALO");
        v = (e && e[0] == '0') ? 0 : 1;
        g_halo.store(v);
    }
    return v != 0;
}

template <int SP, int TI, int TJ, int WC, bool MRG = false>
int launch_halo(const StreamCall &c, hipStream_t s, const HaloMerge mg = HaloMerge{})
{
    constexpr int BM = (4 / WC) * TI * 32, BN = WC * TJ * 32, PH = BM / 8;
    const int tiles_x = (c.cv.win + 7) / 8, tiles_y = (c.cv.hin + PH - 1) / PH;
    const long long npatches = (long long)c.cv.nimg * tiles_y * tiles_x;
    const int nblocks = (c.N + BN - 1) / BN;
    const long long gx = (npatches + 7) / 8 * 8 * nblocks;
    const int nsl = c.cv.cin / kSlice;
    int cslices = 0, gz = 1;
    if (c.ksplit > 1) {
        cslices = (nsl + c.ksplit - 1) / c.ksplit;
        gz = (nsl + cslices - 1) / cslices;
        if (gz <= 1) cslices = 0, gz = 1;
    }
    if (gx > 0x7fffffffLL || gz > 65535 || npatches > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    const bool partial = gz > 1;
    float *out = partial ? c.workspace : c.y;
    hipLaunchKernelGGL((conv3x3_halo_kernel<SP, TI, TJ, WC, MRG>), dim3((unsigned)gx, (unsigned)gz), dim3(kThreads), 0, s, c.x, c.wp,
                       partial ? nullptr : c.bias, partial ? nullptr : c.res, out, c.N, nblocks, (int)npatches, tiles_x, tiles_y,
                       partial ? 0 : c.relu, cslices, c.cv, mg);
    if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    if (partial) {
        const long long mn4 = (long long)c.M * c.N / 4;
        hipLaunchKernelGGL(stream_splitk_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, s, c.workspace, c.bias, c.res,
                           c.y, mn4, c.N / 4, gz, c.relu);
        if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    }
    return TF_MSDA_OK;
}

template <int SP>
int halo_dispatch(const StreamCall &c, hipStream_t s)
{
    const int f = forced_ti();
    const long long pieces = c.ksplit > 1 ? c.ksplit : 1;
    const bool big = (long long)((c.M + 127) / 128) * ((c.N + 255) / 256) * pieces >= 2LL * num_cus();
    // (<= 32 output channels -- the mask head's lay4 / lay5: 32 / 16 at up to 200 x 334 pixels per query --: one column tile, the four
    // waves split the rows; the 64-column shapes leave two of the four waves multiplying padding)
    // 128-row blocks whatever the launch size: at Cin <= 64 a block's life is one or two slices -- cold halo, nine taps of weight
    // fragments from L2, epilogue -- and two resident blocks per CU overlap that better than one of twice the rows (MI355X, 128
    // queries: 32 -> 16 at 200 x 334 902 vs 1 101 us, 64 -> 32 at 100 x 167 402 vs 515; tools/experiments/mask_head_convs.py)
    if (c.N <= 32) return (f >= 2) ? launch_halo<SP, 2, 1, 1>(c, s) : launch_halo<SP, 1, 1, 1>(c, s);
    if (c.N <= 64) return (f ? f >= 2 : big) ? launch_halo<SP, 2, 1, 2>(c, s) : launch_halo<SP, 1, 1, 2>(c, s);
    if (c.N <= 128) return (f ? f >= 4 : big) ? launch_halo<SP, 4, 1, 4>(c, s) : launch_halo<SP, 2, 1, 4>(c, s);
    return launch_halo<SP, 2, 2, 4>(c, s);
}

// the merged form (MRG): the mask head's shapes only -- one column tile of 32 (lay4 / lay5), 64 (lay3) or 128 channels
template <int SP>
int halo_dispatch_merge(const StreamCall &c, hipStream_t s, const HaloMerge &mg)
{
    if (c.N <= 32) return launch_halo<SP, 1, 1, 1, true>(c, s, mg);
    if (c.N <= 64) return launch_halo<SP, 2, 1, 2, true>(c, s, mg);
    if (c.N <= 128) return launch_halo<SP, 2, 1, 4, true>(c, s, mg);
    return launch_halo<SP, 2, 2, 4, true>(c, s, mg);
}

// ---- the LDS-DMA GEMM: shape per call.  g_dma: -1 = TF_LINEAR_DMA or the default, 0 = off (the stream form), 1..4 = a fixed shape
// (1: 128 x 128, 2 slices; 2: 128 x 64, 3 slices; 3: 128 x 128, 3 slices; 4: 256 x 128, 2 slices), 9 = per call shape
std::atomic<int> g_dma{-1};
int dma_mode()
{
    int v = g_dma.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("TF_LINEAR_DMA");
        v = e ? atoi(e) : 0;
        if (v < 0 || v > 9) v = 0;
        g_dma.store(v);
    }
    return v;
}

template <int SP, int TI, int TJ, int STAGES>
int launch_dma(const StreamCall &c, hipStream_t s)
{
    constexpr int BM = 4 * TI * 32, BN = TJ * 32, NB = Split<SP>::NB;
    constexpr size_t lds = (size_t)STAGES * (BM * 128 + TJ * 2 * NB * 1024);
    static_assert(lds <= 160 * 1024, "LDS of a CU");
    const int mblocks = (c.M + BM - 1) / BM, nblocks = (c.N + BN - 1) / BN;
    const long long gx = (long long)((mblocks + 7) / 8) * 8 * nblocks;
    if (gx > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    const void *fn = (const void *)&dma_gemm_kernel<SP, TI, TJ, STAGES>;
    static std::atomic<unsigned long long> raised{0};   // bit d: device d has the dynamic-LDS attribute of THIS instantiation
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TF_MSDA_ERR_LAUNCH;
        raised.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((dma_gemm_kernel<SP, TI, TJ, STAGES>), dim3((unsigned)gx), dim3(kThreads), lds, s, c.x, c.wp, c.bias, c.res, c.y,
                       c.M, c.K, c.N, mblocks, nblocks, c.relu);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

template <int SP>
int dma_dispatch(int mode, const StreamCall &c, hipStream_t s)
{
    if (mode == 9) mode = c.N <= 64 ? 2 : 1;
    switch (mode) {
    case 2: return launch_dma<SP, 1, 2, 3>(c, s);
    case 3: return launch_dma<SP, 1, 4, 3>(c, s);
    case 4: return launch_dma<SP, 2, 4, 2>(c, s);
    default: return launch_dma<SP, 1, 4, 2>(c, s);
    }
}

}  // namespace

namespace tfm {
int linear_dma_set(int v)
{
    const int prev = dma_mode();
    g_dma.store(v >= 0 && v <= 9 ? v : 0);
    return prev;
}
int conv_halo_set(int v)
{
    const int prev = halo_enabled() ? 1 : 0;
    g_halo.store(v != 0 ? 1 : 0);
    return prev;
}
int linear_stream_set_ti(int v)
{
    const int prev = forced_ti();
    g_ti.store(v >= 1 && v <= 4 ? v : 0);
    return prev;
}
}  // namespace tfm

template <bool CONV>
int stream_dispatch_scheme(int sp, const StreamCall &c, hipStream_t s)
{
    switch (sp) {
    case 3: return stream_dispatch<3, CONV>(c, s);
    default: return stream_dispatch<16, CONV>(c, s);
    }
}

extern "C" int tf_conv3x3_merge_packed_f32(const float *low, const float *fpn, const double *gn_workspace, const float *gamma,
                                           const float *beta, int groups, float eps, const void *w_packed, const float *bias, float *y, int nimg,
                                           int q_per_image, int lh, int lw, int H, int W, int cin, int cout, int relu, int terms, void *stream)
{
    if (!low || !fpn || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    if (gn_workspace && (!gamma || !beta)) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (nimg <= 0 || q_per_image <= 0 || (nimg % q_per_image) || lh <= 0 || lw <= 0 || H <= 0 || W <= 0 || cin <= 0 || cin > 320 || (cin % 32) != 0 ||
        cout <= 0 || sp == 0 || (gn_workspace && (groups <= 0 || (cin % groups) != 0)))
        return TF_MSDA_ERR_BAD_DIMS;
    const long long M = (long long)nimg * H * W;
    if ((M + 256) * cout * 4 >= 0xC0000000LL || (long long)nimg * lh * lw * cin * 4 >= 0xC0000000LL ||
        (long long)(nimg / q_per_image) * H * W * cin * 4 >= 0xC0000000LL)
        return TF_MSDA_ERR_BAD_DIMS;
    uintptr_t al = reinterpret_cast<uintptr_t>(low) | reinterpret_cast<uintptr_t>(fpn) | reinterpret_cast<uintptr_t>(w_packed);
    if (gn_workspace) al |= reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta);
    if (al & 15) return TF_MSDA_ERR_BAD_DIMS;
    StreamConv cv{nimg, H, W, cin, H, W, 3, 1, 1};
    StreamCall c{low, static_cast<const u32x4 *>(w_packed), bias, nullptr, y, (int)M, 9 * cin, cout, relu, true, cv, nullptr, 1};
    const HaloMerge mg{fpn, gn_workspace, gamma, beta, q_per_image, lh, lw, gn_workspace ? groups : 1, eps};
    return sp == 3 ? halo_dispatch_merge<3>(c, static_cast<hipStream_t>(stream), mg)
                   : halo_dispatch_merge<16>(c, static_cast<hipStream_t>(stream), mg);
}

#ifdef TF_STREAM_TRACE
extern "C" int tf_debug_stream_trace_buffer(void *device_buffer, int *blocks, int *slices, int *points)
{
    unsigned long long *p = static_cast<unsigned long long *>(device_buffer);
    if (blocks) *blocks = kTraceBlocks;
    if (slices) *slices = kTraceSlices;
    if (points) *points = kTracePoints;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stream_trace), &p, sizeof(p)) == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}
#endif

extern "C" int64_t tf_linear_packed_bytes(int K, int N, int terms)
{
    const int sp = split_scheme(terms);
    if (K <= 0 || N <= 0 || (K % 16) != 0 || sp == 0) return -1;
    const int64_t npad = ((int64_t)N + kBN - 1) / kBN * kBN;
    return npad * K * 2 * scheme_pieces_b(sp) + (sp == 16 ? npad * 4 : 0);   // 16-bit pieces per element [+ a float per output channel]
}

extern "C" int tf_linear_pack_weight_f32(const float *w, void *packed, int K, int N, int terms, void *stream)
{
    if (!w || !packed) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (K <= 0 || N <= 0 || (K % 16) != 0 || sp == 0) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(packed)) & 15) return TF_MSDA_ERR_BAD_DIMS;
    const long long npad = ((long long)N + kBN - 1) / kBN * kBN;
    const long long ntiles = npad / 32;
    const long long total = ntiles * (K >> 4) * 64;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    u32x4 *out = static_cast<u32x4 *>(packed);
    float *rs = reinterpret_cast<float *>(out + total * scheme_pieces_b(sp));   // fp16 scheme only
    switch (sp) {
    case 3: hipLaunchKernelGGL(pack_weight_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, w, out, (const float *)nullptr, K, N, total); break;
    default:
        hipLaunchKernelGGL(pack_scale_kernel, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, s, w, rs, K, N, (int)npad);
        hipLaunchKernelGGL(pack_weight_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, w, out, (const float *)rs, K, N, total);
    }
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

extern "C" int tf_linear_packed_f32(const float *x, const void *w_packed, const float *bias, const float *residual, float *y,
                                    int64_t M, int K, int N, int relu, int terms, void *stream)
{
    if (!x || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (M <= 0 || K <= 0 || N <= 0 || (K % 64) != 0 || M > 0x7fffffffLL || sp == 0) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed)) & 15) return TF_MSDA_ERR_BAD_DIMS;
    if ((long long)(M + 256) * N * 4 >= 0xC0000000LL) return TF_MSDA_ERR_BAD_DIMS;   // buffer-resource offsets (the caller keeps tf_linear_split_f32)
    StreamCall c{x, static_cast<const u32x4 *>(w_packed), bias, residual, y, (int)M, K, N, relu, false, StreamConv{}, nullptr, 1};
    if (const int mode = dma_mode())   // the LDS-DMA GEMM (both operands global -> LDS by DMA, a ring of slices in flight)
        if ((long long)M * K * 4 < 0xC0000000LL)
            return sp == 3 ? dma_dispatch<3>(mode, c, static_cast<hipStream_t>(stream)) : dma_dispatch<16>(mode, c, static_cast<hipStream_t>(stream));
    return stream_dispatch_scheme<false>(sp, c, static_cast<hipStream_t>(stream));
}

extern "C" int tf_conv_packed_f32(const float *x, const void *w_packed, const float *bias, const float *residual, float *y,
                                  float *workspace, int ksplit, int nimg, int hin, int win, int cin, int cout, int ks, int stride,
                                  int relu, int terms, void *stream)
{
    if (!x || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    // (the stream form walks pairs of 32-deep slices: Cin % 64; the halo form -- stride-1 3 x 3 -- any number of them: Cin % 32, which
    // the mask head's lay2 (288 channels) and lay5 (32) need)
    const bool halo = ks == 3 && stride == 1 && halo_enabled();
    if (nimg <= 0 || hin <= 0 || win <= 0 || cin <= 0 || cout <= 0 || (cin % (halo ? 32 : 64)) != 0 || (stride != 1 && stride != 2) ||
        (ks != 1 && ks != 3) || sp == 0 || ksplit < 1 || ksplit > 64)
        return TF_MSDA_ERR_BAD_DIMS;
    if (ksplit > 1 && !workspace) return TF_MSDA_ERR_NULL_POINTER;
    const int pad = ks == 3 ? 1 : 0;
    StreamConv cv{nimg, hin, win, cin, (hin + 2 * pad - ks) / stride + 1, (win + 2 * pad - ks) / stride + 1, ks, pad, stride};
    const long long M = (long long)nimg * cv.hout * cv.wout;
    // every byte offset of the input and the output below the out-of-range marker of the buffer resources (3 GiB)
    if (M <= 0 || (M + 256) * cout * 4 >= 0xC0000000LL || (long long)nimg * hin * win * cin * 4 >= 0xC0000000LL) return TF_MSDA_ERR_BAD_DIMS;
    uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed);
    if (ksplit > 1) {
        if (cout & 3) return TF_MSDA_ERR_BAD_DIMS;
        al |= reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual);
    }
    if (al & 15) return TF_MSDA_ERR_BAD_DIMS;
    StreamCall c{x, static_cast<const u32x4 *>(w_packed), bias, residual, y, (int)M, ks * ks * cin, cout, relu, true, cv, workspace, ksplit};
    if (halo)   // the halo form: every input pixel staged once per channel slice, not once per tap
        return sp == 3 ? halo_dispatch<3>(c, static_cast<hipStream_t>(stream)) : halo_dispatch<16>(c, static_cast<hipStream_t>(stream));
    return stream_dispatch_scheme<true>(sp, c, static_cast<hipStream_t>(stream));
}
