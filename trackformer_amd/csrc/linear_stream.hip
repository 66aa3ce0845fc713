// trackformer_amd/csrc/linear_stream.hip
//
// tf_linear_packed_f32 / tf_linear_pack_weight_f32 (include/tf_fused.h): the same arithmetic as tf_linear_split_f32
// (linear_split.hip: Y = X . W^T + bias as a bf16 split product on v_mfma_f32_32x32x16_bf16, six terms by default, three
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
constexpr int kTJ = 2;                       // MFMA column tiles per wave -> 4 waves x 2 x 32 = 256 columns per block
constexpr int kBN = 4 * kTJ * 32;

// ---- weight packing: one thread per (n-tile, k-step, lane) writes its NP pieces: piece p of (n-tile t, k-step q) at
// ((t KQ + q) NP + p) 64 + lane
template <int NP>
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float *__restrict__ W, u32x4 *__restrict__ out, int K, int N, long long total)
{
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
    }
    bf16x4 pa[NP], pb[NP];   // v_cvt_pk_bf16_f32: round to nearest even, as torch's .to(bfloat16)
    split4<NP>(a, pa);
    split4<NP>(b, pb);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = pa[p][e];
            v[4 + e] = pb[p][e];
        }
        out[(tq * NP + p) * 64 + lane] = __builtin_bit_cast(u32x4, v);
    }
}

template <int NP>
struct WFrags {
    u32x4 v[kTJ][2][NP];   // [column tile][k-step of the slice][hi | mid | lo]
};

constexpr int stream_min_waves(int ti) { return ti <= 3 ? 2 : 1; }   // blocks per CU the register budget is cut for

// BUFST (the default since round 3, linear_bufstore option; see linear_split.hip): the epilogue through a buffer resource
// (no per-store branch / wait): 57.9 -> 44.1 us at 22 223 x 256 -> 1024, 52.1 -> 39.4 us at 1024 -> 256, bit-identical
// (profiles/r03_optin_linear_bufstore.txt).  A variant with transposed accumulators and 16-byte stores measured slower on the
// first shape (58.6 us) and was removed.
template <int NP, int TI, bool RELU, bool BUFST = false>
__global__ void __launch_bounds__(kThreads, (stream_min_waves(TI)))
split_gemm_stream_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias,
                         float *__restrict__ Y, int M, int K, int N, int mblocks, int nblocks)
{
    constexpr int BM = TI * 32;
    constexpr int XV = TI;   // float4 of X per thread and slice: BM * 8 / 256
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][NP][BM * kStride];   // [buffer][hi | mid | lo][row][k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> (row block, column block): ids that are congruent mod 8 land on one XCD; the column blocks of a
    // row block sit 8 ids apart inside a group of 8 * nblocks ids
    const int per = 8 * nblocks;
    const int g = blockIdx.x / per, r = blockIdx.x - g * per;
    const int mb = g * 8 + (r & 7), nb = r >> 3;
    if (mb >= mblocks) return;   // whole block, before any barrier
    const int m0 = mb * BM, n0 = nb * kBN;
    const int S = K / kSlice, KQ = K >> 4;

    f32x16 acc[TI][kTJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < kTJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- activations: thread -> (row, 4 consecutive k) of the slice, XV rows 32 apart
    const float *xp[XV];
    const int arow = tid >> 3, ac4 = tid & 7;
#pragma unroll
    for (int it = 0; it < XV; ++it) {
        const int grow = min(m0 + it * 32 + arow, M - 1);   // rows past M read the last row, never stored
        xp[it] = X + (size_t)grow * K + ac4 * 4;
    }
    auto load_x = [&](int s, f32x4 (&dst)[XV]) {
        const int k0 = min(s, S - 1) * kSlice;   // past the end: a harmless reload of the last slice
#pragma unroll
        for (int it = 0; it < XV; ++it) dst[it] = *reinterpret_cast<const f32x4 *>(xp[it] + k0);
    };
    auto store_x = [&](const f32x4 (&src)[XV], int buf) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            bf16x4 pc[NP];
            split4<NP>(src[it], pc);
            const int o = (it * 32 + arow) * kStride + ac4 * 4;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4 *>(&sA[buf][p][o]) = pc[p];
        }
    };
    // ---- weights: the wave's two column tiles, fragment order (see pack_weight_kernel)
    const u32x4 *wp[kTJ];
#pragma unroll
    for (int j = 0; j < kTJ; ++j) wp[j] = Wp + ((size_t)(nb * 4 * kTJ + wave * kTJ + j) * KQ * NP) * 64 + lane;
    auto load_w = [&](int s, WFrags<NP> &w) {
        const int q0 = min(s, S - 1) * 2;
#pragma unroll
        for (int j = 0; j < kTJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int p = 0; p < NP; ++p) w.v[j][kk][p] = wp[j][((q0 + kk) * NP + p) * 64];
    };

    f32x4 xr[2][XV];   // slice s + 1 lives in xr[(s + 1) & 1], slice s + 2 in the other one
    WFrags<NP> w0, w1;
    {
        f32x4 first[XV];
        load_x(0, first);
        load_w(0, w0);
        load_x(1, xr[1]);
        load_x(2, xr[0]);
        store_x(first, 0);
    }
    __syncthreads();

    // one K-slice; PAR = s & 1 as a compile-time constant so that the register double buffers need no copies
    auto slice = [&](int s, auto par, const WFrags<NP> &cur, WFrags<NP> &nxt) {
        constexpr int PAR = decltype(par)::value;
        load_w(s + 1, nxt);   // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler otherwise sinks them to the end of the
                                             // slice to shorten live ranges, and the next slice starts by waiting for L2
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = kk * 16 + (lane >> 5) * 8;
            bf16x8 af[TI][NP], bfr[kTJ][NP];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int o = (i * 32 + (lane & 31)) * kStride + koff;
#pragma unroll
                for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const bf16x8 *>(&sA[PAR][p][o]);
            }
#pragma unroll
            for (int j = 0; j < kTJ; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) bfr[j][p] = __builtin_bit_cast(bf16x8, cur.v[j][kk][p]);
            // term-major passes over the tiles: consecutive MFMAs never share an accumulator; per accumulator the order is
            // smallest terms first, as in linear_split.hip
            mfma_tiles<NP, TI, kTJ>(acc, af, bfr);
        }
        // slice s + 1 -> the LDS buffer nobody reads in this iteration (its readers passed the previous barrier),
        // then its registers take slice s + 3
        store_x(xr[PAR ^ 1], PAR ^ 1);
        load_x(s + 3, xr[PAR ^ 1]);
        __syncthreads();
    };
    for (int s = 0; s < S; s += 2) {   // S is even (host: K % 64 == 0)
        slice(s, std::integral_constant<int, 0>{}, w0, w1);
        slice(s + 1, std::integral_constant<int, 1>{}, w1, w0);
    }

    // ---- epilogue: C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (BUFST) {
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)((size_t)M * N * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < kTJ; ++j) {
            const int col = n0 + (wave * kTJ + j) * 32 + (lane & 31);
            const bool colok = col < N;
            const float b = (bias && colok) ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int row0 = m0 + i * 32 + 4 * (lane >> 5);
                // rows >= M land beyond num_records (dropped by the hardware); columns >= N start from 3 GiB, which stays
                // out of range and does not wrap for any row delta (host: the tensor is < 3 GiB)
                const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : 0xC0000000u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] + b;
                    if (RELU) v = v > 0.f ? v : 0.f;
                    const unsigned off = base + (unsigned)(((e & 3) + 8 * (e >> 2)) * N) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, off, 0, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kTJ; ++j) {
        const int col = n0 + (wave * kTJ + j) * 32 + (lane & 31);
        if (col >= N) continue;
        const float b = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float v = acc[i][j][e] + b;
                    if (RELU) v = v > 0.f ? v : 0.f;
                    Y[(size_t)row * N + col] = v;
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
        if (v < 2 || v > 4) v = 0;
        g_ti.store(v);
    }
    return v;
}

// rows per block = 32 TI.  Measured at 22 223 rows with three terms (profiles/r02_split_gemm_packed.txt): two row tiles (three
// blocks per CU resident) win at K = 256 -- 59.3 / 67.3 / 84.2 us for TI = 2 / 3 / 4 at N = 1024 -- where a block's eight K-slices
// are too short a loop to hide its own memory latency and the co-resident blocks have to; three win at K = 1024
// (55.6 / 50.5 / 58.0 us), where the loop is long enough and the weight traffic per MFMA counts.
int choose_ti(int K)
{
    const int f = forced_ti();
    if (f) return f;
    return K >= 512 ? 3 : 2;
}

template <int NP, int TI>
int launch_stream(const float *x, const u32x4 *wp, const float *bias, float *y, int M, int K, int N, int relu, hipStream_t s)
{
    const int mblocks = (M + 32 * TI - 1) / (32 * TI), nblocks = (N + kBN - 1) / kBN;
    const long long grid = (long long)((mblocks + 7) / 8) * 8 * nblocks;
    if (grid > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    if ((long long)(M + 256) * N * 4 < 0xC0000000LL) {   // buffer-store epilogue (tensors < 3 GiB)
        if (relu)
            hipLaunchKernelGGL((split_gemm_stream_kernel<NP, TI, true, true>), dim3((unsigned)grid), dim3(kThreads), 0, s, x, wp, bias, y,
                               M, K, N, mblocks, nblocks);
        else
            hipLaunchKernelGGL((split_gemm_stream_kernel<NP, TI, false, true>), dim3((unsigned)grid), dim3(kThreads), 0, s, x, wp, bias, y,
                               M, K, N, mblocks, nblocks);
        return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
    }
    if (relu)
        hipLaunchKernelGGL((split_gemm_stream_kernel<NP, TI, true>), dim3((unsigned)grid), dim3(kThreads), 0, s, x, wp, bias, y, M, K,
                           N, mblocks, nblocks);
    else
        hipLaunchKernelGGL((split_gemm_stream_kernel<NP, TI, false>), dim3((unsigned)grid), dim3(kThreads), 0, s, x, wp, bias, y, M, K,
                           N, mblocks, nblocks);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

}  // namespace

namespace tfm {
int linear_stream_set_ti(int v)
{
    const int prev = forced_ti();
    g_ti.store(v >= 2 && v <= 4 ? v : 0);
    return prev;
}
}  // namespace tfm

extern "C" int64_t tf_linear_packed_bytes(int K, int N, int terms)
{
    const int np = split_pieces(terms);
    if (K <= 0 || N <= 0 || (K % 16) != 0 || np == 0) return -1;
    const int64_t npad = ((int64_t)N + kBN - 1) / kBN * kBN;
    return npad * K * 2 * np;   // np bf16 pieces per element
}

extern "C" int tf_linear_pack_weight_f32(const float *w, void *packed, int K, int N, int terms, void *stream)
{
    if (!w || !packed) return TF_MSDA_ERR_NULL_POINTER;
    const int np = split_pieces(terms);
    if (K <= 0 || N <= 0 || (K % 16) != 0 || np == 0) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(packed)) & 15) return TF_MSDA_ERR_BAD_DIMS;
    const long long ntiles = ((long long)N + kBN - 1) / kBN * (kBN / 32);
    const long long total = ntiles * (K >> 4) * 64;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    if (np == 3)
        hipLaunchKernelGGL(pack_weight_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                           static_cast<u32x4 *>(packed), K, N, total);
    else
        hipLaunchKernelGGL(pack_weight_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                           static_cast<u32x4 *>(packed), K, N, total);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

extern "C" int tf_linear_packed_f32(const float *x, const void *w_packed, const float *bias, float *y, int64_t M, int K, int N,
                                    int relu, int terms, void *stream)
{
    if (!x || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    const int np = split_pieces(terms);
    if (M <= 0 || K <= 0 || N <= 0 || (K % 64) != 0 || M > 0x7fffffffLL || np == 0) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_packed)) & 15) return TF_MSDA_ERR_BAD_DIMS;
    const u32x4 *wp = static_cast<const u32x4 *>(w_packed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto npc) {
        constexpr int NP = decltype(npc)::value;
        switch (choose_ti(K)) {
        case 2: return launch_stream<NP, 2>(x, wp, bias, y, (int)M, K, N, relu, s);
        case 4: return launch_stream<NP, 4>(x, wp, bias, y, (int)M, K, N, relu, s);
        default: return launch_stream<NP, 3>(x, wp, bias, y, (int)M, K, N, relu, s);
        }
    };
    return np == 3 ? go(std::integral_constant<int, 3>{}) : go(std::integral_constant<int, 2>{});
}
