// trackformer_amd/csrc/linear_split.hip
//
// tf_linear_split_f32 (include/tf_fused.h): Y[M, N] = X[M, K] . W[N, K]^T + bias, optionally ReLU -- the
// nn.Linear of the encoder / decoder (reference: models/ops/modules/ms_deform_attn.py:64-88 value_proj /
// sampling_offsets / attention_weights / output_proj, models/deformable_transformer.py:282-297 linear1 / linear2)
// with fp32 inputs and outputs, computed on the bf16 matrix cores as a SPLIT product:
//     x = x_hi + x_mid,  w = w_hi + w_mid  (bf16 pieces, round to nearest even),
//     x . w ~= x_hi . w_hi + x_hi . w_mid + x_mid . w_hi      (three v_mfma_f32_32x32x16_bf16 per K-step, fp32 accumulate)
// The dropped terms are below 2^-16 of the product: the model and the tracker stay inside the parity bar
// (tools/experiments/bf16_split_linear.py: all reference goldens, track ids exact), while three bf16 passes have
// 5x the throughput of the fp32 MFMA instructions the library GEMMs use.  First version (untuned, correct):
// 25.9 us for 22 223 x 256 -> 256 (hipBLASLt fp32, tuned: 32 us), 69.7 us for 256 -> 1024 (109 us),
// 68.2 us for 1024 -> 256 (100 us); profiles/r01_split_gemm_experiment.txt.  OPT-IN (TF_SPLIT_LINEAR=1) until the
// end-to-end numbers are in.
//
//   * 256 threads = 4 waves per 128 x 128 output block, each wave 64 x 64 (2 x 2 MFMA tiles of 32 x 32);
//   * per K-slice of 32: the X tile is loaded as fp32, split into (hi, mid) in registers (v_cvt_pk_bf16_f32)
//     and stored to LDS as bf16; the weight pieces are split once by the caller (constants in inference);
//   * operands are read from LDS with ds_read_b128 (8 consecutive k of one row per lane).  The A and B fragments
//     of v_mfma_f32_32x32x16_bf16 use the same (lane >> 5, element) -> k mapping, so loading the SAME k into the
//     same slot of both makes the result independent of what that mapping is; the C/D mapping is
//     col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
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

constexpr int BK = 32, THREADS = 256;
constexpr int LDS_STRIDE = BK + 8;   // bf16 elements per LDS row: 80 bytes, keeps 16-byte alignment, spreads banks

// Which tile row the q-th group of staging lanes (8 lanes x 8 B of an A row, 4 lanes x 16 B of a weight row) carries:
// within every 8 rows the order 0 4 1 5 2 6 3 7.  An LDS store is serviced in groups of 16 (b64) / 8 (b128) consecutive
// lanes against 32 banks, i.e. two staging groups at a time; with 80-byte rows, rows r and r + 1 share 4 banks (dwords
// 0..15 and 20..35 mod 32) -- every store of the kernels below was a 2-way conflict, SQ_LDS_BANK_CONFLICT = 32 cycles per
// wave and K-slice = a third of SQ_LDS_IDX_ACTIVE (profiles/r03_conv3_pmc.txt) -- while rows r and r + 4 (dwords
// 0..15 and 80..95 = 16..31 mod 32) cover the 32 banks exactly.  Only the lane -> row assignment of the staging loads
// and stores changes; the LDS layout, the fragment reads and the arithmetic do not (bit-identical results).  Time per
// layer: unchanged within noise (profiles/r03_bench_conv_stage_rows.txt) -- the stores were not what the kernels wait for.
__device__ __forceinline__ int stage_row(int q) { return (q & ~7) | ((q & 1) << 2) | ((q & 7) >> 1); }

// BM x BN output block, 4 waves as 2 x 2, each wave (BM / 2) x (BN / 2) = TI x TJ MFMA tiles of 32 x 32.
// PREFETCH: the global loads of K-slice s + 1 are issued before the MFMAs of slice s and written to LDS after them
// (register double buffer), so a block's memory latency hides under its own matrix work instead of relying on a
// second block on the CU being in the other phase.
// RESID: Y = act(X . W^T + bias + R) with R [M, N] fp32 (the identity branch of a bottleneck: `out += identity` before
// the ReLU, torchvision resnet.py Bottleneck.forward) -- only instantiated by split_gemm_res_kernel below.
// BUFST (the default since round 3 for tensors < 3 GiB; tf_msda_set_option("linear_bufstore", 0) / TF_LINEAR_BUFSTORE=0 keeps
// the plain stores; 21.4 -> 18.1 us at 22 223 x 256 -> 256, 26.9 -> 18.9 us at 66 800 x 64 -> 256, bit-identical,
// profiles/r03_optin_linear_bufstore.txt): the epilogue through a buffer resource over Y -- rows >= M
// and columns >= N fall outside num_records and are dropped by the hardware, so the 16 stores of a tile are straight-line
// code.  With `if (row < M) Y[...] = v` every store sits in its own exec-masked block; the compiler's wait-count pass
// re-waits for the bias load at each join, and on gfx9-family hardware vmcnt also counts STORES: the ISA had
// `s_waitcnt vmcnt(0)` in front of every global_store_dword, i.e. each store waited for the previous one to reach L2
// (profiles/r02_split_gemm_astat_trace.txt: 10.5 us of store time per 96 x 256 tile).
// XADD (tf_linear_split_add_f32): the activation is X + X2, added element-wise as the tile is staged -- the layers'
// `with_pos_embed(src, pos)` in front of a projection (deformable_transformer.py:279-283) without its own pass over the tokens.
// SP: the scheme of split_product.h (2 / 3: bf16 pieces, Wsc unused; 16: fp16 pieces -- Whi / Wmid hold the weight's hi and lo pieces,
// Wlo is unused, Wsc the output channels' powers of two).
template <int SP, int BM, int BN, bool RELU, bool PREFETCH, bool RESID, bool BUFST, bool XADD = false>
__device__ __forceinline__ void
split_gemm_body(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                const float *__restrict__ bias, const float *__restrict__ R, float *__restrict__ Y, int M, int K, int N,
                const float *__restrict__ X2 = nullptr)
{
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr bool F16 = Split<SP>::F16;
    const unsigned short *const Wp[3] = {Whi, Wmid, Wlo};
    constexpr int TI = BM / 64, TJ = BN / 64;
    constexpr int XV = (BM * BK / 4) / THREADS;   // float4 of X per thread and slice
    constexpr int WV = (BN * BK / 8) / THREADS;   // 16-byte pieces of each weight tensor per thread and slice
    static_assert(XV >= 1 && WV >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) unsigned short sA[NA][BM * LDS_STRIDE];   // [activation piece][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short sB[NB][BN * LDS_STRIDE];   // [weight piece][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 xr[XV];
    u32x4 wr[NB][WV];
    f32x4 xr2[XADD ? XV : 1];
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;          // float4 index: 8 per row
            const int row = stage_row(idx >> 3), c4 = idx & 7;
            const int grow = min(m0 + row, M - 1);       // rows past M read the last row, never stored
            xr[it] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * K + k0 + c4 * 4);
            // XADD: the second operand stays in its own registers and is added when the slice is staged (store_slice).
            // Written as `xr += load` here, hipcc put the v_pk_add -- and with it `s_waitcnt vmcnt` for BOTH loads --
            // right behind the loads: the prefetch of slice s + 1 was waited for before the MFMAs of slice s (the defect
            // of DESIGN.md section 4.4 again; found by tools/isa_wait_audit.py)
            if constexpr (XADD) xr2[it] = *reinterpret_cast<const f32x4 *>(X2 + (size_t)grow * K + k0 + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;          // 16-byte index: 4 per row
            const int row = stage_row(idx >> 2), c8 = idx & 3;
            const int grow = min(n0 + row, N - 1);
            const size_t g = (size_t)grow * K + k0 + c8 * 8;
#pragma unroll
            for (int q = 0; q < NB; ++q) wr[q][it] = *reinterpret_cast<const u32x4 *>(Wp[q] + g);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 3), c4 = idx & 7;
            u32x2 pc[NA];   // hardware conversion (round to nearest even)
            f32x4 xv = xr[it];
            if constexpr (XADD) xv += xr2[it];
            split4<SP>(xv, pc);
#pragma unroll
            for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x2 *>(&sA[q][row * LDS_STRIDE + c4 * 4]) = pc[q];
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 2), c8 = idx & 3;
#pragma unroll
            for (int q = 0; q < NB; ++q) *reinterpret_cast<u32x4 *>(&sB[q][row * LDS_STRIDE + c8 * 8]) = wr[q][it];
        }
    };

    if constexpr (PREFETCH) load_slice(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        if constexpr (!PREFETCH) load_slice(k0);
        store_slice();
        __syncthreads();
        if constexpr (PREFETCH)
            if (k0 + BK < K) load_slice(k0 + BK);   // in flight during the MFMAs below
        // ---- 2 K-steps of 16: lane -> row (lane & 31) of the 32-row tile, 8 consecutive k from (lane >> 5) * 8
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int koff = kk + (lane >> 5) * 8;
            u32x4 af[TI][NA], bfr[TJ][NB];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int r = (wm + i * 32 + (lane & 31)) * LDS_STRIDE + koff;
#pragma unroll
                for (int q = 0; q < NA; ++q) af[i][q] = *reinterpret_cast<const u32x4 *>(&sA[q][r]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int r = (wn + j * 32 + (lane & 31)) * LDS_STRIDE + koff;
#pragma unroll
                for (int q = 0; q < NB; ++q) bfr[j][q] = *reinterpret_cast<const u32x4 *>(&sB[q][r]);
            }
            mfma_tiles<SP, TI, TJ>(acc, af, bfr);   // smallest terms first
        }
        __syncthreads();
    }
    // ---- epilogue: C/D of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (BUFST) {
        const unsigned ybytes = (unsigned)((size_t)M * N * 4);   // < 4 GiB (host check)
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(RESID ? R : Y), 0, ybytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int col = n0 + wn + j * 32 + (lane & 31);
                const bool colok = col < N;
                const float b = (bias && colok) ? bias[col] : 0.f;
                const float rsc = (F16 && colok) ? Wsc[col] : 1.f;
                const int row0 = m0 + wm + i * 32 + 4 * (lane >> 5);
                // rows >= M: (row * N + col) * 4 >= num_records -> dropped; columns >= N start from 3 GiB, which stays out
                // of range and does not wrap for any row delta (host: the tensor is < 3 GiB)
                const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : 0xC0000000u;
                float rv[16];
                if constexpr (RESID) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned off = base + (unsigned)(((r & 3) + 8 * (r >> 2)) * N) * 4u;
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, off, 0, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = F16 ? __builtin_fmaf(acc[i][j][r], rsc, b) : acc[i][j][r] + b;
                    if constexpr (RESID) v += rv[r];
                    if (RELU) v = v < 0.f ? 0.f : v;
                    const unsigned off = base + (unsigned)(((r & 3) + 8 * (r >> 2)) * N) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, off, 0, tfm::kStoreAux);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float b = bias ? bias[col] : 0.f;
            const float rsc = F16 ? Wsc[col] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float v = F16 ? __builtin_fmaf(acc[i][j][r], rsc, b) : acc[i][j][r] + b;
                    if constexpr (RESID) v += R[(size_t)row * N + col];
                    if (RELU) v = v < 0.f ? 0.f : v;
                    Y[(size_t)row * N + col] = v;
                }
            }
        }
}

template <int SP, int BM, int BN, bool RELU, bool PREFETCH, bool BUFST = false>
__global__ void __launch_bounds__(THREADS)
split_gemm_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                  const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                  const float *__restrict__ bias, float *__restrict__ Y, int M, int K, int N)
{
    split_gemm_body<SP, BM, BN, RELU, PREFETCH, false, BUFST>(X, Whi, Wmid, Wlo, Wsc, bias, nullptr, Y, M, K, N);
}

template <int SP, int BM, int BN, bool RELU, bool PREFETCH, bool BUFST = false>
__global__ void __launch_bounds__(THREADS)
split_gemm_res_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                      const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                      const float *__restrict__ bias, const float *__restrict__ R, float *__restrict__ Y, int M, int K, int N)
{
    split_gemm_body<SP, BM, BN, RELU, PREFETCH, true, BUFST>(X, Whi, Wmid, Wlo, Wsc, bias, R, Y, M, K, N);
}

template <int SP, int BM, int BN, bool PREFETCH, bool BUFST>
__global__ void __launch_bounds__(THREADS)
split_gemm_add_kernel(const float *__restrict__ X, const float *__restrict__ X2, const unsigned short *__restrict__ Whi,
                      const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                      const float *__restrict__ bias, float *__restrict__ Y, int M, int K, int N)
{
    split_gemm_body<SP, BM, BN, false, PREFETCH, false, BUFST, true>(X, Whi, Wmid, Wlo, Wsc, bias, nullptr, Y, M, K, N, X2);
}

// ---- 3 x 3 convolution (padding 1, stride 1 or 2) on channels_last activations as the same split product: an
// implicit GEMM with M = N * Hout * Wout pixels, K = 9 * Cin (tap-major: k = (kh * 3 + kw) * Cin + c -- exactly how a
// channels_last OIHW weight lies in memory, so the folded FrozenBN weight is used as it is), N = Cout.  The only
// difference from split_gemm_kernel is where a row of the A tile comes from: K-slice s lies inside ONE tap (Cin % 32 ==
// 0), its rows are the input pixels shifted by that tap, zeros outside the image.  Register prefetch of the next
// slice, buffer-store epilogue (bias = FrozenBN shift, ReLU).  OPT-IN route of trackformer_amd/backbone.py
// (TF_CONV3X3_SPLIT=1): the bottlenecks' 3 x 3 convolutions (reference: torchvision Bottleneck.conv2 + bn2 + relu).
struct Conv3Args {
    int nimg, hin, win, cin, hout, wout, cout, stride;
    int ks, pad;   // 3 / 1 (the bottlenecks' 3 x 3 convolutions) or 1 / 0 (the strided 1 x 1 projections of the identity branch)
    int kslices;   // 0: a block walks the whole K; > 0 (split-K): block z walks K-slices z kslices .. and writes its partial
                   // sums to Y + z M N (no bias / ReLU) -- few output pixels with a long K (the extra pyramid level, layer4)
};

//
// The A rows and the weight pieces are fetched through buffer resources with 32-bit offsets.  A tap that falls outside the
// image gets an offset beyond num_records and the hardware returns zeros, so nothing is selected AFTER the load: with
// `v = *p; x = ok ? v : 0` the compiler placed the v_cndmask right behind each global_load (s_waitcnt vmcnt(0) between the
// first A load and the remaining five of the slice, vmcnt(4) behind the second), i.e. the prefetch of slice s + 1 waited
// for its own data BEFORE the MFMAs of slice s that were meant to cover it -- two exposed memory latencies per K-slice
// (the ISA is quoted in DESIGN.md section 4.4).  The 64-bit address arithmetic (v_mad_i64 / v_mad_u64 chains per load)
// goes away with it: per slice one uniform tap offset is added to per-thread constants.  Inputs and weights have to lie below
// the out-of-range marker (3 GiB; host check -- larger tensors are refused and the caller keeps the library convolution).
// Measured and NOT kept (profiles/r03_conv3_bufload.txt, r03_conv3_ksplit.txt, r03_conv3_tile128.txt, r03_conv3_ahead2.txt,
// r04_conv3_operands_ahead.txt): pointer loads (the defect above), a second LDS stage with one barrier per slice (1-10 % slower
// at every layer), 128-row output tiles (+3 %), the loads issued two slices ahead with a second register set (+1.5 %), the
// LDS operands read one k-step ahead of their MFMAs with two LDS stages (+7 % over the backbone).
template <int SP, int BM, int BN, bool RELU>
__global__ void __launch_bounds__(THREADS)
split_conv3_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                   const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                   const float *__restrict__ bias, float *__restrict__ Y, const Conv3Args ca)
{
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr bool F16 = Split<SP>::F16;
    constexpr int TI = BM / 64, TJ = BN / 64;
    constexpr int XV = (BM * BK / 4) / THREADS;
    constexpr int WV = (BN * BK / 8) / THREADS;
    static_assert(XV >= 1 && WV >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) unsigned short sA[NA][BM * LDS_STRIDE];   // [activation piece][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short sB[NB][BN * LDS_STRIDE];   // [weight piece][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int M = ca.nimg * ca.hout * ca.wout, K = ca.ks * ca.ks * ca.cin, N = ca.cout;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this thread's XV rows of the A tile: output pixel -> (image, top-left input pixel of its 3 x 3 window)
    int ybase[XV], xbase[XV];
    bool rowok[XV];
    f32x4 xr[XV];
    u32x4 wr[NB][WV];
    // byte offsets of this thread's pieces at tap (0, 0), channel 0 / at k = 0 (wrap-around arithmetic: a border
    // pixel's window starts in front of the image, the sum with a valid tap's offset is back inside)
    constexpr unsigned OOB = 0xC0000000u;   // >= num_records of every resource below (sizes are checked by the host)
    unsigned xoff[XV], woff[WV];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(X), 0, (unsigned)((size_t)ca.nimg * ca.hin * ca.win * ca.cin * 4), 0x00020000);
    const unsigned wbytes = (unsigned)((size_t)N * K * 2);
    const __amdgpu_buffer_rsrc_t wrs[3] = {
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(Whi), 0, wbytes, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(Wmid), 0, wbytes, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(NB > 2 ? Wlo : Whi), 0, wbytes, 0x00020000)};
#pragma unroll
    for (int it = 0; it < XV; ++it) {
        const int idx = it * THREADS + tid;
        const int row = m0 + (stage_row(idx >> 3));
        rowok[it] = row < M;
        const int r = rowok[it] ? row : 0;
        const int img = r / (ca.hout * ca.wout), rem = r - img * (ca.hout * ca.wout);
        const int yo = rem / ca.wout, xo = rem - yo * ca.wout;
        ybase[it] = yo * ca.stride - ca.pad;
        xbase[it] = xo * ca.stride - ca.pad;
        xoff[it] = ((unsigned)((img * ca.hin + ybase[it]) * ca.win + xbase[it]) * (unsigned)ca.cin + (unsigned)((idx & 7) * 4)) * 4u;
    }
#pragma unroll
    for (int it = 0; it < WV; ++it) {
        const int idx = it * THREADS + tid;
        woff[it] = ((unsigned)min(n0 + (stage_row(idx >> 2)), N - 1) * (unsigned)K + (unsigned)((idx & 3) * 8)) * 2u;
    }
    // the slices are walked in order: (channel, dx, dy) of the NEXT slice are stepped instead of dividing k0 per slice
    int nc0 = 0, ndx = 0, ndy = 0;
    auto load_slice = [&](int k0) {
        const int c0 = nc0, dx = ndx, dy = ndy;
        nc0 += BK;
        if (nc0 == ca.cin) {
            nc0 = 0;
            if (++ndx == ca.ks) {
                ndx = 0;
                ++ndy;
            }
        }
        const unsigned tapoff = (unsigned)((dy * ca.win + dx) * ca.cin + c0) * 4u;   // uniform
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const bool ok = rowok[it] && (unsigned)(ybase[it] + dy) < (unsigned)ca.hin && (unsigned)(xbase[it] + dx) < (unsigned)ca.win;
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xoff[it] + tapoff : OOB, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const unsigned o = woff[it] + (unsigned)k0 * 2u;
#pragma unroll
            for (int q = 0; q < NB; ++q) wr[q][it] = __builtin_amdgcn_raw_buffer_load_b128(wrs[q], o, 0, 0);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 3), c4 = idx & 7;
            u32x2 pc[NA];
            split4<SP>(xr[it], pc);
#pragma unroll
            for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x2 *>(&sA[q][row * LDS_STRIDE + c4 * 4]) = pc[q];
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 2), c8 = idx & 3;
#pragma unroll
            for (int q = 0; q < NB; ++q) *reinterpret_cast<u32x4 *>(&sB[q][row * LDS_STRIDE + c8 * 8]) = wr[q][it];
        }
    };

    const int kbeg = ca.kslices > 0 ? (int)blockIdx.z * ca.kslices * BK : 0;
    const int kend = ca.kslices > 0 ? min(K, kbeg + ca.kslices * BK) : K;
    if (ca.kslices > 0) Y += (size_t)blockIdx.z * M * N;   // this split's partial sums
    {
        const int tap = kbeg / ca.cin;
        nc0 = kbeg - tap * ca.cin;
        ndy = tap / ca.ks;
        ndx = tap - ndy * ca.ks;
    }
    load_slice(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        store_slice();
        __syncthreads();
        if (k0 + BK < kend) load_slice(k0 + BK);   // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int koff = kk + (lane >> 5) * 8;
            u32x4 af[TI][NA], bfr[TJ][NB];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int r = (wm + i * 32 + (lane & 31)) * LDS_STRIDE + koff;
#pragma unroll
                for (int q = 0; q < NA; ++q) af[i][q] = *reinterpret_cast<const u32x4 *>(&sA[q][r]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int r = (wn + j * 32 + (lane & 31)) * LDS_STRIDE + koff;
#pragma unroll
                for (int q = 0; q < NB; ++q) bfr[j][q] = *reinterpret_cast<const u32x4 *>(&sB[q][r]);
            }
            mfma_tiles<SP, TI, TJ>(acc, af, bfr);
        }
        __syncthreads();
    }
    // ---- epilogue: buffer stores (rows >= M beyond num_records, columns >= N from 3 GiB)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)((size_t)M * N * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = n0 + wn + j * 32 + (lane & 31);
            const bool colok = col < N;
            const float b = (bias && colok) ? bias[col] : 0.f;
            const float rsc = (F16 && colok) ? Wsc[col] : 1.f;   // (split-K partial sums carry the factor too)
            const int row0 = m0 + wm + i * 32 + 4 * (lane >> 5);
            const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : 0xC0000000u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = F16 ? __builtin_fmaf(acc[i][j][r], rsc, b) : acc[i][j][r] + b;
                if (RELU) v = v < 0.f ? 0.f : v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                      base + (unsigned)(((r & 3) + 8 * (r >> 2)) * N) * 4u, 0, tfm::kStoreAux);
            }
        }
}

// ---- split-K second pass: y = act(sum_z partial[z] + bias), the partials added in the order z = 0, 1, ... (a fixed order:
// the result does not depend on scheduling, unlike atomic accumulation).  One thread per 4 consecutive outputs.
__global__ void __launch_bounds__(256)
conv_splitk_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias, float *__restrict__ y, long long mn4,
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
    if (relu) {
        acc.x = acc.x < 0.f ? 0.f : acc.x;
        acc.y = acc.y < 0.f ? 0.f : acc.y;
        acc.z = acc.z < 0.f ? 0.f : acc.z;
        acc.w = acc.w < 0.f ? 0.f : acc.w;
    }
    reinterpret_cast<f32x4 *>(y)[i] = acc;
}

// ---- few rows (the decoder: 400 / 800 queries): a launch is 28-100 blocks, far fewer than CUs, and each block walks its
// K-slices one memory round trip at a time -- 12 us per 400 x 256 -> 256 linear inside the model (36 such launches per
// frame, profiles/r02_e2e_eager_per_frame.txt) for 0.16 GFLOP.  Variant 7 (the default for every call with <= 4096 rows
// since round 3 -- 6.7 -> 4.9 us at 400 x 256 -> 256, 15.6 -> 12.3 us at 400 x 1024 -> 256, profiles/r03_optin_linear_bufstore.txt;
// tf_msda_set_option("linear_deep", 0) / TF_LINEAR_DEEP=0 goes back to variant 5) keeps a RING OF 8 K-SLICES in registers: all
// of a K = 256 block's global loads are in flight at once (one round trip instead of eight), the loop then only moves
// registers -> LDS (double buffered, one barrier per slice) -> matrix cores; longer K refills the ring slot it has just
// consumed.  64 x 64 blocks, 4 waves as 2 x 2; same arithmetic and accumulation order as split_gemm_kernel (bit-identical
// results); buffer-store epilogue.
// S = K / 32 is a template parameter (8, 9, 32, 36: hidden 256 / 288 and their FFN widths) so that the slice loop is
// straight-line code: with run-time trip counts the compiler's wait-count pass loses track of how many loads are in
// flight across the branches and falls back to vmcnt(0), which would wait for the refill just issued.
template <int SP, bool RELU, int S>
__global__ void __launch_bounds__(THREADS)
split_gemm_deep_kernel(const float *__restrict__ X, const unsigned short *__restrict__ Whi,
                       const unsigned short *__restrict__ Wmid, const unsigned short *__restrict__ Wlo, const float *__restrict__ Wsc,
                       const float *__restrict__ bias, float *__restrict__ Y, int M, int K, int N)
{
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB;
    constexpr int BM = 64, BN = 64, PFD = 8;
    constexpr int XV = (BM * BK / 4) / THREADS;   // 2 float4 of X per thread and slice
    constexpr int WV = (BN * BK / 8) / THREADS;   // 1 16-byte piece of each weight tensor per thread and slice
    const unsigned short *const Wp[3] = {Whi, Wmid, Wlo};
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][NA][BM * LDS_STRIDE];   // [buffer][activation piece][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][NB][BN * LDS_STRIDE];   // [buffer][weight piece][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    f32x4 xr[PFD][XV];
    u32x4 wr[PFD][NB][WV];
    const float *xp[XV];
    size_t wg[WV];
#pragma unroll
    for (int it = 0; it < XV; ++it) {
        const int idx = it * THREADS + tid;
        const int row = stage_row(idx >> 3), c4 = idx & 7;
        xp[it] = X + (size_t)min(m0 + row, M - 1) * K + c4 * 4;   // rows past M read the last row, never stored
    }
#pragma unroll
    for (int it = 0; it < WV; ++it) {
        const int idx = it * THREADS + tid;
        const int row = stage_row(idx >> 2), c8 = idx & 3;
        wg[it] = (size_t)min(n0 + row, N - 1) * K + c8 * 8;
    }
    auto load_slice = [&](int s, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
#pragma unroll
        for (int it = 0; it < XV; ++it) xr[slot][it] = *reinterpret_cast<const f32x4 *>(xp[it] + s * BK);
#pragma unroll
        for (int it = 0; it < WV; ++it)
#pragma unroll
            for (int q = 0; q < NB; ++q) wr[slot][q][it] = *reinterpret_cast<const u32x4 *>(Wp[q] + wg[it] + s * BK);
    };
    auto store_slice = [&](auto slotc, int buf) {
        constexpr int slot = decltype(slotc)::value;
#pragma unroll
        for (int it = 0; it < XV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 3), c4 = idx & 7;
            u32x2 pc[NA];   // round to nearest even
            split4<SP>(xr[slot][it], pc);
#pragma unroll
            for (int q = 0; q < NA; ++q) *reinterpret_cast<u32x2 *>(&sA[buf][q][row * LDS_STRIDE + c4 * 4]) = pc[q];
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) {
            const int idx = it * THREADS + tid;
            const int row = stage_row(idx >> 2), c8 = idx & 3;
#pragma unroll
            for (int q = 0; q < NB; ++q) *reinterpret_cast<u32x4 *>(&sB[buf][q][row * LDS_STRIDE + c8 * 8]) = wr[slot][q][it];
        }
    };
    auto each_slot = [&](auto &&f) {
        f(std::integral_constant<int, 0>{});
        f(std::integral_constant<int, 1>{});
        f(std::integral_constant<int, 2>{});
        f(std::integral_constant<int, 3>{});
        f(std::integral_constant<int, 4>{});
        f(std::integral_constant<int, 5>{});
        f(std::integral_constant<int, 6>{});
        f(std::integral_constant<int, 7>{});
    };

    // ---- every load of the first 8 slices in flight at once
    static_assert(S >= PFD, "at least one full ring");
    each_slot([&](auto jc) { load_slice(decltype(jc)::value, jc); });
    store_slice(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    auto step = [&](auto sc) {
        constexpr int sidx = decltype(sc)::value;
        if constexpr (sidx < S) {
            constexpr int j = sidx % PFD, buf = sidx & 1;
            // slice s + 1 -> the LDS buffer whose readers passed the previous barrier; slot j (consumed one step
            // ago) takes slice s + 8
            if constexpr (sidx + 1 < S) store_slice(std::integral_constant<int, (j + 1) % PFD>{}, buf ^ 1);
            if constexpr (sidx + PFD < S) load_slice(sidx + PFD, std::integral_constant<int, j>{});
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                const int koff = kk + (lane >> 5) * 8;
                const int ra = (wm + (lane & 31)) * LDS_STRIDE + koff, rb = (wn + (lane & 31)) * LDS_STRIDE + koff;
                u32x4 af[NA], bfr[NB];
#pragma unroll
                for (int q = 0; q < NA; ++q) af[q] = *reinterpret_cast<const u32x4 *>(&sA[buf][q][ra]);
#pragma unroll
                for (int q = 0; q < NB; ++q) bfr[q] = *reinterpret_cast<const u32x4 *>(&sB[buf][q][rb]);
                mfma_terms<SP>(acc, af, bfr);   // smallest terms first
            }
            __syncthreads();
        }
    };
    auto steps4 = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        step(std::integral_constant<int, b>{});
        step(std::integral_constant<int, b + 1>{});
        step(std::integral_constant<int, b + 2>{});
        step(std::integral_constant<int, b + 3>{});
    };
    static_assert(S <= 36, "unrolled for up to 36 slices");
    steps4(std::integral_constant<int, 0>{});
    steps4(std::integral_constant<int, 4>{});
    steps4(std::integral_constant<int, 8>{});
    steps4(std::integral_constant<int, 12>{});
    steps4(std::integral_constant<int, 16>{});
    steps4(std::integral_constant<int, 20>{});
    steps4(std::integral_constant<int, 24>{});
    steps4(std::integral_constant<int, 28>{});
    steps4(std::integral_constant<int, 32>{});
    // ---- epilogue (buffer stores: rows >= M fall outside num_records, columns >= N start at 3 GiB)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)((size_t)M * N * 4), 0x00020000);
    const int col = n0 + wn + (lane & 31);
    const bool colok = col < N;
    const float b = (bias && colok) ? bias[col] : 0.f;
    const float rsc = (Split<SP>::F16 && colok) ? Wsc[col] : 1.f;
    const int row0 = m0 + wm + 4 * (lane >> 5);
    const unsigned base = colok ? (unsigned)(row0 * N + col) * 4u : 0xC0000000u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = Split<SP>::F16 ? __builtin_fmaf(acc[r], rsc, b) : acc[r] + b;
        if (RELU) v = v < 0.f ? 0.f : v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                              base + (unsigned)(((r & 3) + 8 * (r >> 2)) * N) * 4u, 0, tfm::kStoreAux);
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
// Tensors whose byte offsets stay below the buffer resources' out-of-range marker (3 GiB) take the buffer-store epilogue; anything
// larger the plain stores (split_gemm_body<..., BUFST = false>).
inline bool fits_bufstore(long long M, int N) { return (M + 256) * N * 4 < 0xC0000000LL; }

// the weight of a call: its 16-bit pieces (hi, mid [, lo] bf16; or, fp16 scheme: hi, lo) [+ the output channels' powers of two]
struct Weight {
    const unsigned short *p[3];
    const float *scale;
};

template <int SP, int BM, int BN, bool PREFETCH>
int launch_gemm(const float *x, const Weight &w, const float *bias, const float *res, float *y, int M, int K, int N, int relu, hipStream_t s)
{
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
    auto go = [&](auto bufst) {
        constexpr bool B = decltype(bufst)::value;
        if (res) {
            if (relu) hipLaunchKernelGGL((split_gemm_res_kernel<SP, BM, BN, true, PREFETCH, B>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, res, y, M, K, N);
            else hipLaunchKernelGGL((split_gemm_res_kernel<SP, BM, BN, false, PREFETCH, B>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, res, y, M, K, N);
        } else {
            if (relu) hipLaunchKernelGGL((split_gemm_kernel<SP, BM, BN, true, PREFETCH, B>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N);
            else hipLaunchKernelGGL((split_gemm_kernel<SP, BM, BN, false, PREFETCH, B>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N);
        }
    };
    if (fits_bufstore(M, N)) go(std::true_type{});
    else go(std::false_type{});
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

template <int SP, int BM, int BN, bool PREFETCH>
int launch_add(const float *x, const float *x2, const Weight &w, const float *bias, float *y, int M, int K, int N, hipStream_t s)
{
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
    if (fits_bufstore(M, N))
        hipLaunchKernelGGL((split_gemm_add_kernel<SP, BM, BN, PREFETCH, true>), grid, dim3(THREADS), 0, s, x, x2, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N);
    else
        hipLaunchKernelGGL((split_gemm_add_kernel<SP, BM, BN, PREFETCH, false>), grid, dim3(THREADS), 0, s, x, x2, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N);
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

// The weight arguments of the C ABI -> scheme (split_product.h): (w_hi, w_mid, w_lo) -> 3 (six bf16 terms), (w_hi, w_mid, NULL,
// w_scale) -> 16 (fp16 pieces hi, lo + the channels' factors); 0: a missing piece, -1: misaligned (16 bytes), a third piece
// beside w_scale, or neither (two bf16 pieces alone were the three-term product of rounds 2-4, removed in round 5)
inline int weight_scheme(const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, Weight &w)
{
    if (!w_hi || !w_mid) return 0;
    if (w_scale && w_lo) return -1;
    if ((reinterpret_cast<uintptr_t>(w_hi) | reinterpret_cast<uintptr_t>(w_mid) | reinterpret_cast<uintptr_t>(w_lo)) & 15) return -1;
    w.p[0] = static_cast<const unsigned short *>(w_hi);
    w.p[1] = static_cast<const unsigned short *>(w_mid);
    w.p[2] = static_cast<const unsigned short *>(w_lo);
    w.scale = w_scale;
    if (!w_scale && !w_lo) return -1;   // (hi, mid) alone was the three-term bf16 product: removed in round 5
    return w_scale ? 16 : 3;
}

// f(integral_constant<int, SP>) for the run-time scheme sp
template <class F>
int with_scheme(int sp, F &&f)
{
    switch (sp) {
    case 3: return f(std::integral_constant<int, 3>{});
    default: return f(std::integral_constant<int, 16>{});
    }
}

// Block shape per call shape (measured in profiles/r02_split_gemm_variants.txt, r03_optin_linear_bufstore.txt):
//   <= 4096 rows (the decoder)      the ring-of-8-slices kernel (64 x 64) when K / 32 is one of its unrolled trip counts and there
//                                   is no residual, else 64 x 64 with register prefetch
//   K >= 512 and N <= 256           128 x 64, prefetch
//   256 < N < 512                   64 x 128, no prefetch
//   else                            64 x 128, prefetch
template <int SP>
int linear_split_sp(const float *x, const Weight &w, const float *bias, const float *res, float *y, int M, int K, int N, int relu,
                    hipStream_t s)
{
    if (M <= 4096) {
        if (!res && fits_bufstore(M, N)) {
            const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
            if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
            const int slices = K / BK;
#define TF_DEEP(SS)                                                                                                                               \
    if (slices == SS) {                                                                                                                           \
        if (relu)                                                                                                                                 \
            hipLaunchKernelGGL((split_gemm_deep_kernel<SP, true, SS>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N);  \
        else                                                                                                                                      \
            hipLaunchKernelGGL((split_gemm_deep_kernel<SP, false, SS>), grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, y, M, K, N); \
        return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;                                                                 \
    }
            TF_DEEP(8) TF_DEEP(9) TF_DEEP(32) TF_DEEP(36)
#undef TF_DEEP
        }
        return launch_gemm<SP, 64, 64, true>(x, w, bias, res, y, M, K, N, relu, s);
    }
    if (K >= 512 && N <= 256) return launch_gemm<SP, 128, 64, true>(x, w, bias, res, y, M, K, N, relu, s);
    if (N > 256 && N < 512) return launch_gemm<SP, 64, 128, false>(x, w, bias, res, y, M, K, N, relu, s);
    return launch_gemm<SP, 64, 128, true>(x, w, bias, res, y, M, K, N, relu, s);
}

int linear_split_impl(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                      const float *res, float *y, int64_t M, int K, int N, int relu, void *stream)
{
    if (!x || !y) return TF_MSDA_ERR_NULL_POINTER;
    Weight w;
    const int sp = weight_scheme(w_hi, w_mid, w_lo, w_scale, w);
    if (sp == 0) return TF_MSDA_ERR_NULL_POINTER;
    if (sp < 0 || M <= 0 || K <= 0 || N <= 0 || (K % BK) != 0 || M > 0x7fffffffLL || (reinterpret_cast<uintptr_t>(x) & 15))
        return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return with_scheme(sp, [&](auto spc) { return linear_split_sp<decltype(spc)::value>(x, w, bias, res, y, (int)M, K, N, relu, s); });
}

template <int SP>
int conv_launch_sp(const float *x, const Weight &w, const float *bias, float *out, const Conv3Args &ca, long long M, unsigned gz, int relu,
                   hipStream_t s)
{
    auto launch = [&](auto kern, unsigned bn) {
        const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((ca.cout + bn - 1) / bn), gz);
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), 0, s, x, w.p[0], w.p[1], w.p[2], w.scale, bias, out, ca);
    };
    if (ca.cout >= 128) {   // output tile 64 x 128 for the wide layers
        relu ? launch(split_conv3_kernel<SP, 64, 128, true>, 128) : launch(split_conv3_kernel<SP, 64, 128, false>, 128);
    } else {
        relu ? launch(split_conv3_kernel<SP, 64, 64, true>, 64) : launch(split_conv3_kernel<SP, 64, 64, false>, 64);
    }
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}

int conv_split_impl(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale, const float *bias,
                    float *y, int nimg, int hin, int win, int cin, int cout, int stride, int ks, int relu, void *stream, int ksplit = 1,
                    float *workspace = nullptr)
{
    if (!x || !y) return TF_MSDA_ERR_NULL_POINTER;
    Weight w;
    const int sp = weight_scheme(w_hi, w_mid, w_lo, w_scale, w);
    if (sp == 0) return TF_MSDA_ERR_NULL_POINTER;
    if (sp < 0 || nimg <= 0 || hin <= 0 || win <= 0 || cin <= 0 || cout <= 0 || (cin % BK) != 0 || (stride != 1 && stride != 2) ||
        (reinterpret_cast<uintptr_t>(x) & 15))
        return TF_MSDA_ERR_BAD_DIMS;
    const int pad = ks == 3 ? 1 : 0;
    Conv3Args ca{nimg, hin, win, cin, (hin + 2 * pad - ks) / stride + 1, (win + 2 * pad - ks) / stride + 1, cout, stride, ks, pad, 0};
    const long long M = (long long)nimg * ca.hout * ca.wout;
    // every byte offset of the input, the weight pieces and the output below the out-of-range marker of the buffer resources
    if (M <= 0 || (M + 256) * cout * 4 >= 0xC0000000LL || (long long)nimg * hin * win * cin * 4 >= 0xC0000000LL ||
        (long long)cout * ks * ks * cin * 2 >= 0xC0000000LL)
        return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // split-K: z-slices of the K loop write partial sums to the workspace [ksplit][M][cout], a second pass adds them in order
    const int slices = ks * ks * cin / BK;
    unsigned gz = 1;
    float *out = y;
    const float *kbias = bias;
    int krelu = relu;
    if (ksplit > 1) {
        if ((cout & 3) || (reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15)
            return TF_MSDA_ERR_BAD_DIMS;
        ca.kslices = (slices + ksplit - 1) / ksplit;
        gz = (unsigned)((slices + ca.kslices - 1) / ca.kslices);   // every z has at least one slice
        out = workspace;
        kbias = nullptr;
        krelu = 0;
    }
    const int rc = with_scheme(sp, [&](auto spc) { return conv_launch_sp<decltype(spc)::value>(x, w, kbias, out, ca, M, gz, krelu, s); });
    if (rc != TF_MSDA_OK) return rc;
    if (ksplit > 1) {
        const long long mn4 = M * cout / 4;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, s, workspace, bias, y, mn4,
                           cout / 4, (int)gz, relu);
        if (hipGetLastError() != hipSuccess) return TF_MSDA_ERR_LAUNCH;
    }
    return TF_MSDA_OK;
}

}  // namespace

extern "C" int tf_linear_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                   const float *bias, float *y, int64_t M, int K, int N, int relu, void *stream)
{
    return linear_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, nullptr, y, M, K, N, relu, stream);
}

extern "C" int tf_linear_split_res_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                       const float *bias, const float *residual, float *y, int64_t M, int K, int N, int relu, void *stream)
{
    if (!residual) return TF_MSDA_ERR_NULL_POINTER;
    return linear_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, residual, y, M, K, N, relu, stream);
}

extern "C" int tf_linear_split_add_f32(const float *x, const float *x2, const void *w_hi, const void *w_mid, const void *w_lo,
                                       const float *w_scale, const float *bias, float *y, int64_t M, int K, int N, void *stream)
{
    if (!x || !x2 || !y) return TF_MSDA_ERR_NULL_POINTER;
    Weight w;
    const int sp = weight_scheme(w_hi, w_mid, w_lo, w_scale, w);
    if (sp == 0) return TF_MSDA_ERR_NULL_POINTER;
    if (sp < 0 || M <= 0 || K <= 0 || N <= 0 || (K % BK) != 0 || M > 0x7fffffffLL ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x2)) & 15))
        return TF_MSDA_ERR_BAD_DIMS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return with_scheme(sp, [&](auto spc) {
        constexpr int SP = decltype(spc)::value;
        // block shapes as linear_split_sp (no deep-prefetch form of the add kernel)
        if (M <= 4096) return launch_add<SP, 64, 64, true>(x, x2, w, bias, y, (int)M, K, N, s);
        if (N > 256 && N < 512) return launch_add<SP, 64, 128, false>(x, x2, w, bias, y, (int)M, K, N, s);
        return launch_add<SP, 64, 128, true>(x, x2, w, bias, y, (int)M, K, N, s);
    });
}

extern "C" int tf_conv3x3_splitk_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                     const float *bias, float *y, float *workspace, int ksplit, int nimg, int hin, int win, int cin,
                                     int cout, int stride, int relu, void *stream)
{
    if (ksplit > 1 && !workspace) return TF_MSDA_ERR_NULL_POINTER;
    if (ksplit < 1 || ksplit > 64) return TF_MSDA_ERR_BAD_DIMS;
    return conv_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, y, nimg, hin, win, cin, cout, stride, 3, relu, stream, ksplit, workspace);
}

extern "C" int tf_conv1x1_splitk_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                     const float *bias, float *y, float *workspace, int ksplit, int nimg, int hin, int win, int cin,
                                     int cout, int stride, int relu, void *stream)
{
    if (ksplit > 1 && !workspace) return TF_MSDA_ERR_NULL_POINTER;
    if (ksplit < 1 || ksplit > 64) return TF_MSDA_ERR_BAD_DIMS;
    return conv_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, y, nimg, hin, win, cin, cout, stride, 1, relu, stream, ksplit, workspace);
}

extern "C" int tf_conv3x3_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                    const float *bias, float *y, int nimg, int hin, int win, int cin, int cout, int stride, int relu,
                                    void *stream)
{
    return conv_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, y, nimg, hin, win, cin, cout, stride, 3, relu, stream);
}

extern "C" int tf_conv1x1_strided_split_f32(const float *x, const void *w_hi, const void *w_mid, const void *w_lo, const float *w_scale,
                                            const float *bias, float *y, int nimg, int hin, int win, int cin, int cout, int stride,
                                            int relu, void *stream)
{
    return conv_split_impl(x, w_hi, w_mid, w_lo, w_scale, bias, y, nimg, hin, win, cin, cout, stride, 1, relu, stream);
}
