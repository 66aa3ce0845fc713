// trackformer_amd/csrc/stem_conv.hip
//
// tf_stem_conv7x7_f32 (include/tf_fused.h): the backbone's first convolution -- 7 x 7, stride 2, padding 3, 3 -> 64 channels
// (reference: models/backbone.py:93-104 -> torchvision resnet50.conv1, with the FrozenBatchNorm2d scale of :45-55 folded into
// the weight by the caller) -- as an implicit GEMM on the matrix cores with the same bf16 split product as the linears
// (split_product.h: fp16 pieces, six bf16 terms, or three in the fast mode; fp32 accumulation).  With it and the bottleneck routes of
// linear_split.hip no convolution of the backbone is left in MIOpen.
//
//   * GEMM view: M = output pixels, N = 64, K = 3 x 7 x 8 = 168 (-> 176 = 11 k-steps of 16): k = (c * 7 + ky) * 8 + kx with the
//     7 taps of a kernel row padded to 8 (zero weight), so that a lane's 8 consecutive k of an MFMA fragment are 8 consecutive
//     input pixels of ONE image row: x[c][2 oy + ky - 3][2 ox - 3 .. 2 ox + 4].
//   * A block computes 4 output rows x 128 output columns as four 4 x 32 tiles; wave w owns output row w of the tile (32
//     pixels x 64 channels: two accumulator tiles).  The weight (64 x 176, packed by tf_linear_pack_weight_f32: 44 fragments
//     of 16 bytes per lane) is loaded ONCE per block and stays in registers.
//   * Per tile the 13 x 70 input patch of the three planes (zero outside the image) is staged in LDS as fp32 (11 KB, two
//     buffers: the next tile's 11 loads per thread are in flight during the current tile's MFMAs); the fragments are read
//     from it with 8-byte reads (lane m starts at pixel 2 m: conflict-free), split into bf16 hi / mid in registers (16
//     vector instructions per 6 MFMAs).
//   * The weight fragment is the A operand (transposed accumulators, see linear_stream.hip): a lane owns 4 consecutive
//     channels of one pixel and stores 16 bytes at a time into the channels_last output; pixels outside the image fall
//     outside the buffer resource.
//
// Per frame (800 x 1333): 2084 tiles x 264 MFMAs = 7 us of matrix time over the chip, 27 MB of patch loads, 68 MB of output:
// HBM-bound at ~12 us against ~78 us for the library kernel it replaces.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "msda_common.h"
#include "split_product.h"
#include "tf_fused.h"
#include "tf_msda.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kCout = 64, kKQ = 11;                  // 176 / 16 k-steps
constexpr int kTH = 4, kTW = 32, kTilesPerBlock = 4; // output rows per block (one per wave), columns per tile, tiles per block
constexpr int kPR = 2 * kTH + 5, kPW = 72;           // patch rows (13), floats per patch row (70 used)
constexpr int kPatch = 3 * kPR * kPW;                // 2808 floats

template <int SP, bool RELU>
__global__ void __launch_bounds__(256)
stem_conv7x7_kernel(const float *__restrict__ X, const u32x4 *__restrict__ Wp, const float *__restrict__ bias, float *Y, int H, int W,
                    int Ho, int Wo)
{
    __shared__ __attribute__((aligned(16))) float sP[2][kPatch];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int oy0 = blockIdx.y * kTH, n = blockIdx.z;
    const float *xin = X + (size_t)n * 3 * H * W;

    // ---- the whole weight: fragment (n-tile t, k-step q, part p) at ((t KQ + q) NB + p) 64 + lane (linear_stream.hip)
    constexpr int NA = Split<SP>::NA, NB = Split<SP>::NB;
    u32x4 wf[2][kKQ][NB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < kKQ; ++q)
#pragma unroll
            for (int p = 0; p < NB; ++p) wf[t][q][p] = Wp[((size_t)(t * kKQ + q) * NB + p) * 64 + lane];

    const unsigned ybytes = (unsigned)((size_t)gridDim.z * Ho * Wo * kCout * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bias ? bias : X), 0, bias ? kCout * 4u : 0u, 0x00020000);
    f32x4 bv[2][4], rv[2][4];   // bias [and, fp16 scheme, power of two] of the lane's channels: tile t, group g -> channels 32 t + 8 g + 4 half + 0..3
    // the packed weight is padded to 256 output channels (8 n-tiles); its per-channel factors lie behind the fragments
    const float *const rsc = reinterpret_cast<const float *>(Wp + (size_t)8 * kKQ * NB * 64);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[t][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(32 * t + 8 * g + 4 * half) * 4u, 0, 0));
            rv[t][g] = f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (Split<SP>::F16) rv[t][g] = *reinterpret_cast<const f32x4 *>(rsc + 32 * t + 8 * g + 4 * half);
        }

    const int oy = oy0 + wave;
    // ---- patch staging: rows 2 oy0 - 3 .. + 12, columns 2 ox0 - 3 .. + 71 of the three planes, zeros outside the image.
    // A thread's kPV elements are loaded together (branch-free, clamped address + select), for the NEXT tile while the
    // current one is in the matrix pipes; two LDS buffers, one barrier per tile.
    constexpr int kPV = (kPatch + 255) / 256;   // 11
    const int iy_base = 2 * oy0 - 3;
    // element it of this thread: (plane, patch row, patch column) -- the same for every tile, only the column base moves
    int poff[kPV], pxx[kPV];      // offset of (plane, clamped image row) in x; patch column
    bool prow[kPV];               // the patch row lies inside the image
#pragma unroll
    for (int it = 0; it < kPV; ++it) {
        const int idx = min(tid + it * 256, kPatch - 1);
        const int c = idx / (kPR * kPW), rem = idx - c * (kPR * kPW);
        const int r = rem / kPW;
        pxx[it] = rem - r * kPW;
        const int iy = iy_base + r;
        prow[it] = iy >= 0 && iy < H;
        poff[it] = (c * H + min(max(iy, 0), H - 1)) * W;
    }
    auto load_patch = [&](int ox0, float (&v)[kPV]) {   // issues the loads only (clamped addresses): nothing waits here
        const int ix_base = 2 * ox0 - 3;
#pragma unroll
        for (int it = 0; it < kPV; ++it) v[it] = xin[poff[it] + min(max(ix_base + pxx[it], 0), W - 1)];
    };
    auto store_patch = [&](int ox0, const float (&v)[kPV], float *dst) {   // zeroes what lies outside the image, writes the patch
        const int ix_base = 2 * ox0 - 3;
#pragma unroll
        for (int it = 0; it < kPV; ++it) {
            const int ix = ix_base + pxx[it];
            const float t = (prow[it] && ix >= 0 && ix < W) ? v[it] : 0.f;
            if (tid + it * 256 < kPatch) dst[tid + it * 256] = t;
        }
    };
    const int tile0 = blockIdx.x * kTilesPerBlock;
    float pv[kPV];
    if (tile0 * kTW < Wo) {   // (always true for a launched block; keeps the structure uniform)
        load_patch(tile0 * kTW, pv);
        store_patch(tile0 * kTW, pv, sP[0]);
    }
    __syncthreads();
    for (int tile = 0; tile < kTilesPerBlock; ++tile) {
        const int ox0 = (tile0 + tile) * kTW;
        if (ox0 >= Wo) break;   // uniform over the block
        const bool has_next = tile + 1 < kTilesPerBlock && ox0 + kTW < Wo;
        if (has_next) load_patch(ox0 + kTW, pv);   // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        const float *cur = sP[tile & 1];

        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int q = 0; q < kKQ; ++q) {
            // this lane's 8 k of the step: chunk j = 2 q + half = (plane c, kernel row ky); j = 21 is the zero padding of K
            constexpr int kLast = 20;
            const int j0 = 2 * q < kLast ? 2 * q : kLast, j1 = 2 * q + 1 < kLast + 1 ? 2 * q + 1 : kLast;
            const int o0 = ((j0 / 7) * kPR + 2 * wave + (j0 % 7)) * kPW, o1 = ((j1 / 7) * kPR + 2 * wave + (j1 % 7)) * kPW;
            const float *src = cur + (half ? o1 : o0) + 2 * m;
            float xv[8];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const f32x2 t2 = *reinterpret_cast<const f32x2 *>(src + 2 * e2);
                xv[2 * e2] = t2.x;
                xv[2 * e2 + 1] = t2.y;
            }
            if (2 * q + 1 > kLast) {   // the last step's upper half is padding: its weights are zero, keep the data finite
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = half ? 0.f : xv[e];
            }
            u32x2 plo[NA], phi[NA];
            split4<SP>(f32x4{xv[0], xv[1], xv[2], xv[3]}, plo);
            split4<SP>(f32x4{xv[4], xv[5], xv[6], xv[7]}, phi);
            u32x4 xp[NA];
#pragma unroll
            for (int p = 0; p < NA; ++p) xp[p] = u32x4{plo[p].x, plo[p].y, phi[p].x, phi[p].y};
            using T = Split<SP>;   // x piece T::A[t] x weight operand T::B[t], smallest terms first
            u32x4 wx[2][T::NBX];
#pragma unroll
            for (int t = 0; t < 2; ++t) expand_weight<SP>(wf[t][q], wx[t]);
#pragma unroll
            for (int tt = 0; tt < T::N; ++tt)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = mfma16<SP>(wx[t][T::B[tt]], xp[T::A[tt]], acc[t]);
        }
        // ---- epilogue: lane -> pixel (oy, ox0 + m); registers 4 g .. 4 g + 3 of tile t -> channels 32 t + 8 g + 4 half + 0..3
        const int ox = ox0 + m;
        const bool ok = oy < Ho && ox < Wo;
        const unsigned pix = (unsigned)(((size_t)n * Ho + oy) * Wo + ox) * (unsigned)(kCout * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = Split<SP>::F16 ? __builtin_fmaf(acc[t][4 * g + e], rv[t][g][e], bv[t][g][e]) : acc[t][4 * g + e] + bv[t][g][e];
                if (RELU) {
                    v.x = v.x < 0.f ? 0.f : v.x;
                    v.y = v.y < 0.f ? 0.f : v.y;
                    v.z = v.z < 0.f ? 0.f : v.z;
                    v.w = v.w < 0.f ? 0.f : v.w;
                }
                const unsigned off = ok ? pix + (unsigned)(32 * t + 8 * g + 4 * half) * 4u : 0xC0000000u;   // outside: dropped
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, tfm::kStoreAux);
            }
        if (has_next) store_patch(ox0 + kTW, pv, sP[(tile + 1) & 1]);   // the buffer of the previous tile: every wave is past it
        __syncthreads();
    }
}

}  // namespace

extern "C" int tf_stem_conv7x7_f32(const float *x, const void *w_packed, const float *bias, float *y, int N, int H, int W, int relu,
                                   int terms, void *stream)
{
    if (!x || !w_packed || !y) return TF_MSDA_ERR_NULL_POINTER;
    const int sp = split_scheme(terms);
    if (N <= 0 || H <= 0 || W <= 0 || N > 65535 || sp == 0) return TF_MSDA_ERR_BAD_DIMS;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((long long)N * Ho * Wo * kCout * 4 >= 0xC0000000LL || (long long)N * 3 * H * W >= (1LL << 31)) return TF_MSDA_ERR_BAD_DIMS;
    if ((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15)
        return TF_MSDA_ERR_BAD_DIMS;
    const dim3 grid((unsigned)((Wo + kTW * kTilesPerBlock - 1) / (kTW * kTilesPerBlock)), (unsigned)((Ho + kTH - 1) / kTH), (unsigned)N);
    if (grid.y > 65535u) return TF_MSDA_ERR_BAD_DIMS;
    const u32x4 *wp = static_cast<const u32x4 *>(w_packed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto go = [&](auto spc) {
        constexpr int SP = decltype(spc)::value;
        if (relu) hipLaunchKernelGGL((stem_conv7x7_kernel<SP, true>), grid, dim3(256), 0, s, x, wp, bias, y, H, W, Ho, Wo);
        else hipLaunchKernelGGL((stem_conv7x7_kernel<SP, false>), grid, dim3(256), 0, s, x, wp, bias, y, H, W, Ho, Wo);
    };
    if (sp == 3) go(std::integral_constant<int, 3>{});
    else go(std::integral_constant<int, 16>{});
    return hipGetLastError() == hipSuccess ? TF_MSDA_OK : TF_MSDA_ERR_LAUNCH;
}
