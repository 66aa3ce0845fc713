// trackformer_amd/csrc/msda_hip.hip
//
// Multi-scale deformable attention (MSDeformAttn) for AMD Instinct MI355X (gfx950 / CDNA4):
// hand-written HIP kernels + the C ABI declared in include/tf_msda.h.  Built with
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC   (see trackformer_amd/build.py)
// into trackformer_amd/lib/libtf_msda.so.  No torch / ATen dependency.
//
// What it replaces in the reference (/root/reference/src/trackformer/models/ops/src/cuda/):
//   ms_deform_attn_cuda.cu:19-86   forward host code  (columns temp + at::sum)   -> one fused kernel
//   ms_deform_attn_cuda.cu:89-168  backward host code (2 kernels per batch chunk) -> one fused kernel
//   ms_deform_im2col_cuda.cuh      im2col / col2im / col2im_coord CUDA kernels
// The arithmetic (pixel mapping, in-range rule, zero padded bilinear taps, the three gradients) is
// the one written out in SURVEY.md Appendix A; the thread mapping, memory staging and reduction
// scheme are designed for 64-wide wavefronts and are unrelated to the reference's.
//
// Kernels in this file (a "pair" is one (batch, query, head) triple: L*P sampling points, one D-float
// output row; the D/4 lanes of a pair each own 4 consecutive channels, so a bilinear tap is one 16-byte
// load per lane and the lanes of a pair read one contiguous row of `value` -- 128 B for D = 32):
//   msda_fwd_rowgather<T,VEC>   any dtype / shape; loc+attn chunk staged in LDS, clamped taps, selects.
//   msda_fwd_f32_buf<P,FUSED>   fp32, D % 4 == 0: buffer loads whose out-of-range offsets give zero
//                               padding in hardware; head-major workgroups (one head per XCD when
//                               M == 8: that head's value rows stay in the XCD's 4 MiB L2).
//   msda_fwd_f32_direct         fp32, D == 32, P == 4 (every TrackFormer config with hidden 256): no
//                               staging prologue, tap arithmetic computed once per pair and shared
//                               inside the wave.  THE DEFAULT for decoder-shaped calls.
//   msda_fwd_f32_direct9        the same for D == 36 (hidden 288): nine lanes per pair.
//   msda_fwd_f32_pquad          encoder shape (Lq == S): persistent workgroups, data-adaptive LDS windows, 4 lanes per
//                               pair (msda_pquad.hip).  THE DEFAULT for encoder-shaped calls.
//   msda_fwd_f32_quad           the one-tile-per-workgroup form of it (msda_fwd_quad.h): taken when pquad declines
//                               (4-d reference points of two-stage models, pquad switched off).
//   msda_bwd_rowgather<T,...>   any dtype; fuses the reference's two backward kernels.
//   msda_bwd_f32_buf<P,ROWATOM> fp32 fast path: buffer loads + buffer atomics (full-row scatter for D == 32).
//   msda_bwd_f32_sorted2        encoder shape, D == 32, P == 4: contributions counting-sorted by destination
//                               row in LDS, one global atomic per row.  THE DEFAULT for encoder backward.
// Removed in round 4 (superseded, no default route reached them; measurements in DESIGN.md section 4.1):
// msda_fwd_f32_win (the first LDS-window kernel, 8 lanes per pair) and msda_bwd_f32_sorted (the first sorted backward).
// Common rules: level geometry comes from the kernel arguments (host-shape entry points) or from the
// reference's device-resident int64 tensor (..._dshapes) -- never a host<->device sync; no kernel
// branches per tap; nothing depends on CUDA-style 32-wide warps.
//
#include <hip/hip_runtime.h>

#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "tf_msda.h"
#include "msda_common.h"
#include "msda_quad_geom.h"

namespace {
using namespace tfm;

constexpr int kThreads = 256;          // 4 wavefronts per workgroup
constexpr int kLdsChunkBudget = 48 * 1024;  // LDS bytes for the loc/attn (and grad) chunk

constexpr int kLevelTableBytes = 3 * TF_MSDA_MAX_LEVELS * (int)sizeof(int);  // 192, multiple of 16

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

thread_local int g_last_hip_error = 0;
thread_local const char *g_last_kernel = "";

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float floor_t(float a) { return __builtin_floorf(a); }
__device__ __forceinline__ double floor_t(double a) { return __builtin_floor(a); }

__device__ __forceinline__ void fill_level_table(int *s_tab, const LevelTable &lt,
                                                 const int64_t *__restrict__ dshapes, int L)
{
    // s_tab: [H[16] | W[16] | start[16]]
    if (threadIdx.x == 0) {
        if (dshapes != nullptr) {
            int acc = 0;
            for (int l = 0; l < L; ++l) {
                const int h = (int)dshapes[2 * l], w = (int)dshapes[2 * l + 1];
                s_tab[l] = h;
                s_tab[TF_MSDA_MAX_LEVELS + l] = w;
                s_tab[2 * TF_MSDA_MAX_LEVELS + l] = acc;
                acc += h * w;
            }
        } else {
            for (int l = 0; l < L; ++l) {
                s_tab[l] = lt.H[l];
                s_tab[TF_MSDA_MAX_LEVELS + l] = lt.W[l];
                s_tab[2 * TF_MSDA_MAX_LEVELS + l] = lt.start[l];
            }
        }
    }
}

// One sampling point: pixel coordinates, clamped tap offsets (in units of pixels within the level),
// bilinear fractions and tap validity.  Follows SURVEY.md Appendix A / cuh:227-229, :24-67.
template <typename T>
struct Tap {
    T fx, fy, gx, gy;       // lw, lh, hw, hh of the reference
    int o1, o2, o3, o4;     // clamped pixel offsets y*W + x of the four taps
    bool k1, k2, k3, k4;    // tap contributes (sample in range AND corner inside the level)
};

template <typename T>
__device__ __forceinline__ Tap<T> make_tap(T lx, T ly, int H, int W)
{
    Tap<T> t;
    // loc*size - 0.5 with a SINGLE rounding (one fma).  The reference's fp32 instantiation rounds twice
    // (cuh:227-228: `loc * spatial` is a float product, then `- 0.5` in double, narrowed to float), its
    // pure-PyTorch path (func.py:40-49, grid_sample at 2*loc-1) rounds three times: the three agree to 1 ulp
    // of the pixel coordinate, i.e. they can differ only for samples within 1 ulp of an integer pixel
    // boundary, where the bilinear weight of the disputed tap is <= 1 ulp as well (outputs differ by ~1e-7;
    // the in-range test at exactly -1 / size can flip, a measure-zero set the golden tests exclude).
    // The oracle (oracle/msda_ref.c) uses the same single rounding; fp64 is exact in all three.
    const T xr = fma_t(lx, (T)W, (T)-0.5);
    const T yr = fma_t(ly, (T)H, (T)-0.5);
    const bool in = (yr > (T)-1) && (xr > (T)-1) && (yr < (T)H) && (xr < (T)W);
    // Out-of-range samples may carry huge / non-finite coordinates: neutralise them so that every
    // derived quantity stays finite (their taps are all invalid anyway).
    const T x = in ? xr : (T)0, y = in ? yr : (T)0;
    const T xf = floor_t(x), yf = floor_t(y);
    t.fx = x - xf;
    t.fy = y - yf;
    t.gx = (T)1 - t.fx;
    t.gy = (T)1 - t.fy;
    const int x0 = (int)xf, y0 = (int)yf;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool kx0 = in && (x0 >= 0), kx1 = in && (x1 <= W - 1);
    const bool ky0 = in && (y0 >= 0), ky1 = in && (y1 <= H - 1);
    const int cx0 = max(x0, 0), cx1 = min(x1, W - 1);
    const int cy0 = max(y0, 0), cy1 = min(y1, H - 1);
    t.o1 = cy0 * W + cx0;
    t.o2 = cy0 * W + cx1;
    t.o3 = cy1 * W + cx0;
    t.o4 = cy1 * W + cx1;
    t.k1 = ky0 && kx0;
    t.k2 = ky0 && kx1;
    t.k3 = ky1 && kx0;
    t.k4 = ky1 && kx1;
    return t;
}

// XCD-aware block order.  The dispatcher places workgroup i on XCD i % 8 (observed behaviour, used
// for speed only).  Remapping i -> (i % 8) * ceil(n/8) + i / 8 hands each XCD one contiguous eighth
// of the pair range, i.e. (for encoder self-attention, where consecutive queries are neighbouring
// pixels) one band of rows per level, whose value rows then fit that XCD's private 4 MiB L2.
// The launch grid is padded to a multiple of 8; returns -1 for the padding workgroups.
constexpr int kXcds = 8;
__device__ __forceinline__ long long logical_block(long long nblocks)
{
    const long long per = (nblocks + kXcds - 1) / kXcds;
    const long long lb = (long long)(blockIdx.x % kXcds) * per + blockIdx.x / kXcds;
    return (blockIdx.x / kXcds < per && lb < nblocks) ? lb : -1;
}

// Cooperative, coalesced copy of `n` elements global -> LDS (or LDS -> global).
template <typename T>
__device__ __forceinline__ void copy_in(T *__restrict__ dst_lds, const T *__restrict__ src, int n)
{
    for (int i = threadIdx.x; i < n; i += kThreads) dst_lds[i] = src[i];
}
template <typename T>
__device__ __forceinline__ void copy_out(T *__restrict__ dst, const T *__restrict__ src_lds, int n)
{
    for (int i = threadIdx.x; i < n; i += kThreads) dst[i] = src_lds[i];
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(kThreads)
msda_fwd_rowgather(const T *__restrict__ value, const T *__restrict__ loc,
                   const T *__restrict__ attn, T *__restrict__ out, const LevelTable lt,
                   const int64_t *__restrict__ dshapes, int S, int M, int D, int L, int Lq, int P,
                   long long total_pairs, int ppb, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    const int LP = L * P;
    T *s_loc = reinterpret_cast<T *>(smem + kLevelTableBytes);
    T *s_attn = s_loc + (size_t)ppb * LP * 2;

    const long long lblk = logical_block((total_pairs + ppb - 1) / ppb);
    if (lblk < 0) return;
    const long long pair0 = lblk * ppb;
    const int npairs = (int)min((long long)ppb, total_pairs - pair0);

    fill_level_table(s_tab, lt, dshapes, L);
    copy_in(s_loc, loc + pair0 * LP * 2, npairs * LP * 2);
    copy_in(s_attn, attn + pair0 * LP, npairs * LP);
    __syncthreads();

    const int pl = threadIdx.x / DV;
    const int dv = threadIdx.x - pl * DV;
    if (pl >= npairs) return;

    const long long pair = pair0 + pl;  // (b*Lq + q)*M + m
    const int m = (int)(pair % M);
    const int b = (int)((pair / M) / Lq);
    const long long pix = (long long)M * D;  // elements between neighbouring pixels
    const T *vb = value + (long long)b * S * pix + (long long)m * D + dv * VEC;
    const T *sl = s_loc + (size_t)pl * LP * 2;
    const T *sa = s_attn + (size_t)pl * LP;

    using P4 = Pack<T, VEC>;
    T acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = (T)0;

    for (int l = 0; l < L; ++l) {
        const int H = s_tab[l], W = s_tab[TF_MSDA_MAX_LEVELS + l];
        const T *vl = vb + (long long)s_tab[2 * TF_MSDA_MAX_LEVELS + l] * pix;
#pragma unroll 4
        for (int p = 0; p < P; ++p) {
            const int s = l * P + p;
            const T lx = sl[2 * s], ly = sl[2 * s + 1];
            const T a = sa[s];
            const Tap<T> t = make_tap(lx, ly, H, W);
            const P4 v1 = *reinterpret_cast<const P4 *>(vl + (long long)t.o1 * pix);
            const P4 v2 = *reinterpret_cast<const P4 *>(vl + (long long)t.o2 * pix);
            const P4 v3 = *reinterpret_cast<const P4 *>(vl + (long long)t.o3 * pix);
            const P4 v4 = *reinterpret_cast<const P4 *>(vl + (long long)t.o4 * pix);
            const T w1 = t.gy * t.gx, w2 = t.gy * t.fx, w3 = t.fy * t.gx, w4 = t.fy * t.fx;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const T a1 = t.k1 ? v1.v[c] : (T)0;
                const T a2 = t.k2 ? v2.v[c] : (T)0;
                const T a3 = t.k3 ? v3.v[c] : (T)0;
                const T a4 = t.k4 ? v4.v[c] : (T)0;
                const T smp = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;
                acc[c] = fma_t(smp, a, acc[c]);
            }
        }
    }
    P4 o;
#pragma unroll
    for (int c = 0; c < VEC; ++c) o.v[c] = acc[c];
    *reinterpret_cast<P4 *>(out + pair * D + dv * VEC) = o;
}

// ---------------------------------------------------------------------------------------------
// forward, fp32 fast path: buffer loads with hardware bounds checking
// ---------------------------------------------------------------------------------------------
// Same pair/lane mapping as msda_fwd_rowgather, but every tap is a `buffer_load_dwordx4` through one
// kernel-uniform buffer descriptor that spans the whole value tensor:
//   * the per-lane address is a 32-bit byte offset (one v_mad_u32_u24 per tap instead of 64-bit
//     multiply-adds);
//   * an invalid tap (outside the level, or an out-of-range sample) gets an offset beyond
//     num_records, for which the hardware returns 0 -- zero padding without selects, without
//     clamping and without divergent code, and a 0*Inf can never be formed;
//   * P is a template parameter so the 4*P loads of a level are issued back to back before the first
//     use (the compiler cannot sink them into conditionals: there are none).
// (u32x4_t / f32x4_t, kOobOffset / kOobBase: msda_common.h)

// Optional fused prologue (FUSED = true): instead of reading finished sampling locations and softmaxed
// attention weights, the kernel takes the raw outputs of the query projections and the reference
// points and performs the arithmetic of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:69-85) while
// it stages the block's chunk in LDS:
//     attn = softmax_{l,p}(logits[q, m, :])
//     loc  = ref[q, l, :2] + off[q, m, l, p, :] / (H_l, W_l)                      (ref_dim == 2; the divisor
//                                                  pairs x with H_l and y with W_l exactly as the reference)
//     loc  = ref[q, l, :2] + off[q, m, l, p, :] / P * ref[q, l, 2:] * 0.5          (ref_dim == 4)
// which removes the softmax, division, multiply and add kernels (and their ~100 MB of traffic per
// encoder layer) that the reference runs between the projection GEMM and the operator.
// (struct FusedArgs: msda_common.h)

template <int PT, bool FUSED>
__global__ void __launch_bounds__(kThreads)
msda_fwd_f32_buf(const float *__restrict__ value, unsigned value_bytes,
                 const float *__restrict__ loc, const float *__restrict__ attn,
                 float *__restrict__ out, const LevelTable lt, const int64_t *__restrict__ dshapes,
                 int S, int M, int D, int L, int Lq, long long total_pairs, int ppb, int DV,
                 const FusedArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    const int LP = L * PT;
    float *s_loc = reinterpret_cast<float *>(smem + kLevelTableBytes);
    float *s_attn = s_loc + (size_t)ppb * LP * 2;

    // Block -> pairs.  head_major == 0: ppb consecutive pairs (all heads of a few queries), blocks in
    // XCD-aware order.  head_major == 1: ppb consecutive QUERIES of ONE head, head = blockIdx % M: with
    // M == 8 every XCD works on a single head, whose value rows (S*D*4 = 2.8 MB) fit its private L2.
    long long pair0 = 0, q0 = 0;
    int npairs = 0, head = 0;
    const long long nlq = total_pairs / M;           // N * Lq
    if (fa.head_major) {
        head = blockIdx.x % M;
        q0 = (long long)(blockIdx.x / M) * ppb;
        if (q0 >= nlq) return;
        npairs = (int)min((long long)ppb, nlq - q0);
    } else {
        const long long lblk = logical_block((total_pairs + ppb - 1) / ppb);
        if (lblk < 0) return;
        pair0 = lblk * ppb;
        npairs = (int)min((long long)ppb, total_pairs - pair0);
    }
    auto pair_of = [&](int pp) -> long long {
        return fa.head_major ? (q0 + pp) * M + head : pair0 + pp;
    };

    fill_level_table(s_tab, lt, dshapes, L);
    if constexpr (!FUSED) {
        if (fa.head_major) {
            const int row = LP * 2;
            for (int i = threadIdx.x; i < npairs * row; i += kThreads) {
                const int pp = i / row, j = i - pp * row;
                s_loc[i] = loc[pair_of(pp) * row + j];
            }
            for (int i = threadIdx.x; i < npairs * LP; i += kThreads) {
                const int pp = i / LP, j = i - pp * LP;
                s_attn[i] = attn[pair_of(pp) * LP + j];
            }
        } else {
            copy_in(s_loc, loc + pair0 * LP * 2, npairs * LP * 2);
            copy_in(s_attn, attn + pair0 * LP, npairs * LP);
        }
        __syncthreads();
    } else {
        __syncthreads();   // level table visible
        for (int sidx = threadIdx.x; sidx < npairs * LP; sidx += kThreads) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
            const int pp = sidx / LP, lp = sidx - pp * LP;
            const int l = lp / PT;
            const long long pr = pair_of(pp);
            const long long bq = pr / M;
            const int mm = (int)(pr - bq * M);
            const float *row = fa.qproj + bq * fa.ld;
            const float2 off = *reinterpret_cast<const float2 *>(row + fa.off_col + (mm * LP + lp) * 2);
            const float *rp = fa.ref + (bq * L + l) * fa.ref_dim;
            float x, y;
            if (fa.ref_dim == 2) {
                x = rp[0] + off.x / (float)s_tab[l];                           // x / H_l  (as written)
                y = rp[1] + off.y / (float)s_tab[TF_MSDA_MAX_LEVELS + l];      // y / W_l
            } else {
                x = rp[0] + off.x / (float)PT * rp[2] * 0.5f;
                y = rp[1] + off.y / (float)PT * rp[3] * 0.5f;
            }
            s_loc[2 * sidx] = x;
            s_loc[2 * sidx + 1] = y;
            s_attn[sidx] = row[fa.logit_col + mm * LP + lp];
        }
        __syncthreads();
        for (int pp = threadIdx.x; pp < npairs; pp += kThreads) {   // softmax over the L*P logits
            float *a = s_attn + (size_t)pp * LP;
            float mx = a[0];
            for (int i = 1; i < LP; ++i) mx = fmaxf(mx, a[i]);
            float sum = 0.f;
            for (int i = 0; i < LP; ++i) {
                const float e = __expf(a[i] - mx);
                a[i] = e;
                sum += e;
            }
            for (int i = 0; i < LP; ++i) a[i] = a[i] / sum;
        }
        __syncthreads();
    }

    const int pl = threadIdx.x / DV;
    const int dv = threadIdx.x - pl * DV;
    if (pl >= npairs) return;

    const long long pair = pair_of(pl);  // (b*Lq + q)*M + m
    const int m = (int)(pair % M);
    const int b = (int)((pair / M) / Lq);
    const unsigned rowbytes = (unsigned)(M * D) * 4u;                                // < 2^24
    const unsigned lane_base = (unsigned)((((long long)b * S * M + m) * D + dv * 4) * 4);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, value_bytes, 0x00020000);
    const float2 *sl = reinterpret_cast<const float2 *>(s_loc + (size_t)pl * LP * 2);
    const float *sa = s_attn + (size_t)pl * LP;

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) {
        const int H = s_tab[l], W = s_tab[TF_MSDA_MAX_LEVELS + l];
        const unsigned lvl_base = lane_base + (unsigned)s_tab[2 * TF_MSDA_MAX_LEVELS + l] * rowbytes;
        const float Wf = (float)W, Hf = (float)H;
        u32x4_t v[PT][4];
        float w[PT][4];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float2 xy = sl[l * PT + p];
            const float a = sa[l * PT + p];
            const float xr = __builtin_fmaf(xy.x, Wf, -0.5f);   // cuh:227-228, single rounding
            const float yr = __builtin_fmaf(xy.y, Hf, -0.5f);
            const bool in = (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);  // cuh:229
            const float x = in ? xr : 0.f, y = in ? yr : 0.f;
            const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
            const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
            const int x0 = (int)xf, y0 = (int)yf;
            const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
            const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
            const int r0 = y0 * W + x0;        // may be "negative": only used when the tap is valid
            const int r1 = r0 + W;
            const unsigned o1 = (ky0 && kx0) ? lvl_base + (unsigned)r0 * rowbytes : kOobOffset;
            const unsigned o2 = (ky0 && kx1) ? lvl_base + (unsigned)(r0 + 1) * rowbytes : kOobOffset;
            const unsigned o3 = (ky1 && kx0) ? lvl_base + (unsigned)r1 * rowbytes : kOobOffset;
            const unsigned o4 = (ky1 && kx1) ? lvl_base + (unsigned)(r1 + 1) * rowbytes : kOobOffset;
            v[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o1, 0, 0);
            v[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o2, 0, 0);
            v[p][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o3, 0, 0);
            v[p][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o4, 0, 0);
            w[p][0] = gy * gx * a;
            w[p][1] = gy * fx * a;
            w[p][2] = fy * gx * a;
            w[p][3] = fy * fx * a;
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc += __builtin_bit_cast(f32x4_t, v[p][t]) * w[p][t];
        }
    }
    *reinterpret_cast<f32x4_t *>(out + pair * D + dv * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// forward, fp32, D == 32, P == 4, L <= 8: no staging prologue, tap arithmetic shared inside the wave
// ---------------------------------------------------------------------------------------------
// Ablations of msda_fwd_f32_buf at the cfg-2 encoder shape (64 us) showed where its time went: ~22 us
// in the prologue (loc/attn chunk -> LDS, barriers: every workgroup first waits a full memory round
// trip) plus the store, and ~31 us of vector-ALU work that all 8 lanes of a pair repeat (the tap
// arithmetic) -- the kernel was instruction-issue bound, not memory bound.  This kernel
//   * has every lane fetch just the sampling points it is responsible for straight into registers
//     at kernel entry (no LDS round trip, no barrier behind a memory access),
//   * computes the tap arithmetic of a point ONCE per pair: per pair of levels, lanes 0-3 of the
//     8-lane group take the 4 points of level l, lanes 4-7 those of level l+1; each lane publishes
//     its 4 byte offsets + 4 weights in a per-wave LDS exchange buffer and the group reads them back
//     with two broadcast ds_read_b128 per point (LDS operations of one wave execute in order, so a
//     wave-scope fence is the only synchronisation).  The vector-ALU work per point drops from ~36 to
//     ~12 instructions per wave; an all-DPP exchange (row_newbcast + bank masks, 16 v_mov_dpp per
//     point) measured the same end-to-end time once the kernel had become vector-memory bound,
//   * optionally (FUSED) performs MSDeformAttn.forward's softmax and sampling-location arithmetic
//     on the fly (softmax statistics by xor butterflies over the 8 lanes).
// What bounds it now is the vector-memory path itself: the 64 row gathers per pair move 1.46 GB per
// launch through the texture-addresser / L1 at <= 64 B/clk/CU (TA_BUSY ~80 % of the kernel's cycles).
// (struct DirectArgs: msda_common.h)

template <int LPAIRS, bool FUSED>   // LPAIRS = ceil(L / 2)
__global__ void __launch_bounds__(kThreads, 4)
msda_fwd_f32_direct(const DirectArgs da, const LevelTable lt, const int64_t *__restrict__ dshapes)
{
    constexpr int PT = 4, D = 32, DV = 8;
    __shared__ int s_tab[3 * TF_MSDA_MAX_LEVELS];
    // per-wave exchange buffers, one 16-byte slot of offsets and one of weights per lane; a pad slot
    // after every 8 lanes keeps the 8 groups of a wave on disjoint banks when they all read slot k
    __shared__ u32x4_t s_xo[(kThreads / 64) * 72];
    __shared__ f32x4_t s_xw[(kThreads / 64) * 72];
    const int L = da.L, M = da.M, LP = L * PT;
    fill_level_table(s_tab, lt, dshapes, L);

    const int head = blockIdx.x % M;                                   // one head per XCD when M == 8
    const long long bq = (long long)(blockIdx.x / M) * (kThreads / DV) + threadIdx.x / DV;
    const int dv = threadIdx.x & 7, sub = dv & 3, which = dv >> 2;
    const bool live = bq < da.nlq;
    const long long bqc = live ? bq : 0;
    const long long pair = bqc * M + head;
    const int b = (int)(bqc / da.Lq);

    // ---- this lane's sampling points: (level 2i + which, point sub) for i < LPAIRS -------------
    float sx[LPAIRS], sy[LPAIRS], sa[LPAIRS];
    bool have[LPAIRS];
#pragma unroll
    for (int i = 0; i < LPAIRS; ++i) {
        const int ml = 2 * i + which;
        have[i] = ml < L;
        const int s = (have[i] ? ml : 0) * PT + sub;
        if constexpr (!FUSED) {
            const float2 xy = *reinterpret_cast<const float2 *>(da.loc + (pair * LP + s) * 2);
            sx[i] = xy.x;
            sy[i] = xy.y;
            sa[i] = da.attn[pair * LP + s];
        } else {
            const float *row = da.fa.qproj + bqc * da.fa.ld;
            const float2 off = *reinterpret_cast<const float2 *>(row + da.fa.off_col + (head * LP + s) * 2);
            sx[i] = off.x;
            sy[i] = off.y;
            sa[i] = have[i] ? row[da.fa.logit_col + head * LP + s] : -__builtin_inff();
        }
    }
    __syncthreads();   // level table (the only LDS use; no memory latency in front of it)

    if constexpr (FUSED) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
        // softmax over the pair's L*P logits: this lane holds LPAIRS of them, the group the rest
        float mx = sa[0];
#pragma unroll
        for (int i = 1; i < LPAIRS; ++i) mx = fmaxf(mx, sa[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < LPAIRS; ++i) {
            sa[i] = have[i] ? __expf(sa[i] - mx) : 0.f;
            sum += sa[i];
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
#pragma unroll
        for (int i = 0; i < LPAIRS; ++i) {
            sa[i] = sa[i] / sum;
            const int ml = have[i] ? 2 * i + which : 0;
            const float *rp = da.fa.ref + (bqc * L + ml) * da.fa.ref_dim;
            if (da.fa.ref_dim == 2) {
                sx[i] = rp[0] + sx[i] / (float)s_tab[ml];                         // x / H_l (as written)
                sy[i] = rp[1] + sy[i] / (float)s_tab[TF_MSDA_MAX_LEVELS + ml];    // y / W_l
            } else {
                sx[i] = rp[0] + sx[i] / (float)PT * rp[2] * 0.5f;
                sy[i] = rp[1] + sy[i] / (float)PT * rp[3] * 0.5f;
            }
        }
    }

    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * da.S * M + head) * D) * 4);
    const unsigned dvb = (unsigned)dv * 16u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(da.value), 0, da.value_bytes, 0x00020000);

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LPAIRS; ++i) {
        // produce: taps of this lane's point of the level pair (2i, 2i+1)
        const int ml = have[i] ? 2 * i + which : 0;
        const int H = s_tab[ml], W = s_tab[TF_MSDA_MAX_LEVELS + ml];
        const unsigned lvl_base = head_base + (unsigned)s_tab[2 * TF_MSDA_MAX_LEVELS + ml] * rowbytes;
        const float Wf = (float)W, Hf = (float)H;
        const float xr = __builtin_fmaf(sx[i], Wf, -0.5f);   // cuh:227-228, single rounding
        const float yr = __builtin_fmaf(sy[i], Hf, -0.5f);
        const bool in = have[i] && live && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
        const float x = in ? xr : 0.f, y = in ? yr : 0.f;
        const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
        const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
        const int x0 = (int)xf, y0 = (int)yf;
        const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
        const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
        const int r0 = y0 * W + x0;
        // invalid taps: kOobBase + dv*16 (<= 0xFFFFFFF0) is still out of range -> hardware zero
        const int po0 = (int)((ky0 && kx0) ? lvl_base + (unsigned)r0 * rowbytes : kOobBase);
        const int po1 = (int)((ky0 && kx1) ? lvl_base + (unsigned)(r0 + 1) * rowbytes : kOobBase);
        const int po2 = (int)((ky1 && kx0) ? lvl_base + (unsigned)(r0 + W) * rowbytes : kOobBase);
        const int po3 = (int)((ky1 && kx1) ? lvl_base + (unsigned)(r0 + W + 1) * rowbytes : kOobBase);
        const float a = in ? sa[i] : 0.f;
        const float pw0 = gy * gx * a, pw1 = gy * fx * a, pw2 = fy * gx * a, pw3 = fy * fx * a;

        // publish: LDS operations of a wave execute in order, so a wave-scope fence is all the
        // synchronisation there is (it also orders the previous level pair's reads before this write)
        const int lane = threadIdx.x & 63;
        const int xbase = (threadIdx.x >> 6) * 72 + (lane >> 3) * 9;   // slot of lane 0 of this group
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        s_xo[xbase + dv] = u32x4_t{(unsigned)po0, (unsigned)po1, (unsigned)po2, (unsigned)po3};
        s_xw[xbase + dv] = f32x4_t{pw0, pw1, pw2, pw3};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // consume: the points of one level (produced by lanes 4*ll .. 4*ll+3 of the group) at a time,
        // 16 row gathers in flight per lane
#pragma unroll
        for (int ll = 0; ll < 2; ++ll) {
            if (2 * i + ll >= L) break;   // uniform
            u32x4_t v[4][4];
            f32x4_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4_t o = s_xo[xbase + ll * 4 + k];
                w[k] = s_xw[xbase + ll * 4 + k];
                v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + dvb, 0, 0);
                v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + dvb, 0, 0);
                v[k][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + dvb, 0, 0);
                v[k][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + dvb, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc += __builtin_bit_cast(f32x4_t, v[k][0]) * w[k].x;
                acc += __builtin_bit_cast(f32x4_t, v[k][1]) * w[k].y;
                acc += __builtin_bit_cast(f32x4_t, v[k][2]) * w[k].z;
                acc += __builtin_bit_cast(f32x4_t, v[k][3]) * w[k].w;
            }
        }
    }
    if (live) *reinterpret_cast<f32x4_t *>(da.out + pair * D + dv * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// forward, fp32, D == 36 (hidden 288: cfg 4), P == 4, L <= 8: msda_fwd_f32_direct's mapping with 9 lanes per pair
// ---------------------------------------------------------------------------------------------
// The decoder calls of the `multi_frame` model (D = 36, L = 8: two frames x 4 levels, 500 + 300 queries) ran
// msda_fwd_f32_buf: every lane repeats the tap arithmetic (965 vector instructions per wave against 380 in
// msda_fwd_f32_direct) behind a staging prologue.  This kernel keeps `direct`'s structure -- points fetched straight
// into registers, the tap arithmetic of a point computed once per pair and published through a per-wave LDS
// exchange, 16 row gathers in flight per lane -- for 144-byte rows: a (query, head) pair is served by NINE lanes
// (4 channels each), seven pairs per wave (lane 63 idles), 28 pairs per 256-thread workgroup.  Lanes 0-7 of a group
// produce (lanes 0-3: the points of level 2i, lanes 4-7: those of level 2i + 1), all nine consume.  The groups are
// not aligned to DPP rows, so the softmax statistics of the fused entry go through wave shuffles (ds_bpermute).
// OPT-IN until it has been timed on hardware: tf_msda_set_option("direct9", 1) / TF_MSDA_DIRECT9=1.
template <int LPAIRS, bool FUSED>   // LPAIRS = ceil(L / 2)
__global__ void __launch_bounds__(kThreads, 4)
msda_fwd_f32_direct9(const DirectArgs da, const LevelTable lt, const int64_t *__restrict__ dshapes)
{
    constexpr int PT = 4, D = 36, GL = 9, GPW = 7;   // lanes per pair, pairs per wave
    __shared__ int s_tab[3 * TF_MSDA_MAX_LEVELS];
    __shared__ u32x4_t s_xo[(kThreads / 64) * 72];   // per wave: 7 groups x 9 slots (8 used) + padding
    __shared__ f32x4_t s_xw[(kThreads / 64) * 72];
    const int L = da.L, M = da.M, LP = L * PT;
    fill_level_table(s_tab, lt, dshapes, L);

    const int head = blockIdx.x % M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / GL, dv = lane - grp * GL;          // lane 63: grp 7, dv 0 (idle)
    const long long bq = (long long)(blockIdx.x / M) * ((kThreads / 64) * GPW) + wave * GPW + grp;
    const bool live = grp < GPW && bq < da.nlq;
    const bool producer = dv < 8 && grp < GPW;
    const int sub = dv & 3, which = (dv >> 2) & 1;
    const long long bqc = live ? bq : 0;
    const long long pair = bqc * M + head;
    const int b = (int)(bqc / da.Lq);

    // ---- this lane's sampling points: (level 2i + which, point sub) for i < LPAIRS (lanes 0-7 of a group) ----
    float sx[LPAIRS], sy[LPAIRS], sa[LPAIRS];
    bool have[LPAIRS];
#pragma unroll
    for (int i = 0; i < LPAIRS; ++i) {
        const int ml = 2 * i + which;
        have[i] = producer && ml < L;
        const int s = (have[i] ? ml : 0) * PT + sub;
        if constexpr (!FUSED) {
            const float2 xy = *reinterpret_cast<const float2 *>(da.loc + (pair * LP + s) * 2);
            sx[i] = xy.x;
            sy[i] = xy.y;
            sa[i] = da.attn[pair * LP + s];
        } else {
            const float *row = da.fa.qproj + bqc * da.fa.ld;
            const float2 off = *reinterpret_cast<const float2 *>(row + da.fa.off_col + (head * LP + s) * 2);
            sx[i] = off.x;
            sy[i] = off.y;
            sa[i] = have[i] ? row[da.fa.logit_col + head * LP + s] : -__builtin_inff();
        }
    }
    __syncthreads();   // level table

    if constexpr (FUSED) {
#pragma clang fp contract(off)   // keep the reference's operation order (no fused multiply-add)
        // softmax over the pair's L*P logits: the eight producer lanes of the group hold LPAIRS of them each
        const int gbase = grp * GL;
        float mxl = sa[0];
#pragma unroll
        for (int i = 1; i < LPAIRS; ++i) mxl = fmaxf(mxl, sa[i]);
        float mx = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, __shfl(mxl, gbase + k));
        float suml = 0.f;
#pragma unroll
        for (int i = 0; i < LPAIRS; ++i) {
            sa[i] = have[i] ? __expf(sa[i] - mx) : 0.f;
            suml += sa[i];
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += __shfl(suml, gbase + k);
#pragma unroll
        for (int i = 0; i < LPAIRS; ++i) {
            sa[i] = sa[i] / sum;
            const int ml = have[i] ? 2 * i + which : 0;
            const float *rp = da.fa.ref + (bqc * L + ml) * da.fa.ref_dim;
            if (da.fa.ref_dim == 2) {
                sx[i] = rp[0] + sx[i] / (float)s_tab[ml];                         // x / H_l (as written)
                sy[i] = rp[1] + sy[i] / (float)s_tab[TF_MSDA_MAX_LEVELS + ml];    // y / W_l
            } else {
                sx[i] = rp[0] + sx[i] / (float)PT * rp[2] * 0.5f;
                sy[i] = rp[1] + sy[i] / (float)PT * rp[3] * 0.5f;
            }
        }
    }

    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * da.S * M + head) * D) * 4);
    const unsigned dvb = (unsigned)dv * 16u;   // <= 128: kOobBase + dvb stays out of range
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(da.value), 0, da.value_bytes, 0x00020000);
    // slot of lane 0 of this group; lane 63 reads along with group 6 (valid offsets, results never stored)
    const int xbase = wave * 72 + (grp < GPW ? grp : GPW - 1) * GL;

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LPAIRS; ++i) {
        // produce: taps of this lane's point of the level pair (2i, 2i+1)
        const int ml = have[i] ? 2 * i + which : 0;
        const int H = s_tab[ml], W = s_tab[TF_MSDA_MAX_LEVELS + ml];
        const unsigned lvl_base = head_base + (unsigned)s_tab[2 * TF_MSDA_MAX_LEVELS + ml] * rowbytes;
        const float Wf = (float)W, Hf = (float)H;
        const float xr = __builtin_fmaf(sx[i], Wf, -0.5f);   // cuh:227-228, single rounding
        const float yr = __builtin_fmaf(sy[i], Hf, -0.5f);
        const bool in = have[i] && live && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
        const float x = in ? xr : 0.f, y = in ? yr : 0.f;
        const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
        const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
        const int x0 = (int)xf, y0 = (int)yf;
        const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
        const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
        const int r0 = y0 * W + x0;
        const int po0 = (int)((ky0 && kx0) ? lvl_base + (unsigned)r0 * rowbytes : kOobBase);
        const int po1 = (int)((ky0 && kx1) ? lvl_base + (unsigned)(r0 + 1) * rowbytes : kOobBase);
        const int po2 = (int)((ky1 && kx0) ? lvl_base + (unsigned)(r0 + W) * rowbytes : kOobBase);
        const int po3 = (int)((ky1 && kx1) ? lvl_base + (unsigned)(r0 + W + 1) * rowbytes : kOobBase);
        const float a = in ? sa[i] : 0.f;
        const float pw0 = gy * gx * a, pw1 = gy * fx * a, pw2 = fy * gx * a, pw3 = fy * fx * a;

        // publish (lanes 0-7 of the seven groups): wave-scope ordering as in `direct`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (producer) {
            s_xo[xbase + dv] = u32x4_t{(unsigned)po0, (unsigned)po1, (unsigned)po2, (unsigned)po3};
            s_xw[xbase + dv] = f32x4_t{pw0, pw1, pw2, pw3};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // consume: one level at a time, 16 row gathers (16 bytes each) in flight per lane
#pragma unroll
        for (int ll = 0; ll < 2; ++ll) {
            if (2 * i + ll >= L) break;   // uniform
            u32x4_t v[4][4];
            f32x4_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4_t o = s_xo[xbase + ll * 4 + k];
                w[k] = s_xw[xbase + ll * 4 + k];
                v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + dvb, 0, 0);
                v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + dvb, 0, 0);
                v[k][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + dvb, 0, 0);
                v[k][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + dvb, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc += __builtin_bit_cast(f32x4_t, v[k][0]) * w[k].x;
                acc += __builtin_bit_cast(f32x4_t, v[k][1]) * w[k].y;
                acc += __builtin_bit_cast(f32x4_t, v[k][2]) * w[k].z;
                acc += __builtin_bit_cast(f32x4_t, v[k][3]) * w[k].w;
            }
        }
    }
    if (live) *reinterpret_cast<f32x4_t *>(da.out + pair * D + dv * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// 2-D query tiles of the sorted backward kernel (512 threads, 8 lanes per (query, head) pair, up to two passes of 64
// pairs).  The forward kernel that introduced this geometry (msda_fwd_f32_win, round 1: 8 lanes per pair, all windows
// resident) was superseded by the 4-lanes-per-pair LDS-window kernels (msda_fwd_quad.h, msda_pquad.hip) and removed
// in round 4; its measurements stay in DESIGN.md section 4.1.
// ---------------------------------------------------------------------------------------------
constexpr int kWinThreads = 512;
constexpr int kWinWaves = kWinThreads / 64;
#ifndef TF_BWD_PASSES
#define TF_BWD_PASSES 2   // tools/build_variant.py --source msda_hip.hip -DTF_BWD_PASSES=4: 256-query tiles (experiment)
#endif
#ifndef TF_BWD_ABLATE
#define TF_BWD_ABLATE 0   // timing ablations of msda_bwd_f32_sorted2 (tools only): 1 no global atomics, 2 no value gathers, 4 no sort / reduce
#endif
constexpr int kWinPasses = TF_BWD_PASSES;
constexpr int kWinPairs = kWinThreads / 8;               // (query, head) pairs per pass
constexpr int kWinMaxQueries = kWinPasses * kWinPairs;   // queries per tile
constexpr int kWinLevels = 4;
constexpr int kWinHeaderBytes = 1664;                    // level table | query partition | boxes | geometry
constexpr int kWinXchSlots = kWinWaves * 72;
constexpr int kWinRowsOffset = kWinHeaderBytes + 3 * kWinXchSlots * 16;   // multiple of 128

struct WinGeom {
    int TH, TW;        // tile size in level-0 pixels
    int HY, HX;        // the adaptive windows are clamped to the tile footprint +- this many pixels
    int tiles_y, tiles_x;
    int cap_rows;      // LDS rows for windows, all levels together (multiple of 8)
};

// ---------------------------------------------------------------------------------------------
// forward, encoder shape, fp32, D == 32, P == 4, L <= 4: LDS windows, 4 lanes per pair (DPP quads)
// ---------------------------------------------------------------------------------------------
#include "msda_quad_dev.h"
#include "msda_fwd_quad.h"

// ---------------------------------------------------------------------------------------------
// backward, encoder shape, fp32, D == 32, P == 4, L <= 4: grad_value contributions sorted by row in LDS
// ---------------------------------------------------------------------------------------------
// msda_bwd_f32_buf is bound by the L2 atomic units, which are occupied ~25 cycles per cache line touched:
// 64 taps per (query, head) pair = 11.4 M row updates per cfg-2 encoder launch.  Neighbouring queries
// hit the same rows (~18 contributions per row inside a tile), so this kernel merges them BEFORE they
// reach L2 -- without floating-point LDS atomics (ds_add_f32 costs ~100 cycles per wave instruction on
// gfx950): per 2-D query tile, head and level the tap contributions (destination row, weight x
// attention, source query) are counting-sorted by destination row with integer LDS atomics, then one
// half wave per destination row sums its segment in registers (lane = channel, grad_out of the tile
// staged in LDS) and issues ONE full-row global atomic.  Rows outside the binned window (bounding box
// of the tile's taps, clamped; at most kSortRowsCap rows per level) are scattered directly, also as
// full rows.  grad_loc / grad_attn as in the row kernels.
constexpr int kSortRowsCap = 1024;                      // destination rows binned per level and tile
constexpr int kSortItems = kWinMaxQueries * 4 * 4;      // taps per level: queries x points x taps
constexpr unsigned kSortInvalid = 0xFFFFFFFFu;          // item key: [31] direct scatter, [30:23] query, [22:0] row
constexpr int kSortQueryShift = 23;
constexpr unsigned kSortRowMask = (1u << kSortQueryShift) - 1u, kSortQueryMask = 0xFFu;
static_assert(kWinMaxQueries <= 256, "item key: 8 bits of query");
constexpr int kSortOffCnt = 768;                                     // u32[kSortRowsCap]
constexpr int kSortOffStart = kSortOffCnt + 4 * kSortRowsCap;        // u32[kSortRowsCap + 1]
constexpr int kSortOffItem = kSortOffStart + 4224;                   // {key, weight}[kSortItems]
constexpr int kSortOffSorted = kSortOffItem + 8 * kSortItems;        // the same, sorted by row
constexpr int kSortOffGo = kSortOffSorted + 8 * kSortItems;          // grad_out of the tile [128][32]
constexpr int kSortLdsBytes = kSortOffGo + kWinMaxQueries * 128;     // 58240: two workgroups per CU

struct BwdSortArgs {
    const float *value;
    unsigned value_bytes;
    const float *loc, *attn, *grad_out;
    float *grad_value, *grad_loc, *grad_attn;
    int S, M, L;
};

// ---------------------------------------------------------------------------------------------
// msda_bwd_f32_sorted2: the algorithm above with ~half the vector instructions of its first implementation
// (msda_bwd_f32_sorted, rounds 1-2: 292 -> 223 us at cfg 2, profiles/r03_optin_msda_variants.txt; removed in round 4).  That
// kernel was bound by vector-ALU issue (profiles/r01_msda_bwd_sorted_pmc.json: 84 % of its cycles, 9.9 k instructions per
// wave), spent in phase b and e:
//   * phase b: the tap arithmetic of a point was repeated by the 8 lanes of its pair.  Here the lane that loaded the
//     point in phase A (level 2i + which, point sub) computes it once, files the point's four grad_value items, and
//     publishes offsets / weights through a per-wave LDS exchange (as msda_fwd_f32_direct);
//   * phase b: the three channel sums per point (cuh:365-376) are linear in s_t = <grad_out, value row of tap t>:
//     grad_attn = sum_t w_t s_t,  d/dx = gy (s2 - s1) + fy (s4 - s3),  d/dy = gx (s3 - s1) + fx (s4 - s2).
//     Four dot products of 4 channels per lane and four 8-lane DPP reductions replace three element-wise
//     combinations of the four rows (60 -> 16 multiply-adds per lane and point);
//   * phase e: eight destination rows per wave at a time, 8 lanes x 4 channels per row (one ds_read_b128 of the
//     tile's grad_out per item instead of a ds_read_b32 per channel and half wave), items as {byte offset of the
//     query's grad_out row, weight}; the sums are transposed through LDS so that the global atomics still add to
//     complete 128-byte rows (two rows per instruction).
// The first implementation's LDS layout (constants above) plus 1.5 KB per wave (exchange / transposition buffer): 70 528 B.
constexpr int kSort2XchPerWave = 1536;
constexpr int kSort2OffXch = kSortLdsBytes;
constexpr int kSort2LdsBytes = kSort2OffXch + kWinWaves * kSort2XchPerWave;

__global__ void __launch_bounds__(kWinThreads, (kWinPasses > 2 ? 2 : 4))
msda_bwd_f32_sorted2(const BwdSortArgs ba, const LevelTable lt, const WinGeom wg)
{
    constexpr int PT = 4, D = 32, LPAIRS = kWinLevels / 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);                    // H | W | start          (48 ints)
    int *s_q = s_tab + 3 * TF_MSDA_MAX_LEVELS;                     // ya | yb | xa | xb
    int *s_bb = s_q + 5 * TF_MSDA_MAX_LEVELS + 4;                  // xmin xmax ymin ymax per level
    int *s_direct = s_bb + 4 * kWinLevels;                         // per level: taps filed for direct scatter?
    unsigned *s_cnt = reinterpret_cast<unsigned *>(smem + kSortOffCnt);
    unsigned *s_start = reinterpret_cast<unsigned *>(smem + kSortOffStart);
    uint2 *s_item = reinterpret_cast<uint2 *>(smem + kSortOffItem);
    uint2 *s_sorted = reinterpret_cast<uint2 *>(smem + kSortOffSorted);
    float *s_go = reinterpret_cast<float *>(smem + kSortOffGo);

    const int L = ba.L, M = ba.M, S = ba.S, LP = L * PT;
    const int m = blockIdx.x % M;
    int t = blockIdx.x / M;
    const int tx = t % wg.tiles_x;
    t /= wg.tiles_x;
    const int ty = t % wg.tiles_y;
    const int b = t / wg.tiles_y;

    if (threadIdx.x < 4 * kWinLevels) {   // as msda_fwd_f32_win: exact partition of every level
        const int l = threadIdx.x >> 2, k = threadIdx.x & 3;
        if (l < L) {
            const unsigned H0 = (unsigned)lt.H[0], W0 = (unsigned)lt.W[0];
            const unsigned Hl = (unsigned)lt.H[l], Wl = (unsigned)lt.W[l];
            const unsigned y0 = (unsigned)ty * wg.TH, y1 = min(H0, y0 + (unsigned)wg.TH);
            const unsigned x0 = (unsigned)tx * wg.TW, x1 = min(W0, x0 + (unsigned)wg.TW);
            const unsigned num = k == 0 ? 2u * y0 * Hl + H0 - 1u : k == 1 ? 2u * y1 * Hl + H0 - 1u
                                 : k == 2 ? 2u * x0 * Wl + W0 - 1u : 2u * x1 * Wl + W0 - 1u;
            s_q[k * TF_MSDA_MAX_LEVELS + l] = (int)(num / (k < 2 ? 2u * H0 : 2u * W0));
            if (k == 0) {
                s_tab[l] = lt.H[l];
                s_tab[TF_MSDA_MAX_LEVELS + l] = lt.W[l];
                s_tab[2 * TF_MSDA_MAX_LEVELS + l] = lt.start[l];
            }
        }
        s_bb[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
        if (threadIdx.x < kWinLevels) s_direct[threadIdx.x] = 0;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dv = threadIdx.x & 7, sub = dv & 3, which = dv >> 2;
    const int pl = threadIdx.x >> 3;   // pair of this lane group inside a pass
    int qoff[kWinLevels + 1];
    qoff[0] = 0;
#pragma unroll
    for (int l = 0; l < kWinLevels; ++l)
        qoff[l + 1] = qoff[l] + (l < L ? (s_q[TF_MSDA_MAX_LEVELS + l] - s_q[l]) *
                                             (s_q[3 * TF_MSDA_MAX_LEVELS + l] - s_q[2 * TF_MSDA_MAX_LEVELS + l])
                                       : 0);
    const int nq = qoff[kWinLevels];

    long long bqs[kWinPasses];
    bool live[kWinPasses];
    f32x4_t gA[kWinPasses];
#pragma unroll
    for (int ps = 0; ps < kWinPasses; ++ps) {
        const int tq = ps * kWinPairs + pl;
        int q = 0;
        live[ps] = tq < nq;
        if (live[ps]) {
            int l = 0, base = 0;
#pragma unroll
            for (int k = 1; k < kWinLevels; ++k)
                if (tq >= qoff[k] && k < L) {
                    l = k;
                    base = qoff[k];
                }
            const int r = tq - base;
            const int nx = s_q[3 * TF_MSDA_MAX_LEVELS + l] - s_q[2 * TF_MSDA_MAX_LEVELS + l];
            const int yy = r / nx, xx = r - yy * nx;
            q = s_tab[2 * TF_MSDA_MAX_LEVELS + l] + (s_q[l] + yy) * s_tab[TF_MSDA_MAX_LEVELS + l] +
                s_q[2 * TF_MSDA_MAX_LEVELS + l] + xx;
        }
        bqs[ps] = (long long)b * S + q;
        gA[ps] = *reinterpret_cast<const f32x4_t *>(ba.grad_out + (bqs[ps] * M + m) * D + dv * 4);
        if (!live[ps]) gA[ps] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4_t *>(s_go + (size_t)tq * D + dv * 4) = gA[ps];   // tile of grad_out
    }

    // ---- phase A: this lane's points (level 2i + which, point sub) stay in registers; bounding box of the valid
    //      taps per level
    float sx[LPAIRS][kWinPasses], sy[LPAIRS][kWinPasses], sa[LPAIRS][kWinPasses];
#pragma unroll
    for (int i = 0; i < LPAIRS; ++i) {
        int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
        const bool have = 2 * i + which < L;
        const int ml = have ? 2 * i + which : 0;
        const int H = s_tab[ml], W = s_tab[TF_MSDA_MAX_LEVELS + ml];
        const float Wf = (float)W, Hf = (float)H;
#pragma unroll
        for (int ps = 0; ps < kWinPasses; ++ps) {
            const long long pi = (bqs[ps] * M + m) * LP + ml * PT + sub;
            const float2 xy = *reinterpret_cast<const float2 *>(ba.loc + pi * 2);
            sx[i][ps] = xy.x;
            sy[i][ps] = xy.y;
            sa[i][ps] = ba.attn[pi];
            const float xr = __builtin_fmaf(xy.x, Wf, -0.5f);
            const float yr = __builtin_fmaf(xy.y, Hf, -0.5f);
            const bool in = live[ps] && have && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);
            const int x0 = (int)__builtin_floorf(in ? xr : 0.f), y0 = (int)__builtin_floorf(in ? yr : 0.f);
            if (in) {
                mnx = min(mnx, x0 >= 0 ? x0 : x0 + 1);
                mxx = max(mxx, (x0 + 1 <= W - 1) ? x0 + 1 : x0);
                mny = min(mny, y0 >= 0 ? y0 : y0 + 1);
                mxy = max(mxy, (y0 + 1 <= H - 1) ? y0 + 1 : y0);
            }
        }
        mnx = min(mnx, dpp_i<kDppQuadXor1>(mnx)); mxx = max(mxx, dpp_i<kDppQuadXor1>(mxx));
        mny = min(mny, dpp_i<kDppQuadXor1>(mny)); mxy = max(mxy, dpp_i<kDppQuadXor1>(mxy));
        mnx = min(mnx, dpp_i<kDppQuadXor2>(mnx)); mxx = max(mxx, dpp_i<kDppQuadXor2>(mxx));
        mny = min(mny, dpp_i<kDppQuadXor2>(mny)); mxy = max(mxy, dpp_i<kDppQuadXor2>(mxy));
        mnx = min(mnx, dpp_i<kDppRowRor8>(mnx)); mxx = max(mxx, dpp_i<kDppRowRor8>(mxx));
        mny = min(mny, dpp_i<kDppRowRor8>(mny)); mxy = max(mxy, dpp_i<kDppRowRor8>(mxy));
        if ((lane & 0xB) == 0 && have && mnx != INT_MAX) {
            atomicMin(&s_bb[4 * ml + 0], mnx);
            atomicMax(&s_bb[4 * ml + 1], mxx);
            atomicMin(&s_bb[4 * ml + 2], mny);
            atomicMax(&s_bb[4 * ml + 3], mxy);
        }
    }
    __syncthreads();

    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * S * M + m) * D) * 4);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(ba.value), 0, ba.value_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g =
        __builtin_amdgcn_make_buffer_rsrc(ba.grad_value, 0, ba.value_bytes, 0x00020000);
    const unsigned la = (unsigned)dv * 16u;
    const int half = lane >> 5, ch = lane & 31;
    const int H0 = __builtin_amdgcn_readfirstlane(s_tab[0]);
    const int W0 = __builtin_amdgcn_readfirstlane(s_tab[TF_MSDA_MAX_LEVELS]);
    const float rH0 = __builtin_amdgcn_rcpf((float)H0), rW0 = __builtin_amdgcn_rcpf((float)W0);
    const int y0t = ty * wg.TH, y1t = min(H0, y0t + wg.TH);
    const int x0t = tx * wg.TW, x1t = min(W0, x0t + wg.TW);
    float rdx[kWinPasses][2], rdy[kWinPasses][2], rdot[kWinPasses][2];
#pragma unroll
    for (int ps = 0; ps < kWinPasses; ++ps)
#pragma unroll
        for (int h = 0; h < 2; ++h) rdx[ps][h] = rdy[ps][h] = rdot[ps][h] = 0.f;

    // per-wave exchange buffer: [8 pairs of the wave][4 points][offsets | weights | fx fy a -]; later the
    // transposition buffer of phase e: [8 rows][32 channels] floats + 8 item counts
    unsigned char *xw = smem + kSort2OffXch + wave * kSort2XchPerWave;
    u32x4_t *xch = reinterpret_cast<u32x4_t *>(xw) + (lane >> 3) * 12;
    float *s_tr = reinterpret_cast<float *>(xw);
    unsigned *s_trn = reinterpret_cast<unsigned *>(xw + 1024);
    const unsigned char *s_gob = reinterpret_cast<const unsigned char *>(s_go);

#pragma unroll
    for (int l = 0; l < kWinLevels; ++l) {
        if (l >= L) break;   // uniform
        // ---- window of this level: bounding box of the tile's taps, clamped to the tile footprint
        //      +- (HY, HX) and to kSortRowsCap rows (wave-uniform)
        const int H = __builtin_amdgcn_readfirstlane(s_tab[l]);
        const int W = __builtin_amdgcn_readfirstlane(s_tab[TF_MSDA_MAX_LEVELS + l]);
        const unsigned lvl_base =
            head_base + (unsigned)__builtin_amdgcn_readfirstlane(s_tab[2 * TF_MSDA_MAX_LEVELS + l]) * rowbytes;
        const int bx0 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 0]);
        const int bx1 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 1]);
        const int by0 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 2]);
        const int by1 = __builtin_amdgcn_readfirstlane(s_bb[4 * l + 3]);
        const int ny0 = (int)__builtin_floorf((float)y0t * (float)H * rH0 - 0.5f) - wg.HY;
        const int ny1 = (int)__builtin_floorf((float)y1t * (float)H * rH0 - 0.5f) + 1 + wg.HY;
        const int nx0 = (int)__builtin_floorf((float)x0t * (float)W * rW0 - 0.5f) - wg.HX;
        const int nx1 = (int)__builtin_floorf((float)x1t * (float)W * rW0 - 0.5f) + 1 + wg.HX;
        const int wx0 = max(max(bx0, nx0), 0), wy0 = max(max(by0, ny0), 0);
        int ww = min(min(bx1, nx1), W - 1) - wx0 + 1, wh = min(min(by1, ny1), H - 1) - wy0 + 1;
        if (ww <= 0 || wh <= 0 || bx0 == INT_MAX || ww > kSortRowsCap) {
            ww = 1;
            wh = 0;
        }
        if (wh * ww > kSortRowsCap) wh = kSortRowsCap / ww;
        const int wx1 = wx0 + ww - 1, wy1 = wy0 + wh - 1, nrows = wh * ww;
        const float Wf = (float)W, Hf = (float)H;

        // ---- a. reset counters and items
        for (int i = threadIdx.x; i < nrows; i += kWinThreads) s_cnt[i] = 0u;
        for (int i = threadIdx.x; i < kSortItems; i += kWinThreads) s_item[i].x = kSortInvalid;
        __syncthreads();

        // ---- b. taps of this level: grad_loc / grad_attn, and the grad_value items
        const bool owner = which == (l & 1);   // this lane loaded point `sub` of level l in phase A
#pragma unroll
        for (int ps = 0; ps < kWinPasses; ++ps) {
            if (ps * kWinPairs >= nq) break;   // uniform
            const int tq = ps * kWinPairs + pl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous round's reads are done
            if (owner) {
                const float a = sa[l >> 1][ps];
                const float xr = __builtin_fmaf(sx[l >> 1][ps], Wf, -0.5f);   // cuh:350-351
                const float yr = __builtin_fmaf(sy[l >> 1][ps], Hf, -0.5f);
                const bool in = live[ps] && (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);   // cuh:359
                const float x = in ? xr : 0.f, y = in ? yr : 0.f;
                const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
                const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
                const int x0 = (int)xf, y0 = (int)yf;
                const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
                const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
                const bool kt[4] = {ky0 && kx0, ky0 && kx1, ky1 && kx0, ky1 && kx1};
                const int r0 = y0 * W + x0;
                const unsigned t1 = lvl_base + (unsigned)r0 * rowbytes, t2 = t1 + rowbytes;
                const unsigned t3 = t1 + (unsigned)W * rowbytes, t4 = t3 + rowbytes;
                const float w[4] = {gy * gx, gy * fx, fy * gx, fy * fx};
                xch[sub * 3 + 0] = u32x4_t{kt[0] ? t1 : kOobBase, kt[1] ? t2 : kOobBase, kt[2] ? t3 : kOobBase,
                                           kt[3] ? t4 : kOobBase};
                xch[sub * 3 + 1] = __builtin_bit_cast(u32x4_t, f32x4_t{w[0], w[1], w[2], w[3]});
                xch[sub * 3 + 2] = __builtin_bit_cast(u32x4_t, f32x4_t{fx, fy, in ? a : 0.f, 0.f});
                // grad_value items of the point's four taps (cuh:279,296-301)
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int tx_ = x0 + (tp & 1), ty_ = y0 + (tp >> 1);
                    const float wt = w[tp] * a;
                    if (kt[tp] && wt != 0.f) {
                        const bool inside = tx_ >= wx0 && tx_ <= wx1 && ty_ >= wy0 && ty_ <= wy1;
                        const unsigned row = inside ? (unsigned)((ty_ - wy0) * ww + (tx_ - wx0))
                                                    : (0x80000000u | (unsigned)(ty_ * W + tx_));
                        s_item[(tq * PT + sub) * 4 + tp] =
                            uint2{row | ((unsigned)tq << kSortQueryShift), __builtin_bit_cast(unsigned, wt)};
                        if (inside)
                            atomicAdd(&s_cnt[row], 1u);
                        else
                            s_direct[l] = 1;   // benign race: every writer stores the same value
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int s = l * PT + p;
                const u32x4_t o = xch[p * 3 + 0];
                const f32x4_t w = __builtin_bit_cast(f32x4_t, xch[p * 3 + 1]);
                const f32x4_t f = __builtin_bit_cast(f32x4_t, xch[p * 3 + 2]);
                const bool gather = !(TF_BWD_ABLATE & 2);
                const f32x4_t vz = __builtin_bit_cast(f32x4_t, o);   // (ablation 2: no loads, the arithmetic stays)
                const f32x4_t v1 = gather ? __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, o.x + la, 0, 0)) : vz;
                const f32x4_t v2 = gather ? __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, o.y + la, 0, 0)) : vz;
                const f32x4_t v3 = gather ? __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, o.z + la, 0, 0)) : vz;
                const f32x4_t v4 = gather ? __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, o.w + la, 0, 0)) : vz;
                // s_t = <grad_out, value row of tap t>: this lane's 4 channels, then the 8 lanes of the pair
                const f32x4_t g = gA[ps];
                float s1 = (g.x * v1.x + g.y * v1.y) + (g.z * v1.z + g.w * v1.w);
                float s2 = (g.x * v2.x + g.y * v2.y) + (g.z * v2.z + g.w * v2.w);
                float s3 = (g.x * v3.x + g.y * v3.y) + (g.z * v3.z + g.w * v3.w);
                float s4 = (g.x * v4.x + g.y * v4.y) + (g.z * v4.z + g.w * v4.w);
                s1 += dpp_f<kDppQuadXor1>(s1); s2 += dpp_f<kDppQuadXor1>(s2);
                s3 += dpp_f<kDppQuadXor1>(s3); s4 += dpp_f<kDppQuadXor1>(s4);
                s1 += dpp_f<kDppQuadXor2>(s1); s2 += dpp_f<kDppQuadXor2>(s2);
                s3 += dpp_f<kDppQuadXor2>(s3); s4 += dpp_f<kDppQuadXor2>(s4);
                s1 += dpp_f<kDppRowHalfMirror>(s1); s2 += dpp_f<kDppRowHalfMirror>(s2);
                s3 += dpp_f<kDppRowHalfMirror>(s3); s4 += dpp_f<kDppRowHalfMirror>(s4);
                const float fx = f.x, fy = f.y, a = f.z, gx = 1.f - fx, gy = 1.f - fy;
                const float dot = (w.x * s1 + w.y * s2) + (w.z * s3 + w.w * s4);   // cuh:365,376
                const float dx = (s2 - s1) * gy + (s4 - s3) * fy;                   // cuh:150-160
                const float dy = (s3 - s1) * gx + (s4 - s2) * fx;                   // cuh:139-149
                if ((s & 7) == dv) {   // lane dv keeps points dv and dv + 8
                    rdx[ps][s >> 3] = dx * a * Wf;      // cuh:371,373
                    rdy[ps][s >> 3] = dy * a * Hf;      // cuh:371,374
                    rdot[ps][s >> 3] = dot;             // cuh:376
                }
            }
        }
        __syncthreads();

        if (TF_BWD_ABLATE & 4) continue;   // (ablation 4: no sort, no row reduction, no atomics; uniform)
        // ---- c. exclusive prefix sum of the row counts (wave 0; 16 consecutive rows per lane)
        if (wave == 0) {
            unsigned c[16], sum = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = lane * 16 + k;
                c[k] = r < nrows ? s_cnt[r] : 0u;
                sum += c[k];
            }
            unsigned incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, off);
                if (lane >= off) incl += up;
            }
            unsigned run = incl - sum;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = lane * 16 + k;
                if (r < nrows) {
                    s_start[r] = run;
                    s_cnt[r] = 0u;   // reused as the placement cursor
                }
                run += c[k];
            }
            if (lane == 63) s_start[nrows] = incl;   // number of binned items
        }
        __syncthreads();

        // ---- d. place the binned items in row order, as {byte offset of the query's grad_out row in s_go, weight}
        for (int i = threadIdx.x; i < kSortItems; i += kWinThreads) {
            const uint2 it = s_item[i];
            if (it.x != kSortInvalid && !(it.x & 0x80000000u)) {
                const unsigned row = it.x & kSortRowMask;
                s_sorted[s_start[row] + atomicAdd(&s_cnt[row], 1u)] = uint2{((it.x >> kSortQueryShift) & kSortQueryMask) * (unsigned)(D * 4), it.y};
            }
        }
        __syncthreads();

        // ---- e. eight destination rows per wave at a time (8 lanes x 4 channels per row): sum each row's segment,
        //         transpose through LDS, one atomic per complete 128-byte row
        const float inv_ww = __builtin_amdgcn_rcpf((float)ww);
        const int rg = lane >> 3, cq = lane & 7;
        for (int r0 = wave * 8; r0 < nrows; r0 += 8 * kWinWaves) {
            const int row = r0 + rg;
            unsigned beg = 0, n = 0;
            if (row < nrows) {
                beg = s_start[row];
                n = s_start[row + 1] - beg;
            }
            if (!__any(n != 0u)) continue;   // wave-uniform: eight empty rows
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            for (unsigned k = 0; __any(k < n); k += 2) {   // two items in flight
                const bool p0 = k < n, p1 = k + 1 < n;
                const uint2 i0 = s_sorted[p0 ? beg + k : 0u], i1 = s_sorted[p1 ? beg + k + 1 : 0u];
                const f32x4_t g0 = *reinterpret_cast<const f32x4_t *>(s_gob + (p0 ? i0.x : 0u) + (unsigned)cq * 16u);
                const f32x4_t g1 = *reinterpret_cast<const f32x4_t *>(s_gob + (p1 ? i1.x : 0u) + (unsigned)cq * 16u);
                acc += g0 * (p0 ? __builtin_bit_cast(float, i0.y) : 0.f);
                acc += g1 * (p1 ? __builtin_bit_cast(float, i1.y) : 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous group's reads of s_tr are done
            *reinterpret_cast<f32x4_t *>(s_tr + rg * D + cq * 4) = acc;
            if (cq == 0) s_trn[rg] = n;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = 2 * j + half, rowj = r0 + rr;
                const float v = s_tr[rr * D + ch];
                if (rowj < nrows && s_trn[rr] != 0u) {
                    int wy = (int)(((float)rowj + 0.5f) * inv_ww);
                    int wx = rowj - wy * ww;
                    if (wx < 0) { --wy; wx += ww; }
                    if (wx >= ww) { ++wy; wx -= ww; }
                    if (!(TF_BWD_ABLATE & 1))
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                            v, rsrc_g, lvl_base + (unsigned)((wy0 + wy) * W + wx0 + wx) * rowbytes + (unsigned)ch * 4u, 0, 0);
                    else if (v == 12345.678f)   // (ablation: keep the sum alive)
                        s_tr[rr * D + ch] = v;
                }
            }
        }
        // ---- f. taps outside the window: scattered directly, still one full row per half wave
        if (s_direct[l]) {
            for (int i = wave * 2 + half; i < kSortItems; i += 2 * kWinWaves) {
                const uint2 it = s_item[i];
                if (it.x != kSortInvalid && (it.x & 0x80000000u)) {
                    const float v = __builtin_bit_cast(float, it.y) * s_go[((it.x >> kSortQueryShift) & kSortQueryMask) * D + ch];
                    if (!(TF_BWD_ABLATE & 1))
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                            v, rsrc_g, lvl_base + (it.x & kSortRowMask) * rowbytes + (unsigned)ch * 4u, 0, 0);
                }
            }
        }
        __syncthreads();   // items / counters / exchange buffers are reused by the next level
    }

#pragma unroll
    for (int ps = 0; ps < kWinPasses; ++ps) {
        if (!live[ps]) continue;
        const long long pair = bqs[ps] * M + m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = dv + 8 * h;
            if (s < LP) {
                *reinterpret_cast<float2 *>(ba.grad_loc + (pair * LP + s) * 2) = float2{rdx[ps][h], rdy[ps][h]};
                ba.grad_attn[pair * LP + s] = rdot[ps][h];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward (grad_value via atomics, grad_loc / grad_attn via wave reduction), fused
// ---------------------------------------------------------------------------------------------
#ifdef TF_EXPERIMENT_WG_SCOPE_ATOMICS
#define ATOMIC_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif

template <typename T, int VEC, bool POW2>
__global__ void __launch_bounds__(kThreads)
msda_bwd_rowgather(const T *__restrict__ value, const T *__restrict__ loc,
                   const T *__restrict__ attn, const T *__restrict__ grad_out,
                   T *__restrict__ grad_value, T *__restrict__ grad_loc,
                   T *__restrict__ grad_attn, const LevelTable lt,
                   const int64_t *__restrict__ dshapes, int S, int M, int D, int L, int Lq, int P,
                   long long total_pairs, int ppb, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    const int LP = L * P;
    T *s_loc = reinterpret_cast<T *>(smem + kLevelTableBytes);
    T *s_attn = s_loc + (size_t)ppb * LP * 2;
    T *s_gloc = s_attn + (size_t)ppb * LP;
    T *s_gattn = s_gloc + (size_t)ppb * LP * 2;

    const long long lblk = logical_block((total_pairs + ppb - 1) / ppb);
    if (lblk < 0) return;
    const long long pair0 = lblk * ppb;
    const int npairs = (int)min((long long)ppb, total_pairs - pair0);

    fill_level_table(s_tab, lt, dshapes, L);
    copy_in(s_loc, loc + pair0 * LP * 2, npairs * LP * 2);
    copy_in(s_attn, attn + pair0 * LP, npairs * LP);
    if (!POW2) {
        for (int i = threadIdx.x; i < ppb * LP * 3; i += kThreads) s_gloc[i] = (T)0;  // gloc+gattn
    }
    __syncthreads();

    const int pl = threadIdx.x / DV;
    const int dv = threadIdx.x - pl * DV;
    if (pl < npairs) {
        const long long pair = pair0 + pl;
        const int m = (int)(pair % M);
        const int b = (int)((pair / M) / Lq);
        const long long pix = (long long)M * D;
        const long long voff = (long long)b * S * pix + (long long)m * D + dv * VEC;
        const T *vb = value + voff;
        T *gvb = grad_value + voff;
        const T *sl = s_loc + (size_t)pl * LP * 2;
        const T *sa = s_attn + (size_t)pl * LP;
        T *sgl = s_gloc + (size_t)pl * LP * 2;
        T *sga = s_gattn + (size_t)pl * LP;

        using P4 = Pack<T, VEC>;
        const P4 g = *reinterpret_cast<const P4 *>(grad_out + pair * D + dv * VEC);

        for (int l = 0; l < L; ++l) {
            const int H = s_tab[l], W = s_tab[TF_MSDA_MAX_LEVELS + l];
            const long long loff = (long long)s_tab[2 * TF_MSDA_MAX_LEVELS + l] * pix;
            const T *vl = vb + loff;
            T *gvl = gvb + loff;
#pragma unroll 2
            for (int p = 0; p < P; ++p) {
                const int s = l * P + p;
                const T lx = sl[2 * s], ly = sl[2 * s + 1];
                const T a = sa[s];
                const Tap<T> t = make_tap(lx, ly, H, W);
                const long long e1 = (long long)t.o1 * pix, e2 = (long long)t.o2 * pix;
                const long long e3 = (long long)t.o3 * pix, e4 = (long long)t.o4 * pix;
                const P4 v1 = *reinterpret_cast<const P4 *>(vl + e1);
                const P4 v2 = *reinterpret_cast<const P4 *>(vl + e2);
                const P4 v3 = *reinterpret_cast<const P4 *>(vl + e3);
                const P4 v4 = *reinterpret_cast<const P4 *>(vl + e4);
                const T w1 = t.gy * t.gx, w2 = t.gy * t.fx, w3 = t.fy * t.gx, w4 = t.fy * t.fx;
                T dot = (T)0, dx = (T)0, dy = (T)0;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const T a1 = t.k1 ? v1.v[c] : (T)0;
                    const T a2 = t.k2 ? v2.v[c] : (T)0;
                    const T a3 = t.k3 ? v3.v[c] : (T)0;
                    const T a4 = t.k4 ? v4.v[c] : (T)0;
                    const T gc = g.v[c];
                    dot = fma_t(gc, w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4, dot);      // cuh:365
                    dx = fma_t(gc, t.gy * (a2 - a1) + t.fy * (a4 - a3), dx);          // cuh:150-160
                    dy = fma_t(gc, t.gx * (a3 - a1) + t.fx * (a4 - a2), dy);          // cuh:139-149
                    const T top = gc * a;                                             // cuh:279
                    if (t.k1) ATOMIC_ADD(gvl + e1 + c, w1 * top);                // cuh:296-301
                    if (t.k2) ATOMIC_ADD(gvl + e2 + c, w2 * top);
                    if (t.k3) ATOMIC_ADD(gvl + e3 + c, w3 * top);
                    if (t.k4) ATOMIC_ADD(gvl + e4 + c, w4 * top);
                }
                dx *= a * (T)W;  // cuh:371,373
                dy *= a * (T)H;  // cuh:371,374
                if (POW2) {
                    for (int off = DV >> 1; off > 0; off >>= 1) {
                        dot += __shfl_xor(dot, off);
                        dx += __shfl_xor(dx, off);
                        dy += __shfl_xor(dy, off);
                    }
                    if (dv == 0) {
                        sgl[2 * s] = dx;
                        sgl[2 * s + 1] = dy;
                        sga[s] = dot;
                    }
                } else {
                    atomicAdd(&sgl[2 * s], dx);
                    atomicAdd(&sgl[2 * s + 1], dy);
                    atomicAdd(&sga[s], dot);
                }
            }
        }
    }
    __syncthreads();
    copy_out(grad_loc + pair0 * LP * 2, s_gloc, npairs * LP * 2);
    copy_out(grad_attn + pair0 * LP, s_gattn, npairs * LP);
}

// ---------------------------------------------------------------------------------------------
// backward, fp32 fast path: buffer loads + buffer atomics with hardware bounds checking
// ---------------------------------------------------------------------------------------------
// Same pair / lane mapping as msda_fwd_f32_buf (head-major blocks).  Per sampling point a lane group
//   * loads the 4 taps with buffer_load_dwordx4 (invalid taps: out-of-range offset -> 0),
//   * forms its share of d(out)/d(attn), d(out)/d(x), d(out)/d(y) over its 4 channels and reduces
//     them over the D/4 lanes of the pair with an xor butterfly (the reference loops serially over
//     the channels, cuh:356-372),
//   * scatters grad_value with buffer_atomic_add_f32: invalid taps get an out-of-range offset and
//     are DROPPED by the hardware, so there is no divergent code around the 16 atomics of a point;
//     for the atomics lane dv owns channels dv, dv+DV, dv+2DV, dv+3DV, so that one atomic
//     instruction touches D/4 consecutive floats of each row instead of every fourth float;
//   * ROWATOM (D == 32): the taps' row offsets and weights go through LDS and one HALF WAVE scatters a
//     pair's contributions, lane = channel: every atomic instruction then adds to 2 complete 128-byte
//     rows instead of 32-byte pieces of 8 rows, which halves the time (the L2 atomic units are
//     occupied per cache line touched).
// Tried and removed: pre-reducing grad_value in LDS windows (the tiling of msda_fwd_f32_win in reverse,
// ds_add_f32 into zeroed windows, one flush per window row).  Correct, 15x fewer global atomics -- and
// no faster: ds_add_f32 costs ~100 cycles per wave instruction on gfx950, which makes the LDS phase as
// slow as the L2 atomics it replaces (1.9 ms vs 1.9 ms at the cfg-2 encoder shape, init pattern; the
// kernel without its LDS atomics ran in 0.18 ms).
template <int PT, bool ROWATOM>
__global__ void __launch_bounds__(kThreads)
msda_bwd_f32_buf(const float *__restrict__ value, unsigned value_bytes,
                 const float *__restrict__ loc, const float *__restrict__ attn,
                 const float *__restrict__ grad_out, float *__restrict__ grad_value,
                 float *__restrict__ grad_loc, float *__restrict__ grad_attn, const LevelTable lt,
                 const int64_t *__restrict__ dshapes, int S, int M, int D, int L, int Lq,
                 long long total_pairs, int ppb, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    const int LP = L * PT;
    float *s_loc = reinterpret_cast<float *>(smem + kLevelTableBytes);
    float *s_attn = s_loc + (size_t)ppb * LP * 2;
    float *s_gloc = s_attn + (size_t)ppb * LP;
    float *s_gattn = s_gloc + (size_t)ppb * LP * 2;
    // ROWATOM: per (pair, point) the 4 row offsets and the 4 weights (x attention weight) of the taps
    u32x4_t *s_tapo = reinterpret_cast<u32x4_t *>(s_gattn + (size_t)ppb * LP);
    f32x4_t *s_tapw = reinterpret_cast<f32x4_t *>(s_tapo + (size_t)ppb * LP);

    const long long nlq = total_pairs / M;
    const int head = blockIdx.x % M;
    const long long q0 = (long long)(blockIdx.x / M) * ppb;
    if (q0 >= nlq) return;
    const int npairs = (int)min((long long)ppb, nlq - q0);
    auto pair_of = [&](int pp) -> long long { return (q0 + pp) * M + head; };

    fill_level_table(s_tab, lt, dshapes, L);
    {
        const int row = LP * 2;
        for (int i = threadIdx.x; i < npairs * row; i += kThreads) {
            const int pp = i / row, j = i - pp * row;
            s_loc[i] = loc[pair_of(pp) * row + j];
        }
        for (int i = threadIdx.x; i < npairs * LP; i += kThreads) {
            const int pp = i / LP, j = i - pp * LP;
            s_attn[i] = attn[pair_of(pp) * LP + j];
        }
    }
    __syncthreads();

    const int pl = threadIdx.x / DV;
    const int dv = threadIdx.x - pl * DV;
    if (pl < npairs) {
        const long long pair = pair_of(pl);
        const int b = (int)((pair / M) / Lq);
        const unsigned rowbytes = (unsigned)(M * D) * 4u;
        const unsigned head_base = (unsigned)((((long long)b * S * M + head) * D) * 4);
        const __amdgpu_buffer_rsrc_t rsrc_v =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, value_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_g =
            __builtin_amdgcn_make_buffer_rsrc(grad_value, 0, value_bytes, 0x00020000);
        const float2 *sl = reinterpret_cast<const float2 *>(s_loc + (size_t)pl * LP * 2);
        const float *sa = s_attn + (size_t)pl * LP;
        float *sgl = s_gloc + (size_t)pl * LP * 2;
        float *sga = s_gattn + (size_t)pl * LP;

        const float *go = grad_out + pair * D;
        const f32x4_t gA = *reinterpret_cast<const f32x4_t *>(go + dv * 4);   // channels 4dv..4dv+3
        float gB[4];                                                           // channels dv + c*DV
#pragma unroll
        for (int c = 0; c < 4; ++c) gB[c] = go[dv + c * DV];

        for (int l = 0; l < L; ++l) {
            const int H = s_tab[l], W = s_tab[TF_MSDA_MAX_LEVELS + l];
            const unsigned lvl_base = head_base + (unsigned)s_tab[2 * TF_MSDA_MAX_LEVELS + l] * rowbytes;
            const float Wf = (float)W, Hf = (float)H;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int s = l * PT + p;
                const float2 xy = sl[s];
                const float a = sa[s];
                const float xr = __builtin_fmaf(xy.x, Wf, -0.5f);   // cuh:350-351
                const float yr = __builtin_fmaf(xy.y, Hf, -0.5f);
                const bool in = (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);  // cuh:359
                const float x = in ? xr : 0.f, y = in ? yr : 0.f;
                const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
                const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
                const int x0 = (int)xf, y0 = (int)yf;
                const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= W - 1);
                const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= H - 1);
                const int r0 = y0 * W + x0;
                const bool k1 = ky0 && kx0, k2 = ky0 && kx1, k3 = ky1 && kx0, k4 = ky1 && kx1;
                const unsigned t1 = lvl_base + (unsigned)r0 * rowbytes;        // row byte offsets
                const unsigned t2 = t1 + rowbytes;
                const unsigned t3 = t1 + (unsigned)W * rowbytes;
                const unsigned t4 = t3 + rowbytes;
                const unsigned la = (unsigned)dv * 16u;
                const f32x4_t v1 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, k1 ? t1 + la : kOobOffset, 0, 0));
                const f32x4_t v2 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, k2 ? t2 + la : kOobOffset, 0, 0));
                const f32x4_t v3 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, k3 ? t3 + la : kOobOffset, 0, 0));
                const f32x4_t v4 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, k4 ? t4 + la : kOobOffset, 0, 0));
                const float w1 = gy * gx, w2 = gy * fx, w3 = fy * gx, w4 = fy * fx;
                // grad wrt value: cuh:279,296-301 with the weights of cuh:84-93
                if constexpr (ROWATOM) {
                    if (dv == 0) {   // scattered below, one full 128-byte row per half wave
                        s_tapo[(size_t)pl * LP + s] = u32x4_t{k1 ? t1 : kOobBase, k2 ? t2 : kOobBase,
                                                              k3 ? t3 : kOobBase, k4 ? t4 : kOobBase};
                        s_tapw[(size_t)pl * LP + s] = f32x4_t{w1 * a, w2 * a, w3 * a, w4 * a};
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned lb = (unsigned)(dv + c * DV) * 4u;
                        const float top = gB[c] * a;
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w1 * top, rsrc_g, k1 ? t1 + lb : kOobOffset, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w2 * top, rsrc_g, k2 ? t2 + lb : kOobOffset, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w3 * top, rsrc_g, k3 ? t3 + lb : kOobOffset, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w4 * top, rsrc_g, k4 ? t4 + lb : kOobOffset, 0, 0);
                    }
                }
                // grad wrt attention weight / location: partial sums over this lane's 4 channels
                const f32x4_t smp = v1 * w1 + v2 * w2 + v3 * w3 + v4 * w4;             // cuh:365
                const f32x4_t ddx = (v2 - v1) * gy + (v4 - v3) * fy;                   // cuh:150-160
                const f32x4_t ddy = (v3 - v1) * gx + (v4 - v2) * fx;                   // cuh:139-149
                const f32x4_t pd = gA * smp, px = gA * ddx, py = gA * ddy;
                float dot = (pd.x + pd.y) + (pd.z + pd.w);
                float dx = (px.x + px.y) + (px.z + px.w);
                float dy = (py.x + py.y) + (py.z + py.w);
                for (int off = DV >> 1; off > 0; off >>= 1) {
                    dot += __shfl_xor(dot, off);
                    dx += __shfl_xor(dx, off);
                    dy += __shfl_xor(dy, off);
                }
                if (dv == 0) {
                    sgl[2 * s] = dx * a * Wf;       // cuh:371,373
                    sgl[2 * s + 1] = dy * a * Hf;   // cuh:371,374
                    sga[s] = dot;                   // cuh:376
                }
            }
        }
    }
    if constexpr (ROWATOM) {
        // D == 32: a half wave (32 lanes = the 32 channels of one row) scatters the taps of one pair, so
        // every atomic instruction adds to 2 complete 128-byte rows instead of 32-byte pieces of 8 rows
        // (the L2 atomic units are occupied per cache line touched).  The 8 pairs a wave computed above
        // are the ones it scatters: wave-scope ordering is enough.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int lane = threadIdx.x & 63, half = lane >> 5, ch = lane & 31;
        const __amdgpu_buffer_rsrc_t rsrc_g =
            __builtin_amdgcn_make_buffer_rsrc(grad_value, 0, value_bytes, 0x00020000);
        const unsigned cb = (unsigned)ch * 4u;
        for (int j = 0; j < 4; ++j) {
            const int pp = (threadIdx.x >> 6) * 8 + 2 * j + half;
            if (pp >= npairs) continue;
            const float g = grad_out[pair_of(pp) * D + ch];
            const u32x4_t *to = s_tapo + (size_t)pp * LP;
            const f32x4_t *tw = s_tapw + (size_t)pp * LP;
            for (int s = 0; s < LP; ++s) {
                const u32x4_t o = to[s];
                const f32x4_t w = tw[s];
                // taps with an exactly zero weight add nothing (a default-initialised model samples at
                // integer pixel offsets: 3 of 4 bilinear weights are 0): skip the instruction when that
                // holds for both pairs of the wave
                if (w.x != 0.f) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w.x * g, rsrc_g, o.x + cb, 0, 0);
                if (w.y != 0.f) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w.y * g, rsrc_g, o.y + cb, 0, 0);
                if (w.z != 0.f) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w.z * g, rsrc_g, o.z + cb, 0, 0);
                if (w.w != 0.f) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w.w * g, rsrc_g, o.w + cb, 0, 0);
            }
        }
    }
    __syncthreads();
    {
        const int row = LP * 2;
        for (int i = threadIdx.x; i < npairs * row; i += kThreads) {
            const int pp = i / row, j = i - pp * row;
            grad_loc[pair_of(pp) * row + j] = s_gloc[i];
        }
        for (int i = threadIdx.x; i < npairs * LP; i += kThreads) {
            const int pp = i / LP, j = i - pp * LP;
            grad_attn[pair_of(pp) * LP + j] = s_gattn[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename... Args>
hipError_t launch(const void *fn, unsigned grid, size_t lds, hipStream_t stream, Args... args)
{
    void *argv[] = {(void *)&args...};
    return hipLaunchKernel(fn, dim3(grid), dim3(kThreads), argv, lds, stream);
}

int record_hip(hipError_t e)
{
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return TF_MSDA_ERR_LAUNCH;
    }
    return TF_MSDA_OK;
}

int build_level_table(const int64_t *shapes_host, int L, int S, LevelTable *lt)
{
    long long acc = 0;
    for (int l = 0; l < TF_MSDA_MAX_LEVELS; ++l) lt->H[l] = lt->W[l] = lt->start[l] = 0;
    for (int l = 0; l < L; ++l) {
        const int64_t h = shapes_host[2 * l], w = shapes_host[2 * l + 1];
        if (h <= 0 || w <= 0 || h > INT32_MAX || w > INT32_MAX) return TF_MSDA_ERR_BAD_DIMS;
        if (acc > INT32_MAX) return TF_MSDA_ERR_BAD_DIMS;
        lt->H[l] = (int)h;
        lt->W[l] = (int)w;
        lt->start[l] = (int)acc;
        acc += h * w;
    }
    if (acc != (long long)S) return TF_MSDA_ERR_SHAPE_SUM;
    return TF_MSDA_OK;
}

struct Plan {
    int vec, DV, ppb;
    size_t lds;
    unsigned grid;
};

template <typename T>
int make_plan(int N, int M, int D, int L, int Lq, int P, int lds_elems_per_sample, bool aligned,
              Plan *pl)
{
    pl->vec = (D % 4 == 0 && aligned) ? 4 : 1;
    pl->DV = D / pl->vec;
    if (pl->DV > kThreads) return TF_MSDA_ERR_BAD_DIMS;  // D > 1024 (or > 256 unaligned)
    const long long LP = (long long)L * P;
    const long long bytes_per_pair = LP * lds_elems_per_sample * (long long)sizeof(T);
    if (bytes_per_pair > kLdsChunkBudget) return TF_MSDA_ERR_BAD_DIMS;
    long long ppb = kThreads / pl->DV;
    if (ppb * bytes_per_pair > kLdsChunkBudget) ppb = kLdsChunkBudget / bytes_per_pair;
    pl->ppb = (int)ppb;
    pl->lds = (size_t)kLevelTableBytes + (size_t)(ppb * bytes_per_pair);
    const long long total_pairs = (long long)N * Lq * M;
    long long grid = (total_pairs + ppb - 1) / ppb;
    grid = (grid + kXcds - 1) / kXcds * kXcds;  // see logical_block()
    if (grid > 0x7fffffffLL) return TF_MSDA_ERR_BAD_DIMS;
    pl->grid = (unsigned)grid;
    return TF_MSDA_OK;
}

bool is_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Block -> pair mapping of the buffer-load forward kernel (TF_MSDA_HEAD_MAJOR=0/1, default 1).
bool head_major_enabled()
{
    static const int on = [] { const char *e = getenv("TF_MSDA_HEAD_MAJOR"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}
unsigned head_major_grid(int N, int Lq, int M, int ppb)
{
    const long long chunks = ((long long)N * Lq + ppb - 1) / ppb;
    return (unsigned)(chunks * M);
}

// fp32 fast path eligibility: everything addressable with 32-bit byte offsets / 24-bit multiplies.
bool buf_path_ok(const LevelTable &lt, bool host_shapes, int N, int S, int M, int D, int L)
{
    const long long bytes = (long long)N * S * M * D * 4;
    if (bytes >= (long long)kOobBase) return false;
    if ((long long)M * D * 4 >= (1 << 24)) return false;
    if (S >= (1 << 24)) return false;   // also bounds every level's H*W (and start) below 2^24
    if (host_shapes)
        for (int l = 0; l < L; ++l)
            if ((long long)lt.H[l] * lt.W[l] >= (1 << 24)) return false;
    return true;
}

// Encoder-shaped calls (Lq == S, host shapes, D == 32 / 36, P == 4, L <= 4) run the LDS-window kernels by default
// (msda_fwd_f32_pquad, or msda_fwd_f32_quad where that declines); TF_MSDA_TILED=0 / tf_msda_set_tiled(0) selects
// msda_fwd_f32_direct.  (Mode 1 selected msda_fwd_f32_win, removed in round 4: it now means 2.)  Round-1 measurements
// at the cfg-2 encoder shape (HIP graph of 20 launches, us per launch, plain / fused entry;
// profiles/r01_msda_fwd_quad_harness.txt):
//                                  init          local         uniform
//   msda_fwd_f32_direct            51.8 / 54.4   62.1 / 65.9   66.6 / 73.6
//   msda_fwd_f32_win               47.3 / 54.4   71.4 / 77.3   82.2 / 88.4
//   msda_fwd_f32_quad (default)    36.0 / 44.4   60.0 / 70.4   76.4 / 89.5
// init = what a default-initialised model produces (bench.py), local = reference point + N(0, 2 px),
// uniform = rand over the level (no locality at all: every level falls back to buffer loads).
std::atomic<int> g_tiled_mode{-1};   // -1: follow the environment, 0: row gathers (msda_fwd_f32_direct), 2: the LDS-window kernels
int tiled_mode()
{
    const int g = g_tiled_mode.load(std::memory_order_relaxed);
    if (g >= 0) return g;
    static const int env_mode = [] {
        const char *e = getenv("TF_MSDA_TILED");   // unset: the LDS-window kernels; 0: off
        if (!e || !e[0]) return 2;
        return e[0] == '0' ? 0 : 2;
    }();
    return env_mode;
}

// Largest number of queries any TH x TW tile holds (exact, same integer partition as the kernel).
long long tile_max_queries(const LevelTable &lt, int L, int th, int tw)
{
    const int H0 = lt.H[0], W0 = lt.W[0];
    long long max_nq = 0;   // exact, same integer partition as the kernel
    for (int y0 = 0; y0 < H0; y0 += th)
        for (int x0 = 0; x0 < W0; x0 += tw) {
            const int y1 = (y0 + th < H0) ? y0 + th : H0, x1 = (x0 + tw < W0) ? x0 + tw : W0;
            long long nq = 0;
            for (int l = 0; l < L; ++l) {
                const long long Hl = lt.H[l], Wl = lt.W[l];
                const long long ny = (2 * y1 * Hl + H0 - 1) / (2LL * H0) - (2 * y0 * Hl + H0 - 1) / (2LL * H0);
                const long long nx = (2 * x1 * Wl + W0 - 1) / (2LL * W0) - (2 * x0 * Wl + W0 - 1) / (2LL * W0);
                nq += ny * nx;
            }
            if (nq > max_nq) max_nq = nq;
        }
    return max_nq;
}

// Tile plan of msda_bwd_f32_sorted2: the tile with the most queries that still fits kWinMaxQueries.
bool plan_sorted(const LevelTable &lt, int L, int D, int P, WinGeom *wg)
{
    static const int on = [] { const char *e = getenv("TF_MSDA_BWD_SORTED"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!on || D != 32 || P != 4 || L > kWinLevels) return false;
    for (int l = 0; l < L; ++l)
        if (lt.H[l] >= 32768 || lt.W[l] >= 32768 || (long long)lt.H[l] * lt.W[l] >= (1 << kSortQueryShift)) return false;
    struct Memo {
        bool valid = false, ok = false;
        int L = 0;
        LevelTable lt;
        WinGeom wg;
    };
    static thread_local Memo memo;
    if (memo.valid && memo.L == L && memcmp(&memo.lt, &lt, sizeof(lt)) == 0) {
        *wg = memo.wg;
        return memo.ok;
    }
    memo.valid = true;
    memo.ok = false;
    memo.L = L;
    memo.lt = lt;
    int hy = 8, hx = 14, th = 0, tw = 0;
    if (const char *e = getenv("TF_MSDA_BWD_HALO")) sscanf(e, "%d,%d", &hy, &hx);
    if (const char *e = getenv("TF_MSDA_BWD_TILE")) sscanf(e, "%d,%d", &th, &tw);
    if (hy < 0 || hx < 0 || th < 0 || tw < 0) return false;
    long long best = 0;
    int bth = 0, btw = 0;
    const int tws[5] = {8, 16, 4, 2, 1};
    for (int k = 0; k < 5; ++k) {
        const int ctw = tw ? tw : tws[k];
        for (int cth = th ? th : 16; cth >= (th ? th : 1); --cth) {
            const long long nq = tile_max_queries(lt, L, cth, ctw);
            if (nq >= 1 && nq <= kWinMaxQueries && nq > best) {
                best = nq;
                bth = cth;
                btw = ctw;
            }
        }
        if (tw || best >= kWinMaxQueries * 2 / 3) break;
    }
    if (!best) return false;
    wg->TH = bth;
    wg->TW = btw;
    wg->HY = hy;
    wg->HX = hx;
    wg->tiles_y = (lt.H[0] + bth - 1) / bth;
    wg->tiles_x = (lt.W[0] + btw - 1) / btw;
    wg->cap_rows = kSortRowsCap;
    memo.wg = *wg;
    memo.ok = true;
    return true;
}

bool raise_dynamic_lds_limit(const void *fn);   // per (function, device), below

// ---- msda_fwd_f32_quad: options, tile plan, launch ----------------------------------------------------
// Performance knobs (process-wide; tf_msda_set_option / TF_MSDA_QUAD="ta=12,waves=8,npass=1,lds=53,...").
enum QuadOpt { kQoTaMask, kQoWaves, kQoNpass, kQoLdsKb, kQoHaloY, kQoHaloX, kQoTileH, kQoTileW, kQoSplit, kQoCount };
const char *const kQuadOptNames[kQoCount] = {"quad_ta_mask", "quad_waves", "quad_npass", "quad_lds_kb", "quad_halo_y",
                                             "quad_halo_x",  "quad_tile_h", "quad_tile_w", "quad_split"};
const char *const kQuadEnvKeys[kQoCount] = {"ta", "waves", "npass", "lds", "hy", "hx", "th", "tw", "split"};
constexpr int kQuadOptDefaults[kQoCount] = {0, 4, 3, 40, 6, 10, 0, 0, 1};
std::atomic<int> g_quad_opt[kQoCount];
std::atomic<int> g_quad_epoch{0};   // bumped by every change: invalidates the per-thread tile plans
std::atomic<unsigned long long *> g_quad_trace{nullptr};   // tf_msda_debug_trace_buffer

void quad_opts_init()
{
    static const bool once = [] {
        for (int i = 0; i < kQoCount; ++i) g_quad_opt[i].store(kQuadOptDefaults[i]);
        if (const char *e = getenv("TF_MSDA_QUAD")) {
            // comma-separated key=value list
            const char *p = e;
            while (*p) {
                const char *eq = strchr(p, '=');
                if (!eq) break;
                for (int i = 0; i < kQoCount; ++i)
                    if ((size_t)(eq - p) == strlen(kQuadEnvKeys[i]) && strncmp(p, kQuadEnvKeys[i], eq - p) == 0)
                        g_quad_opt[i].store(atoi(eq + 1));
                const char *c = strchr(eq, ',');
                if (!c) break;
                p = c + 1;
            }
        }
        return true;
    }();
    (void)once;
}

struct QuadPlan {
    QuadGeom geom;
    size_t lds;
    int ta_mask, waves, npass, split;
};

bool plan_quad(const LevelTable &lt, int L, int D, int P, QuadPlan *qp)
{
    if (tiled_mode() != 2 || D != 32 || P != 4 || L > kQuadLevels) return false;
    quad_opts_init();
    int o[kQoCount];
    for (int i = 0; i < kQoCount; ++i) o[i] = g_quad_opt[i].load(std::memory_order_relaxed);
    const int epoch = g_quad_epoch.load(std::memory_order_relaxed);
    const int ta = o[kQoTaMask], waves = o[kQoWaves], npass = o[kQoNpass];
    const int split = o[kQoSplit];
    if ((ta != 0 && ta != 12) || (waves != 4 && waves != 8) || npass < 1 || npass > 3 ||
        (waves == 8 && npass != 1) || (split != 0 && split != 1))
        return false;
    if (o[kQoLdsKb] < 8 || o[kQoLdsKb] > 160 || o[kQoHaloY] < 0 || o[kQoHaloX] < 0 || o[kQoTileH] < 0 || o[kQoTileW] < 0)
        return false;
    for (int l = 0; l < L; ++l)
        if (lt.H[l] >= 32768 || lt.W[l] >= 32768) return false;   // 32-bit tile arithmetic in the kernel
    const int cap_rows = (int)(((size_t)o[kQoLdsKb] * 1024 - kQuadHdrBytes) / 128 - 2) & ~7;
    if (cap_rows < 8) return false;
    const size_t need = (size_t)kQuadHdrBytes + (size_t)(2 + cap_rows) * 128;
    struct Memo {
        bool valid = false, ok = false;
        int L = 0, epoch = -1;
        LevelTable lt;
        QuadPlan qp;
    };
    static thread_local Memo memo;
    if (memo.valid && memo.L == L && memo.epoch == epoch && memcmp(&memo.lt, &lt, sizeof(lt)) == 0) {
        *qp = memo.qp;
        return memo.ok;
    }
    memo.valid = true;
    memo.ok = false;
    memo.L = L;
    memo.epoch = epoch;
    memo.lt = lt;
    // tile: the most queries per (estimated) window row among the tiles that fill the workgroup
    const long long cap_q = (long long)waves * 16 * npass;
    int bth = 0, btw = 0;
    if (o[kQoTileH] > 0 && o[kQoTileW] > 0) {
        const long long nq = tile_max_queries(lt, L, o[kQoTileH], o[kQoTileW]);
        if (nq >= 1 && nq <= cap_q) {
            bth = o[kQoTileH];
            btw = o[kQoTileW];
        }
    } else {
        double best = 0.0;
        long long best_nq = 0;
        for (int pass = 0; pass < 2 && !bth; ++pass)
            for (int th = 1; th <= 32; ++th)
                for (int tw = 2; tw <= 32; tw += 2) {
                    if ((long long)th * tw > cap_q) continue;
                    const long long nq = tile_max_queries(lt, L, th, tw);
                    if (nq < 1 || nq > cap_q) continue;
                    if (pass == 0 && nq * 4 < cap_q * 3) continue;        // first pass: well-filled tiles only
                    const double eff = (double)nq / ((double)(th + 4) * (double)(tw + 8));
                    if (eff > best || (eff == best && nq > best_nq)) {
                        best = eff;
                        best_nq = nq;
                        bth = th;
                        btw = tw;
                    }
                }
    }
    if (!bth) return false;
    QuadPlan r{};
    r.geom.TH = bth;
    r.geom.TW = btw;
    r.geom.HY = o[kQoHaloY];
    r.geom.HX = o[kQoHaloX];
    r.geom.tiles_y = (lt.H[0] + bth - 1) / bth;
    r.geom.tiles_x = (lt.W[0] + btw - 1) / btw;
    r.geom.cap_rows = cap_rows;
    r.lds = need;
    r.ta_mask = ta;
    r.waves = waves;
    r.npass = npass;
    r.split = split;
    memo.qp = r;
    memo.ok = true;
    *qp = r;
    static const bool verbose = getenv("TF_MSDA_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[tf_msda] quad plan: tile %dx%d (%lld queries max of %lld), %dx%d tiles, ta_mask %d, %d waves x %d "
                        "passes, split %d, %d window rows, %zu B LDS\n", bth, btw, tile_max_queries(lt, L, bth, btw), cap_q,
                r.geom.tiles_y, r.geom.tiles_x, ta, waves, npass, split, cap_rows, need);
    return true;
}

template <bool FUSED, int TA, int ROUND0>
const void *quad_kernel_wn(int waves, int npass)
{
    if (waves == 8) return (const void *)&msda_fwd_f32_quad<FUSED, TA, 8, 1, ROUND0>;
    return npass == 1   ? (const void *)&msda_fwd_f32_quad<FUSED, TA, 4, 1, ROUND0>
           : npass == 2 ? (const void *)&msda_fwd_f32_quad<FUSED, TA, 4, 2, ROUND0>
                        : (const void *)&msda_fwd_f32_quad<FUSED, TA, 4, 3, ROUND0>;
}
template <bool FUSED, int ROUND0>
const void *quad_kernel_ta(int ta, int waves, int npass)
{
    return ta == 0 ? quad_kernel_wn<FUSED, 0, ROUND0>(waves, npass) : quad_kernel_wn<FUSED, 12, ROUND0>(waves, npass);
}
template <bool FUSED>
const void *quad_kernel(int ta, int waves, int npass, int split)
{
    // split: level 0's window alone in a first round, the other levels' windows reuse its LDS rows
    return split ? quad_kernel_ta<FUSED, 0x1>(ta, waves, npass) : quad_kernel_ta<FUSED, 0xF>(ta, waves, npass);
}

// The dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is an attribute of a function ON A DEVICE:
// raised once per (function, current device), remembered in a small lock-protected table.
bool raise_dynamic_lds_limit(const void *fn)
{
    struct Entry { const void *fn; int dev; };
    static Entry table[256];
    static std::atomic<int> count{0};
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const int n = count.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (table[i].fn == fn && table[i].dev == dev) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    while (lock.test_and_set(std::memory_order_acquire)) {}
    const int k = count.load(std::memory_order_relaxed);
    if (k < 256) {
        table[k] = Entry{fn, dev};
        count.store(k + 1, std::memory_order_release);
    }
    lock.clear(std::memory_order_release);
    return true;
}

// Launch msda_fwd_f32_quad for encoder-shaped calls (Lq == S, host shapes).  Returns false if not taken.
bool launch_quad(bool fused, const DirectArgs &da, const LevelTable &lt, int N, int D, int P,
                 hipStream_t stream, hipError_t *err)
{
    if (da.Lq != da.S) return false;
    // the kernel addresses loc / attn / qproj / out with 32-bit byte offsets
    if (fused && (long long)N * da.Lq * da.fa.ld * 4 >= (1LL << 32)) return false;
    QuadPlan qp;
    if (!plan_quad(lt, da.L, D, P, &qp)) return false;
    const long long grid = (long long)N * qp.geom.tiles_y * qp.geom.tiles_x * da.M;
    if (grid > 0x7fffffffLL) return false;
    const void *fn = fused ? quad_kernel<true>(qp.ta_mask, qp.waves, qp.npass, qp.split)
                           : quad_kernel<false>(qp.ta_mask, qp.waves, qp.npass, qp.split);
    if (!raise_dynamic_lds_limit(fn)) return false;
    qp.geom.trace = g_quad_trace.load(std::memory_order_relaxed);
    void *argv[] = {(void *)&da, (void *)&lt, (void *)&qp.geom};
    *err = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(qp.waves * 64), argv, qp.lds, stream);
    note_kernel(fused ? "msda_fwd_f32_quad<fused>" : "msda_fwd_f32_quad<plain>");
    return true;
}

// Launch msda_fwd_f32_direct when the shape qualifies (D == 32, P == 4, L <= 8).  Returns false if not.
bool direct_enabled()
{
    static const int on = [] { const char *e = getenv("TF_MSDA_DIRECT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

// dynamic LDS above 64 KB needs the per-function, per-device attribute: set once per device for msda_bwd_f32_sorted2
bool raise_dynamic_lds(const void *fn)
{
    static std::atomic<unsigned long long> done{0};   // bit d: device d has the attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

std::atomic<int> g_direct9{-1};   // -1: environment (TF_MSDA_DIRECT9, default 1 since round 3: cfg-4 decoder forward 18.8-19.7 -> 12.7-14.0 us,
                                  // profiles/r03_optin_msda_variants.txt)
bool direct9_enabled()
{
    const int v = g_direct9.load(std::memory_order_relaxed);
    if (v >= 0) return v != 0;
    static const int env = [] { const char *e = getenv("TF_MSDA_DIRECT9"); return (e && e[0] == '0') ? 0 : 1; }();
    return env != 0;
}

bool launch_direct(bool fused, const DirectArgs &da, const LevelTable &lt, const int64_t *shapes_dev,
                   int D, int P, hipStream_t stream, hipError_t *err)
{
    if (!direct_enabled() || P != 4 || da.L > 8) return false;
    const int lpairs = (da.L + 1) / 2;
    const void *fn = nullptr;
    if (D == 36) {   // 9 lanes per pair, 28 pairs per workgroup
        if (!direct9_enabled()) return false;
        if (fused)
            fn = lpairs == 1   ? (const void *)&msda_fwd_f32_direct9<1, true>
                 : lpairs == 2 ? (const void *)&msda_fwd_f32_direct9<2, true>
                 : lpairs == 3 ? (const void *)&msda_fwd_f32_direct9<3, true>
                               : (const void *)&msda_fwd_f32_direct9<4, true>;
        else
            fn = lpairs == 1   ? (const void *)&msda_fwd_f32_direct9<1, false>
                 : lpairs == 2 ? (const void *)&msda_fwd_f32_direct9<2, false>
                 : lpairs == 3 ? (const void *)&msda_fwd_f32_direct9<3, false>
                               : (const void *)&msda_fwd_f32_direct9<4, false>;
        const long long ppb = (kThreads / 64) * 7;
        const long long grid9 = (da.nlq + ppb - 1) / ppb * da.M;
        if (grid9 > 0x7fffffffLL) return false;
        *err = launch(fn, (unsigned)grid9, 0, stream, da, lt, shapes_dev);
        note_kernel(fused ? "msda_fwd_f32_direct9<fused>" : "msda_fwd_f32_direct9<plain>");
        return true;
    }
    if (D != 32) return false;
    if (fused)
        fn = lpairs == 1   ? (const void *)&msda_fwd_f32_direct<1, true>
             : lpairs == 2 ? (const void *)&msda_fwd_f32_direct<2, true>
             : lpairs == 3 ? (const void *)&msda_fwd_f32_direct<3, true>
                           : (const void *)&msda_fwd_f32_direct<4, true>;
    else
        fn = lpairs == 1   ? (const void *)&msda_fwd_f32_direct<1, false>
             : lpairs == 2 ? (const void *)&msda_fwd_f32_direct<2, false>
             : lpairs == 3 ? (const void *)&msda_fwd_f32_direct<3, false>
                           : (const void *)&msda_fwd_f32_direct<4, false>;
    const long long chunks = (da.nlq + (kThreads / 8) - 1) / (kThreads / 8);
    const long long grid = chunks * da.M;
    if (grid > 0x7fffffffLL) return false;
    *err = launch(fn, (unsigned)grid, 0, stream, da, lt, shapes_dev);
    note_kernel(fused ? "msda_fwd_f32_direct<fused>" : "msda_fwd_f32_direct<plain>");
    return true;
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes_host, const int64_t *shapes_dev,
                 const T *loc, const T *attn, T *out, int N, int S, int M, int D, int L, int Lq,
                 int P, void *stream_v)
{
    if (!value || !loc || !attn || !out || (!shapes_host && !shapes_dev))
        return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0 ||
        L > TF_MSDA_MAX_LEVELS)
        return TF_MSDA_ERR_BAD_DIMS;
    LevelTable lt{};
    if (shapes_host) {
        const int rc = build_level_table(shapes_host, L, S, &lt);
        if (rc != TF_MSDA_OK) return rc;
    }
    const size_t va = sizeof(T) * 4;
    const bool aligned = is_aligned(value, va) && is_aligned(out, va);
    Plan pl;
    const int rc = make_plan<T>(N, M, D, L, Lq, P, 3, aligned, &pl);
    if (rc != TF_MSDA_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const long long total_pairs = (long long)N * Lq * M;
    hipError_t e;
    if constexpr (sizeof(T) == 4) {
        if (pl.vec == 4 && (P == 1 || P == 2 || P == 4 || P == 8) &&
            buf_path_ok(lt, shapes_host != nullptr, N, S, M, D, L)) {
            const unsigned vbytes = (unsigned)((long long)N * S * M * D * 4);
            {
                DirectArgs da{};
                da.value = value;
                da.value_bytes = vbytes;
                da.loc = loc;
                da.attn = attn;
                da.out = out;
                da.S = S;
                da.M = M;
                da.L = L;
                da.Lq = Lq;
                da.nlq = (long long)N * Lq;
                if (is_aligned(loc, 8) && shapes_dev == nullptr && tiled_mode() == 2 &&
                    launch_pquad(false, da, lt, N, D, P, stream, &e))
                    return record_hip(e);
                if (is_aligned(loc, 8) && shapes_dev == nullptr && launch_quad(false, da, lt, N, D, P, stream, &e))
                    return record_hip(e);
                if (is_aligned(loc, 8) && launch_direct(false, da, lt, shapes_dev, D, P, stream, &e))
                    return record_hip(e);
            }
            const void *fn = P == 1   ? (const void *)&msda_fwd_f32_buf<1, false>
                             : P == 2 ? (const void *)&msda_fwd_f32_buf<2, false>
                             : P == 4 ? (const void *)&msda_fwd_f32_buf<4, false>
                                      : (const void *)&msda_fwd_f32_buf<8, false>;
            FusedArgs none{};
            none.head_major = head_major_enabled() ? 1 : 0;
            const unsigned grid = none.head_major ? head_major_grid(N, Lq, M, pl.ppb) : pl.grid;
            e = launch(fn, grid, pl.lds, stream, value, vbytes, loc, attn, out, lt, shapes_dev,
                       S, M, D, L, Lq, total_pairs, pl.ppb, pl.DV, none);
            note_kernel("msda_fwd_f32_buf<plain>");
            return record_hip(e);
        }
    }
    const void *fn = pl.vec == 4 ? (const void *)&msda_fwd_rowgather<T, 4>
                                 : (const void *)&msda_fwd_rowgather<T, 1>;
    e = launch(fn, pl.grid, pl.lds, stream, value, loc, attn, out, lt, shapes_dev, S, M, D, L, Lq,
               P, total_pairs, pl.ppb, pl.DV);
    note_kernel(sizeof(T) == 4 ? "msda_fwd_rowgather<f32>" : "msda_fwd_rowgather<f64>");
    return record_hip(e);
}

int forward_fused_impl(const float *value, const int64_t *shapes_host, const float *ref, int ref_dim,
                       const float *qproj, int ld, int off_col, int logit_col, float *out, int N,
                       int S, int M, int D, int L, int Lq, int P, void *stream_v)
{
    if (!value || !shapes_host || !ref || !qproj || !out) return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0 ||
        L > TF_MSDA_MAX_LEVELS || (ref_dim != 2 && ref_dim != 4))
        return TF_MSDA_ERR_BAD_DIMS;
    const int LP = L * P;
    if (off_col < 0 || logit_col < 0 || (off_col & 1) || ld < off_col + M * LP * 2 ||
        ld < logit_col + M * LP)
        return TF_MSDA_ERR_BAD_DIMS;
    if ((P != 1 && P != 2 && P != 4 && P != 8) || (D & 3)) return TF_MSDA_ERR_BAD_DIMS;
    LevelTable lt{};
    int rc = build_level_table(shapes_host, L, S, &lt);
    if (rc != TF_MSDA_OK) return rc;
    if (!is_aligned(value, 16) || !is_aligned(out, 16) || !is_aligned(qproj, 8) || (ld & 1) ||
        !buf_path_ok(lt, true, N, S, M, D, L))
        return TF_MSDA_ERR_BAD_DIMS;
    Plan pl;
    rc = make_plan<float>(N, M, D, L, Lq, P, 3, true, &pl);
    if (rc != TF_MSDA_OK) return rc;
    const unsigned vbytes = (unsigned)((long long)N * S * M * D * 4);
    const long long total_pairs = (long long)N * Lq * M;
    {
        DirectArgs da{};
        da.value = value;
        da.value_bytes = vbytes;
        da.out = out;
        da.fa = FusedArgs{ref, qproj, ref_dim, ld, off_col, logit_col, 1};
        da.S = S;
        da.M = M;
        da.L = L;
        da.Lq = Lq;
        da.nlq = (long long)N * Lq;
        hipError_t de;
        if (tiled_mode() == 2 && launch_pquad(true, da, lt, N, D, P, static_cast<hipStream_t>(stream_v), &de))
            return record_hip(de);
        if (launch_quad(true, da, lt, N, D, P, static_cast<hipStream_t>(stream_v), &de))
            return record_hip(de);
        if (launch_direct(true, da, lt, nullptr, D, P, static_cast<hipStream_t>(stream_v), &de))
            return record_hip(de);
    }
    const void *fn = P == 1   ? (const void *)&msda_fwd_f32_buf<1, true>
                     : P == 2 ? (const void *)&msda_fwd_f32_buf<2, true>
                     : P == 4 ? (const void *)&msda_fwd_f32_buf<4, true>
                              : (const void *)&msda_fwd_f32_buf<8, true>;
    const int hm = head_major_enabled() ? 1 : 0;
    const FusedArgs fa{ref, qproj, ref_dim, ld, off_col, logit_col, hm};
    const float *nul = nullptr;
    const int64_t *nod = nullptr;
    const unsigned grid = hm ? head_major_grid(N, Lq, M, pl.ppb) : pl.grid;
    const hipError_t e = launch(fn, grid, pl.lds, static_cast<hipStream_t>(stream_v), value,
                                vbytes, nul, nul, out, lt, nod, S, M, D, L, Lq, total_pairs, pl.ppb,
                                pl.DV, fa);
    note_kernel("msda_fwd_f32_buf<fused>");
    return record_hip(e);
}

template <typename W>
__global__ void __launch_bounds__(256) zero_words_kernel(W *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) p[i] = W(0);
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes_host, const int64_t *shapes_dev,
                  const T *loc, const T *attn, const T *grad_out, T *grad_value, T *grad_loc,
                  T *grad_attn, int N, int S, int M, int D, int L, int Lq, int P, void *stream_v)
{
    if (!value || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn ||
        (!shapes_host && !shapes_dev))
        return TF_MSDA_ERR_NULL_POINTER;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0 ||
        L > TF_MSDA_MAX_LEVELS)
        return TF_MSDA_ERR_BAD_DIMS;
    LevelTable lt{};
    if (shapes_host) {
        const int rc = build_level_table(shapes_host, L, S, &lt);
        if (rc != TF_MSDA_OK) return rc;
    }
    const size_t va = sizeof(T) * 4;
    const bool aligned = is_aligned(value, va) && is_aligned(grad_out, va);
    Plan pl;
    int rc = make_plan<T>(N, M, D, L, Lq, P, 6, aligned, &pl);
    if (rc != TF_MSDA_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    // grad_value is accumulated by atomics: zeros first -- by a KERNEL, not hipMemsetAsync.  Inside a captured HIP graph a memset
    // node in front of a kernel node was seen NOT to be ordered before that kernel's atomics on ROCm 7.2 (the group-norm
    // statistics of the two-graph detector, profiles/r05_graph_memset_groupnorm.txt; and this very call: a captured backward
    // replayed onto uninitialised memory, tests/test_dropin_compiled_gpu.py).  Kernel -> kernel order is an ordinary edge.
    {
        const size_t bytes = sizeof(T) * (size_t)N * S * M * D;
        const bool wide = bytes % 16 == 0 && is_aligned(grad_value, 16);
        const size_t n = wide ? bytes / 16 : bytes / 4;   // (T is float or double: a multiple of 4 bytes)
        size_t blocks = (n + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;   // grid-stride above 8192 workgroups
        if (wide)
            hipLaunchKernelGGL(zero_words_kernel<u32x4_t>, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<u32x4_t *>(grad_value), n);
        else
            hipLaunchKernelGGL(zero_words_kernel<unsigned>, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<unsigned *>(grad_value), n);
        rc = record_hip(hipGetLastError());
        if (rc != TF_MSDA_OK) return rc;
    }
    const long long total_pairs = (long long)N * Lq * M;
    const bool pow2 = (pl.DV & (pl.DV - 1)) == 0 && pl.DV <= 64;
    if constexpr (sizeof(T) == 4) {
        if (pl.vec == 4 && pow2 && (P == 1 || P == 2 || P == 4 || P == 8) &&
            buf_path_ok(lt, shapes_host != nullptr, N, S, M, D, L) && is_aligned(grad_value, 16)) {
            const unsigned vbytes = (unsigned)((long long)N * S * M * D * 4);
            WinGeom sgeom;
            if (shapes_host && Lq == S && is_aligned(loc, 8) && is_aligned(grad_loc, 8) &&
                plan_sorted(lt, L, D, P, &sgeom) &&
                (long long)N * sgeom.tiles_y * sgeom.tiles_x * M <= 0x7fffffffLL) {
                BwdSortArgs ba{value, vbytes, loc, attn, grad_out, grad_value, grad_loc, grad_attn, S, M, L};
                const void *sfn = (const void *)&msda_bwd_f32_sorted2;
                const size_t slds = (size_t)kSort2LdsBytes;
                if (slds > 64 * 1024 && !raise_dynamic_lds(sfn)) return record_hip(hipErrorInvalidValue);
                void *argv[] = {(void *)&ba, (void *)&lt, (void *)&sgeom};
                const unsigned sgrid = (unsigned)((long long)N * sgeom.tiles_y * sgeom.tiles_x * M);
                note_kernel("msda_bwd_f32_sorted2");
                return record_hip(hipLaunchKernel(sfn, dim3(sgrid), dim3(kWinThreads), argv, slds, stream));
            }
            static const int rowatom_on = [] { const char *e = getenv("TF_MSDA_BWD_ROWATOM"); return (e && e[0] == '0') ? 0 : 1; }();
            const bool rowatom = rowatom_on && D == 32 && pl.DV == 8 && pl.ppb == kThreads / 8;
            const void *bfn =
                rowatom ? (P == 1   ? (const void *)&msda_bwd_f32_buf<1, true>
                           : P == 2 ? (const void *)&msda_bwd_f32_buf<2, true>
                           : P == 4 ? (const void *)&msda_bwd_f32_buf<4, true>
                                    : (const void *)&msda_bwd_f32_buf<8, true>)
                        : (P == 1   ? (const void *)&msda_bwd_f32_buf<1, false>
                           : P == 2 ? (const void *)&msda_bwd_f32_buf<2, false>
                           : P == 4 ? (const void *)&msda_bwd_f32_buf<4, false>
                                    : (const void *)&msda_bwd_f32_buf<8, false>);
            if (rowatom) pl.lds = ((pl.lds + 15) & ~(size_t)15) + (size_t)pl.ppb * L * P * 32;
            const hipError_t be = launch(bfn, head_major_grid(N, Lq, M, pl.ppb), pl.lds, stream,
                                         value, vbytes, loc, attn, grad_out, grad_value, grad_loc,
                                         grad_attn, lt, shapes_dev, S, M, D, L, Lq, total_pairs,
                                         pl.ppb, pl.DV);
            note_kernel(rowatom ? "msda_bwd_f32_buf<rowatom>" : "msda_bwd_f32_buf");
            return record_hip(be);
        }
    }
    const void *fn;
    if (pl.vec == 4)
        fn = pow2 ? (const void *)&msda_bwd_rowgather<T, 4, true>
                  : (const void *)&msda_bwd_rowgather<T, 4, false>;
    else
        fn = pow2 ? (const void *)&msda_bwd_rowgather<T, 1, true>
                  : (const void *)&msda_bwd_rowgather<T, 1, false>;
    const hipError_t e = launch(fn, pl.grid, pl.lds, stream, value, loc, attn, grad_out, grad_value,
                                grad_loc, grad_attn, lt, shapes_dev, S, M, D, L, Lq, P, total_pairs,
                                pl.ppb, pl.DV);
    note_kernel(sizeof(T) == 4 ? "msda_bwd_rowgather<f32>" : "msda_bwd_rowgather<f64>");
    return record_hip(e);
}

}  // namespace

namespace tfm {
void note_kernel(const char *name) { g_last_kernel = name; }
}  // namespace tfm

// ---------------------------------------------------------------------------------------------
// C ABI (include/tf_msda.h)
// ---------------------------------------------------------------------------------------------
extern "C" {

int tf_msda_abi_version(void) { return TF_MSDA_ABI_VERSION; }

const char *tf_msda_last_kernel(void) { return g_last_kernel; }

const char *tf_msda_strerror(int status)
{
    switch (status) {
    case TF_MSDA_OK: return "ok";
    case TF_MSDA_ERR_NULL_POINTER: return "a required pointer was NULL";
    case TF_MSDA_ERR_BAD_DIMS: return "invalid dimension (<=0, too many levels, or too large)";
    case TF_MSDA_ERR_SHAPE_SUM: return "sum of H_l*W_l over levels does not equal S";
    case TF_MSDA_ERR_LAUNCH: return "HIP error while enqueueing work";
    case TF_MSDA_ERR_NO_DEVICE: return "no HIP device available";
    default: return "unknown tf_msda status";
    }
}

int tf_msda_last_hip_error(void) { return g_last_hip_error; }

int tf_msda_set_tiled(int mode)
{
    return g_tiled_mode.exchange(mode < 0 ? -1 : (mode ? 2 : 0));
}

void tf_msda_debug_trace_buffer(void *device_buffer)
{
    g_quad_trace.store(static_cast<unsigned long long *>(device_buffer));
    pquad_set_trace(static_cast<unsigned long long *>(device_buffer));
}

int tf_msda_set_option(const char *name, int value)
{
    if (!name) return INT_MIN;
    if (strcmp(name, "tiled") == 0) return tf_msda_set_tiled(value);
    quad_opts_init();
    for (int i = 0; i < kQoCount; ++i)
        if (strcmp(name, kQuadOptNames[i]) == 0) {
            const int prev = g_quad_opt[i].exchange(value);
            g_quad_epoch.fetch_add(1);
            return prev;
        }
    if (strcmp(name, "direct9") == 0) return g_direct9.exchange(value < 0 ? -1 : (value ? 1 : 0));
    if (strcmp(name, "ffn_ti") == 0) return ffn_set_ti(value);
    if (strcmp(name, "ffn_tail_split") == 0) return ffn_set_tail_split(value);
    if (strcmp(name, "linln_ti") == 0) return linln_set_ti(value);
    if (strcmp(name, "linear_stream_ti") == 0) return linear_stream_set_ti(value);
    if (strcmp(name, "conv_halo") == 0) return conv_halo_set(value);
    if (strcmp(name, "linear_dma") == 0) return linear_dma_set(value);
    if (strcmp(name, "mha_mfma") == 0) return mha_set_mfma(value);
    if (strncmp(name, "pquad", 5) == 0) {
        const int prev = pquad_set_option(name, value);
        return prev == -1 ? INT_MIN : prev;
    }
    return INT_MIN;
}

int tf_msda_forward_fused_f32(const float *value, const int64_t *shapes_hw_host,
                              const float *ref_points, int ref_dim, const float *qproj, int ld,
                              int off_col, int logit_col, float *out, int N, int S, int M, int D,
                              int L, int Lq, int P, void *stream)
{
    return forward_fused_impl(value, shapes_hw_host, ref_points, ref_dim, qproj, ld, off_col,
                              logit_col, out, N, S, M, D, L, Lq, P, stream);
}

int tf_msda_forward_f32(const float *value, const int64_t *shapes_hw_host, const float *loc,
                        const float *attn, float *out, int N, int S, int M, int D, int L, int Lq,
                        int P, void *stream)
{
    if (!shapes_hw_host) return TF_MSDA_ERR_NULL_POINTER;
    return forward_impl<float>(value, shapes_hw_host, nullptr, loc, attn, out, N, S, M, D, L, Lq, P,
                               stream);
}
int tf_msda_forward_f64(const double *value, const int64_t *shapes_hw_host, const double *loc,
                        const double *attn, double *out, int N, int S, int M, int D, int L, int Lq,
                        int P, void *stream)
{
    if (!shapes_hw_host) return TF_MSDA_ERR_NULL_POINTER;
    return forward_impl<double>(value, shapes_hw_host, nullptr, loc, attn, out, N, S, M, D, L, Lq,
                                P, stream);
}
int tf_msda_forward_f32_dshapes(const float *value, const int64_t *shapes_hw_dev, const float *loc,
                                const float *attn, float *out, int N, int S, int M, int D, int L,
                                int Lq, int P, void *stream)
{
    if (!shapes_hw_dev) return TF_MSDA_ERR_NULL_POINTER;
    return forward_impl<float>(value, nullptr, shapes_hw_dev, loc, attn, out, N, S, M, D, L, Lq, P,
                               stream);
}
int tf_msda_forward_f64_dshapes(const double *value, const int64_t *shapes_hw_dev,
                                const double *loc, const double *attn, double *out, int N, int S,
                                int M, int D, int L, int Lq, int P, void *stream)
{
    if (!shapes_hw_dev) return TF_MSDA_ERR_NULL_POINTER;
    return forward_impl<double>(value, nullptr, shapes_hw_dev, loc, attn, out, N, S, M, D, L, Lq, P,
                                stream);
}

int tf_msda_backward_f32(const float *value, const int64_t *shapes_hw_host, const float *loc,
                         const float *attn, const float *grad_out, float *grad_value,
                         float *grad_loc, float *grad_attn, int N, int S, int M, int D, int L,
                         int Lq, int P, void *stream)
{
    if (!shapes_hw_host) return TF_MSDA_ERR_NULL_POINTER;
    return backward_impl<float>(value, shapes_hw_host, nullptr, loc, attn, grad_out, grad_value,
                                grad_loc, grad_attn, N, S, M, D, L, Lq, P, stream);
}
int tf_msda_backward_f64(const double *value, const int64_t *shapes_hw_host, const double *loc,
                         const double *attn, const double *grad_out, double *grad_value,
                         double *grad_loc, double *grad_attn, int N, int S, int M, int D, int L,
                         int Lq, int P, void *stream)
{
    if (!shapes_hw_host) return TF_MSDA_ERR_NULL_POINTER;
    return backward_impl<double>(value, shapes_hw_host, nullptr, loc, attn, grad_out, grad_value,
                                 grad_loc, grad_attn, N, S, M, D, L, Lq, P, stream);
}
int tf_msda_backward_f32_dshapes(const float *value, const int64_t *shapes_hw_dev, const float *loc,
                                 const float *attn, const float *grad_out, float *grad_value,
                                 float *grad_loc, float *grad_attn, int N, int S, int M, int D,
                                 int L, int Lq, int P, void *stream)
{
    if (!shapes_hw_dev) return TF_MSDA_ERR_NULL_POINTER;
    return backward_impl<float>(value, nullptr, shapes_hw_dev, loc, attn, grad_out, grad_value,
                                grad_loc, grad_attn, N, S, M, D, L, Lq, P, stream);
}
int tf_msda_backward_f64_dshapes(const double *value, const int64_t *shapes_hw_dev,
                                 const double *loc, const double *attn, const double *grad_out,
                                 double *grad_value, double *grad_loc, double *grad_attn, int N,
                                 int S, int M, int D, int L, int Lq, int P, void *stream)
{
    if (!shapes_hw_dev) return TF_MSDA_ERR_NULL_POINTER;
    return backward_impl<double>(value, nullptr, shapes_hw_dev, loc, attn, grad_out, grad_value,
                                 grad_loc, grad_attn, N, S, M, D, L, Lq, P, stream);
}

}  // extern "C"
