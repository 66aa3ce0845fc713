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
// Kernel design (v1, "row gather"):
//   * A "pair" is one (batch, query, head) triple: it owns L*P sampling points and one D-float
//     output row.  pairs are contiguous in loc/attn/out memory, so a workgroup that owns
//     `ppb` consecutive pairs reads ONE contiguous chunk of loc and attn (fully coalesced) into
//     LDS and writes ONE contiguous chunk of out.
//   * Inside a pair, D/VEC lanes each own VEC (=4) consecutive channels, so a bilinear tap is one
//     16-byte load per lane and the D/VEC lanes of a pair together read one contiguous
//     D*sizeof(T)-byte row of `value` (128 B for D=32: exactly one cache line).  With M=8, D=32 a
//     64-lane wavefront is exactly one query (8 heads x 8 lanes).
//   * Every tap address is clamped into the level, so all 4*L*P loads of a lane are unconditional
//     and independent (deep memory-level parallelism, no divergent branches); validity is applied
//     with selects on the loaded values (bit-exact zero padding, no 0*Inf leaks).
//   * Level geometry lives in a 192-byte LDS table filled either from the kernel arguments
//     (host-shape entry points) or from the reference's device-resident int64 tensor
//     (..._dshapes entry points) -- never a host<->device sync.
//   * Backward fuses the reference's two kernels: the D-reduction for grad_loc / grad_attn is a
//     wave shuffle (xor butterfly over the D/VEC lanes of a pair) instead of a serial channel loop,
//     results are staged in LDS and written back coalesced; grad_value uses hardware fp atomics.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "tf_msda.h"

namespace {

constexpr int kThreads = 256;          // 4 wavefronts per workgroup
constexpr int kLdsChunkBudget = 48 * 1024;  // LDS bytes for the loc/attn (and grad) chunk

struct LevelTable {
    int H[TF_MSDA_MAX_LEVELS];
    int W[TF_MSDA_MAX_LEVELS];
    int start[TF_MSDA_MAX_LEVELS];
};
constexpr int kLevelTableBytes = 3 * TF_MSDA_MAX_LEVELS * (int)sizeof(int);  // 192, multiple of 16

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

thread_local int g_last_hip_error = 0;

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
    // loc*size - 0.5 with a single rounding == the reference's double-literal expression narrowed
    // to T (cuh:227-228): the product is exact in double for any float loc and int size.
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
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0xFFFFFFF0u;  // >= num_records for every supported tensor
constexpr unsigned kOobBase = 0xFFFFFF00u;    // ... and so is kOobBase + (lane slice offset < 0xF0)

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
struct FusedArgs {
    const float *ref;     // [N, Lq, L, ref_dim]
    const float *qproj;   // [N*Lq, ld]: per query, M*L*P*2 raw offsets at off_col, M*L*P logits at logit_col
    int ref_dim, ld, off_col, logit_col;
    int head_major;       // block -> pair mapping, see msda_fwd_f32_buf
};

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
struct DirectArgs {
    const float *value;
    unsigned value_bytes;
    const float *loc, *attn;   // plain operator inputs (FUSED == false)
    float *out;
    FusedArgs fa;              // FUSED == true
    int S, M, L, Lq;
    long long nlq;             // N * Lq
};

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
// forward, encoder self-attention shape (Lq == S): 2-D query tiles + LDS-staged sampling windows
// ---------------------------------------------------------------------------------------------
// In the encoder every pyramid pixel is a query and samples a small neighbourhood of its own position
// in each level, so a value row is fetched ~64 times (16 samples x 4 taps per query and head).  Through
// the vector-memory path that is 1.46 GB of 128-byte row gathers per launch at <= 64 B/clk/CU (measured:
// TA busy 16 cycles per dwordx4 wave-load); LDS serves the same gathers at 256 B/clk/CU.  This kernel
//   * forms 512-thread workgroups from 2-D tiles of queries: a tile is a TH x TW rectangle of level-0
//     pixels together with the pixels of every other level whose centres fall into the same
//     normalised rectangle (an exact partition of all S queries), for ONE head (blockIdx % M: with
//     M == 8 each XCD serves one head, whose 2.8 MB of value rows stay in that XCD's 4 MiB L2);
//   * gives each (query, head) pair to TWO lanes (even / odd 16-byte channel slices): the tap
//     arithmetic is done twice per sampling point instead of D/4 times as in the row-gather kernels,
//     and 8 wavefronts per CU keep every SIMD two-deep;
//   * walks the value levels with two LDS windows: while level l is gathered from one window the
//     tile's nominal window of level l+1 (tile extent mapped into the level plus a halo) streams into
//     the other one by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no LDS store
//     instructions).  Rows are stored with a stride of an ODD number of 16-byte slots (144 B for
//     D = 32) so that ds_read_b128 of one slice of neighbouring rows is bank-conflict free;
//   * taps outside the level read a zero row kept in LDS (zero padding with no selects); sampling
//     points whose taps leave the staged window -- possible for any input, the window is only a
//     guess -- take buffer loads under a wave-uniform branch.
// Correctness never depends on the tile/window/halo heuristics (tests sweep adversarial inputs).
constexpr int kV4Threads = 512;
constexpr int kV4Waves = kV4Threads / 64;
constexpr int kV4Pairs = kV4Threads / 2;     // queries per tile (two lanes each)

struct TileGeom {
    int TH, TW;        // tile size in level-0 pixels
    int HY, HX;        // window halo in pixels (every level)
    int tiles_y, tiles_x;
    int cap_even, cap_odd;   // LDS window capacities in rows: even levels use buffer A, odd ones B
    int debug;               // timing experiments only (wrong results): 1 = skip staging, 2 = skip gather
};

template <int PT, int NCH>   // NCH = D / 4 sixteen-byte channel slices per row
__global__ void __launch_bounds__(kV4Threads)
msda_fwd_f32_tiled(const float *__restrict__ value, unsigned value_bytes,
                   const float *__restrict__ loc, const float *__restrict__ attn,
                   float *__restrict__ out, const LevelTable lt,
                   const int64_t *__restrict__ dshapes, int S, int M, int L, const TileGeom tg)
{
    constexpr int D = NCH * 4;
    constexpr int kSlots = (NCH % 2) ? NCH : NCH + 1;      // odd number of 16-byte slots per LDS row
    constexpr unsigned kStride = kSlots * 16u;             // LDS row stride in bytes
    constexpr int kMine = (NCH + 1) / 2;                   // slices per lane (even lane gets the extra one)
    static_assert(PT == 4, "tiled kernel is written for 4 sampling points per level");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);                    // H | W | start   (192 B)
    int *s_q = s_tab + 3 * TF_MSDA_MAX_LEVELS;                     // ya | yb | xa | xb | qoff(17)
    constexpr int kQInts = 5 * TF_MSDA_MAX_LEVELS + 4;             // 84 ints -> header = 528 B
    unsigned char *s_rows = reinterpret_cast<unsigned char *>(s_q + kQInts);
    // s_rows: [zero row][window A: cap_even rows][window B: cap_odd rows], all with stride kStride

    const int m = blockIdx.x % M;
    int t = blockIdx.x / M;
    const int tx = t % tg.tiles_x;
    t /= tg.tiles_x;
    const int ty = t % tg.tiles_y;
    const int b = t / tg.tiles_y;

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
    if (threadIdx.x < kSlots * 4) reinterpret_cast<float *>(s_rows)[threadIdx.x] = 0.f;
    __syncthreads();

    const int H0 = s_tab[0], W0 = s_tab[TF_MSDA_MAX_LEVELS];
    const int y0t = ty * tg.TH, y1t = min(H0, y0t + tg.TH);
    const int x0t = tx * tg.TW, x1t = min(W0, x0t + tg.TW);
    if (threadIdx.x == 0) {
        // pixels of level l whose centre lies in [y0t/H0, y1t/H0) x [x0t/W0, x1t/W0): integer exact
        int acc = 0;
        for (int l = 0; l < L; ++l) {
            const int Hl = s_tab[l], Wl = s_tab[TF_MSDA_MAX_LEVELS + l];
            const int ya = (int)((2LL * y0t * Hl + H0 - 1) / (2LL * H0));
            const int yb = (int)((2LL * y1t * Hl + H0 - 1) / (2LL * H0));
            const int xa = (int)((2LL * x0t * Wl + W0 - 1) / (2LL * W0));
            const int xb = (int)((2LL * x1t * Wl + W0 - 1) / (2LL * W0));
            s_q[l] = ya;
            s_q[TF_MSDA_MAX_LEVELS + l] = yb;
            s_q[2 * TF_MSDA_MAX_LEVELS + l] = xa;
            s_q[3 * TF_MSDA_MAX_LEVELS + l] = xb;
            s_q[4 * TF_MSDA_MAX_LEVELS + l] = acc;
            acc += (yb - ya) * (xb - xa);
        }
        s_q[4 * TF_MSDA_MAX_LEVELS + L] = acc;
    }
    __syncthreads();
    const int nq = s_q[4 * TF_MSDA_MAX_LEVELS + L];   // <= kV4Pairs (host computed the maximum)

    // this lane's query (tile-local index = thread / 2; level-major, row-major inside the tile)
    const int tq = threadIdx.x >> 1, half = threadIdx.x & 1;
    int q = -1;
    if (tq < nq) {
        int l = 0;
        while (l + 1 < L && tq >= s_q[4 * TF_MSDA_MAX_LEVELS + l + 1]) ++l;
        const int r = tq - s_q[4 * TF_MSDA_MAX_LEVELS + l];
        const int nx = s_q[3 * TF_MSDA_MAX_LEVELS + l] - s_q[2 * TF_MSDA_MAX_LEVELS + l];
        const int yy = s_q[l] + r / nx, xx = s_q[2 * TF_MSDA_MAX_LEVELS + l] + r % nx;
        q = s_tab[2 * TF_MSDA_MAX_LEVELS + l] + yy * s_tab[TF_MSDA_MAX_LEVELS + l] + xx;
    }
    const long long pair = ((long long)b * S + max(q, 0)) * M + m;
    const int LP = L * PT;
    const float *lp_base = loc + pair * LP * 2;
    const float *ap_base = attn + pair * LP;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned rowbytes = (unsigned)(M * D) * 4u;
    const unsigned head_base = (unsigned)((((long long)b * S * M + m) * D) * 4);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, value_bytes, 0x00020000);

    struct Geom {
        int H, W, wy0, wx0, wy1, wx1, ww, nrows;
        unsigned lvl_base, win_off;   // win_off: byte offset of the level's window inside s_rows
    };
    auto level_geom = [&](int l) {
        Geom g;
        g.H = s_tab[l];
        g.W = s_tab[TF_MSDA_MAX_LEVELS + l];
        int wy0 = (int)__builtin_floorf((float)y0t * g.H / H0 - 0.5f) - tg.HY;
        int wy1 = (int)__builtin_floorf((float)y1t * g.H / H0 - 0.5f) + 1 + tg.HY;
        int wx0 = (int)__builtin_floorf((float)x0t * g.W / W0 - 0.5f) - tg.HX;
        int wx1 = (int)__builtin_floorf((float)x1t * g.W / W0 - 0.5f) + 1 + tg.HX;
        g.wy0 = max(wy0, 0);
        g.wx0 = max(wx0, 0);
        wy1 = min(wy1, g.H - 1);
        g.wx1 = min(wx1, g.W - 1);
        g.ww = g.wx1 - g.wx0 + 1;
        const int cap = (l & 1) ? tg.cap_odd : tg.cap_even;
        int wh = wy1 - g.wy0 + 1;
        if (wh * g.ww > cap) wh = cap / g.ww;            // cap >= ww is ensured by the host
        g.wy1 = g.wy0 + wh - 1;
        g.nrows = wh * g.ww;
        g.lvl_base = head_base + (unsigned)s_tab[2 * TF_MSDA_MAX_LEVELS + l] * rowbytes;
        g.win_off = kStride * (1u + ((l & 1) ? (unsigned)tg.cap_even : 0u));
        return g;
    };

    // window of level l -> LDS by LDS-DMA: one wave-instruction moves 64 sixteen-byte slots (1 KiB of
    // LDS, contiguous) gathered from 64 lane-supplied global offsets; padding slots and slots past
    // the window get an out-of-range offset (the hardware writes zeros).
    auto issue_window_dma = [&](const Geom &g) {
        if (tg.debug == 1) return;
        const int nslots = g.nrows * kSlots;
        const float inv_ww = 1.0f / (float)g.ww;
        for (int chunk = wave; chunk * 64 < nslots; chunk += kV4Waves) {
            const int sidx = chunk * 64 + lane;
            const int row = sidx / kSlots, c = sidx - row * kSlots;
            int wy = (int)(((float)row + 0.5f) * inv_ww);      // exact for row < 2^22
            const int wx = row - wy * g.ww;
            const unsigned off = (row < g.nrows && c < NCH)
                ? g.lvl_base + (unsigned)((g.wy0 + wy) * g.W + g.wx0 + wx) * rowbytes + (unsigned)c * 16u
                : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void *)(s_rows + g.win_off + chunk * 1024),
                16, off, 0, 0, 0);
        }
    };

    f32x4_t acc[kMine];
#pragma unroll
    for (int k = 0; k < kMine; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    Geom gnext = level_geom(0);
    f32x4_t lxy0 = *reinterpret_cast<const f32x4_t *>(lp_base);
    f32x4_t lxy1 = *reinterpret_cast<const f32x4_t *>(lp_base + 4);
    f32x4_t law = *reinterpret_cast<const f32x4_t *>(ap_base);
    issue_window_dma(gnext);
    for (int l = 0; l < L; ++l) {
        const Geom g = gnext;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA + location loads landed
        __syncthreads();      // ... everybody's; and everybody is done gathering level l-1
        const f32x4_t cxy0 = lxy0, cxy1 = lxy1, caw = law;
        if (l + 1 < L) {      // next level streams into the other window while this one is gathered
            gnext = level_geom(l + 1);
            lxy0 = *reinterpret_cast<const f32x4_t *>(lp_base + (size_t)(l + 1) * PT * 2);
            lxy1 = *reinterpret_cast<const f32x4_t *>(lp_base + (size_t)(l + 1) * PT * 2 + 4);
            law = *reinterpret_cast<const f32x4_t *>(ap_base + (size_t)(l + 1) * PT);
            issue_window_dma(gnext);
        }
        if (q < 0 || tg.debug == 2) continue;  // no barrier below this point inside the iteration

        const float Wf = (float)g.W, Hf = (float)g.H;
        const float lx[4] = {cxy0.x, cxy0.z, cxy1.x, cxy1.z};
        const float ly[4] = {cxy0.y, cxy0.w, cxy1.y, cxy1.w};
        const float aw[4] = {caw.x, caw.y, caw.z, caw.w};
        const unsigned char *rbase = s_rows + half * 16;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float xr = __builtin_fmaf(lx[p], Wf, -0.5f);   // cuh:227-228, single rounding
            const float yr = __builtin_fmaf(ly[p], Hf, -0.5f);
            const bool in = (yr > -1.f) && (xr > -1.f) && (yr < Hf) && (xr < Wf);  // cuh:229
            const float x = in ? xr : 0.f, y = in ? yr : 0.f;
            const float xf = __builtin_floorf(x), yf = __builtin_floorf(y);
            const float fx = x - xf, fy = y - yf, gx = 1.f - fx, gy = 1.f - fy;
            const int x0 = (int)xf, y0 = (int)yf;
            const bool kx0 = in && (x0 >= 0), kx1 = in && (x0 + 1 <= g.W - 1);
            const bool ky0 = in && (y0 >= 0), ky1 = in && (y0 + 1 <= g.H - 1);
            // lowest / highest VALID tap coordinate must lie inside the staged window
            const int xlo = kx0 ? x0 : x0 + 1, xhi = kx1 ? x0 + 1 : x0;
            const int ylo = ky0 ? y0 : y0 + 1, yhi = ky1 ? y0 + 1 : y0;
            const bool staged = !in || (xlo >= g.wx0 && xhi <= g.wx1 && ylo >= g.wy0 && yhi <= g.wy1);
            const float w1 = gy * gx * aw[p], w2 = gy * fx * aw[p];
            const float w3 = fy * gx * aw[p], w4 = fy * fx * aw[p];
            const unsigned o = g.win_off + (unsigned)((y0 - g.wy0) * g.ww + (x0 - g.wx0)) * kStride;
            const unsigned o1 = (staged && ky0 && kx0) ? o : 0u;                 // 0 = the zero row
            const unsigned o2 = (staged && ky0 && kx1) ? o + kStride : 0u;
            const unsigned o3 = (staged && ky1 && kx0) ? o + (unsigned)g.ww * kStride : 0u;
            const unsigned o4 = (staged && ky1 && kx1) ? o + (unsigned)(g.ww + 1) * kStride : 0u;
#pragma unroll
            for (int k = 0; k < kMine; ++k) {
                if (2 * k + 1 >= NCH && half) break;     // odd lane has one slice less when NCH is odd
                const f32x4_t v1 = *reinterpret_cast<const f32x4_t *>(rbase + o1 + k * 32);
                const f32x4_t v2 = *reinterpret_cast<const f32x4_t *>(rbase + o2 + k * 32);
                const f32x4_t v3 = *reinterpret_cast<const f32x4_t *>(rbase + o3 + k * 32);
                const f32x4_t v4 = *reinterpret_cast<const f32x4_t *>(rbase + o4 + k * 32);
                acc[k] += v1 * w1;
                acc[k] += v2 * w2;
                acc[k] += v3 * w3;
                acc[k] += v4 * w4;
            }
            if (__any(in && !staged)) {   // rare: the point left the window -> global gather for it
                const bool gl = in && !staged;
                const int r0 = y0 * g.W + x0;
                const unsigned hb = (unsigned)half * 16u;
                const unsigned b1 = (gl && ky0 && kx0) ? g.lvl_base + (unsigned)r0 * rowbytes + hb : kOobOffset;
                const unsigned b2 = (gl && ky0 && kx1) ? g.lvl_base + (unsigned)(r0 + 1) * rowbytes + hb : kOobOffset;
                const unsigned b3 = (gl && ky1 && kx0) ? g.lvl_base + (unsigned)(r0 + g.W) * rowbytes + hb : kOobOffset;
                const unsigned b4 = (gl && ky1 && kx1) ? g.lvl_base + (unsigned)(r0 + g.W + 1) * rowbytes + hb : kOobOffset;
#pragma unroll
                for (int k = 0; k < kMine; ++k) {
                    if (2 * k + 1 >= NCH && half) break;
                    acc[k] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b1 == kOobOffset ? b1 : b1 + k * 32, 0, 0)) * w1;
                    acc[k] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b2 == kOobOffset ? b2 : b2 + k * 32, 0, 0)) * w2;
                    acc[k] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b3 == kOobOffset ? b3 : b3 + k * 32, 0, 0)) * w3;
                    acc[k] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b4 == kOobOffset ? b4 : b4 + k * 32, 0, 0)) * w4;
                }
            }
        }
    }
    if (q >= 0) {
        float *op = out + pair * D + half * 4;
#pragma unroll
        for (int k = 0; k < kMine; ++k) {
            if (2 * k + 1 >= NCH && half) break;
            *reinterpret_cast<f32x4_t *>(op + k * 8) = acc[k];
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
//     instruction touches D/4 consecutive floats of each row instead of every fourth float.
template <int PT>
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
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned lb = (unsigned)(dv + c * DV) * 4u;
                    const float top = gB[c] * a;
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w1 * top, rsrc_g, k1 ? t1 + lb : kOobOffset, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w2 * top, rsrc_g, k2 ? t2 + lb : kOobOffset, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w3 * top, rsrc_g, k3 ? t3 + lb : kOobOffset, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(w4 * top, rsrc_g, k4 ? t4 + lb : kOobOffset, 0, 0);
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

// The LDS-tiled encoder kernel is opt-in (tf_msda_set_tiled(1) or TF_MSDA_TILED=1): on MI355X it is
// currently slower than the row-gather kernel (81 vs 58 us at the cfg-2 encoder shape).
int g_tiled_mode = -1;   // -1: follow the environment, 0: off, 1: on
bool tiled_enabled()
{
    if (g_tiled_mode >= 0) return g_tiled_mode != 0;
    static const int env_on = [] { const char *e = getenv("TF_MSDA_TILED"); return (e && e[0] == '1') ? 1 : 0; }();
    return env_on != 0;
}

// Tile / window plan of msda_fwd_f32_tiled.  Returns false when the shape does not suit the kernel
// (then the row-gather kernels are used).  Everything here is a performance heuristic.
constexpr size_t kTiledLdsBudget = 160 * 1024;   // one 512-thread workgroup per CU owns the whole LDS
constexpr int kTiledHeaderBytes = kLevelTableBytes + (5 * TF_MSDA_MAX_LEVELS + 4) * (int)sizeof(int);

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

bool plan_tiles(const LevelTable &lt, int L, int D, TileGeom *tg, size_t *lds)
{
    if (!tiled_enabled()) return false;
    if (D != 32 && D != 36) return false;      // instantiated row widths (hidden 256 / 288, 8 heads)
    const int nch = D / 4, slots = (nch % 2) ? nch : nch + 1;
    const size_t row = (size_t)slots * 16;
    int hy = 3, hx = 7;   // default halo: the (H,W)-divisor quirk stretches x offsets by W/H (~1.67)
    int th = 0, tw = 16;
    if (const char *e = getenv("TF_MSDA_HALO")) sscanf(e, "%d,%d", &hy, &hx);
    if (const char *e = getenv("TF_MSDA_TILE")) sscanf(e, "%d,%d", &th, &tw);
    if (hy < 0 || hx < 0 || th < 0 || tw < 1) return false;
    const int H0 = lt.H[0], W0 = lt.W[0];
    // tallest tile (most queries per workgroup) whose queries fit the lanes and whose two largest
    // windows (even / odd levels) fit the LDS
    const int th_first = th ? th : 16, th_last = th ? th : 4;
    for (int cand = th_first; cand >= th_last; --cand) {
        const long long nq = tile_max_queries(lt, L, cand, tw);
        if (nq < 1 || nq > kV4Pairs) continue;
        long long cap[2] = {1, 1}, max_ww = 1;
        for (int l = 0; l < L; ++l) {
            long long wh = ((long long)cand * lt.H[l] + H0 - 1) / H0 + 2 * hy + 3;
            long long ww = ((long long)tw * lt.W[l] + W0 - 1) / W0 + 2 * hx + 3;
            if (wh > lt.H[l]) wh = lt.H[l];
            if (ww > lt.W[l]) ww = lt.W[l];
            if (wh * ww > cap[l & 1]) cap[l & 1] = wh * ww;
            if (ww > max_ww) max_ww = ww;
        }
        // windows are filled in whole 1-KiB DMA chunks: round the capacities up to 64 slots
        for (int k = 0; k < 2; ++k) cap[k] = ((cap[k] * slots + 63) / 64 * 64 + slots - 1) / slots;
        const size_t need = (size_t)kTiledHeaderBytes + row * (size_t)(1 + cap[0] + cap[1]) + 1024;
        if (need > kTiledLdsBudget) continue;
        tg->TH = cand;
        tg->TW = tw;
        tg->HY = hy;
        tg->HX = hx;
        tg->tiles_y = (H0 + cand - 1) / cand;
        tg->tiles_x = (W0 + tw - 1) / tw;
        tg->cap_even = (int)cap[0];
        tg->cap_odd = (int)cap[1];
        tg->debug = 0;
        if (const char *e = getenv("TF_MSDA_TILED_DEBUG")) tg->debug = atoi(e);
        *lds = need;
        return true;
    }
    return false;
}

// Launch msda_fwd_f32_direct when the shape qualifies (D == 32, P == 4, L <= 8).  Returns false if not.
bool direct_enabled()
{
    static const int on = [] { const char *e = getenv("TF_MSDA_DIRECT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

bool launch_direct(bool fused, const DirectArgs &da, const LevelTable &lt, const int64_t *shapes_dev,
                   int D, int P, hipStream_t stream, hipError_t *err)
{
    if (!direct_enabled() || D != 32 || P != 4 || da.L > 8) return false;
    const int lpairs = (da.L + 1) / 2;
    const void *fn = nullptr;
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
            TileGeom tg;
            size_t tiled_lds = 0;
            if (shapes_host && Lq == S && P == 4 && is_aligned(loc, 16) && is_aligned(attn, 16) &&
                plan_tiles(lt, L, D, &tg, &tiled_lds)) {
                const long long grid = (long long)N * tg.tiles_y * tg.tiles_x * M;
                if (grid <= 0x7fffffffLL) {
                    const void *tfn = D == 32 ? (const void *)&msda_fwd_f32_tiled<4, 8>
                                              : (const void *)&msda_fwd_f32_tiled<4, 9>;
                    void *argv[] = {(void *)&value, (void *)&vbytes, (void *)&loc, (void *)&attn,
                                    (void *)&out,   (void *)&lt,     (void *)&shapes_dev, (void *)&S,
                                    (void *)&M,     (void *)&L,      (void *)&tg};
                    e = hipLaunchKernel(tfn, dim3((unsigned)grid), dim3(kV4Threads), argv, tiled_lds,
                                        stream);
                    return record_hip(e);
                }
            }
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
            return record_hip(e);
        }
    }
    const void *fn = pl.vec == 4 ? (const void *)&msda_fwd_rowgather<T, 4>
                                 : (const void *)&msda_fwd_rowgather<T, 1>;
    e = launch(fn, pl.grid, pl.lds, stream, value, loc, attn, out, lt, shapes_dev, S, M, D, L, Lq,
               P, total_pairs, pl.ppb, pl.DV);
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
    return record_hip(e);
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
    rc = record_hip(hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)N * S * M * D, stream));
    if (rc != TF_MSDA_OK) return rc;
    const long long total_pairs = (long long)N * Lq * M;
    const bool pow2 = (pl.DV & (pl.DV - 1)) == 0 && pl.DV <= 64;
    if constexpr (sizeof(T) == 4) {
        if (pl.vec == 4 && pow2 && (P == 1 || P == 2 || P == 4 || P == 8) &&
            buf_path_ok(lt, shapes_host != nullptr, N, S, M, D, L) && is_aligned(grad_value, 16)) {
            const unsigned vbytes = (unsigned)((long long)N * S * M * D * 4);
            const void *bfn = P == 1   ? (const void *)&msda_bwd_f32_buf<1>
                              : P == 2 ? (const void *)&msda_bwd_f32_buf<2>
                              : P == 4 ? (const void *)&msda_bwd_f32_buf<4>
                                       : (const void *)&msda_bwd_f32_buf<8>;
            const hipError_t be = launch(bfn, head_major_grid(N, Lq, M, pl.ppb), pl.lds, stream,
                                         value, vbytes, loc, attn, grad_out, grad_value, grad_loc,
                                         grad_attn, lt, shapes_dev, S, M, D, L, Lq, total_pairs,
                                         pl.ppb, pl.DV);
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
    return record_hip(e);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI (include/tf_msda.h)
// ---------------------------------------------------------------------------------------------
extern "C" {

int tf_msda_abi_version(void) { return TF_MSDA_ABI_VERSION; }

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
    const int prev = g_tiled_mode;
    g_tiled_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
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
