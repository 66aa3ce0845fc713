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

template <int PT>
__global__ void __launch_bounds__(kThreads)
msda_fwd_f32_buf(const float *__restrict__ value, unsigned value_bytes,
                 const float *__restrict__ loc, const float *__restrict__ attn,
                 float *__restrict__ out, const LevelTable lt, const int64_t *__restrict__ dshapes,
                 int S, int M, int D, int L, int Lq, long long total_pairs, int ppb, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_tab = reinterpret_cast<int *>(smem);
    const int LP = L * PT;
    float *s_loc = reinterpret_cast<float *>(smem + kLevelTableBytes);
    float *s_attn = s_loc + (size_t)ppb * LP * 2;

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
// backward (grad_value via atomics, grad_loc / grad_attn via wave reduction), fused
// ---------------------------------------------------------------------------------------------
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
                    if (t.k1) unsafeAtomicAdd(gvl + e1 + c, w1 * top);                // cuh:296-301
                    if (t.k2) unsafeAtomicAdd(gvl + e2 + c, w2 * top);
                    if (t.k3) unsafeAtomicAdd(gvl + e3 + c, w3 * top);
                    if (t.k4) unsafeAtomicAdd(gvl + e4 + c, w4 * top);
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

// fp32 fast path eligibility: everything addressable with 32-bit byte offsets / 24-bit multiplies.
bool buf_path_ok(const LevelTable &lt, bool host_shapes, int N, int S, int M, int D, int L)
{
    const long long bytes = (long long)N * S * M * D * 4;
    if (bytes >= (long long)kOobOffset) return false;
    if ((long long)M * D * 4 >= (1 << 24)) return false;
    if (S >= (1 << 24)) return false;   // also bounds every level's H*W (and start) below 2^24
    if (host_shapes)
        for (int l = 0; l < L; ++l)
            if ((long long)lt.H[l] * lt.W[l] >= (1 << 24)) return false;
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
            const void *fn = P == 1   ? (const void *)&msda_fwd_f32_buf<1>
                             : P == 2 ? (const void *)&msda_fwd_f32_buf<2>
                             : P == 4 ? (const void *)&msda_fwd_f32_buf<4>
                                      : (const void *)&msda_fwd_f32_buf<8>;
            e = launch(fn, pl.grid, pl.lds, stream, value, vbytes, loc, attn, out, lt, shapes_dev,
                       S, M, D, L, Lq, total_pairs, pl.ppb, pl.DV);
            return record_hip(e);
        }
    }
    const void *fn = pl.vec == 4 ? (const void *)&msda_fwd_rowgather<T, 4>
                                 : (const void *)&msda_fwd_rowgather<T, 1>;
    e = launch(fn, pl.grid, pl.lds, stream, value, loc, attn, out, lt, shapes_dev, S, M, D, L, Lq,
               P, total_pairs, pl.ppb, pl.DV);
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
