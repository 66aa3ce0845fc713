// tests/emu/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: stands in for <hip/hip_runtime.h> when the repo's
// .hip sources are compiled as plain C++ for the host and run under the SIMT emulator (hipemu.h).  It maps the
// HIP / amdgcn vocabulary the kernels use onto the emulator: work-item indices, barriers, wave operations
// (DPP, readlane, shuffles, votes, MFMA), buffer resources with hardware bounds checking, LDS-DMA, atomics.
#ifndef TF_HIPEMU_HIP_RUNTIME_H_
#define TF_HIPEMU_HIP_RUNTIME_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <memory>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>

#include "hipemu.h"

// ---- qualifiers -----------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

#define threadIdx (::hipemu::thread_idx())
#define blockIdx (::hipemu::block_idx())
#define blockDim (::hipemu::block_dim())
#define gridDim (::hipemu::grid_dim())
typedef ::hipemu::Dim3 dim3;

// ---- vector types ---------------------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

// ---- runtime API (the handful of calls the launch code makes) -------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
typedef void *hipStream_t;
struct hipDeviceProp_t {
    int multiProcessorCount;
    size_t sharedMemPerBlock;
    size_t maxSharedMemoryPerMultiProcessor;
    char name[64];
    char gcnArchName[64];
};
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    p->multiProcessorCount = ::hipemu::num_cus();
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    strcpy(p->name, "hipemu");
    strcpy(p->gcnArchName, "gfx950");
    return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }

namespace hipemu {
// kernels reached through hipLaunchKernel(const void *, ...) are registered (with their parameter types) where
// their address is taken: the build script rewrites `(const void *)&kernel<...>` into HIPEMU_FN(kernel<...>)
typedef std::function<std::function<void()>(void **)> BodyMaker;
inline std::unordered_map<const void *, BodyMaker> &registry()
{
    static std::unordered_map<const void *, BodyMaker> r;
    return r;
}
inline std::mutex &registry_mutex()
{
    static std::mutex m;
    return m;
}
template <class... P, size_t... I>
inline std::function<void()> make_body(void (*k)(P...), void **argv, std::index_sequence<I...>)
{
    auto args = std::make_shared<std::tuple<std::decay_t<P>...>>(*reinterpret_cast<std::decay_t<P> *>(argv[I])...);
    return [k, args]() { k(std::get<I>(*args)...); };
}
template <class... P>
inline const void *register_kernel(void (*k)(P...))
{
    std::lock_guard<std::mutex> g(registry_mutex());
    const void *key = reinterpret_cast<const void *>(k);
    if (!registry().count(key))
        registry()[key] = [k](void **argv) { return make_body(k, argv, std::index_sequence_for<P...>{}); };
    return key;
}
}  // namespace hipemu
#define HIPEMU_FN(...) (::hipemu::register_kernel(&__VA_ARGS__))

static inline hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **argv, size_t lds, hipStream_t)
{
    ::hipemu::BodyMaker mk;
    {
        std::lock_guard<std::mutex> g(::hipemu::registry_mutex());
        auto it = ::hipemu::registry().find(fn);
        if (it == ::hipemu::registry().end()) {
            fprintf(stderr, "hipemu: hipLaunchKernel of an unregistered kernel %p\n", fn);
            return hipErrorInvalidValue;
        }
        mk = it->second;
    }
    return ::hipemu::launch(grid, block, lds, mk(argv)) == 0 ? hipSuccess : hipErrorLaunchFailure;
}
template <class... P, class... A>
static inline void hipLaunchKernelGGL(void (*k)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t, A &&...a)
{
    auto args = std::make_shared<std::tuple<std::decay_t<P>...>>(std::forward<A>(a)...);
    std::function<void()> body = [k, args]() { std::apply(k, *args); };
    ::hipemu::launch(grid, block, lds, body);
}

// ---- work-group / wave primitives -----------------------------------------------------------------------------
#define __syncthreads() ::hipemu::barrier()
#define __builtin_amdgcn_s_barrier() ::hipemu::barrier()
#define __builtin_amdgcn_wave_barrier() ::hipemu::wave_sync(__LINE__)
#define __builtin_amdgcn_fence(order, scope) ::hipemu::wave_sync(__LINE__)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_nop(n) ((void)0)
#define __builtin_amdgcn_s_memrealtime() ::hipemu::realtime()
#define __builtin_amdgcn_s_memtime() ::hipemu::realtime()
// s_waitcnt simm16 (gfx9): vmcnt = [3:0] | [15:14] << 4
#define __builtin_amdgcn_s_waitcnt(v) ::hipemu::waitcnt_vm((int)(((v) & 0xF) | ((((v) >> 14) & 3) << 4)))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_readfirstlane(v) ::hipemu::first_lane((v), __LINE__)
#define __builtin_amdgcn_readlane(v, l) ::hipemu::shuffle_from((v), (int)(l), (v), __LINE__)
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) ::hipemu::mov_dpp((v), (v), (ctrl), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) ::hipemu::mov_dpp((old), (v), (ctrl), (rm), (bm), (bc), __LINE__)
// ds_bpermute_b32: lane i <- v of lane (addr_i / 4) mod 64
#define __builtin_amdgcn_ds_bpermute(addr, v) ::hipemu::shuffle_from((int)(v), (int)((((unsigned)(addr)) >> 2) & 63u), (int)(v), __LINE__)
// ds_swizzle_b32, bit-mask mode (pattern bit 15 == 0): inside each group of 32 lanes, lane i <- lane ((i & and) | or) ^ xor
#define __builtin_amdgcn_ds_swizzle(v, pat) ::hipemu::shuffle_from((int)(v), (::hipemu::lane_id() & 32) | (((((::hipemu::lane_id() & 31) & ((pat) & 31)) | (((pat) >> 5) & 31)) ^ (((pat) >> 10) & 31)) & 31), (int)(v), __LINE__)
typedef short hipemu_s16x2 __attribute__((ext_vector_type(2)));
static inline hipemu_s16x2 hipemu_cvt_pk_i16(int a, int b)   // v_cvt_pk_i16_i32: {sat16(a), sat16(b)}
{
    auto sat = [](int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); };
    return hipemu_s16x2{sat(a), sat(b)};
}
#define __builtin_amdgcn_cvt_pk_i16(a, b) hipemu_cvt_pk_i16((a), (b))

static inline int __lane_id() { return ::hipemu::lane_id(); }
static inline bool hipemu_any(bool p, unsigned site) { return ::hipemu::ballot(p, nullptr, site) != 0; }
static inline bool hipemu_all(bool p, unsigned site)
{
    uint64_t exec = 0;
    const uint64_t b = ::hipemu::ballot(p, &exec, site);
    return b == exec;
}
#define __any(p) hipemu_any((p), __LINE__)
#define __all(p) hipemu_all((p), __LINE__)
#define __ballot(p) ::hipemu::ballot((p), nullptr, __LINE__)

template <class T>
static inline T hipemu_shfl(T v, int src, int width, unsigned site)
{
    const int self = ::hipemu::lane_id();
    return ::hipemu::shuffle_from<T>(v, (src & (width - 1)) + (self & ~(width - 1)), v, site);
}
template <class T>
static inline T hipemu_shfl_xor(T v, int mask, int width, unsigned site)
{
    const int self = ::hipemu::lane_id();
    int idx = self ^ mask;
    if (idx >= ((self + width) & ~(width - 1))) idx = self;
    return ::hipemu::shuffle_from<T>(v, idx, v, site);
}
template <class T>
static inline T hipemu_shfl_up(T v, unsigned delta, int width, unsigned site)
{
    const int self = ::hipemu::lane_id();
    int idx = self - (int)delta;
    if (idx < (self & ~(width - 1))) idx = self;
    return ::hipemu::shuffle_from<T>(v, idx, v, site);
}
template <class T>
static inline T hipemu_shfl_down(T v, unsigned delta, int width, unsigned site)
{
    const int self = ::hipemu::lane_id();
    int idx = self + (int)delta;
    if ((int)((self & (width - 1)) + delta) >= width) idx = self;
    return ::hipemu::shuffle_from<T>(v, idx, v, site);
}
#define HIPEMU_PICK3(_1, _2, _3, NAME, ...) NAME
#define hipemu_shfl2(v, s) hipemu_shfl((v), (s), 64, __LINE__)
#define hipemu_shfl3(v, s, w) hipemu_shfl((v), (s), (w), __LINE__)
#define __shfl(...) HIPEMU_PICK3(__VA_ARGS__, hipemu_shfl3, hipemu_shfl2, )(__VA_ARGS__)
#define hipemu_shfl_xor2(v, s) hipemu_shfl_xor((v), (s), 64, __LINE__)
#define hipemu_shfl_xor3(v, s, w) hipemu_shfl_xor((v), (s), (w), __LINE__)
#define __shfl_xor(...) HIPEMU_PICK3(__VA_ARGS__, hipemu_shfl_xor3, hipemu_shfl_xor2, )(__VA_ARGS__)
#define hipemu_shfl_up2(v, s) hipemu_shfl_up((v), (s), 64, __LINE__)
#define hipemu_shfl_up3(v, s, w) hipemu_shfl_up((v), (s), (w), __LINE__)
#define __shfl_up(...) HIPEMU_PICK3(__VA_ARGS__, hipemu_shfl_up3, hipemu_shfl_up2, )(__VA_ARGS__)
#define hipemu_shfl_down2(v, s) hipemu_shfl_down((v), (s), 64, __LINE__)
#define hipemu_shfl_down3(v, s, w) hipemu_shfl_down((v), (s), (w), __LINE__)
#define __shfl_down(...) HIPEMU_PICK3(__VA_ARGS__, hipemu_shfl_down3, hipemu_shfl_down2, )(__VA_ARGS__)

// ---- math -----------------------------------------------------------------------------------------------------
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : x > 1.f ? 1.f : x; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

// ---- atomics (workgroups may run on several host threads: real atomics) --------------------------------------
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class F, class U>
static inline F hipemu_atomic_fadd(F *p, F v)
{
    U old_bits, new_bits;
    F old;
    __atomic_load(reinterpret_cast<U *>(p), &old_bits, __ATOMIC_RELAXED);
    do {
        memcpy(&old, &old_bits, sizeof(F));
        const F nv = old + v;
        memcpy(&new_bits, &nv, sizeof(F));
    } while (!__atomic_compare_exchange(reinterpret_cast<U *>(p), &old_bits, &new_bits, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
static inline float atomicAdd(float *p, float v) { return hipemu_atomic_fadd<float, uint32_t>(p, v); }
static inline double atomicAdd(double *p, double v) { return hipemu_atomic_fadd<double, uint64_t>(p, v); }
static inline float unsafeAtomicAdd(float *p, float v) { return atomicAdd(p, v); }
static inline double unsafeAtomicAdd(double *p, double v) { return atomicAdd(p, v); }
static inline int atomicMin(int *p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int *p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMin(unsigned *p, unsigned v)
{
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMax(unsigned *p, unsigned v)
{
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- buffer resources: base + size, out-of-range loads return 0, out-of-range stores / atomics are dropped ---
struct hipemu_buffer_rsrc {
    unsigned char *base;
    uint32_t num_records;
};
typedef hipemu_buffer_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned int hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int hipemu_u32x2 __attribute__((ext_vector_type(2)));

template <class T>
static inline hipemu_buffer_rsrc hipemu_make_rsrc(T *p, int /*stride*/, unsigned num_records, int /*flags*/)
{
    return hipemu_buffer_rsrc{reinterpret_cast<unsigned char *>(const_cast<std::remove_const_t<T> *>(p)), num_records};
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hipemu_make_rsrc((p), (stride), (unsigned)(num), (flags))

// raw buffer, offen: the range check covers voffset (+ the instruction offset), not soffset (gfx9 rule)
static inline bool hipemu_in_range(const hipemu_buffer_rsrc &r, uint32_t voff, uint32_t bytes)
{
    return (uint64_t)voff + bytes <= (uint64_t)r.num_records;
}
template <class V>
static inline V hipemu_buffer_load(const hipemu_buffer_rsrc &r, uint32_t voff, uint32_t soff)
{
    V v;
    memset(&v, 0, sizeof(V));
    if (hipemu_in_range(r, voff, sizeof(V))) memcpy(&v, r.base + (size_t)voff + soff, sizeof(V));
    return v;
}
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) hipemu_buffer_load<hipemu_u32x4>((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) hipemu_buffer_load<hipemu_u32x2>((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) hipemu_buffer_load<unsigned int>((r), (voff), (soff))
template <class V>
static inline void hipemu_buffer_store(V v, const hipemu_buffer_rsrc &r, uint32_t voff, uint32_t soff)
{
    if (hipemu_in_range(r, voff, sizeof(V))) memcpy(r.base + (size_t)voff + soff, &v, sizeof(V));
}
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux) hipemu_buffer_store((v), (r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, aux) hipemu_buffer_store((v), (r), (voff), (soff))
static inline float hipemu_buffer_atomic_fadd(float v, const hipemu_buffer_rsrc &r, uint32_t voff, uint32_t soff)
{
    if (!hipemu_in_range(r, voff, 4)) return 0.f;
    return atomicAdd(reinterpret_cast<float *>(r.base + (size_t)voff + soff), v);
}
#define __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, voff, soff, aux) hipemu_buffer_atomic_fadd((v), (r), (voff), (soff))

// buffer_load ... lds: lane l's `size` bytes go to (wave-uniform LDS base) + l * size, asynchronously (hipemu.h)
template <class LdsPtr>
static inline void hipemu_buffer_load_lds(const hipemu_buffer_rsrc &r, LdsPtr lds, int size, uint32_t voff, uint32_t soff,
                                          int imm, unsigned site)
{
    if (imm != 0 || (size != 16 && size != 4)) {
        fprintf(stderr, "hipemu: buffer_load ... lds with size %d / instruction offset %d is not modelled\n", size, imm);
        abort();
    }
    unsigned char data[16] = {0};
    if (hipemu_in_range(r, voff, (uint32_t)size)) memcpy(data, r.base + (size_t)voff + soff, size);
    ::hipemu::OpReq q{::hipemu::kOpLdsDma, site, data, nullptr, size, 0, (uint64_t)(uintptr_t)lds, nullptr, nullptr};
    ::hipemu::wave_op(q);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, size, voff, soff, imm, aux) \
    hipemu_buffer_load_lds((r), (lds), (size), (voff), (soff), (imm), __LINE__)

// ---- MFMA (gfx950 bf16 shapes + the 32 x 32 x 16 fp16 one): executed once per wave by the emulator with the hardware's fragment layout ---
template <class AB, class CD>
static inline CD hipemu_mfma(int kind, AB a, AB b, CD c, unsigned site)
{
    CD d;
    ::hipemu::OpReq q{kind, site, &a, &d, 0, 0, 0, &b, &c};
    ::hipemu::wave_op(q);
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma(::hipemu::kOpMfma32x32x16Bf16, (a), (b), (c), __LINE__)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu_mfma(::hipemu::kOpMfma16x16x32Bf16, (a), (b), (c), __LINE__)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu_mfma(::hipemu::kOpMfma32x32x16F16, (a), (b), (c), __LINE__)
template <class CD>
static inline CD hipemu_mfma_f32(float a, float b, CD c, unsigned site)
{
    CD d;
    ::hipemu::OpReq q{::hipemu::kOpMfma16x16x4F32, site, &a, &d, 0, 0, 0, &b, &c};
    ::hipemu::wave_op(q);
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_f32((float)(a), (float)(b), (c), __LINE__)

#endif  // TF_HIPEMU_HIP_RUNTIME_H_
