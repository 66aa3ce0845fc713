// tests/emu/hipemu/hipemu.h -- TEST INFRASTRUCTURE, not product code.
//
// A small SIMT emulator that runs the repo's HIP kernels (trackformer_amd/csrc/*.hip, compiled as plain C++
// for the host through the shim tests/emu/hipemu/hip/hip_runtime.h) on the CPU, so that the kernels' own
// source -- tile plans, LDS windows, DPP exchanges, LDS-DMA staging, MFMA fragment indexing -- can be checked
// against the oracle without a GPU (tests/test_emu_*.py, `-m "not gpu"`).  Nothing under trackformer_amd/
// includes, links or loads any of this.
//
// Model.  One workgroup at a time per host thread; every work-item of the workgroup is a fiber (its own
// stack, hand-written context switch).  A fiber runs until it reaches
//   * a workgroup barrier (__syncthreads / s_barrier): resumes when every live work-item of the workgroup
//     has arrived;
//   * a wavefront operation (DPP move, readlane / readfirstlane, shuffles, votes, MFMA, LDS-DMA, wave
//     barrier / wavefront fence): the operation is executed ONCE for the wave, by the scheduler, when all
//     live lanes of the 64-wide wave have deposited their operands -- lane l then sees exactly what the
//     hardware's lock-step execution would give it;
//   * its end.
// Lanes of one wave that wait at DIFFERENT wave operations (divergent control flow around a cross-lane
// operation) are executed group by group with the other lanes inactive and counted in
// stats().divergent_ops; a cross-lane read of an inactive lane yields 0 and is counted in
// stats().inactive_reads.  The tests require both to be 0.
//
// LDS.  The dynamic LDS of a workgroup is a 160 KB arena mapped below 4 GiB, so the kernels' idiom of
// keeping LDS addresses as 32-bit integers ((unsigned)(size_t)(address_space(3) T *)p and back) works
// unchanged.  `__shared__` arrays are static thread_local storage of the host thread (one workgroup at a time).
//
// LDS-DMA (buffer_load ... lds) is ASYNCHRONOUS here as on the hardware: the data lands at the issuing lane's
// next s_waitcnt with vmcnt == 0 (or at its end), not at the call -- a kernel that reads a staged window
// without waiting sees stale LDS contents in the emulation too.
#ifndef TF_HIPEMU_H_
#define TF_HIPEMU_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>

namespace hipemu {

struct Dim3 {
    unsigned x, y, z;
    constexpr Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

enum OpKind : int {
    kOpSync = 1,       // wave barrier / wavefront-scope fence
    kOpShuffle,        // out = in of lane `src` (src < 0: out = `fallback`); size bytes
    kOpFirstLane,      // out = in of the first active lane
    kOpVote,           // out (uint64) = ballot of (in != 0); aux out: exec mask
    kOpMfma32x32x16Bf16,
    kOpMfma16x16x32Bf16,
    kOpMfma32x32x16F16,
    kOpMfma16x16x4F32,  // in / in2: ONE float of A (row lane % 16, k lane / 16) / of B (column lane % 16, k lane / 16); in3 / out: 4 floats
    kOpLdsDma,         // in: 16/4 bytes of data fetched by the lane; lds base = first active lane's pointer
    kOpLdsTrack16,     // in: the lane's 32-bit LDS address of a ds_read_b128; accounting only (HIPEMU_LDS_TRACK=1)
};

struct OpReq {
    int kind;
    unsigned site;
    const void *in;
    void *out;
    int size;
    int src;             // kOpShuffle: source lane (0..63) or -1
    uint64_t aux;        // kOpShuffle: fallback bits; kOpLdsDma: lds base pointer; kOpVote: receives exec mask
    const void *in2, *in3;
};

struct Fiber;
struct Stats {
    uint64_t wave_ops, barriers, divergent_ops, inactive_reads, lds_dma_bytes, blocks, switches;
    uint64_t lds_b128_reads, lds_b128_cycles;   // tracked ds_read_b128 wave instructions / their bank-conflict cycles
};

Fiber *cur();
const Dim3 &thread_idx();
const Dim3 &block_idx();
const Dim3 &block_dim();
const Dim3 &grid_dim();
int lane_id();          // 0..63 within the wave
void *dynamic_lds();    // base of the workgroup's dynamic LDS (below 4 GiB, 1 KiB aligned)
void barrier();
void wave_op(OpReq &r);
void waitcnt_vm(int vmcnt);                        // lands this lane's pending LDS-DMA writes beyond `vmcnt`
void defer_lds_write(void *dst, const void *src, int bytes);
uint64_t realtime();
int num_cus();          // HIPEMU_CUS (default 4): what hipGetDeviceProperties reports
Stats &stats();         // process-wide, accumulated over launches (reset_stats() clears)
void reset_stats();
void track_lds_read16(unsigned lds_addr);          // optional bank-conflict accounting of a ds_read_b128

// Runs body() once per work-item of a grid x block launch.  Returns 0, or -1 when a workgroup dead-locked
// (some work-items wait at a barrier that the others can never reach).
int launch(Dim3 grid, Dim3 block, size_t dynamic_lds_bytes, const std::function<void()> &body);

// ---- wave-level primitives on top of wave_op() ---------------------------------------------------------------
template <class T>
inline T shuffle_from(T v, int src_lane, T fallback, unsigned site)
{
    static_assert(sizeof(T) <= 8, "32- or 64-bit payloads");
    T out;
    uint64_t fb = 0;
    memcpy(&fb, &fallback, sizeof(T));
    OpReq r{kOpShuffle, site, &v, &out, (int)sizeof(T), src_lane, fb, nullptr, nullptr};
    wave_op(r);
    return out;
}
template <class T>
inline T first_lane(T v, unsigned site)
{
    static_assert(sizeof(T) <= 8, "32- or 64-bit payloads");
    T out;
    OpReq r{kOpFirstLane, site, &v, &out, (int)sizeof(T), 0, 0, nullptr, nullptr};
    wave_op(r);
    return out;
}
inline uint64_t ballot(bool pred, uint64_t *exec_mask, unsigned site)
{
    int p = pred ? 1 : 0;
    uint64_t out = 0;
    OpReq r{kOpVote, site, &p, &out, 4, 0, 0, nullptr, nullptr};
    wave_op(r);
    if (exec_mask) *exec_mask = r.aux;
    return out;
}
inline void wave_sync(unsigned site)
{
    OpReq r{kOpSync, site, nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
    wave_op(r);
}

// v_mov_b32_dpp source-lane selection (gfx9 DPP control codes); returns -1 when the source is out of range
// (bound_ctrl then decides between 0 and `old`).
int dpp_source_lane(int lane, int ctrl);

template <class T>
inline T mov_dpp(T old, T v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, unsigned site)
{
    static_assert(sizeof(T) == 4, "32-bit DPP");
    const int lane = lane_id();
    int src = dpp_source_lane(lane, ctrl);
    const bool enabled = ((row_mask >> ((lane >> 4) & 3)) & 1) && ((bank_mask >> ((lane >> 2) & 3)) & 1);
    T zero;
    memset(&zero, 0, sizeof(T));
    T got = shuffle_from<T>(v, src, bound_ctrl ? zero : old, site);
    return enabled ? got : old;
}

}  // namespace hipemu

#endif  // TF_HIPEMU_H_
