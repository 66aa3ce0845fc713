// trackformer_amd/csrc/msda_common.h -- types shared by the translation units of libtf_msda.so
// (msda_hip.hip: every kernel but one + the C ABI; msda_pquad.hip: the persistent encoder forward kernel).
#ifndef TF_MSDA_COMMON_H_
#define TF_MSDA_COMMON_H_

#include <hip/hip_runtime.h>

#include "tf_msda.h"

namespace tfm {

struct LevelTable {
    int H[TF_MSDA_MAX_LEVELS];
    int W[TF_MSDA_MAX_LEVELS];
    int start[TF_MSDA_MAX_LEVELS];
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// Cache policy of the output stores of the dense / element-wise kernels (buffer-store aux bits on gfx940+: 1 = sc0, 2 = nt, 16 = sc1).
// Measured in round 6 (profiles/r06_nt_stores.txt): non-temporal stores make an ISOLATED streaming kernel faster (the dirty output
// otherwise waits for the write-back at the end of the kernel: msda_fwd_f32_pquad2 40.6 -> 38.1 us in the harness), but in the frame
// every output is the next kernel's input, and with nt stores in all dense kernels the frame is SLOWER (335 vs 348 frames/s): the
// consumer then misses the caches.  Plain stores stay the default; the macro is the A/B switch (tools/build_variant_all.py).
#ifndef TF_STORE_AUX
#define TF_STORE_AUX 0
#endif
constexpr int kStoreAux = TF_STORE_AUX;
template <typename V>
__device__ __forceinline__ void stream_store(V *p, V v)
{
    if constexpr (kStoreAux == 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

constexpr unsigned kOobOffset = 0xFFFFFFF0u;  // >= num_records for every supported tensor
constexpr unsigned kOobBase = 0xFFFFFF00u;    // ... and so is kOobBase + (lane slice offset < 0xF0)

// Fused prologue (ms_deform_attn.py:69-86 inside the kernel): raw projection outputs + reference points.
struct FusedArgs {
    const float *ref;     // [N, Lq, L, ref_dim]
    const float *qproj;   // [N*Lq, ld]: per query, M*L*P*2 raw offsets at off_col, M*L*P logits at logit_col
    int ref_dim, ld, off_col, logit_col;
    int head_major;       // block -> pair mapping, see msda_fwd_f32_buf
};

struct DirectArgs {
    const float *value;
    unsigned value_bytes;
    const float *loc, *attn;   // plain operator inputs (FUSED == false)
    float *out;
    FusedArgs fa;              // FUSED == true
    int S, M, L, Lq;
    long long nlq;             // N * Lq
};

// the kernel the calling thread dispatched last (tf_msda_last_kernel, include/tf_msda.h); `name` must be a string literal
void note_kernel(const char *name);

// msda_pquad.hip: the persistent LDS-window encoder forward (msda_fwd_f32_pquad).  launch_pquad returns
// false when the call does not qualify (the caller falls through to the next kernel).
bool launch_pquad(bool fused, const DirectArgs &da, const LevelTable &lt, int N, int D, int P, hipStream_t stream,
                  hipError_t *err);
int pquad_set_option(const char *name, int value);   // -1: unknown name, else the previous value
void pquad_set_trace(unsigned long long *device_buffer);

// ffn_fused.hip: row tiles per block of tf_ffn_fused_f32 (1..3, default 3); returns the previous value
int ffn_set_ti(int v);
int ffn_tail_split();            // tf_ffn_fused_f32: the rows behind the full rounds of 64-row blocks go to 32-row blocks (0 / 1, default 1)
int ffn_set_tail_split(int v);
int linln_set_ti(int v);   // tf_linear_res_ln_f32 (1..3; 0 = by row count)
// linear_stream.hip: row tiles per block of tf_linear_packed_f32 (2..4; 0 = per shape); returns the previous value
int linear_stream_set_ti(int v);
// linear_stream.hip: 1 = the halo form of the stride-1 3 x 3 convolutions (default), 0 = the stream form; returns the previous value
int conv_halo_set(int v);
// linear_stream.hip: the LDS-DMA GEMM behind tf_linear_packed_f32: 0 = off (the stream form), 1..4 = a fixed block shape, 9 = per call
int linear_dma_set(int v);
// mha_core.hip: 1 = the matrix-core kernel (default), 0 = the vector kernel of round 4; returns the previous value
int mha_set_mfma(int v);

}  // namespace tfm

#endif  // TF_MSDA_COMMON_H_
